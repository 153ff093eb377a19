"""`import crowd_sim` of the reference (crowd_sim/__init__.py:8-26 registers the gym ids): here the four accelerated ids map
to single-env objects that are E = 1 views over the device simulator (crowdnav_prediction_attngraph_amd.gym_env).  gym itself
is not needed; `crowd_sim.make(id)` / `crowd_sim.registry` stand in for gym.make / gym's registry."""
from crowdnav_prediction_attngraph_amd.gym_env import (CrowdSimPred, CrowdSimPredRealGST, CrowdSimVarNum,  # noqa: F401
                                                       CrowdSimVarNumCollect, make, registry)

__all__ = ["registry", "make", "CrowdSimVarNum", "CrowdSimPred", "CrowdSimPredRealGST", "CrowdSimVarNumCollect"]
