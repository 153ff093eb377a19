"""Registers the three accelerated gym ids (crowd_sim/__init__.py:8-26 in the reference).  gym itself is not needed:
the ids are resolved by crowdnav_prediction_attngraph_amd.vec_env.make_vec_envs."""
from crowdnav_prediction_attngraph_amd._abi import ENV_KINDS as registry  # noqa: F401

__all__ = ["registry"]
