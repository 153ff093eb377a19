"""Import path of the reference's rl/networks/envs.py (`from rl.networks.envs import make_vec_envs`, train.py:14): one batched device simulator instead of N subprocess envs."""
from crowdnav_prediction_attngraph_amd.vec_env import BatchedCrowdSim, make_vec_envs  # noqa: F401
