"""Import path of the reference's rl/ppo package (`from rl import ppo` ... `ppo.PPO(...)`, train.py:11,148)."""
from crowdnav_prediction_attngraph_amd.ppo import PPO  # noqa: F401
