"""Import path of the reference's rl/networks/storage.py (`from rl.networks.storage import RolloutStorage`, train.py:16)."""
from crowdnav_prediction_attngraph_amd.storage import RolloutStorage  # noqa: F401
