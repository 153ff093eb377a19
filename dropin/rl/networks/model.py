"""Import path of the reference's rl/networks/model.py (`from rl.networks.model import Policy`, train.py:15): same state-dict keys and seeded init."""
from crowdnav_prediction_attngraph_amd.policy import Policy  # noqa: F401
