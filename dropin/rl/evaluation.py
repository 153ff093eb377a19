"""Import path of the reference's rl/evaluation.py (`from rl.evaluation import evaluate`, test.py:10): the drop-in mirror lives in the package."""
from crowdnav_prediction_attngraph_amd.evaluation import evaluate, evaluate_batched  # noqa: F401
