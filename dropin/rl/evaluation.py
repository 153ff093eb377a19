from crowdnav_prediction_attngraph_amd.evaluation import evaluate, evaluate_batched  # noqa: F401
