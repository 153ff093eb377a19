from crowdnav_prediction_attngraph_amd.storage import RolloutStorage  # noqa: F401
