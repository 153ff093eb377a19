from crowdnav_prediction_attngraph_amd.policy import Policy  # noqa: F401
