from crowdnav_prediction_attngraph_amd.vec_env import BatchedCrowdSim, make_vec_envs  # noqa: F401
