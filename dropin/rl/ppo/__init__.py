from crowdnav_prediction_attngraph_amd.ppo import PPO  # noqa: F401
