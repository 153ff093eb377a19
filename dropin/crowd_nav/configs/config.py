"""`crowd_nav.configs.config.Config` for the reference's train.py: the MI355X package's Config, bound to the parsed
command line like the reference does (config.py:11), with the env-id / predictor consistency the reference asks the
user to keep by hand (README.md:66-69) derived from --env-name."""
from arguments import get_args
from crowdnav_prediction_attngraph_amd.config import Config as _Config


class Config(_Config):
    def __init__(self):
        args = get_args()
        super().__init__(args=args)
        self.env.num_processes = args.num_processes
        self.training.device = "cuda:0" if args.cuda else "cpu"
        self.training.load_path = args.load_path   # the reference never wires --load-path (train.py:106 bug); do it here
        self.sim.predict_method = {"CrowdSimVarNum-v0": "none", "CrowdSimPred-v0": "const_vel",
                                   "CrowdSimPredRealGST-v0": "inferred"}.get(args.env_name, "none")
        self.env.use_wrapper = self.sim.predict_method == "inferred"
