"""`arguments.get_args()` with the reference's flag names and defaults (/root/reference/arguments.py:6-217), `--env-name`
included (CrowdSimPredRealGST-v0).  The one place where the shims decide something the reference leaves to the user is
dropin/crowd_nav/configs/config.py: `sim.predict_method` / `env.use_wrapper` follow --env-name instead of being edited by hand
(README.md:66-69 of the reference asks for exactly that consistency)."""
import argparse

import torch

# (flag, kwargs) -- table form of the reference's parser; dest names are what train.py / Policy / PPO read
_FLAGS = [
    ("--output_dir", dict(type=str, default="trained_models/my_model")),
    ("--resume", dict(default=False, action="store_true")),
    ("--load-path", dict(default="trained_models/GST_predictor_non_rand/checkpoints/41200.pt")),
    ("--overwrite", dict(default=True, action="store_true")),
    ("--num_threads", dict(type=int, default=1)),
    ("--phase", dict(type=str, default="test")),
    ("--cuda-deterministic", dict(action="store_true", default=False)),
    ("--no-cuda", dict(action="store_true", default=False)),
    ("--seed", dict(type=int, default=425)),
    ("--num-processes", dict(type=int, default=16)),
    ("--num-mini-batch", dict(type=int, default=2)),
    ("--num-steps", dict(type=int, default=30)),
    ("--recurrent-policy", dict(action="store_true", default=True)),
    ("--ppo-epoch", dict(type=int, default=5)),
    ("--clip-param", dict(type=float, default=0.2)),
    ("--value-loss-coef", dict(type=float, default=0.5)),
    ("--entropy-coef", dict(type=float, default=0.0)),
    ("--lr", dict(type=float, default=4e-5)),
    ("--eps", dict(type=float, default=1e-5)),
    ("--alpha", dict(type=float, default=0.99)),
    ("--max-grad-norm", dict(type=float, default=0.5)),
    ("--num-env-steps", dict(type=float, default=20e6)),
    ("--use-linear-lr-decay", dict(action="store_true", default=False)),
    ("--algo", dict(default="ppo")),
    ("--save-interval", dict(type=int, default=200)),
    ("--use-gae", dict(action="store_true", default=True)),
    ("--gae-lambda", dict(type=float, default=0.95)),
    ("--log-interval", dict(type=int, default=20)),
    ("--gamma", dict(type=float, default=0.99)),
    ("--use-proper-time-limits", dict(action="store_true", default=False)),
    ("--human_node_rnn_size", dict(type=int, default=128)),
    ("--human_human_edge_rnn_size", dict(type=int, default=256)),
    ("--aux-loss", dict(action="store_true", default=False)),
    ("--human_node_input_size", dict(type=int, default=3)),
    ("--human_human_edge_input_size", dict(type=int, default=2)),
    ("--human_node_output_size", dict(type=int, default=256)),
    ("--human_node_embedding_size", dict(type=int, default=64)),
    ("--human_human_edge_embedding_size", dict(type=int, default=64)),
    ("--attention_size", dict(type=int, default=64)),
    ("--seq_length", dict(type=int, default=30)),
    ("--use_self_attn", dict(type=bool, default=True)),
    ("--use_hr_attn", dict(type=bool, default=True)),
    ("--env-name", dict(default="CrowdSimPredRealGST-v0")),
    ("--sort_humans", dict(type=bool, default=True)),
]


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="RL")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    args = parser.parse_args(argv)
    args.cuda = not args.no_cuda and torch.cuda.is_available()
    assert args.algo in ["ppo"], "only PPO is on the accelerated path"
    return args
