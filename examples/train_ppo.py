#!/usr/bin/env python3
"""Train the attention-graph policy with PPO on the batched MI355X simulator (the reference's train.py flow).

    python examples/train_ppo.py --env-name CrowdSimVarNum-v0 --num-processes 4096 --updates 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ppo.py ...   # DP over env shards
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-name", default="CrowdSimVarNum-v0")
    ap.add_argument("--num-processes", type=int, default=4096)
    ap.add_argument("--num-steps", type=int, default=30)
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--humans", type=int, default=20)
    ap.add_argument("--randomized", action="store_true")
    ap.add_argument("--seed", type=int, default=425)
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    import torch
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.trainer import train
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    cfg = (C.Config if a.randomized else C.non_randomized)(**{"sim.human_num": a.humans})
    rank = int(os.environ.get("RANK", "0"))
    hist, pol = train(a.env_name, a.num_processes, a.num_steps, a.updates, a.seed, config=cfg,
                      log=(lambda r: print(json.dumps(r))) if rank == 0 else None)
    if a.save and rank == 0:
        torch.save(pol.state_dict(), a.save)


if __name__ == "__main__":
    main()
