#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X: env-steps/s of (policy forward -> crowd_sim step) at 20 humans.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole batch: the attention-graph policy's forward on the current
observations (fp32, sampled action), the ORCA/crowd-sim step for all E envs with in-launch auto-reset, and the done
mask for the next forward.  Inputs (env state, observations, weights) are resident in HBM before the timed region.
Workload at N=1: BASELINE configs[1] -- CrowdSimVarNum-v0, 20 humans, 4096 envs, HH+HR attention on.  With N ranks
every rank owns 4096 envs (weak scaling, global env indices rank*4096.., no data-path collective in the rollout).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic policy-forward FLOPs per env-step (unfolded reference graph) and compulsory HBM bytes
def flops_per_env_step(H, D):
    return 2 * (H * (D * 128 + 128 * 512 + 3 * 512 ** 2 + 3 * 512 ** 2 + 512 ** 2 + 512 * 256 + 256 * 64) + 2 * 8 * H * H * 64
                + 9 * 256 + 256 * 64 * 3 + H * 64 + H * 256 + 3 * 128 * 256 + 128 * 256 + 4 * 256 ** 2 + 256 + 512)


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0


def cpu_baseline_multi(H, workers, envs=48, steps=60):
    """The same bounded sample in `workers` independent single-threaded processes at once (the shape of the reference's
    SubprocVecEnv path: one process per group of envs) -> aggregate rate over the host cores actually used."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(H), str(envs), str(steps)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT) for _ in range(workers)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    wall = time.perf_counter() - t0
    recs = [json.loads(o.decode().strip().splitlines()[-1]) for o in outs if o.strip()]
    if len(recs) != workers:
        raise RuntimeError("%d of %d CPU workers returned a result" % (len(recs), workers))
    busy = max(r["seconds"] for r in recs)          # timed region of the slowest worker (start-up and imports excluded)
    total = sum(r["env_steps"] for r in recs)
    return {"value": round(total / busy, 1), "unit": "env-steps/s", "cores": workers, "kind": "port",
            "sample": "%d single-threaded processes x (%d envs x %d steps) of the same workload (H=%d) run concurrently; slowest worker %.1f s "
                      "(wall incl. start-up %.1f s); one process alone: %s" % (workers, envs, steps, H, busy, wall, recs[0]["sample_single"])}


def cpu_baseline(H, envs=48, steps=60):
    """The oracle (kind='port': scalar C sim + numpy policy forward) on ONE host core, bounded sample."""
    import numpy as np
    from oracle import oracle as O
    from oracle import policy_oracle as P
    from tests import policy_util as PU
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    shapes = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "policy_varnum_e4_h20.npz"))["meta"]))["shapes"]
    sd = PU.formula_state_dict({k: tuple(v) for k, v in shapes.items()})
    cfg = O.default_config(human_num=H, nenv=envs)
    oenvs = [O.OracleEnv(cfg, 425 + i) for i in range(envs)]
    obs = [e.reset() for e in oenvs]
    h = np.zeros((envs, 128))
    masks = np.ones((envs, 1))
    rs = np.random.RandomState(0)
    std = np.exp(sd["dist.logstd._bias"].astype(np.float64).reshape(1, 2))

    def run(n):
        nonlocal obs, h, masks
        t_sim = t_pol = 0.0
        for _ in range(n):
            t0 = time.perf_counter()
            batch = {k: np.stack([o[k] for o in obs]) for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
            _, mean, _, h, _ = P.act(sd, batch, h, masks)
            act = (mean + std * rs.standard_normal((envs, 2))).astype(np.float32)
            t1 = time.perf_counter()
            dones = []
            for i, e in enumerate(oenvs):
                ob, _, d, _ = e.step(act[i], autoreset=True)
                obs[i] = ob
                dones.append(d)
            masks = 1.0 - np.array(dones, dtype=np.float64).reshape(envs, 1)
            t2 = time.perf_counter()
            t_pol += t1 - t0
            t_sim += t2 - t1
        return t_sim, t_pol

    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        run(2)
        t_sim, t_pol = run(steps)
    finally:
        if ctx is not None:
            ctx.unregister() if hasattr(ctx, "unregister") else None
    n = envs * steps
    return {"value": round(n / (t_sim + t_pol), 2), "unit": "env-steps/s", "cores": 1, "kind": "port", "seconds": t_sim + t_pol, "env_steps": n,
            "sample": "%d envs x %d steps of the same workload (H=%d): scalar C sim %.0f env-steps/s, numpy fp64 policy forward %.0f env-steps/s, 1 thread"
                      % (envs, steps, H, n / t_sim, n / t_pol)}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":      # child of cpu_baseline_multi: no torch, no GPU
        r = cpu_baseline(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        print(json.dumps({"seconds": r["seconds"], "env_steps": r["env_steps"], "sample_single": "%.0f env-steps/s" % r["value"]}))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--humans", type=int, default=20)
    ap.add_argument("--env-name", default="CrowdSimVarNum-v0")
    ap.add_argument("--randomized", action="store_true", help="randomize_attributes + random_goal_changing (BASELINE configs[4] stress shape)")
    ap.add_argument("--max-placement-attempts", type=int, default=0,
                    help="bound of the reference's unbounded rejection sampling of human positions / goals (0 = library default 65536); dense "
                         "randomised crowds (configs[4]) need a small bound or the batch waits for its unluckiest env")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="CPU baseline: number of concurrent single-threaded oracle processes (-1 = min(64, host cores / 2); 1 = one process)")
    ap.add_argument("--no-ppo", action="store_true", help="skip the PPO samples/sec leg (rollout + update, 3 updates of T=30)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing test: put every rank on GPU 0 (use with --dist-backend gloo)")
    ap.add_argument("--gemm", choices=["fused", "bf16x3", "fp32"], default="fused",
                    help="human-human block: 'fused' = one persistent kernel, bf16x3 split-precision MFMA (default); 'bf16x3' = the same arithmetic "
                         "as separate launches (round-1 path); 'fp32' = exact fp32 MFMA, separate launches")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev_index = 0 if args.same_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend)

    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch, HipPolicy
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces

    E, H = args.envs, args.humans
    kind = A.ENV_KINDS[args.env_name]
    cfg = A.default_env_config(human_num=H, env_kind=kind, nenv=E * world, randomize_attributes=int(args.randomized),
                               random_goal_changing=int(args.randomized), max_placement_attempts=args.max_placement_attempts)
    env = HipEnvBatch(cfg, E, 425, first_env_index=rank * E)
    D = env.D
    torch.manual_seed(425)
    ob_space, act_space = make_spaces(H, D)
    net = Policy(ob_space.spaces, act_space, base_kwargs=dict(env_name=args.env_name, num_processes=E), base="selfAttn_merge_srnn").cuda()
    pol = HipPolicy(H, D, E)
    pol.set_gemm_mode(args.gemm)
    pol.set_weights(net.state_dict())
    obs = env.reset()
    gst = None
    if args.env_name == "CrowdSimPredRealGST-v0":
        # configs[3]: GST predictor + VecPretextNormalize in the loop (random-init predictor weights: throughput is value independent)
        from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
        from crowdnav_prediction_attngraph_amd.hip import HipGST
        gst = HipGST(H, E)
        gst.set_weights(GSTPredictor().cuda().state_dict())
        gst.wrapper_reset(E)
        pol_obs = dict(obs)
        pol_obs["spatial_edges"] = torch.empty(E, H, D, device="cuda")
        gst_rew = torch.zeros(E, device="cuda")
        gst.wrapper_step(obs, gst_rew, 0.6, -20.0, out=pol_obs["spatial_edges"])
    else:
        pol_obs = obs
    hxs = [torch.zeros(E, 1, 128, device="cuda"), torch.zeros(E, 1, 128, device="cuda")]
    out = dict(value=torch.empty(E, 1, device="cuda"), action=torch.empty(E, 2, device="cuda"), logp=torch.empty(E, 1, device="cuda"), hxs=hxs[1])
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    eps = torch.empty(E, 2, device="cuda")

    masks2 = [torch.ones(E, 1, device="cuda"), torch.ones(E, 1, device="cuda")]

    def step(i):
        eps.normal_(generator=gen)
        out["hxs"] = hxs[(i + 1) & 1]
        pol.act(pol_obs, hxs[i & 1], masks2[i & 1], eps=eps, out=out)
        _, reward, done, _, _, _ = env.step(out["action"], not_done=masks2[(i + 1) & 1])   # done mask for the next forward
        if gst is not None:
            gst.wrapper_step(obs, reward, 0.6, -20.0, out=pol_obs["spatial_edges"])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    pol.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    pol.set_profiling(False)
    prof_ms, prof_n = pol.get_profile()
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda" if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # second half of BASELINE.json's metric: PPO samples/sec = T * E_total / wall time of (rollout + GAE + update), the
    # reference's train.py loop (rl/ppo.py defaults: T = 30, 5 epochs x 2 recurrent minibatches); every rank runs it,
    # gradients are all-reduced once per optimiser step (one flat bucket), time = max over ranks of the last update
    ppo = None
    if not args.no_ppo and args.env_name != "CrowdSimPredRealGST-v0":
        del env, pol, net
        torch.cuda.empty_cache()
        from crowdnav_prediction_attngraph_amd.trainer import train
        from crowdnav_prediction_attngraph_amd import config as CFG
        over = {"sim.human_num": H}
        if args.env_name == "CrowdSimPred-v0":
            over["sim.predict_method"] = "const_vel"
        tcfg = CFG.Config(**dict(over, **{"humans.end_goal_changing": True})) if args.randomized else CFG.non_randomized(**over)
        try:
            hist, _ = train(env_name=args.env_name, num_processes=E, num_steps=30, num_updates=3, seed=425, config=tcfg, log=None)
            last = hist[-1]
            tt = torch.tensor([last["rollout_s"], last["update_s"]], device="cuda" if (dist is None or args.dist_backend == "nccl") else "cpu", dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            r_s, u_s = float(tt[0]), float(tt[1])
            ppo = {"samples_per_s": round(30 * E * world / (r_s + u_s), 1), "rollout_s": round(r_s, 5), "update_s": round(u_s, 5),
                   "config": "T=30 steps x %d envs per GPU, ppo_epoch 5, num_mini_batch 2, Adam; 3 updates run, the last one timed" % E,
                   "value_loss": round(last["value_loss"], 6)}
        except Exception as exc:   # the headline line is still printed; a failure here is the same on every rank
            ppo = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    total_env_steps = E * world * args.steps
    value = total_env_steps / elapsed
    # dominant kernel: the folded QKV projection GEMM [M,512]x[512,1536] (fp32 MFMA), timed with HIP events on its stream
    M = prof_n[1] / max(prof_n[0], 1)      # mean live (env, human) rows per step (device-side counter): padded humans are not computed
    fused = args.gemm == "fused"
    # fused: the timed kernel is the whole human-human block; its algorithmic work = the three dense layers on the live rows
    # (embedding_layer.2 128->512, folded q|k|v 512->1536, folded out_proj∘spatial_linear 512->256); the D->128 input layer and the
    # attention core (<2 % of it) are left out of the count
    qkv_flops = 2.0 * M * (128 * 512 + 512 * 1536 + 512 * 256) if fused else 2.0 * M * 512 * 1536
    qkv_ms = prof_ms[0] / max(prof_n[0], 1)
    achieved = qkv_flops / (qkv_ms * 1e-3) / 1e12 if qkv_ms > 0 else 0.0
    F = flops_per_env_step(H, D)
    split = args.gemm in ("bf16x3", "fused")
    # the split runs 3 bf16 MFMA passes per algorithmic product, so its attainable algorithmic rate is the bf16 peak / 3
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    kname = ("hh_fused_kernel (v_mfma_f32_16x16x32_bf16, 3 passes hi*hi+hi*lo+lo*hi): embedding -> q|k|v -> attention -> out_proj∘spatial_linear in one launch" if fused else
             "gemm3_nt_kernel<128,NONE> (v_mfma_f32_32x32x16_bf16, 3 passes hi*hi+hi*lo+lo*hi): folded q|k|v projection" if split
             else "gemm_nt_kernel<128,NONE> (v_mfma_f32_32x32x2_f32): folded q|k|v projection")
    # HBM traffic of the dominant kernel comes from the committed PMC passes (bench.py cannot run rocprofv3 on itself)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if args.gemm == "bf16x3" and E == 4096 and H == 20 and os.path.exists(tpath):
        traffic = json.load(open(tpath))["hbm_bytes_per_launch_corrected"]
    line = {
        "metric": "env-steps/sec (sim+policy fwd) at %d humans" % H, "value": round(value, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if not split else "f32 (big GEMMs as bf16x3 split-precision MFMA, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "%s%s, %d humans, %d parallel envs per GPU, HH+HR attention on, "
                               "policy forward + ORCA sim step + auto-reset per step" % (
                                   "BASELINE configs[1]: " if (args.env_name, H, E, args.randomized) == ("CrowdSimVarNum-v0", 20, 4096, False) else ("randomized humans, " if args.randomized else ""), args.env_name, H, E),
                   "envs_per_gpu": E, "humans": H, "parallelism": "dp%d (envs sharded, no rollout collective)" % world,
                   "policy_init": "orthogonal, torch.manual_seed(425)", "sampled_actions": True},
        "roofline": {"bound": "mfma", "kernel": "%s, M=%d live rows of %d%s" % (kname, M, E * H, "" if fused else ", N=1536 K=512"),
                     "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "note": ("achieved = algorithmic 2*M*(128*512+512*1536+512*256) / launch time (hipEvents on the kernel's stream); peak = 2500 TFLOP/s dense bf16 "
                              "MFMA / 3 passes of the hi/lo split; executed bf16 MFMA rate = %.1f TFLOP/s" % (3 * achieved)) if fused else
                             ("achieved = algorithmic 2*M*N*K / launch time (hipEvents on the kernel's stream); peak = 2500 TFLOP/s dense bf16 "
                              "MFMA / 3 passes of the hi/lo split; executed bf16 MFMA rate = %.1f TFLOP/s" % (3 * achieved)) if split else
                             "achieved = algorithmic 2*M*N*K / launch time on exact fp32 MFMA",
                     "launch_ms": round(qkv_ms, 4), "launches": int(prof_n[0]), "mean_detected_humans": round(M / E, 3),
                     "whole_step": {"reference_graph_flops_per_env_step": F,
                                    "reference_graph_tflops_equivalent": round(value / world * F / 1e12, 2),
                                    "note": "env-steps/s x the FLOPs of the reference's dense, unfolded forward; NOT a hardware utilisation "
                                            "(padded humans are not computed and affine pairs are folded)"}},
    }
    if ppo is not None:
        line["ppo"] = ppo
    if not args.no_cpu_baseline and world == 1:
        workers = args.cpu_workers if args.cpu_workers > 0 else max(1, min(64, (os.cpu_count() or 2) // 2))
        cb = None
        if workers > 1:
            try:
                cb = cpu_baseline_multi(H, workers)
            except Exception as exc:      # fall back to the single-process measurement
                cb = None
                sys.stderr.write("cpu_baseline_multi failed (%s); measuring one process\n" % exc)
        if cb is None:
            cb = cpu_baseline(H)
            cb.pop("seconds", None); cb.pop("env_steps", None)
        line["cpu_baseline"] = cb
        line["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        line["gpu_over_cpu"] = round(value / cb["value"], 1)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
