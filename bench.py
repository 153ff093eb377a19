#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X: env-steps/s of (policy forward -> crowd_sim step) at 20 humans.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E]        (N > 1 without a launcher: re-executes itself under
                                                                            torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole batch: the attention-graph policy's forward on the current
observations (fp32, sampled action), the ORCA/crowd-sim step for all E envs with in-launch auto-reset, and the done
mask for the next forward.  Inputs (env state, observations, weights) are resident in HBM before the timed region.
Workload at N=1: BASELINE configs[1] -- CrowdSimVarNum-v0, 20 humans, 4096 envs, HH+HR attention on.  With N ranks
every rank owns 4096 envs (weak scaling, global env indices rank*4096.., no data-path collective in the rollout).
Rank 0 prints ONE JSON line.
"""
import argparse
import datetime
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


def cpu_baseline_threads(H, env_name="CrowdSimVarNum-v0", threads=None, target_s=12.0):
    """the reference-equivalent CPU path on THIS box's host cores -- scalar C simulator (the oracle, one env per
    call, envs sharded over P worker threads; ctypes releases the GIL) + the policy forward as a torch CPU fp32 graph with
    torch.set_num_threads(P) -- on a bounded sample of the same workload.  kind = 'port' (the reference's Python cannot travel)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import torch
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from oracle import oracle as O
    logical = os.cpu_count() or 2
    P = threads if threads and threads > 0 else max(1, min(128, logical // 2))     # physical cores (SMT pairs counted once)
    E = max(8 * P, 4096 // P * P)                 # the GPU run's batch (4096 envs) unless the box has > 512 cores
    kind = {"CrowdSimVarNum-v0": 0, "CrowdSimPred-v0": 1, "CrowdSimPredRealGST-v0": 2}[env_name]
    cfg = O.default_config(human_num=H, nenv=E, env_kind=kind)
    D = O.obs_width(cfg)
    envs = [O.OracleEnv(cfg, 425 + i) for i in range(E)]
    obs0 = [e.reset() for e in envs]
    L = O.lib()
    handles = (C.c_void_p * E)(*[e._h for e in envs])
    buf = dict(robot_node=np.zeros((E, 1, 7), np.float32), temporal_edges=np.zeros((E, 1, 2), np.float32), spatial_edges=np.zeros((E, H, D), np.float32),
               detected_human_num=np.zeros((E, 1), np.float32), visible=np.zeros((E, H), np.uint8), rewards=np.zeros(E, np.float32),
               dones=np.zeros(E, np.uint8), infos=np.zeros(E, np.uint8))
    for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num"):
        buf[k][:] = np.stack([o[k] for o in obs0]).reshape(buf[k].shape)
    actions = np.zeros((E, 2), np.float32)
    fp, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    per = E // P

    def sim_chunk(w):
        lo = w * per
        hp = C.cast(C.byref(handles, lo * C.sizeof(C.c_void_p)), C.POINTER(C.c_void_p))
        L.orc_env_batch_step(hp, per, actions[lo:].ctypes.data_as(fp), buf["robot_node"][lo:].ctypes.data_as(fp), buf["temporal_edges"][lo:].ctypes.data_as(fp),
                             buf["spatial_edges"][lo:].ctypes.data_as(fp), buf["detected_human_num"][lo:].ctypes.data_as(fp), buf["visible"][lo:].ctypes.data_as(u8),
                             buf["rewards"][lo:].ctypes.data_as(fp), buf["dones"][lo:].ctypes.data_as(u8), buf["infos"][lo:].ctypes.data_as(u8))

    nthreads_before = torch.get_num_threads()
    torch.set_num_threads(P)
    torch.manual_seed(425)
    ob_space, act_space = make_spaces(H, D)
    net = Policy(ob_space.spaces, act_space, base_kwargs=dict(env_name=env_name, num_processes=E), base="selfAttn_merge_srnn")
    hxs = {"human_node_rnn": torch.zeros(E, 1, 128), "human_human_edge_rnn": None}
    masks = torch.ones(E, 1)
    pool = ThreadPoolExecutor(P)
    t_sim = t_pol = 0.0

    def one_step():
        nonlocal hxs, masks, t_sim, t_pol
        t0 = time.perf_counter()
        obs = {k: torch.from_numpy(buf[k]) for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
        _, action, _, h = net.act(obs, hxs, masks)
        actions[:] = action.numpy()
        hxs = {"human_node_rnn": h["human_node_rnn"], "human_human_edge_rnn": None}
        t1 = time.perf_counter()
        list(pool.map(sim_chunk, range(P)))
        masks = torch.from_numpy(1.0 - buf["dones"].astype(np.float32)).view(E, 1)
        t2 = time.perf_counter()
        t_pol += t1 - t0
        t_sim += t2 - t1

    try:
        for _ in range(3):
            one_step()
        per_step = (t_sim + t_pol) / 3
        steps = int(max(5, min(400, target_s / max(per_step, 1e-4))))
        t_sim = t_pol = 0.0
        for _ in range(steps):
            one_step()
    finally:
        pool.shutdown()
        torch.set_num_threads(nthreads_before)
    n = E * steps
    return {"value": round(n / (t_sim + t_pol), 1), "unit": "env-steps/s", "cores": P, "kind": "port",
            "sample": "%d envs x %d steps of the same workload (H=%d, %s): scalar C simulator sharded over %d threads %.0f env-steps/s, torch CPU fp32 policy "
                      "forward (set_num_threads(%d)) %.0f env-steps/s; %.1f s of CPU work" % (E, steps, H, env_name, P, n / t_sim, P, n / t_pol, t_sim + t_pol),
            "host_logical_cpus": logical}


def pmc_traffic_live(argv_tail, timeout_s=170):
    """HBM-side traffic of the dominant kernel measured IN this run: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE -- they do not fit one
    pass, MI355X_MICROARCH.md) over a short child run of this very script on this box and build; per launch = the mean over every
    hh_fused_kernel dispatch of the child.  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B, same guide); WRITE_SIZE
    as reported.  Returns a dict or raises; the caller falls back to the committed constant."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    out = {}
    child_line = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cn_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 --pmc %s exited with %d: %s" % (counter, r.returncode, (r.stderr or "")[-300:]))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if lines:
                child_line = json.loads(lines[-1])
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if not dbs:
                raise RuntimeError("no rocpd database from the %s pass" % counter)
            cur = sqlite3.connect(dbs[0]).cursor()
            row = list(cur.execute(
                "select count(*), avg(p.value) from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id "
                "join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                "where s.kernel_name like '%hh_fused_kernel%' and i.name = ?", (counter,)))[0]
            if not row[0]:
                raise RuntimeError("no %s samples of hh_fused_kernel" % counter)
            out[counter] = (int(row[0]), float(row[1]))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_kb, write_kb = out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]
    res = {"hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024), "fetch_size_kb": round(fetch_kb, 1), "write_size_kb": round(write_kb, 1),
           "launches": out["FETCH_SIZE"][0], "child_live_rows": None}
    if child_line is not None:
        import re
        m = re.search(r"M=(\d+) live rows", child_line["roofline"]["kernel"])
        res["child_live_rows"] = int(m.group(1)) if m else None
    return res


def other_configs_leg(timeout_s=150, budget_s=330):
    """The single-GPU shares of BASELINE configs[2], [3], [4] (extra keys, after the headline): each a short child run of this script on
    this box and build (a separate process: its own simulator batch, policy and -- configs[3] -- the GST predictor + VecPretextNormalize
    in the loop), env-steps/s, ms per step and the per-kernel medians of its step."""
    import subprocess
    common = ["--gpus", "1", "--no-cpu-baseline", "--no-ppo", "--no-worst-case", "--no-dropin", "--no-pmc-traffic", "--no-other-configs"]
    legs = [
        ("configs[2]: CrowdSimPred-v0, 20 humans, const_vel predictor, 4096 envs per GPU", ["--env-name", "CrowdSimPred-v0", "--envs", "4096", "--steps", "60", "--warmup", "20"]),
        ("configs[3]: CrowdSimPredRealGST-v0, 20 humans, GST predictor + wrapper in the loop, 2048 envs per GPU",
         ["--env-name", "CrowdSimPredRealGST-v0", "--envs", "2048", "--steps", "60", "--warmup", "20"]),
        ("configs[4]: CrowdSimVarNum-v0, 50 randomised humans, random_goal_changing, 8192 envs per GPU, default placement bound",
         ["--humans", "50", "--randomized", "--envs", "8192", "--steps", "40", "--warmup", "10", "--dephase", "120"]),
    ]
    out = []
    t_begin = time.perf_counter()
    for name, extra in legs:
        rec = {"config": name, "argv": " ".join(extra)}
        try:
            # the three legs share one wall-clock budget (they are extras of the driver's command: a slow box must cost a leg, not the line)
            left = budget_s - (time.perf_counter() - t_begin)
            if left < 20:
                raise RuntimeError("skipped: the %d s budget of the extra legs is spent" % budget_s)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + common + extra, capture_output=True, text=True, timeout=min(timeout_s, left))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError("exit %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
            d = json.loads(lines[-1])
            rec.update({"env_steps_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                        "hh_kernel": d["roofline"]["kernel"].split(":")[0].split(" (")[0], "hh_launch_ms": d["roofline"]["launch_ms"], "hh_frac": d["roofline"]["frac"],
                        "mean_detected_humans": d["roofline"]["mean_detected_humans"],
                        "kernel_median_us": (d.get("step_decomposition") or {}).get("median_us"),
                        "device_step_interval_us": (d.get("device_step_interval_us") or {}).get("median")})
        except Exception as exc:
            rec["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
        out.append(rec)
    return out


def dropin_leg(E, H, env_name, steps=24):
    """The rollout loop in the shape the reference's train.py runs it (train.py:152-189), through the reference-compatible interfaces
    (make_vec_envs / Policy.act / envs.step -> CPU rewards, numpy dones, infos list / RolloutStorage.insert) at this bench's batch size.
    Everything the zero-sync path (trainer.collect_rollout) avoids is in here: one device-to-host transfer + stream synchronisation per
    step in envs.step(), and train.py's OWN per-env Python (two list comprehensions over the envs building FloatTensors, the infos loop)."""
    import torch
    from crowdnav_prediction_attngraph_amd import config as CFG
    from crowdnav_prediction_attngraph_amd.policy import Policy
    from crowdnav_prediction_attngraph_amd.storage import RolloutStorage
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    dev = torch.device("cuda", torch.cuda.current_device())
    over = {"sim.human_num": H}
    if env_name == "CrowdSimPred-v0":
        over["sim.predict_method"] = "const_vel"
    cfg = CFG.non_randomized(**over)
    envs = make_vec_envs(env_name, 425, E, 0.99, None, dev, False, config=cfg)
    torch.manual_seed(425)
    ac = Policy(envs.observation_space.spaces, envs.action_space, base="selfAttn_merge_srnn",
                base_kwargs=dict(env_name=env_name, num_processes=E, num_mini_batch=2, seq_length=steps)).to(dev)
    ro = RolloutStorage(steps, E, envs.observation_space.spaces, envs.action_space, 128, 256)
    obs = envs.reset()
    for k in obs:
        ro.obs[k][0].copy_(obs[k])
    ro.to(dev)
    ep_rewards = []
    t = dict(act=0.0, env_step=0.0, loop_python=0.0, insert=0.0)

    def one(step, timed):
        c0 = time.perf_counter()
        with torch.no_grad():
            o = {k: ro.obs[k][step] for k in ro.obs}
            hx = {k: ro.recurrent_hidden_states[k][step] for k in ro.recurrent_hidden_states}
            value, action, logp, hxs = ac.act(o, hx, ro.masks[step])
        c1 = time.perf_counter()
        obs, reward, done, infos = envs.step(action)
        c2 = time.perf_counter()
        for info in infos:
            if "episode" in info.keys():
                ep_rewards.append(info["episode"]["r"])
        masks = torch.FloatTensor([[0.0] if d else [1.0] for d in done])
        bad_masks = torch.FloatTensor([[0.0] if "bad_transition" in info.keys() else [1.0] for info in infos])
        c3 = time.perf_counter()
        ro.insert(obs, hxs, action, logp, value, reward, masks, bad_masks)
        c4 = time.perf_counter()
        if timed:
            t["act"] += c1 - c0; t["env_step"] += c2 - c1; t["loop_python"] += c3 - c2; t["insert"] += c4 - c3
    for s_ in range(steps):          # warm-up pass over the storage, then the timed pass
        one(s_, False)
    ro.after_update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        one(s_, True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    envs.close()
    return {"env_steps_per_s": round(E * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
            "host_ms_per_step": {k: round(v / steps * 1e3, 4) for k, v in t.items()},
            "what": "train.py-shaped rollout loop through make_vec_envs / Policy.act / envs.step (CPU rewards, numpy dones, infos) / RolloutStorage.insert; "
                    "env_step includes the per-step device-to-host transfer and synchronisation, loop_python is train.py's own per-env list building"}


def main():
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
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="CPU baseline: worker threads P for the C simulator and torch.set_num_threads (-1 = physical host cores, capped at 128)")
    ap.add_argument("--dephase", type=int, default=240,
                    help="untimed pre-roll steps before the warm-up so that the envs are spread over their episodes (all envs start an episode "
                         "together at reset; SURVEY 8d asks for a de-phased steady-state window)")
    ap.add_argument("--kernel-events-every", type=int, default=0,
                    help="the dominant kernel is timed live with a pair of HIP events on its stream around every n-th launch of the timed "
                         "window (0 = every launch when --steps <= 64, else every 4th: an event record on the critical stream costs a few "
                         "microseconds of dispatch gap).  Every launch is ALSO timed by the kernel's own device-clock stamps, which cost nothing")
    ap.add_argument("--tail", choices=["deferred", "inline"], default="inline",
                    help="where a sim step's side-stream tail (ORCA fallback programs + episode pre-generation) is enqueued: 'inline' = by the sim step "
                         "itself (default, fastest: 0.287 ms per step), 'deferred' = right behind the policy's human-human kernel (cn_env_set_tail_deferral + "
                         "cn_policy_set_post_hh_hook: every kernel runs undisturbed and 5-12 us shorter, but the cross-stream event that releases the "
                         "tail costs 15-25 us and its chain ends after the robot-node kernel: 0.325-0.33 ms per step, profiles/HISTORY.md)")
    ap.add_argument("--pregen-budget-us", type=float, default=None, help="time budget of one launch of the episode pre-generation kernel (library default 40)")
    ap.add_argument("--timeline-out", default=None, help="write the stamped timeline of the decomposition window (all kernels of 24 steps) to this file")
    ap.add_argument("--no-worst-case", action="store_true", help="skip the second timed window with every human detected (all H rows live)")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc child runs of this script, ~1 min); the committed "
                         "constant of the same kernel source is printed instead when there is one")
    ap.add_argument("--no-dropin", action="store_true", help="skip the leg that times the reference-shaped rollout loop through the drop-in interfaces")
    ap.add_argument("--no-ppo", action="store_true", help="skip the PPO samples/sec leg (rollout + update, 3 updates of T=30)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the legs that time the single-GPU shares of BASELINE configs[2], [3], [4] (three short child runs of this script, ~1 min)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing test: put every rank on GPU 0 (use with --dist-backend gloo)")
    ap.add_argument("--gemm", choices=["fused", "bf16x3", "fp32"], default="fused",
                    help="human-human block: 'fused' = one persistent kernel, bf16x3 split-precision MFMA (default); 'bf16x3' = the same arithmetic "
                         "as separate launches (round-1 path); 'fp32' = exact fp32 MFMA, separate launches")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks (one process per GPU) under torch.distributed.run on this node
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would not describe the run" % (args.gpus, world))
    if not args.same_gpu and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d device(s) visible (one rank per GPU; --same-gpu is a plumbing test only)" % (world, torch.cuda.device_count()))
    dev_index = 0 if args.same_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group(args.dist_backend, timeout=datetime.timedelta(minutes=10))

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
    if args.tail == "deferred":
        env.set_tail_deferral(True)
        pol.attach_env_tail(env)
    if args.pregen_budget_us is not None:
        env.set_pregen_budget(int(args.pregen_budget_us * 100))
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
    NOISE_BLOCK = 30                      # action noise is drawn a rollout's worth (30 steps, train.py --num-steps) at a time, like the trainer does
    eps = torch.empty(NOISE_BLOCK, E, 2, device="cuda")

    masks2 = [torch.ones(E, 1, device="cuda"), torch.ones(E, 1, device="cuda")]

    force_all_detected = [False]
    noise_i = [0]

    def forward(i):
        if noise_i[0] % NOISE_BLOCK == 0:
            eps.normal_(generator=gen)
        out["hxs"] = hxs[(i + 1) & 1]
        # the simulator's row plan belongs to the observation env.step() just wrote: not to the GST-wrapped one, nor to one whose
        # detected_human_num was overwritten for the worst-case leg (the kernel then cuts the envs in order itself)
        plan = env.row_plan if (gst is None and not force_all_detected[0]) else None
        pol.act(pol_obs, hxs[i & 1], masks2[i & 1], eps=eps[noise_i[0] % NOISE_BLOCK], out=out, row_plan=plan)
        noise_i[0] += 1

    def step(i):
        """One pass of the hot path: the crowd-sim step of all E envs under the actions of the last forward (ORCA humans, reward, auto-reset,
        new observation, done mask) and the policy forward on the observation it produced.  A timed window of K steps therefore starts
        with a simulator launch and ends with the last forward's heads: the ORCA solve a sim step leaves running for ITS successor (side
        stream, overlapped with the forward) is over before that forward ends, so nothing of the K steps is left in flight at the closing
        synchronize() and nothing of an earlier step is."""
        _, reward, done, _, _, _ = env.step(out["action"], not_done=masks2[i & 1])   # done mask for the forward below
        if gst is not None:
            gst.wrapper_step(obs, reward, 0.6, -20.0, out=pol_obs["spatial_edges"])
        if force_all_detected[0]:
            pol_obs["detected_human_num"].fill_(float(H))     # worst case of the state-dependent work: every (env, human) row is live
        forward(i)

    forward(1)               # the first action (untimed): forward on the reset observation; step i then runs sim -> forward
    from crowdnav_prediction_attngraph_amd.hip import StepStamps
    ev_every = args.kernel_events_every if args.kernel_events_every > 0 else (1 if args.steps <= 64 else 4)
    it = 0
    # untimed pre-roll: every env starts an episode at reset, so the first ~40 steps are a lock-step transient (few humans in sensor
    # range, no resets); after a few hundred steps of sampled actions the envs are spread over their episodes
    for _ in range(args.dephase + (args.dephase & 1)):
        step(it); it += 1
    # the measurement instruments are switched on BEFORE the warm-up so that their one-off costs (the first record of a timing event
    # switches the queue's profiling on: some hundred microseconds; the first launches with a stamp argument) fall into it, not into
    # the timed window; what they collected during the warm-up is dropped below
    pol.set_profiling(ev_every)
    stamps = StepStamps(args.steps, ("hh_fused", "rn_fused"))
    for _ in range(args.warmup + (args.warmup & 1)):
        step(it); it += 1
    torch.cuda.synchronize()
    pol.reset_profile()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        stamps.next()
        step(it + i)
    t_enq = time.perf_counter() - t0       # host time to ENQUEUE the K steps (well below `elapsed` = the host runs ahead of the device)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stamps.close()
    pol.set_profiling(False)
    prof_ms, prof_n = pol.get_profile()
    ev_samples = sorted(pol.get_profile_samples())
    dev_hh = sorted(x * 1e-3 for x in stamps.durations_us("hh_fused"))     # ms, every launch of the window
    dev_rn = sorted(x * 1e-3 for x in stamps.durations_us("rn_fused"))
    dev_rows = stamps.counts("hh_fused")
    hh_starts = [a_ for (_, k_, a_, _) in stamps.table() if k_ == "hh_fused"]
    step_intervals = [round(b_ - a_, 1) for a_, b_ in zip(hh_starts[:-1], hh_starts[1:])]     # us between consecutive human-human launches
    it += args.steps            # the hxs / masks ping-pong follows the step index: advance by exactly the steps taken
    plan_hdr = env.row_plan[:8].tolist() if (gst is None and env.row_plan.numel() >= 8) else None   # [1] workgroups [6] tiles of the last planned launch
    per_rank = None
    rank_devices = None
    if dist is not None:
        # which device every rank really sits on (the line must describe the run: N ranks on N DISTINCT GPUs unless --same-gpu)
        prop = torch.cuda.get_device_properties(dev_index)
        me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": prop.name,
              "uuid": str(getattr(prop, "uuid", "")), "pci_bus_id": getattr(prop, "pci_bus_id", None), "cus": prop.multi_processor_count}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, me)
        tdev0 = "cuda" if args.dist_backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed], device=tdev0, dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank = [round(E * args.steps / float(x.item()), 1) for x in allt]
        elapsed = max(float(x.item()) for x in allt)    # max over ranks
    # step decomposition: a separate short window (NOT the timed one) in which every kernel of the step stamps the device clock --
    # per-kernel durations and, because the clock is global, where each sits on the step's timeline (gaps, side-stream overlap)
    decomp = None
    if rank == 0:
        DEC = 24
        dst = StepStamps(DEC)
        for i in range(DEC):
            dst.next()
            step(it + i)
        torch.cuda.synchronize()
        dst.close()
        it += DEC
        med = lambda v: (sorted(v)[len(v) // 2] if v else None)   # noqa: E731
        decomp = {"window": "%d extra steps after the timed window, every kernel stamped (device clock, 10 ns)" % DEC, "median_us": {}}
        for k in ("env_step", "orca_lane", "row_plan", "hh_fused", "rn_fused", "orca_lp3", "env_pregen", "other"):
            d = dst.durations_us(k)
            if d:   # slot "other" = env_post_kernel (dense crowds: the deferred goal changes, on the side stream in front of ORCA)
                decomp["median_us"]["env_post" if k == "other" else k] = round(med(d), 2)
                if k == "other":
                    decomp["env_post_max_us"] = round(max(d), 2)
        tab = dst.table()
        # critical path of a step on the caller's stream: env_step -> orca_lane (+ row plan) -> hh_fused -> rn_fused -> next env_step
        by = {}
        for s_, k, a_, b_ in tab:
            by[(s_, k)] = (a_, b_)
        gaps = {"env_step_to_orca_lane": [], "orca_lane_to_hh": [], "hh_to_rn": [], "rn_to_env_step": [], "step": []}
        for s_ in range(DEC - 1):
            try:
                gaps["env_step_to_orca_lane"].append(by[(s_, "orca_lane")][0] - by[(s_, "env_step")][1])
                gaps["orca_lane_to_hh"].append(by[(s_, "hh_fused")][0] - max(by[(s_, "orca_lane")][1], by.get((s_, "row_plan"), (0, 0))[1]))
                gaps["hh_to_rn"].append(by[(s_, "rn_fused")][0] - by[(s_, "hh_fused")][1])
                gaps["rn_to_env_step"].append(by[(s_ + 1, "env_step")][0] - by[(s_, "rn_fused")][1])
                gaps["step"].append(by[(s_ + 1, "env_step")][0] - by[(s_, "env_step")][0])
            except KeyError:
                pass
        decomp["median_gap_us"] = {k: round(med(v), 2) for k, v in gaps.items() if v and k != "step"}
        if gaps["step"]:
            decomp["median_step_us"] = round(med(gaps["step"]), 2)
        if args.timeline_out:
            with open(args.timeline_out, "w") as f:
                f.write("# device-clock stamps (s_memrealtime, 10 ns) of every kernel of %d consecutive rollout steps; python bench.py %s\n" % (DEC, " ".join(sys.argv[1:])))
                f.write("# step kernel start_us end_us duration_us\n")
                for s_, k, a_, b_ in tab:
                    f.write("%3d %-11s %10.2f %10.2f %8.2f\n" % (s_, k, a_, b_, b_ - a_))
                # when do the human-human kernel's workgroups get their CUs?  (it needs whole CUs: 160 KB of LDS -- a CU on which a
                # wavefront of another launch still holds LDS admits it only when that wavefront is gone).  Start stamps of its
                # workgroups 0..63 and end stamps by workgroup & 63, relative to the end of the ORCA lane launch of the same step
                import numpy as np
                raw = dst.ring.cpu().numpy()
                kid = A.PROF_KERNEL_IDS
                f.write("# hh_fused workgroups 0..63 of each step, us after the end of the step's orca_lane launch: first instruction min / median / p90 / max | "
                        "last wavefront out (by workgroup & 63) min / median / max | env_pregen's last wavefront out\n")
                for s_ in range(DEC):
                    late = raw[s_, kid["hh_fused"], 8:1024:16].astype(np.uint64)
                    st0 = raw[s_, kid["hh_fused"], 0:1024:16].astype(np.uint64)
                    en0 = raw[s_, kid["hh_fused"], 1024:2048:16].astype(np.uint64)
                    lane_end = max(raw[s_, kid["orca_lane"], 1024:2048:16].astype(np.uint64).max(), raw[s_, kid["row_plan"], 1024:2048:16].astype(np.uint64).max())
                    pg_end = raw[s_, kid["env_pregen"], 1024:2048:16].astype(np.uint64).max()
                    ok = st0 != np.uint64(0xFFFFFFFFFFFFFFFF)
                    if not ok.any() or lane_end == 0:
                        continue
                    a_ = np.sort((st0[ok].astype(np.int64) - np.int64(lane_end)) * 0.01)
                    b_ = np.sort((en0[en0 > 0].astype(np.int64) - np.int64(lane_end)) * 0.01)
                    l_ = np.sort((late[late > 0].astype(np.int64) - np.int64(lane_end)) * 0.01)
                    f.write("#   step %2d: start %6.2f %6.2f %6.2f %6.2f | latest start per slot (all workgroups) %6.2f %6.2f %6.2f | end %7.2f %7.2f %7.2f | pregen end %7.2f\n" % (
                        s_, a_[0], a_[len(a_) // 2], a_[int(len(a_) * 0.9)], a_[-1], l_[0] if len(l_) else -1, l_[len(l_) // 2] if len(l_) else -1, l_[-1] if len(l_) else -1,
                        b_[0], b_[len(b_) // 2], b_[-1], (np.int64(pg_end) - np.int64(lane_end)) * 0.01))
    worst = None
    if not args.no_worst_case and rank == 0:
        force_all_detected[0] = True
        pol_obs["detected_human_num"].fill_(float(H))
        for _ in range(10):
            step(it); it += 1
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for i in range(args.steps):
            step(it + i)
        torch.cuda.synchronize()
        tw = time.perf_counter() - tw
        force_all_detected[0] = False
        worst = {"value": round(E * args.steps / tw, 1), "unit": "env-steps/s (this rank)", "ms_per_step": round(tw / args.steps * 1e3, 4), "mean_detected_humans": float(H),
                 "note": "second timed window, same steps: every env reports all %d humans detected (%d live rows instead of the natural count) -- "
                         "the upper bound of the state-dependent human-human work" % (H, E * H)}
    # second half of BASELINE.json's metric: PPO samples/sec = T * E_total / wall time of (rollout + GAE + update), the
    # reference's train.py loop (rl/ppo.py defaults: T = 30, 5 epochs x 2 recurrent minibatches); every rank runs it,
    # gradients are all-reduced once per optimiser step (one flat bucket), time = max over ranks of the last update
    ppo = None
    pol.attach_env_tail(None)
    if not args.no_ppo and args.env_name != "CrowdSimPredRealGST-v0":
        del env, pol, net
        torch.cuda.empty_cache()
        from crowdnav_prediction_attngraph_amd.trainer import train
        from crowdnav_prediction_attngraph_amd import config as CFG
        over = {"sim.human_num": H}
        if args.env_name == "CrowdSimPred-v0":
            over["sim.predict_method"] = "const_vel"
        tcfg = CFG.Config(**dict(over, **{"humans.end_goal_changing": True})) if args.randomized else CFG.non_randomized(**over)
        # a failure on ONE rank (out of memory, a collective error) must not leave the others waiting in a collective forever: the
        # process group was created with a finite timeout, every rank reports an error flag, and the leg counts only if all succeeded
        err, last = None, None
        try:
            hist, _ = train(env_name=args.env_name, num_processes=E, num_steps=30, num_updates=3, seed=425, config=tcfg, log=None)
            last = hist[-1]
        except Exception as exc:
            err = "%s: %s" % (type(exc).__name__, exc)
        tdev = "cuda" if (dist is None or args.dist_backend == "nccl") else "cpu"
        tt = torch.tensor([last["rollout_s"] if last else 0.0, last["update_s"] if last else 0.0, 1.0 if err else 0.0], device=tdev, dtype=torch.float64)
        rows_local = float(last["live_rows"]) if last else 0.0
        ppo_per_rank = None
        if dist is not None:
            try:
                mine_t = tt.clone()
                allp = [torch.zeros_like(mine_t) for _ in range(world)]
                dist.all_gather(allp, mine_t)             # every rank's own clock, next to the max the headline number uses
                ppo_per_rank = [round(30 * E / max(float(x[0] + x[1]), 1e-9), 1) if float(x[2]) == 0 else None for x in allp]
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            except Exception as exc:
                err = err or "%s: %s" % (type(exc).__name__, exc)
        if err or float(tt[2]) > 0:
            ppo = {"error": err or "the PPO leg failed on another rank"}
        else:
            r_s, u_s = float(tt[0]), float(tt[1])
            ppo = {"samples_per_s": round(30 * E * world / (r_s + u_s), 1), "rollout_s": round(r_s, 5), "update_s": round(u_s, 5),
                   "config": "T=30 steps x %d envs per GPU, ppo_epoch 5, num_mini_batch 2, Adam; 3 updates run, the last one timed" % E,
                   "value_loss": round(last["value_loss"], 6)}
            # roofline of the update on THIS rank (every rank does the same work on its own shard; the time is the max over ranks):
            # algorithmic FLOPs of forward + backward over ppo_epoch passes / update_s, against the bf16x3 MFMA peak
            from crowdnav_prediction_attngraph_amd.trainer import update_flops
            uf = update_flops(rows_local, 30 * E, 5)
            ppo["roofline"] = {"bound": "mfma", "achieved": round(uf / u_s / 1e12, 2), "peak": round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1), "unit": "TFLOP/s",
                               "frac": round(uf / u_s / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4), "live_rows": int(rows_local),
                               "note": "3 (fwd + dX + dW) x ppo_epoch 5 x [live rows x 1.966 MFLOP (the three dense layers of the human-human block) + "
                                       "30 x %d samples x 0.79 MFLOP (robot-node layers)] / update_s; peak = 2500 TFLOP/s dense bf16 / 3 split passes" % E}
            if last.get("allreduce_ms") is not None:   # N > 1: one flat 10 MB gradient all-reduce per optimiser step (this rank's mean)
                ppo["grad_allreduce_ms_per_step"] = round(float(last["allreduce_ms"]), 4)
            if ppo_per_rank is not None:
                ppo["per_rank_samples_per_s"] = ppo_per_rank   # 30 x E / (rollout_s + update_s) on each rank's own clock
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    dropin = None
    if not args.no_dropin and args.env_name != "CrowdSimPredRealGST-v0":
        try:
            dropin = dropin_leg(E, H, args.env_name)
        except Exception as exc:
            dropin = {"error": "%s: %s" % (type(exc).__name__, exc)}
    total_env_steps = E * world * args.steps
    value = total_env_steps / elapsed
    # dominant kernel: timed (a) by HIP event brackets on its stream, one per launch (median over the window; a bracket also contains
    # the dispatch of the launch, and whatever the launch waits for when the host is not ahead), and (b) by the kernel's own stamps of
    # the device clock (first workgroup in -> last wavefront out, every launch).  The roofline uses the MEDIAN event bracket.
    M = (sum(dev_rows) / len(dev_rows)) if dev_rows else prof_n[1] / max(prof_n[0], 1)   # mean live (env, human) rows per launch
    fused = args.gemm == "fused"
    # fused: the timed kernel is the whole human-human block; its algorithmic work = the three dense layers on the live rows
    # (embedding_layer.2 128->512, folded q|k|v 512->1536, folded out_proj∘spatial_linear 512->256); the D->128 input layer and the
    # attention core (<2 % of it) are left out of the count
    qkv_flops = 2.0 * M * (128 * 512 + 512 * 1536 + 512 * 256) if fused else 2.0 * M * 512 * 1536
    mid = lambda v: v[len(v) // 2] if v else 0.0    # noqa: E731  (v sorted)
    qkv_ms = mid(ev_samples) if ev_samples else prof_ms[0] / max(prof_n[0], 1)
    achieved = qkv_flops / (qkv_ms * 1e-3) / 1e12 if qkv_ms > 0 else 0.0
    F = flops_per_env_step(H, D)
    split = args.gemm in ("bf16x3", "fused")
    # the split runs 3 bf16 MFMA passes per algorithmic product, so its attainable algorithmic rate is the bf16 peak / 3
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    team_max = int(os.environ.get("CN_HH_TEAM_MAX", "63"))   # hh_fused.hip: the two-team kernel takes crowds of <= 63 humans (round 6), the wide one 64
    kname = (("hh_fused_wide_kernel" if (H > team_max or os.environ.get("CN_HH_WIDE", "0") not in ("", "0")) else "hh_fused_kernel") + " (v_mfma_f32_16x16x32_bf16, 3 passes hi*hi+hi*lo+lo*hi): embedding -> q|k|v -> attention -> out_proj∘spatial_linear in one launch" if fused else
             "gemm3_nt_kernel<128,NONE> (v_mfma_f32_32x32x16_bf16, 3 passes hi*hi+hi*lo+lo*hi): folded q|k|v projection" if split
             else "gemm_nt_kernel<128,NONE> (v_mfma_f32_32x32x2_f32): folded q|k|v projection")
    # HBM traffic of the dominant kernel: bench.py cannot run rocprofv3 on itself, so the number comes from the committed PMC passes of
    # the same command (tools/profile_step.sh -> tools/mk_traffic.py) -- but only while their stamp (sha256 of csrc/hh_fused.hip at the
    # time of the measurement) matches the kernel source of this build; a stale constant is reported as null with the reason
    traffic = None
    traffic_note = None
    if fused and world == 1 and not args.no_pmc_traffic:
        try:
            child = ["--gpus", "1", "--steps", "10", "--warmup", "10", "--dephase", str(args.dephase), "--envs", str(E), "--humans", str(H), "--env-name", args.env_name,
                     "--no-cpu-baseline", "--no-ppo", "--no-worst-case", "--no-dropin", "--no-pmc-traffic", "--no-other-configs", "--tail", args.tail] + (["--randomized"] if args.randomized else [])
            tl = pmc_traffic_live(child)
            traffic = tl["hbm_bytes_per_launch"]
            rows_c = tl["child_live_rows"] or M
            alg_c = rows_c * (D + 256) * 4 + 3932160
            traffic_note = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child runs of this script on this box, %d "
                            "hh_fused_kernel launches each): FETCH_SIZE %.0f KB x 2 (gfx950 correction) + WRITE_SIZE %.0f KB per launch = %.2fx the algorithmic "
                            "%.1f MB (%d live rows x (%d input + 256 output floats) + the 3.93 MB weight image once)"
                            % (tl["launches"], tl["fetch_size_kb"], tl["write_size_kb"], traffic / alg_c, alg_c / 1e6, rows_c, D))
        except Exception as exc:
            traffic_note = "live PMC measurement failed (%s: %s)" % (type(exc).__name__, str(exc)[:200])
    if traffic is None and fused and (args.env_name, E, H, args.randomized) == ("CrowdSimVarNum-v0", 4096, 20, False):
        # fall back to the committed PMC passes of the same command (tools/profile_step.sh -> tools/mk_traffic.py) -- but only while their stamp
        # (sha256 of csrc/hh_fused.hip at the time of the measurement) matches the kernel source of this build
        import glob
        import hashlib
        src = os.path.join(ROOT, "crowdnav_prediction_attngraph_amd", "csrc", "hh_fused.hip")
        stamp = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] if os.path.exists(src) else None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        tj = json.load(open(cands[-1])) if cands else None
        prev = (traffic_note + "; ") if traffic_note else ""
        if tj is not None and stamp is not None and tj.get("kernel_source_sha16") == stamp:
            traffic = tj["hbm_bytes_per_launch_corrected"]
            traffic_note = prev + ("constant from the committed passes of the same kernel source (%s, stamp %s; %.2fx the algorithmic bytes)"
                                   % (os.path.basename(cands[-1]), stamp, tj["ratio"]))
        else:
            traffic_note = prev + ("null: the newest committed PMC result (%s) was measured on a different hh_fused.hip (stamp %s, this build %s)"
                                   % (os.path.basename(cands[-1]) if cands else "none", tj.get("kernel_source_sha16") if tj else None, stamp))
    line = {
        "metric": "env-steps/sec (sim+policy fwd) at %d humans" % H, "value": round(value, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if not split else "f32 (big GEMMs as bf16x3 split-precision MFMA, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "%s%s, %d humans, %d parallel envs per GPU, HH+HR attention on, "
                               "policy forward + ORCA sim step + auto-reset per step" % (
                                   "BASELINE configs[1]: " if (args.env_name, H, E, args.randomized) == ("CrowdSimVarNum-v0", 20, 4096, False) else ("randomized humans, " if args.randomized else ""), args.env_name, H, E),
                   "envs_per_gpu": E, "humans": H, "parallelism": "dp%d (envs sharded, no rollout collective)" % world,
                   "policy_init": "orthogonal, torch.manual_seed(425)", "sampled_actions": True, "side_stream_tail": args.tail},
        "roofline": {"bound": "mfma", "kernel": "%s, M=%d live rows of %d%s" % (kname, M, E * H, "" if fused else ", N=1536 K=512"),
                     "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "note": ("achieved = algorithmic 2*M*(128*512+512*1536+512*256) / launch time (hipEvents on the kernel's stream); peak = 2500 TFLOP/s dense bf16 "
                              "MFMA / 3 passes of the hi/lo split; executed bf16 MFMA rate = %.1f TFLOP/s" % (3 * achieved)) if fused else
                             ("achieved = algorithmic 2*M*N*K / launch time (hipEvents on the kernel's stream); peak = 2500 TFLOP/s dense bf16 "
                              "MFMA / 3 passes of the hi/lo split; executed bf16 MFMA rate = %.1f TFLOP/s" % (3 * achieved)) if split else
                             "achieved = algorithmic 2*M*N*K / launch time on exact fp32 MFMA",
                     "traffic_note": traffic_note, "launch_ms": round(qkv_ms, 4), "launches": len(ev_samples), "mean_detected_humans": round(M / E, 3),
                     "launch_ms_events": {"median": round(mid(ev_samples), 4), "min": round(ev_samples[0], 4) if ev_samples else None,
                                          "max": round(ev_samples[-1], 4) if ev_samples else None, "mean": round(sum(ev_samples) / max(len(ev_samples), 1), 4),
                                          "samples": len(ev_samples), "every": ev_every,
                                          "what": "hipEventElapsedTime of a (record, launch, record) bracket on the kernel's stream, per launch of the timed window"},
                     "launch_ms_device": {"median": round(mid(dev_hh), 4), "min": round(dev_hh[0], 4) if dev_hh else None,
                                          "max": round(dev_hh[-1], 4) if dev_hh else None, "samples": len(dev_hh),
                                          "frac_at_median": round(qkv_flops / (mid(dev_hh) * 1e-3) / 1e12 / peak, 4) if dev_hh and fused else None,
                                          "what": "the kernel's own stamps of the 100 MHz device clock: first workgroup in -> last wavefront out, every launch of the timed window"},
                     "rn_fused_launch_ms_device": {"median": round(mid(dev_rn), 4), "min": round(dev_rn[0], 4) if dev_rn else None,
                                                   "max": round(dev_rn[-1], 4) if dev_rn else None, "samples": len(dev_rn)},
                     "whole_step": {"reference_graph_flops_per_env_step": F,
                                    "reference_graph_tflops_equivalent": round(value / world * F / 1e12, 2),
                                    "note": "env-steps/s x the FLOPs of the reference's dense, unfolded forward; NOT a hardware utilisation "
                                            "(padded humans are not computed and affine pairs are folded)"}},
    }
    line["config"]["dephase_steps"] = args.dephase
    if fused:
        rl = line["roofline"]
        step_s = elapsed / args.steps
        rn_flops = 0.79e6 * E                                  # robot-node kernel: 0.79 MFLOP per env
        rl["frac_of_dense_bf16_algorithmic"] = round(achieved / PEAK_BF16_MFMA_TFLOPS, 4)
        rl["whole_step_frac"] = round((qkv_flops + rn_flops) / step_s / 1e12 / peak, 4)
        rl["fractions_note"] = ("frac = dominant kernel against the three-pass ceiling (dense bf16 / 3); frac_of_dense_bf16_algorithmic = the same algorithmic rate "
                                "against the dense bf16 peak itself (%.0f TFLOP/s); whole_step_frac = (human-human + robot-node FLOPs of a step) / ms_per_step "
                                "against the three-pass ceiling: the simulator kernels and the launch gaps of the step count as time, not as work" % PEAK_BF16_MFMA_TFLOPS)
        if plan_hdr is not None and plan_hdr[0] == 0x52504C4E and qkv_ms > 0:
            tiles = int(plan_hdr[6])
            rl["l2_weight_stream_TBps"] = round(tiles * 3932160 / (qkv_ms * 1e-3) / 1e12, 2)
            rl["l2_weight_stream_note"] = "%d tiles of the last planned launch x the 3.93 MB weight image each tile streams L2 -> registers / launch time" % tiles
        if decomp is not None:
            rl["step_decomposition_us"] = dict(decomp.get("median_us", {}), **{"gap_" + k: v for k, v in decomp.get("median_gap_us", {}).items()})
            if "median_step_us" in decomp:
                rl["step_decomposition_us"]["step"] = decomp["median_step_us"]
    if world > 1:
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ver = None
        line["collectives"] = {"backend": args.dist_backend, "rccl_ranks": world if args.dist_backend == "nccl" else 0, "rccl_version": ver,
                               "rollout": "none (envs sharded by global index)", "update": "one 3-double all-reduce + ONE flat fp32 gradient all-reduce per optimiser step",
                               "grad_allreduce_ms_per_step": (ppo or {}).get("grad_allreduce_ms_per_step"), "rank_devices": rank_devices}
        ids = [(d_.get("uuid") or d_.get("pci_bus_id") or d_.get("device_index")) for d_ in (rank_devices or [])]
        line["collectives"]["self_check"] = {
            "rccl_ranks_equal_n_gpus": line["collectives"]["rccl_ranks"] == world,
            "one_distinct_device_per_rank": len(set(ids)) == world,
            "every_rank_reported": per_rank is not None and len(per_rank) == world and all(x > 0 for x in per_rank),
            "gradient_allreduce_timed": (ppo or {}).get("grad_allreduce_ms_per_step") is not None or args.no_ppo,
            "value_is_sum_of_ranks_over_slowest_clock": True,
            "note": "weak scaling: every rank owns --envs envs (global env indices rank * E ..); value = E x N x steps / max over ranks of the timed window; "
                    "no scaling efficiency is claimed in this line -- the driver derives it from the per-N values"}
    line["host_enqueue_ms_per_step"] = round(t_enq / args.steps * 1e3, 4)   # Python + launch cost of one step; the device needs ms_per_step
    if step_intervals:
        si = sorted(step_intervals)
        line["device_step_interval_us"] = {"median": si[len(si) // 2], "min": si[0], "max": si[-1],
                                           "first": step_intervals[:4], "what": "start-to-start of consecutive hh_fused launches in the timed window (device clock)"}
    if decomp is not None:
        line["step_decomposition"] = decomp
    if dropin is not None:
        if "env_steps_per_s" in dropin:
            dropin["fraction_of_zero_sync_path"] = round(dropin["env_steps_per_s"] / (value / world), 3)
        line["dropin_train_loop"] = dropin
    if per_rank is not None:
        line["per_rank_env_steps_per_s"] = per_rank     # each rank's own clock over the same K steps (value uses the slowest)
    if worst is not None:
        line["worst_case_all_detected"] = worst
    if ppo is not None:
        line["ppo"] = ppo
    if not args.no_other_configs and world == 1 and (args.env_name, H, E, args.randomized) == ("CrowdSimVarNum-v0", 20, 4096, False):
        line["other_baseline_configs_1gpu"] = other_configs_leg()
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline_threads(H, args.env_name, args.cpu_threads)
        line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        # SURVEY 8d CPU baseline (ii) / BASELINE.md 4.1: the REAL reference Python, timed in the build container (it cannot travel to this
        # box) by tools/ref_python_cpu_baseline.py -- a labelled constant read from the committed result file, not a measurement of this run
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_python_cpu.json")))
            rj = json.load(open(cands[-1]))
            rec = [r for r in rj["env_step"] if r["env"] == args.env_name and r["humans"] == H]
            if rec:
                r0 = rec[0]
                cores_here = line["cpu_baseline"]["cores"]
                line["cpu_baseline"]["reference_python_container"] = {
                    "env_steps_per_s_single_process": r0["single_process_steps_per_s"], "env_steps_per_s_%d_workers" % r0["workers"]: r0["workers_aggregate_steps_per_s"],
                    "env_steps_per_s_per_worker": r0["per_worker_steps_per_s"], "container_logical_cpus": rj["host"]["logical_cpus"],
                    "policy_act_cpu_1thread_batch16_env_steps_per_s": rj["policy_act_cpu_1thread_batch16"]["env_steps_per_s"],
                    "what": "env.step of the reference's own Python (rvo2 = shim over the oracle's C RVO2), NOT measured in this run: constant from %s" % os.path.basename(cands[-1])}
                line["speedup_vs_reference"] = {
                    "vs_reference_python_scaled_to_this_box": round(value / (r0["per_worker_steps_per_s"] * cores_here), 1),
                    "scaling": "%.1f env-steps/s per worker process x %d physical cores of this box (env.step only: the reference adds its policy forward on top)" % (r0["per_worker_steps_per_s"], cores_here),
                    "vs_published_training_fps_176.5": round(value / 176.5, 1),
                    "note": "north_star target: >= 50x; published fps = rollout + policy + PPO update, 16 worker processes, unstated hardware (BASELINE.md 1)"}
        except Exception as exc:   # a missing / malformed constant must not cost the line
            line["cpu_baseline"]["reference_python_container"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
