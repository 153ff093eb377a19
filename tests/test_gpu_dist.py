"""Two ranks on ONE MI355X (gloo process group, CUDA tensors): the branch of PPO.update() that RCCL ranks execute.

* flat gradient bucket -> ONE all-reduce per optimiser step -> cn_adam_clip_step(grad_scale = 1 / world) on every rank
  (ppo.py, `on_gpu` branch), global advantage statistics through cn_adv_stats partial sums + all-reduce;
  result == the single-process update over the union of the two env shards, and both ranks hold identical weights;
* env shards through make_vec_envs: rank r owns global env indices [r * E, (r + 1) * E) (seed + global index), so the union
  of the shards reproduces the single-process batch bit for bit.

The reference is single process (train.py:112 discards DataParallel); BASELINE north_star asks for env shards + one
gradient all-reduce.  gloo stands in for RCCL because both ranks share one device here; the code path above the
collective is the same.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

H, D, T = 5, 2, 6
E_TOTAL = 8


def _build(lo, hi, dev):
    """Seeded policy on the GPU + rollout storage holding envs [lo, hi) of a deterministic synthetic rollout of E_TOTAL envs."""
    sys.path.insert(0, ROOT)
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from crowdnav_prediction_attngraph_amd.storage import RolloutStorage
    from tests import policy_util as PU
    ob_space, act_space = make_spaces(H, D)
    torch.manual_seed(7)
    E = hi - lo
    pol = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn",
                 base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=E, num_mini_batch=1, seq_length=T)).to(dev)
    ro = RolloutStorage(T, E, ob_space.spaces, act_space, 128, 256)
    rs = np.random.RandomState(3)
    full = dict(rewards=rs.uniform(-1, 1, (T, E_TOTAL, 1)), values=rs.uniform(-1, 1, (T + 1, E_TOTAL, 1)),
                returns=rs.uniform(-1, 1, (T + 1, E_TOTAL, 1)), logp=rs.uniform(-3, -1, (T, E_TOTAL, 1)),
                actions=rs.uniform(-1, 1, (T, E_TOTAL, 2)), masks=(rs.uniform(size=(T + 1, E_TOTAL, 1)) > 0.2).astype(np.float64),
                hx=rs.uniform(-1, 1, (E_TOTAL, 1, 128)))
    obs = [PU.synth_obs(E_TOTAL, H, D, seed=50 + s) for s in range(T + 1)]
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, lo:hi]).astype(np.float32))  # noqa: E731
    for s in range(T + 1):
        for k, v in obs[s].items():
            ro.obs[k][s].copy_(torch.from_numpy(v[lo:hi]))
    ro.rewards.copy_(f32(full["rewards"])); ro.value_preds.copy_(f32(full["values"])); ro.returns.copy_(f32(full["returns"]))
    ro.action_log_probs.copy_(f32(full["logp"])); ro.actions.copy_(f32(full["actions"])); ro.masks.copy_(f32(full["masks"]))
    ro.recurrent_hidden_states["human_node_rnn"][0].copy_(torch.from_numpy(full["hx"][lo:hi].astype(np.float32)))
    ro.to(dev)
    return pol, ro


def _update(pol, ro):
    from crowdnav_prediction_attngraph_amd.ppo import PPO
    agent = PPO(pol, 0.2, 2, 1, 0.5, 0.0, lr=1e-3, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(11)
    losses = agent.update(ro)
    assert agent._flat is not None, "the GPU update must run on the flat buckets"
    flat = agent._flat["p"].detach().clone()
    return losses, flat, agent


def _env_trace(E, dev, steps=12):
    """A short rollout of this process's env shard under fixed actions (a function of the GLOBAL env index)."""
    from crowdnav_prediction_attngraph_amd import config as CFG
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    envs = make_vec_envs("CrowdSimVarNum-v0", 425, E, 0.99, None, dev, False, config=CFG.non_randomized(**{"sim.human_num": H}))
    obs = envs.reset()
    gidx = torch.arange(rank * E, (rank + 1) * E, device=dev, dtype=torch.float32)
    tr = [torch.cat([obs["robot_node"].reshape(E, -1), obs["spatial_edges"].reshape(E, -1)], 1).clone()]
    for s in range(steps):
        a = torch.stack([torch.cos(0.3 * gidx + 0.1 * s), torch.sin(0.2 * gidx - 0.05 * s)], 1) * 0.7
        obs, rew, done, infos = envs.step(a)
        tr.append(torch.cat([obs["robot_node"].reshape(E, -1), obs["spatial_edges"].reshape(E, -1), rew.to(dev).reshape(E, 1)], 1).clone())
    envs.close()
    return [t.cpu() for t in tr]


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = E_TOTAL // world
    pol, ro = _build(rank * per, (rank + 1) * per, dev)
    losses, flat, agent = _update(pol, ro)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    trace = _env_trace(per, dev)
    torch.save(dict(losses=losses, flat=flat.cpu(), same=all(torch.equal(gathered[0], g) for g in gathered), trace=trace,
                    allreduce_ms=agent.last_allreduce_ms), out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_bucket_update_and_env_shards_equal_the_single_process_union(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(2)]
    assert got[0]["same"], "ranks diverged after the all-reduced update"
    assert got[0]["allreduce_ms"] is not None and got[0]["allreduce_ms"] > 0, "the flat-bucket branch must have all-reduced the gradients"
    dev = torch.device("cuda", 0)
    pol, ro = _build(0, E_TOTAL, dev)
    losses, flat, _ = _update(pol, ro)
    # the three reported losses are averages over ranks of per-rank minibatch means == the union's mean (equal shard sizes)
    np.testing.assert_allclose(got[0]["losses"], losses, rtol=2e-5, atol=2e-6)
    # same weights as one process that saw all 8 envs: the summation order of the weight-gradient products differs (two partial
    # products + all-reduce vs one product), hence the tolerance; lr 1e-3 x 2 steps
    np.testing.assert_allclose(got[0]["flat"].numpy(), flat.cpu().numpy(), rtol=0, atol=5e-6)
    # env shards: rank r's envs are the union's envs r*4 .. r*4+3, bit for bit
    per = E_TOTAL // 2
    import torch.distributed as dist
    assert not dist.is_initialized()
    union = _env_trace(E_TOTAL, dev)
    for step, u in enumerate(union):
        for r in range(2):
            assert torch.equal(got[r]["trace"][step], u[r * per:(r + 1) * per]), "env shard %d differs from the union at step %d" % (r, step)


def _rccl_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CN_FORCE_DIST="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # "nccl" IS RCCL on ROCm
    assert dist.get_backend() == "nccl"
    pol, ro = _build(0, E_TOTAL, dev)
    losses, flat, agent = _update(pol, ro)
    # the three collectives of the data-parallel design on their real dtypes / sizes, on the compute stream:
    bucket = torch.arange(2_501_255, device=dev, dtype=torch.float32) * 1e-6      # the 10 MB flat gradient bucket
    want = bucket.clone()
    dist.all_reduce(bucket)
    stats = torch.tensor([1.5, 2.25, 1228800.0], device=dev, dtype=torch.float64)  # (sum a, sum a^2, n) of the advantage statistics
    dist.all_reduce(stats)
    from crowdnav_prediction_attngraph_amd.trainer import EpisodeStats
    es = EpisodeStats(dev)
    es.acc += 1.0
    popped = es.pop()
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, flat=flat.cpu(), allreduce_ms=agent.last_allreduce_ms, bucket_ok=bool(torch.equal(bucket, want)),
                    stats=stats.cpu(), episodes=popped["episodes"]), out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_single_rank_runs_the_collectives_of_the_update(tmp_path):
    """RCCL itself, on the one GPU a box has: the nccl backend is initialised with world_size 1 and PPO.update() takes its collective
    branch (CN_FORCE_DIST=1): advantage statistics all-reduce (3 doubles), ONE flat-bucket gradient all-reduce per optimiser step on the
    compute stream, the loss all-reduce -- plus the 10 MB bucket and the episode statistics stand-alone.  With one rank every all-reduce is
    the identity and grad_scale = 1, so the result must equal the plain single-process update bit for bit; what this run adds is that RCCL
    initialises, accepts the dtypes / sizes / stream the design uses, and that its ordering against the HIP kernels around it holds."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(1, port, out), nprocs=1, join=True)
    got = torch.load(out, weights_only=False)
    assert got["allreduce_ms"] is not None and got["allreduce_ms"] > 0, "update() must have gone through the gradient all-reduce"
    assert got["bucket_ok"] and got["episodes"] == 1
    np.testing.assert_array_equal(got["stats"].numpy(), np.array([1.5, 2.25, 1228800.0]))
    dev = torch.device("cuda", 0)
    pol, ro = _build(0, E_TOTAL, dev)
    losses, flat, _ = _update(pol, ro)
    np.testing.assert_array_equal(got["flat"].numpy(), flat.cpu().numpy())
    np.testing.assert_array_equal(np.asarray(got["losses"]), np.asarray(losses))


def test_bench_under_the_distributed_launcher_reproduces_the_plain_run():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...` (how the driver's scaling sweep starts its N = 1 point) against
    the plain `python bench.py --gpus 1 ...` on the same box: the same line shape and the same throughput within the box's run-to-run noise, so
    that the N = 1 value of a scaling curve is the BENCH value.  And `--gpus 2 --same-gpu` over gloo carries the self-check block (plumbing:
    two ranks time-slice the one GPU, which the block reports as NOT one distinct device per rank)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--steps", "40", "--warmup", "10", "--dephase", "60", "--no-ppo", "--no-cpu-baseline", "--no-dropin", "--no-pmc-traffic", "--no-other-configs",
             "--no-worst-case"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=root)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert r.returncode == 0 and lines, (r.returncode, r.stderr[-600:])
        return json.loads(lines[-1])

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    plain = run([sys.executable, "bench.py", "--gpus", "1"] + flags)
    launched = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                    "bench.py", "--gpus", "1"] + flags)
    assert launched["n_gpus"] == plain["n_gpus"] == 1 and launched["metric"] == plain["metric"] and launched["config"]["workload"] == plain["config"]["workload"]
    assert abs(launched["value"] - plain["value"]) <= 0.12 * plain["value"], (launched["value"], plain["value"])
    two = run([sys.executable, "bench.py", "--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--envs", "512"] + flags)
    c = two["collectives"]
    assert two["n_gpus"] == 2 and len(c["rank_devices"]) == 2 and [d["rank"] for d in c["rank_devices"]] == [0, 1]
    sc = c["self_check"]
    assert sc["every_rank_reported"] and not sc["one_distinct_device_per_rank"] and not sc["rccl_ranks_equal_n_gpus"]     # gloo on one GPU: says so
    assert len(two["per_rank_env_steps_per_s"]) == 2
