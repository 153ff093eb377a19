"""PPO training loop over the device-resident simulator -- the flow of the reference's train.py:140-242 with the
per-step host round trips removed.

`collect_rollout` is the fused rollout: for each of the T steps the policy forward writes value / action / log-prob /
next hidden state straight into the RolloutStorage rows and the simulator writes the next observation, reward and
done mask straight into row t+1 -- no clones, no host synchronisation, no Python work proportional to E.
(The reference-compatible `envs.step()` path, which returns CPU rewards / numpy dones / info dicts, still exists for
an unmodified train.py; it costs one D2H sync and O(E) Python per step.)
"""
import time

import torch

from .policy import Policy
from .ppo import PPO
from .storage import RolloutStorage
from .vec_env import make_vec_envs


class EpisodeStats:
    """Device-side bench.Monitor aggregate: sum of finished-episode returns / lengths / outcome counts."""

    def __init__(self, device):
        self.acc = torch.zeros(8, dtype=torch.float64, device=device)  # n_done, sum_ret, sum_len, timeout, collision, goal, -, -

    def update(self, done, info, ep_return, ep_len):
        if (done.is_cuda and done.dtype == torch.uint8 and info.dtype == torch.uint8 and ep_return.dtype == torch.float64 and ep_len.dtype == torch.int32
                and all(t.is_contiguous() for t in (done, info, ep_return, ep_len))):
            from . import _abi as A      # the step outputs of HipEnvBatch as they are: ONE launch, fixed summation order
            A.check(A.lib().cn_episode_stats_update(done.numel(), A.ptr(done), A.ptr(info), A.ptr(ep_return), A.ptr(ep_len), A.ptr(self.acc), A.stream_ptr()),
                    "cn_episode_stats_update")
            return
        d = done.to(torch.float64)
        self.acc[0] += d.sum()
        self.acc[1] += (ep_return * d).sum()
        self.acc[2] += (ep_len.to(torch.float64) * d).sum()
        for code in (1, 2, 3):
            self.acc[2 + code] += ((info == code).to(torch.float64) * d).sum()

    def pop(self):
        """Aggregate since the last pop, over ALL ranks when torch.distributed is initialised (each rank steps its own shard of envs)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            acc = self.acc if dist.get_backend() == "nccl" else self.acc.cpu()
            dist.all_reduce(acc)
            if acc is not self.acc:
                self.acc.copy_(acc)
        a = self.acc.cpu().tolist()
        self.acc.zero_()
        n = max(a[0], 1.0)
        return dict(episodes=int(a[0]), eprewmean=a[1] / n, eplenmean=a[2] / n, timeout=a[3] / n, collision=a[4] / n, success=a[5] / n)


def collect_rollout(envs, actor_critic, rollouts, stats=None, generator=None):
    """Fill `rollouts` (row 0 must hold the current observation / hidden state / mask).  Zero host syncs."""
    env = envs._env
    E, dev = envs.num_envs, envs.device
    pol = actor_critic._hip_policy(E, dev)
    T = rollouts.num_steps
    hx = rollouts.recurrent_hidden_states["human_node_rnn"]
    eps = torch.empty(T, E, 2, device=dev).normal_(generator=generator)   # the action noise of the whole rollout in one launch
    # the simulator writes a row plan beside every observation (hip.HipEnvBatch.row_plan): row t of the storage IS the newest observation of
    # `env` at every t of this loop (row 0: the last one of the previous rollout, or the reset), so the plan in the buffer is the one made
    # for it.  Not through the GST wrapper, which post-processes the observation.
    base = actor_critic.base
    plan = env.row_plan if (envs._pretext is None and base.use_self_attn and base.sort_humans) else None
    for t in range(T):
        obs_t = {k: rollouts.obs[k][t] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
        out = dict(value=rollouts.value_preds[t], action=rollouts.actions[t], logp=rollouts.action_log_probs[t], hxs=hx[t + 1])
        pol_obs = obs_t
        if not base.sort_humans:   # args.sort_humans = False: attention masked by visible_masks -> visible humans first + their count
            pol_obs = base.counted_inputs(dict(obs_t, visible_masks=rollouts.obs["visible_masks"][t]))
        pol.act(pol_obs, hx[t], rollouts.masks[t], eps=eps[t], out=out, row_plan=plan)
        obs_n = {k: rollouts.obs[k][t + 1] for k in obs_t}
        obs_n["visible_masks"] = None
        if "visible_masks" in rollouts.obs:
            obs_n["visible_masks"] = rollouts.obs["visible_masks"][t + 1].view(torch.uint8)
        if envs._pretext is None:
            _, reward, done, info, ep_ret, ep_len = env.step(rollouts.actions[t], obs=obs_n, not_done=rollouts.masks[t + 1],
                                                             reward=rollouts.rewards[t])   # rows written in place
        else:
            # GST wrapper in the loop: predictions / sort / social penalty post-process the raw observation
            o, reward, done, info, ep_ret, ep_len = envs.step_device(rollouts.actions[t])
            for k in obs_t:
                rollouts.obs[k][t + 1].copy_(o[k].view_as(rollouts.obs[k][t + 1]))
            if "visible_masks" in rollouts.obs:
                rollouts.obs["visible_masks"][t + 1].copy_(o["visible_masks"].to(torch.bool))
        if envs._pretext is not None:
            rollouts.rewards[t].copy_(reward.view(E, 1))
            rollouts.masks[t + 1].copy_((done == 0).view(E, 1))
        if stats is not None:
            stats.update(done, info, ep_ret, ep_len)
    rollouts.step = 0


def bootstrap_value(actor_critic, rollouts):
    obs = {k: rollouts.obs[k][-1] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num", "visible_masks") if k in rollouts.obs}
    hxs = {"human_node_rnn": rollouts.recurrent_hidden_states["human_node_rnn"][-1]}
    return actor_critic.get_value(obs, hxs, rollouts.masks[-1])


def save_checkpoint(save_dir, update, actor_critic, agent, envs, rollouts, stats, device):
    """`<save_dir>/checkpoints/<update:05d>.pt` is the policy state_dict exactly as train.py:213-219 writes it (loadable by the
    reference's test.py and by --resume); `<update:05d>.resume.pt` beside it holds what a BIT-EXACT continuation needs on top:
    optimiser state, both torch RNG streams, the simulator snapshot (cn_env_save), row 0 of the rollout storage and the episode
    statistics accumulator.  One pair per rank under torch.distributed (`.rankN` suffix) since env shards and noise streams differ."""
    import os
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    d = os.path.join(save_dir, "checkpoints")
    os.makedirs(d, exist_ok=True)
    stem = os.path.join(d, "%.5i" % update)
    if rank == 0:
        torch.save(actor_critic.state_dict(), stem + ".pt")
    agent._sync_optimizer_state()
    ck = {"format": 1, "update": update, "optimizer": agent.optimizer.state_dict(), "rng_cpu": torch.get_rng_state(),
          "rng_cuda": torch.cuda.get_rng_state(device), "env": envs.state_dict(),
          "rollout0": {"obs": {k: v[0].clone() for k, v in rollouts.obs.items()},
                       "hxs": rollouts.recurrent_hidden_states["human_node_rnn"][0].clone(), "masks": rollouts.masks[0].clone(),
                       "bad_masks": rollouts.bad_masks[0].clone(),
                       # derived data of that observation (hip.HipEnvBatch.row_plan): with it the first forward after a resume walks the
                       # same tiles, i.e. sums in the same order, as the uninterrupted run
                       "row_plan": envs._env.row_plan.clone()},
          "stats": stats.acc.clone()}
    path = stem + (".resume.pt" if rank == 0 else ".resume.rank%d.pt" % rank)
    torch.save(ck, path)
    return stem + ".pt", path


def load_checkpoint(policy_path, resume_path, actor_critic, agent, envs, rollouts, stats, device):
    """Inverse of save_checkpoint; returns the index of the next update."""
    actor_critic.load_state_dict(torch.load(policy_path, map_location=device))          # train.py:105-108
    ck = torch.load(resume_path, map_location="cpu")
    agent.optimizer.load_state_dict(ck["optimizer"])       # PPO re-binds its flat buckets to the restored moments at the next update()
    envs.load_state_dict(ck["env"])
    r0 = ck["rollout0"]
    for k in rollouts.obs:
        rollouts.obs[k][0].copy_(r0["obs"][k])
    rollouts.recurrent_hidden_states["human_node_rnn"][0].copy_(r0["hxs"])
    rollouts.masks[0].copy_(r0["masks"]); rollouts.bad_masks[0].copy_(r0["bad_masks"])
    if "row_plan" in r0 and r0["row_plan"].numel() == envs._env.row_plan.numel():   # (load_state_dict above cleared it)
        envs._env.row_plan.copy_(r0["row_plan"])
    stats.acc.copy_(ck["stats"])
    torch.set_rng_state(ck["rng_cpu"])
    torch.cuda.set_rng_state(ck["rng_cuda"], device)
    return int(ck["update"]) + 1


def update_flops(live_rows, samples, ppo_epoch):
    """Algorithmic FLOPs of one PPO.update(): ppo_epoch passes over all samples, forward + backward (dX and dW: 2 x forward) of
      * the three dense layers of the human-human block on the live rows (embedding_layer.2 128->512, folded q|k|v 512->1536, folded
        out_proj∘spatial_linear 512->256: 1.966 MFLOP per row, the figure bench.py prices the rollout kernel with), and
      * the per-sample robot-node layers (robot_linear 9->256, [u | enc] 256->320, edge embed 256->64, GRU 2 x 128->384, folded
        output_linear∘(actor.0 | critic.0) 128->512, actor.2 / critic.2 256->256, heads): 0.79 MFLOP per sample.
    The attention cores, the D->128 input layer and all pointwise work are left out (< 3 %)."""
    hh = 2.0 * (128 * 512 + 512 * 1536 + 512 * 256)
    rn = 2.0 * (9 * 256 + 256 * 320 + 256 * 64 + 2 * 128 * 384 + 128 * 512 + 2 * 256 * 256 + 256 + 512)
    return 3.0 * ppo_epoch * (live_rows * hh + samples * rn)


def train(env_name="CrowdSimVarNum-v0", num_processes=4096, num_steps=30, num_updates=10, seed=425, config=None, ppo_epoch=5,
          num_mini_batch=2, lr=4e-5, eps=1e-5, clip_param=0.2, value_loss_coef=0.5, entropy_coef=0.0, max_grad_norm=0.5, gamma=0.99,
          gae_lambda=0.95, log=print, device=None, save_dir=None, save_interval=0, resume=None, use_self_attn=True, sort_humans=None):
    """Returns a list of per-update dicts (losses, timings, episode stats).  Works single- or multi-GPU (one process per
    GPU, torch.distributed initialised by the caller).  save_dir / save_interval: write checkpoints like train.py:213-219 (every
    `save_interval` updates and after the last one); resume = path of a `NNNNN.pt` written by this function: continue that run
    (updates NNNNN+1 .. num_updates-1) with results bit-identical to the uninterrupted run."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(seed)
    envs = make_vec_envs(env_name, seed, num_processes, gamma, None, device, False, config=config, phase="train")
    # arguments.py:189, :206: use_self_attn, and sort_humans -- by default whatever the config's args say (it also decides the simulator's row order)
    if sort_humans is None:
        sort_humans = bool(getattr(getattr(config, "args", None), "sort_humans", True))
    # the simulator's row order follows config.args.sort_humans (config.to_env_config): the policy must read the observation the same way, and the
    # unsorted mode masks by `visible_masks`, which CrowdSimPred-v0's observation does not have (crowd_sim_pred.py:36-48) -- refuse both here,
    # before anything is allocated, rather than with a KeyError in the middle of the first rollout
    cfg_sort = bool(getattr(getattr(config, "args", None), "sort_humans", True))
    if bool(sort_humans) != cfg_sort:
        raise ValueError("train(sort_humans=%s) disagrees with config.args.sort_humans=%s, which decides the simulator's row order" % (sort_humans, cfg_sort))
    if not sort_humans and "visible_masks" not in envs.observation_space.spaces:
        raise ValueError("sort_humans=False needs the `visible_masks` observation, which %s does not provide" % env_name)
    base_kwargs = dict(env_name=env_name, num_processes=num_processes, num_mini_batch=num_mini_batch, seq_length=num_steps, use_self_attn=use_self_attn,
                       sort_humans=sort_humans)
    actor_critic = Policy(envs.observation_space.spaces, envs.action_space, base_kwargs=base_kwargs, base="selfAttn_merge_srnn").to(device)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        # identical weights on every rank (seeded above), but each env shard explores with its own action noise
        torch.cuda.manual_seed(seed + 1000003 * torch.distributed.get_rank())
    rollouts = RolloutStorage(num_steps, num_processes, envs.observation_space.spaces, envs.action_space, 128, 256)
    rollouts.to(device)
    agent = PPO(actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr=lr, eps=eps, max_grad_norm=max_grad_norm)
    obs = envs.reset_device()
    for k in rollouts.obs:
        rollouts.obs[k][0].copy_(obs[k].view_as(rollouts.obs[k][0]) if k != "visible_masks" else obs[k].to(torch.bool))
    stats = EpisodeStats(device)
    history = []
    first = 0
    if resume is not None:
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        side = resume[:-3] + (".resume.pt" if rank == 0 else ".resume.rank%d.pt" % rank)
        first = load_checkpoint(resume, side, actor_critic, agent, envs, rollouts, stats, device)
    for j in range(first, num_updates):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        collect_rollout(envs, actor_critic, rollouts, stats)
        next_value = bootstrap_value(actor_critic, rollouts)
        rollouts.compute_returns(next_value, True, gamma, gae_lambda, False)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        # live (env, human) rows of the T x E samples the update trains on: the human-human block runs on these (see update_flops)
        live_rows = rollouts.obs["detected_human_num"][:num_steps].clamp(1, envs.human_num).sum()
        value_loss, action_loss, dist_entropy = agent.update(rollouts)
        rollouts.after_update()
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        rec = dict(update=j, value_loss=value_loss, action_loss=action_loss, entropy=dist_entropy, rollout_s=t1 - t0, update_s=t2 - t1,
                   allreduce_ms=getattr(agent, "last_allreduce_ms", None), live_rows=int(live_rows.item()),
                   samples_per_s=num_steps * num_processes / (t2 - t0), **stats.pop())
        history.append(rec)
        if log:
            log(rec)
        if save_dir is not None and ((save_interval and j % save_interval == 0) or j == num_updates - 1):
            save_checkpoint(save_dir, j, actor_critic, agent, envs, rollouts, stats, device)
    envs.close()
    return history, actor_critic
