"""PPO -- mirror of rl/ppo/ppo.py:6-101 (same constructor, `update(rollouts) -> (value_loss, action_loss, entropy)`).

Data-parallel extension (new, the reference is single process): when torch.distributed is initialised with world > 1
every rank owns a disjoint shard of envs and
  * the advantage statistics (sum, sum of squares, count) are all-reduced once per update() so mean / unbiased std are
    global (ppo.py:38-39 semantics over all T x E_total samples);
  * per optimiser step ONE all-reduce of a single flat fp32 gradient bucket (all parameters, ~10 MB) over RCCL/xGMI,
    averaged, then the grad-norm clip and Adam run identically on every rank.
"""
import torch
import torch.nn as nn
import torch.optim as optim

from . import hip


def _dist():
    """torch.distributed when this process is one of several ranks.  CN_FORCE_DIST=1 (test aid) also takes the collective branch with a
    single rank: tests/test_gpu_dist.py runs the RCCL all-reduces of update() that way on a box with one GPU."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("CN_FORCE_DIST") == "1"):
        return dist
    return None


class PPO():
    def __init__(self, actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr=None, eps=None,
                 max_grad_norm=None, use_clipped_value_loss=True):
        self.actor_critic = actor_critic
        self.clip_param = clip_param
        self.ppo_epoch = ppo_epoch
        self.num_mini_batch = num_mini_batch
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.optimizer = optim.Adam(actor_critic.parameters(), lr=lr, eps=eps)
        self._flat = None
        self._step = 0

    # ---- flat buckets (GPU path) -------------------------------------------------------------------------------------
    # ONE fp32 buffer each for the parameters, their gradients and the two Adam moments; every p.data / p.grad /
    # optimizer.state[p]['exp_avg'|'exp_avg_sq'] is a view into them.  The gradient all-reduce then needs no packing and the
    # clip + Adam step is a single pair of launches (cn_adam_clip_step).  `self.optimizer` stays the owner of the optimiser
    # state: its state_dict() / load_state_dict() / param_groups (learning-rate schedule of train.py:148-152) keep working.
    def _params(self):
        return [p for p in self.actor_critic.parameters() if p.requires_grad]

    def _bound(self, params):
        if self._flat is None or self._flat["p"].device != params[0].device or len(params) != len(self._flat["views"]):
            return False
        st = self.optimizer.state
        for p, (pv, gv, mv, vv) in zip(params, self._flat["views"]):
            if p.data_ptr() != pv.data_ptr() or p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                return False
            s = st.get(p)
            if not s or s["exp_avg"].data_ptr() != mv.data_ptr() or s["exp_avg_sq"].data_ptr() != vv.data_ptr():
                return False
        return True

    def _bind_flat(self):
        params = self._params()
        if self._bound(params):
            return
        dev = params[0].device
        # every parameter starts on a 16-byte boundary of the bucket: its .data pointer is handed to kernels that load rows as float4
        # (cn_split_bf16, cn_embed0_*, cn_gru_seq_*, bias vectors ...); the 1- and 2-element tensors (critic_linear.bias, fc_mean.bias,
        # logstd) would otherwise leave everything behind them 4-byte aligned.  The padding stays zero in all four buckets: no
        # gradient, no moment, no update, nothing added to the gradient norm.
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        flat = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k in ("p", "g", "m", "v")}
        views, off, step = [], 0, 0
        for p in params:
            k = p.numel()
            pv, gv, mv, vv = (flat[key][off:off + k].view_as(p) for key in ("p", "g", "m", "v"))
            pv.copy_(p.data)
            s = self.optimizer.state.get(p)
            if s:                                  # state restored by optimizer.load_state_dict() or left by the CPU path
                mv.copy_(s["exp_avg"]); vv.copy_(s["exp_avg_sq"])
                step = max(step, int(float(s["step"])))
            p.data = pv
            p.grad = gv
            self.optimizer.state[p] = {"step": torch.tensor(float(step)), "exp_avg": mv, "exp_avg_sq": vv}
            views.append((pv, gv, mv, vv))
            off += (k + 3) // 4 * 4
        flat["views"] = views
        flat["ws"] = torch.empty(hip.A.lib().cn_adam_workspace_doubles(), dtype=torch.float64, device=dev)
        self._flat = flat
        self._step = step
        self._weights_changed()

    def _weights_changed(self):
        f = getattr(self.actor_critic, "weights_changed", None)
        if f is not None:
            f()        # the rollout-side weight snapshot (cn_policy_set_weights) must be refreshed: raw-pointer writes bump no version

    def _sync_optimizer_state(self):
        for s in self.optimizer.state.values():
            s["step"].fill_(float(self._step))

    def _advantages(self, rollouts):
        ret, val = rollouts.returns, rollouts.value_preds
        T, N = rollouts.rewards.shape[0], rollouts.rewards.shape[1]
        d = _dist()
        if ret.is_cuda:
            stats = hip.adv_stats(ret, val, T * N)          # HIP kernel (first T rows are contiguous)
            if d is not None:
                d.all_reduce(stats)
            adv = torch.empty(T, N, 1, device=ret.device)
            hip.adv_normalize(ret, val, stats, T * N, adv)
            return adv
        # CPU tensors (unit tests / --no-cuda plumbing): ppo.py:37-39 in torch ops
        adv = ret[:-1] - val[:-1]
        if d is not None:
            a64 = adv.double()
            stats = torch.stack([a64.sum(), (a64 * a64).sum(), torch.tensor(float(adv.numel()), dtype=torch.float64)])
            d.all_reduce(stats)
            mean = stats[0] / stats[2]
            std = ((stats[1] - stats[2] * mean * mean) / (stats[2] - 1)).clamp(min=0).sqrt()
            return (adv - mean.float()) / (std.float() + 1e-5)
        return (adv - adv.mean()) / (adv.std() + 1e-5)

    def _losses(self, values, action_log_probs, old_logp, adv_targ, value_preds, returns):
        """(value_loss, action_loss) of ppo.py:66-84.  GPU tensors: one fused HIP kernel each way (cn_ppo_loss_fwd/bwd)."""
        if values.is_cuda:
            losses = hip.PPOLoss.apply(values, action_log_probs, old_logp, adv_targ, value_preds, returns, self.clip_param,
                                       self.use_clipped_value_loss)
            return losses[0], losses[1]
        ratio = torch.exp(action_log_probs - old_logp)
        surr1 = ratio * adv_targ
        surr2 = torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_targ
        action_loss = -torch.min(surr1, surr2).mean()
        if self.use_clipped_value_loss:
            value_pred_clipped = value_preds + (values - value_preds).clamp(-self.clip_param, self.clip_param)
            value_losses = (values - returns).pow(2)
            value_losses_clipped = (value_pred_clipped - returns).pow(2)
            value_loss = 0.5 * torch.max(value_losses, value_losses_clipped).mean()
        else:
            value_loss = 0.5 * (returns - values).pow(2).mean()
        return value_loss, action_loss

    # ---- one boundary call per minibatch (default network on the GPU) -------------------------------------------------
    # cn_ppo_minibatch_step gathers the minibatch from the storage by env index, runs the train-mode forward, the losses and the whole
    # backward, and WRITES every parameter gradient into the flat bucket: no autograd graph, no framework kernel between the rollout
    # storage and the Adam step.  use_minibatch_step = False (or CN_PPO_MINIBATCH_STEP=0) keeps the autograd-joined path below, which
    # the non-default variants (use_self_attn / sort_humans off, fp32 arithmetic, CPU tensors) always take.
    use_minibatch_step = True

    def _fast_path(self, rollouts):
        import os
        if not self.use_minibatch_step or os.environ.get("CN_PPO_MINIBATCH_STEP", "1") == "0":
            return False
        if not self.actor_critic.is_recurrent or not hasattr(self.actor_critic, "base"):
            return False
        return hip.MinibatchStepper.supported(self.actor_critic, rollouts)

    def _update_fast(self, rollouts, advantages, d):
        flat = self._flat
        if getattr(self, "_stepper", None) is None or self._stepper.policy is not self.actor_critic:
            self._stepper = hip.MinibatchStepper(self.actor_critic)
        stepper = self._stepper
        for p, (_, gv, _, _) in zip(self._params(), flat["views"]):     # the bucket views must still be the gradients (checked once per update())
            if p.grad is not gv and (p.grad is None or p.grad.data_ptr() != gv.data_ptr()):
                p.grad = gv
        E = rollouts.rewards.shape[1]
        dev = rollouts.rewards.device
        assert E >= self.num_mini_batch, (
            "PPO requires the number of processes ({}) to be greater than or equal to the number of PPO mini batches ({}).".format(E, self.num_mini_batch))
        npb = E // self.num_mini_batch
        totals = stepper.row_totals(rollouts)                    # the ONE readback of update(): rows per env -> rows per minibatch on the host
        starts = list(range(0, E, npb))
        losses = torch.zeros(self.ppo_epoch * len(starts), 3, device=dev)
        hyper = (self.clip_param, self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss)
        ar_events, k = [], 0
        for e in range(self.ppo_epoch):
            perm = torch.randperm(E)                             # storage.py:193: one permutation per epoch, same generator draw
            for start in starts:
                if start + npb > E:                              # storage.py:209-210 raises here, after the complete groups (see storage.py)
                    raise IndexError("index {} is out of bounds for dimension 0 with size {}".format(E, E))
                idx = perm[start:start + npb]
                rows = int(totals[idx].sum())
                stepper.step(rollouts, advantages, idx.to(device=dev, dtype=torch.int32), rows, hyper, losses[k])
                scale = 1.0
                if d is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                    d.all_reduce(flat["g"])                      # ONE collective per optimiser step (RCCL over xGMI)
                    ev[1].record()
                    ar_events.append(ev)
                    scale = 1.0 / d.get_world_size()
                g = self.optimizer.param_groups[0]
                self._step += 1
                hip.adam_clip_step(flat["p"], flat["g"], flat["m"], flat["v"], self._step, g["lr"], g["betas"], g["eps"],
                                   self.max_grad_norm, grad_scale=scale, workspace=flat["ws"])
                k += 1
        self._weights_changed()
        self._sync_optimizer_state()
        if ar_events:
            torch.cuda.synchronize()
            self.last_allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)
        sums = losses[:k].sum(0)
        if d is not None:
            d.all_reduce(sums)
            sums /= d.get_world_size()
        v, a, ent = (sums / (self.ppo_epoch * self.num_mini_batch)).tolist()   # single host sync per update()
        return v, a, ent

    def update(self, rollouts):
        advantages = self._advantages(rollouts)
        dev = rollouts.rewards.device
        sums = torch.zeros(3, device=dev)
        d = _dist()
        on_gpu = dev.type == "cuda"
        if on_gpu:
            self._bind_flat()
        num_steps = 0
        ar_events = []
        self.last_allreduce_ms = None
        if on_gpu and self._fast_path(rollouts):
            return self._update_fast(rollouts, advantages, d)
        for e in range(self.ppo_epoch):
            if not self.actor_critic.is_recurrent:
                raise NotImplementedError("feed-forward policies are out of scope")
            for sample in rollouts.recurrent_generator(advantages, self.num_mini_batch):
                obs_batch, hxs_batch, actions_batch, value_preds_batch, return_batch, masks_batch, old_logp_batch, adv_targ = sample
                values, action_log_probs, dist_entropy, _ = self.actor_critic.evaluate_actions(obs_batch, hxs_batch, masks_batch, actions_batch)
                value_loss, action_loss = self._losses(values, action_log_probs, old_logp_batch, adv_targ, value_preds_batch, return_batch)
                total_loss = value_loss * self.value_loss_coef + action_loss - dist_entropy * self.entropy_coef
                if on_gpu:
                    flat = self._flat
                    flat["g"].zero_()                         # optimizer.zero_grad(): the views stay bound
                    total_loss.backward()
                    if num_steps == 0:
                        # autograd accumulates into the bound views; checked once per update() (not per optimiser step: the walk is
                        # O(parameters) of Python) in case something replaced a .grad instead of accumulating into the bucket
                        for p, (_, gv, _, _) in zip(self._params(), flat["views"]):
                            if p.grad is not gv and p.grad.data_ptr() != gv.data_ptr():
                                gv.copy_(p.grad)
                                p.grad = gv
                    scale = 1.0
                    if d is not None:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record()
                        d.all_reduce(flat["g"])               # ONE collective per optimiser step (RCCL over xGMI)
                        ev[1].record()
                        ar_events.append(ev)
                        scale = 1.0 / d.get_world_size()
                    g = self.optimizer.param_groups[0]
                    self._step += 1
                    hip.adam_clip_step(flat["p"], flat["g"], flat["m"], flat["v"], self._step, g["lr"], g["betas"], g["eps"],
                                       self.max_grad_norm, grad_scale=scale, workspace=flat["ws"])
                    self._weights_changed()
                else:
                    self.optimizer.zero_grad()
                    total_loss.backward()
                    if d is not None:                          # CPU tensors (gloo tests): pack, one all-reduce, unpack
                        ps = [p for p in self._params() if p.grad is not None]
                        bucket = torch.cat([p.grad.reshape(-1) for p in ps])
                        d.all_reduce(bucket)
                        bucket /= d.get_world_size()
                        off = 0
                        for p in ps:
                            p.grad.copy_(bucket[off:off + p.numel()].view_as(p))
                            off += p.numel()
                    nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
                    self.optimizer.step()
                sums += torch.stack([value_loss.detach(), action_loss.detach(), dist_entropy.detach()])
                num_steps += 1
        if on_gpu:
            self._sync_optimizer_state()
        if ar_events:      # mean duration of the gradient all-reduce on this rank's stream (the events complete with the host sync below)
            torch.cuda.synchronize()
            self.last_allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)
        num_updates = self.ppo_epoch * self.num_mini_batch
        if d is not None:
            d.all_reduce(sums)
            sums /= d.get_world_size()
        v, a, ent = (sums / num_updates).tolist()   # single host sync per update()
        return v, a, ent
