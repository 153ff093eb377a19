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
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
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

    # one flat gradient bucket; p.grad are views into it, so the all-reduce needs no packing copies
    def _bind_flat_grads(self):
        params = [p for p in self.actor_critic.parameters() if p.requires_grad]
        if self._flat is not None and self._flat.device == params[0].device and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                                                                                   for p, v in zip(params, self._views)):
            return
        n = sum(p.numel() for p in params)
        self._flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        self._views, off = [], 0
        for p in params:
            v = self._flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self._views.append(v)
            off += p.numel()

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

    def update(self, rollouts):
        advantages = self._advantages(rollouts)
        dev = rollouts.rewards.device
        sums = torch.zeros(3, device=dev)
        d = _dist()
        self._bind_flat_grads()
        for e in range(self.ppo_epoch):
            if not self.actor_critic.is_recurrent:
                raise NotImplementedError("feed-forward policies are out of scope")
            for sample in rollouts.recurrent_generator(advantages, self.num_mini_batch):
                obs_batch, hxs_batch, actions_batch, value_preds_batch, return_batch, masks_batch, old_logp_batch, adv_targ = sample
                values, action_log_probs, dist_entropy, _ = self.actor_critic.evaluate_actions(obs_batch, hxs_batch, masks_batch, actions_batch)
                ratio = torch.exp(action_log_probs - old_logp_batch)
                surr1 = ratio * adv_targ
                surr2 = torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_targ
                action_loss = -torch.min(surr1, surr2).mean()
                if self.use_clipped_value_loss:
                    value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-self.clip_param, self.clip_param)
                    value_losses = (values - return_batch).pow(2)
                    value_losses_clipped = (value_pred_clipped - return_batch).pow(2)
                    value_loss = 0.5 * torch.max(value_losses, value_losses_clipped).mean()
                else:
                    value_loss = 0.5 * (return_batch - values).pow(2).mean()
                self._flat.zero_()
                total_loss = value_loss * self.value_loss_coef + action_loss - dist_entropy * self.entropy_coef
                total_loss.backward()
                if d is not None:
                    d.all_reduce(self._flat)
                    self._flat.div_(d.get_world_size())
                nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
                self.optimizer.step()
                sums += torch.stack([value_loss.detach(), action_loss.detach(), dist_entropy.detach()])
        num_updates = self.ppo_epoch * self.num_mini_batch
        if d is not None:
            d.all_reduce(sums)
            sums /= d.get_world_size()
        v, a, ent = (sums / num_updates).tolist()   # single host sync per update()
        return v, a, ent
