"""RolloutStorage -- mirror of rl/networks/storage.py:13-253 with device-resident, batched internals.

Same constructor signature and attribute names (`obs`, `recurrent_hidden_states`, `rewards`, `value_preds`, `returns`,
`action_log_probs`, `actions`, `masks`, `bad_masks`, `step`) so `train.py` and `PPO.update` use it unchanged.
Differences that are invisible to callers:
  * `recurrent_hidden_states['human_human_edge_rnn']` is always zero in the reference (selfAttn_srnn_temp_node.py:389-393)
    but stored as [T+1,N,H+1,256] (2.7 GB at 4096 envs); here it is a stride-0 view of one zero.
  * `compute_returns` runs the GAE scan as one HIP kernel (cn_gae) when the buffers live on the GPU.
  * `recurrent_generator` gathers each minibatch with one index_select per tensor instead of a Python loop over envs.
"""
import torch

from . import hip


class RolloutStorage(object):
    def __init__(self, num_steps, num_processes, obs_shape, action_space, human_node_rnn_size, human_human_edge_rnn_size):
        if not isinstance(obs_shape, dict):
            raise NotImplementedError("only Dict observation spaces (the crowd-sim envs) are supported")
        self.obs = {}
        for key in obs_shape:
            dtype = torch.bool if key == "visible_masks" else torch.float32
            self.obs[key] = torch.zeros(num_steps + 1, num_processes, *obs_shape[key].shape, dtype=dtype)
        self.human_num = obs_shape["spatial_edges"].shape[0]
        self._edge_shape = (num_steps + 1, num_processes, self.human_num + 1, human_human_edge_rnn_size)
        self.recurrent_hidden_states = {
            "human_node_rnn": torch.zeros(num_steps + 1, num_processes, 1, human_node_rnn_size),
            "human_human_edge_rnn": torch.zeros(1, 1, 1, 1).expand(*self._edge_shape),
        }
        self.rewards = torch.zeros(num_steps, num_processes, 1)
        self.value_preds = torch.zeros(num_steps + 1, num_processes, 1)
        self.returns = torch.zeros(num_steps + 1, num_processes, 1)
        self.action_log_probs = torch.zeros(num_steps, num_processes, 1)
        if action_space.__class__.__name__ == "Discrete":
            raise NotImplementedError("Discrete action spaces are out of scope (the robot action is Box(2))")
        self.actions = torch.zeros(num_steps, num_processes, action_space.shape[0])
        self.masks = torch.ones(num_steps + 1, num_processes, 1)
        self.bad_masks = torch.ones(num_steps + 1, num_processes, 1)
        self.num_steps = num_steps
        self.step = 0

    def to(self, device):
        for key in self.obs:
            self.obs[key] = self.obs[key].to(device)
        self.recurrent_hidden_states["human_node_rnn"] = self.recurrent_hidden_states["human_node_rnn"].to(device)
        self.recurrent_hidden_states["human_human_edge_rnn"] = torch.zeros(1, 1, 1, 1, device=device).expand(*self._edge_shape)
        for name in ("rewards", "value_preds", "returns", "action_log_probs", "actions", "masks", "bad_masks"):
            setattr(self, name, getattr(self, name).to(device))

    def insert(self, obs, recurrent_hidden_states, actions, action_log_probs, value_preds, rewards, masks, bad_masks):
        for key in self.obs:
            if key in obs:
                self.obs[key][self.step + 1].copy_(obs[key].view_as(self.obs[key][self.step + 1]), non_blocking=True)
        self.recurrent_hidden_states["human_node_rnn"][self.step + 1].copy_(
            recurrent_hidden_states["human_node_rnn"].view_as(self.recurrent_hidden_states["human_node_rnn"][self.step + 1]))
        self.actions[self.step].copy_(actions)
        self.action_log_probs[self.step].copy_(action_log_probs)
        self.value_preds[self.step].copy_(value_preds)
        self.rewards[self.step].copy_(rewards.view_as(self.rewards[self.step]), non_blocking=True)
        self.masks[self.step + 1].copy_(masks.view_as(self.masks[self.step + 1]), non_blocking=True)
        self.bad_masks[self.step + 1].copy_(bad_masks.view_as(self.bad_masks[self.step + 1]), non_blocking=True)
        self.step = (self.step + 1) % self.num_steps

    def after_update(self):
        for key in self.obs:
            self.obs[key][0].copy_(self.obs[key][-1])
        self.recurrent_hidden_states["human_node_rnn"][0].copy_(self.recurrent_hidden_states["human_node_rnn"][-1])
        self.masks[0].copy_(self.masks[-1])
        self.bad_masks[0].copy_(self.bad_masks[-1])

    def compute_returns(self, next_value, use_gae, gamma, gae_lambda, use_proper_time_limits=True):
        T = self.rewards.size(0)
        if use_gae and not use_proper_time_limits and self.rewards.is_cuda:
            self.value_preds[-1] = next_value
            hip.gae(self.rewards, self.value_preds, self.masks, gamma, gae_lambda, self.returns)  # HIP kernel, no fallback
            return
        # remaining branches of storage.py:104-137 (CPU tensors, or flag combinations no BASELINE config uses), torch ops
        if use_gae:
            self.value_preds[-1] = next_value
            gae = 0
            for step in reversed(range(T)):
                delta = self.rewards[step] + gamma * self.value_preds[step + 1] * self.masks[step + 1] - self.value_preds[step]
                gae = delta + gamma * gae_lambda * self.masks[step + 1] * gae
                if use_proper_time_limits:
                    gae = gae * self.bad_masks[step + 1]
                self.returns[step] = gae + self.value_preds[step]
        else:
            self.returns[-1] = next_value
            for step in reversed(range(T)):
                if use_proper_time_limits:
                    self.returns[step] = (self.returns[step + 1] * gamma * self.masks[step + 1] + self.rewards[step]) * self.bad_masks[step + 1] \
                        + (1 - self.bad_masks[step + 1]) * self.value_preds[step]
                else:
                    self.returns[step] = self.returns[step + 1] * gamma * self.masks[step + 1] + self.rewards[step]

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        raise NotImplementedError("the policy is recurrent (ppo.py:47): only recurrent_generator is on the path")

    def recurrent_generator(self, advantages, num_mini_batch):
        """Minibatches of whole env trajectories: storage.py:184-253 (same torch.randperm draw, same T-major flattening)."""
        num_processes = self.rewards.size(1)
        assert num_processes >= num_mini_batch, (
            "PPO requires the number of processes ({}) to be greater than or equal to the number of PPO mini batches ({}).".format(
                num_processes, num_mini_batch))
        # storage.py:190-192: groups of num_processes // num_mini_batch envs, as many as range(0, num_processes, npb) gives -- 6 envs
        # with 4 mini-batches run as 6 groups of one env, 8 with 3 as 4 groups of two
        npb = num_processes // num_mini_batch
        perm = torch.randperm(num_processes)
        T = self.num_steps
        dev = self.rewards.device
        for start in range(0, num_processes, npb):
            if start + npb > num_processes:
                # a last, incomplete group: the reference's per-env loop reads perm[start + offset] past the end here and raises
                # IndexError in the middle of the epoch, after the complete groups have been yielded (storage.py:209-210)
                raise IndexError("index {} is out of bounds for dimension 0 with size {}".format(num_processes, num_processes))
            idx = perm[start:start + npb].to(dev)
            N = idx.numel()

            def take(x):  # [T(+1), num_processes, ...] -> [T*N, ...]
                g = x[:T].index_select(1, idx)
                return g.reshape(T * N, *g.shape[2:])

            obs_batch = {key: take(self.obs[key]) for key in self.obs}
            hxs = {"human_node_rnn": self.recurrent_hidden_states["human_node_rnn"][0].index_select(0, idx),
                   "human_human_edge_rnn": torch.zeros(1, 1, 1, device=dev).expand(N, *self._edge_shape[2:])}
            yield (obs_batch, hxs, take(self.actions), take(self.value_preds), take(self.returns), take(self.masks),
                   take(self.action_log_probs), take(advantages))
