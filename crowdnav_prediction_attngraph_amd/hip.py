"""Thin object wrappers over the C ABI handles (cn_env_batch, cn_policy) and the rollout-math entry points."""
import ctypes as C

import torch

from . import _abi as A


def _need_cuda():
    if not torch.cuda.is_available():
        raise A.CnError("no GPU visible: the crowd-sim / policy hot path only runs on MI355X (no CPU fallback)")


class HipEnvBatch:
    """E device-resident environments (cn_env_batch).  Mirrors VecEnv reset/step at tensor level; no host sync."""

    def __init__(self, cfg, num_envs, seed, first_env_index=0, device=None):
        _need_cuda()
        self.cfg = cfg
        self.E = int(num_envs)
        self.H = int(cfg.human_num) + int(cfg.human_num_range)   # observation rows (the crowd holds <= H humans)
        self.D = A.lib().cn_env_obs_width(C.byref(cfg))
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_create(C.byref(cfg), self.E, int(seed), int(first_env_index), C.byref(h)), "cn_env_create")
        self._h = h
        E, H, D, dev = self.E, self.H, self.D, self.device
        self.obs = {
            "robot_node": torch.zeros(E, 1, 7, device=dev), "temporal_edges": torch.zeros(E, 1, 2, device=dev),
            "spatial_edges": torch.zeros(E, H, D, device=dev), "detected_human_num": torch.zeros(E, 1, device=dev),
            "visible_masks": torch.zeros(E, H, dtype=torch.uint8, device=dev),
        }
        # derived data of the newest observation, written by every reset() / step() beside it: row offsets + a packing of the envs into
        # equally filled tiles for the policy's fused human-human kernel (csrc/row_plan.h).  Pass it to HipPolicy.act(..., row_plan=) ONLY
        # together with the observation it was made for (the one the last reset() / step() of THIS batch returned or wrote).
        self.row_plan = torch.zeros(int(A.lib().cn_row_plan_words(E)), dtype=torch.int32, device=dev)
        # the per-step host-visible outputs live in ONE buffer (ep_return f64 | reward f32 | ep_len i32 | done u8 | info u8): a caller that
        # needs them on the host (the reference's VecPyTorch contract, vec_env.BatchedCrowdSim.step) fetches all five with one transfer
        self._packed = torch.zeros(18 * E, dtype=torch.uint8, device=dev)
        self.ep_return = self._packed[0:8 * E].view(torch.float64)
        self.reward = self._packed[8 * E:12 * E].view(torch.float32)
        self.ep_len = self._packed[12 * E:16 * E].view(torch.int32)
        self.done = self._packed[16 * E:17 * E]
        self.info = self._packed[17 * E:18 * E]
        self._packed_host = None
        self._reward_in_packed = False    # did the last step() write its reward into the packed buffer (fetch_step_outputs reads it there)?
        self._tail_policies = []          # weak references to the HipPolicy objects whose post-hh hook points at this batch (attach_env_tail)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # a policy whose hook still holds this handle would call cn_env_launch_tail on freed memory at its next forward: detach first
            for ref in list(getattr(self, "_tail_policies", ())):
                pol = ref()
                if pol is not None and getattr(pol, "_tail_env", None) is self:
                    pol.attach_env_tail(None)
            self._tail_policies = []
            A.lib().cn_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, obs=None):
        obs = self.obs if obs is None else obs
        o = A.obs_struct(obs, self.row_plan)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_reset(self._h, C.byref(o), A.stream_ptr()), "cn_env_reset")
        return obs

    def step(self, actions, obs=None, not_done=None, reward=None):
        """actions [E,2] float32 on the device -> (obs, reward [E], done [E] u8, info [E] u8, ep_return, ep_len).
        not_done: optional float32 [E] / [E,1] tensor that receives 1 - done (the rollout `masks`);
        reward: optional float32 [E] / [E,1] tensor written instead of the internal reward buffer."""
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        for name, t in (("not_done", not_done), ("reward", reward)):
            if t is not None and (t.dtype != torch.float32 or t.numel() != self.E or not t.is_contiguous()):
                raise A.CnError("%s must be a contiguous float32 tensor of %d elements" % (name, self.E))
        if actions.dtype != torch.float32 or actions.shape != (self.E, 2):
            raise A.CnError("actions must be float32 [%d,2]" % self.E)
        actions = actions.contiguous()
        o = A.obs_struct(obs, self.row_plan)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_step(self._h, A.ptr(actions), C.byref(o), A.ptr(reward), A.ptr(self.done), A.ptr(self.info),
                                        A.ptr(self.ep_return), A.ptr(self.ep_len), A.ptr(not_done), A.stream_ptr()), "cn_env_step")
        self._reward_in_packed = reward.data_ptr() == self.reward.data_ptr()
        return obs, reward, self.done, self.info, self.ep_return, self.ep_len

    def fetch_step_outputs(self):
        """(reward f32 [E], done bool [E], info u8 [E], ep_return f64 [E], ep_len i32 [E]) of the last step() as numpy arrays: ONE
        device-to-host transfer into a pinned buffer and one stream synchronisation.  The arrays are views of that buffer: valid until the
        next call.  The reward is whatever the packed buffer holds at the time of the call: the raw env reward, or -- when a wrapper
        (gst.PretextProcessor / HipGST.wrapper_step) has added its prediction penalty to that buffer in place -- the reward after the penalty.
        After a step(reward=<caller's tensor>) the packed buffer does not hold that step's reward at all, and the call raises."""
        if not self._reward_in_packed:
            raise A.CnError("fetch_step_outputs: the last step() wrote its reward into a caller-supplied tensor (or no step ran yet), "
                            "the packed buffer holds an older one; read the tensor passed as step(reward=...)")
        E = self.E
        if self._packed_host is None:
            self._packed_host = torch.empty(18 * E, dtype=torch.uint8, pin_memory=True)
        self._packed_host.copy_(self._packed, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        h = self._packed_host.numpy()
        return (h[8 * E:12 * E].view("float32"), h[16 * E:17 * E].view("bool"), h[17 * E:18 * E], h[0:8 * E].view("float64"),
                h[12 * E:16 * E].view("int32"))

    def set_tail_deferral(self, enabled):
        """Hold the side-stream tail of every step (ORCA fallback programs + episode pre-generation) back until launch_tail() -- or until
        HipPolicy.attach_env_tail(env) makes the policy release it right behind its human-human kernel (cn_env_set_tail_deferral)."""
        A.check(A.lib().cn_env_set_tail_deferral(self._h, int(bool(enabled))), "cn_env_set_tail_deferral")

    def launch_tail(self):
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_launch_tail(self._h, A.stream_ptr()), "cn_env_launch_tail")

    def set_pregen_budget(self, ticks_10ns):
        """Time budget (x 10 ns) of one launch of the episode pre-generation kernel; the episodes do not depend on it (cn_env_set_pregen_budget)."""
        A.check(A.lib().cn_env_set_pregen_budget(self._h, int(ticks_10ns)), "cn_env_set_pregen_budget")

    def join(self):
        """Order the library's side-stream work (ORCA of the current state, episode pre-generation) before what the caller enqueues next
        on the current stream: needed to close a graph capture of a block of steps."""
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_join(self._h, A.stream_ptr()), "cn_env_join")

    def get_state(self):
        humans = torch.zeros(self.E, self.H, 8, dtype=torch.float64, device=self.device)
        robot = torch.zeros(self.E, 8, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_get_state(self._h, A.ptr(humans), A.ptr(robot), A.stream_ptr()), "cn_env_get_state")
        return humans, robot

    def set_case_counters(self, counters):
        """counters: int64/uint64 [E] tensor; the next reset of env e generates test/train case `counters[e]` (+ its seed)."""
        c = torch.as_tensor(counters, dtype=torch.int64).to(self.device).contiguous()
        if c.numel() != self.E or bool((c < 0).any()):
            raise A.CnError("counters must be %d non-negative integers" % self.E)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_set_case_counters(self._h, A.ptr(c), A.stream_ptr()), "cn_env_set_case_counters")

    def state_dict(self):
        """Snapshot of the whole simulator state (cn_env_save) as one uint8 CPU tensor -- torch.save()-able next to the policy."""
        n = int(A.lib().cn_env_snapshot_bytes(self._h))
        buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_save(self._h, A.ptr(buf), A.stream_ptr()), "cn_env_save")
        return buf.cpu()

    def load_state_dict(self, snap):
        """Restore a snapshot taken from a batch of the same configuration / shape / seed / shard (cn_env_load)."""
        n = int(A.lib().cn_env_snapshot_bytes(self._h))
        snap = torch.as_tensor(snap, dtype=torch.uint8)
        if snap.numel() != n:
            raise A.CnError("snapshot has %d bytes, this batch needs %d (different env count / humans / config?)" % (snap.numel(), n))
        buf = snap.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_load(self._h, A.ptr(buf), A.stream_ptr()), "cn_env_load")
        self.row_plan[:8].zero_()   # the plan belongs to an observation of the state that was just replaced

    def get_danger_min_dist(self):
        """Danger.min_dist of the last step per env (float64 [E]); non-zero only in the test phase."""
        out = torch.zeros(self.E, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_get_danger_min_dist(self._h, A.ptr(out), A.stream_ptr()), "cn_env_get_danger_min_dist")
        return out

    def get_human_counts(self):
        """len(self.humans) per env (int32 [E]); constant unless sim.human_num_range > 0."""
        out = torch.zeros(self.E, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_get_human_counts(self._h, A.ptr(out), A.stream_ptr()), "cn_env_get_human_counts")
        return out

    def get_human_actions(self):
        out = torch.zeros(self.E, self.H, 2, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_env_get_human_actions(self._h, A.ptr(out), A.stream_ptr()), "cn_env_get_human_actions")
        return out


def orca_solve(self_state, others, neighbor_dist=10.0, max_neighbors=None, time_horizon=5.0, time_step=0.25):
    """Batched stand-alone ORCA (rvo2 replacement).  self_state [B,8], others [B,n,5] float32 device tensors -> [B,2]."""
    _need_cuda()
    B, n = others.shape[0], others.shape[1]
    out = torch.zeros(B, 2, device=self_state.device)
    A.check(A.lib().cn_orca_solve(B, n, A.ptr(self_state.contiguous()), A.ptr(others.contiguous()), float(neighbor_dist),
                                  n if max_neighbors is None else int(max_neighbors), float(time_horizon), float(time_step),
                                  A.ptr(out), A.stream_ptr()), "cn_orca_solve")
    return out


class HipPolicy:
    """cn_policy handle: rollout-time forward (act / get_value) of the attention-graph policy."""

    def __init__(self, human_num, edge_width, max_envs, device=None):
        _need_cuda()
        self.H, self.D, self.maxE = int(human_num), int(edge_width), int(max_envs)
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_policy_create(self.H, self.D, self.maxE, C.byref(h)), "cn_policy_create")
        self._h = h
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            A.lib().cn_policy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_self_attention(self, enabled):
        """args.use_self_attn (cn_policy_set_self_attention): False = no human-human attention, spatial_linear is the two-layer MLP
        Linear(D, 128) - ReLU - Linear(128, 256) - ReLU on the spatial edges.  Call before set_weights."""
        A.check(A.lib().cn_policy_set_self_attention(self._h, int(bool(enabled))), "cn_policy_set_self_attention")
        self._self_attn = bool(enabled)

    def set_weights(self, state_dict):
        """state_dict: reference key -> float32 device tensor (the library snapshots + folds them)."""
        w = A.PolicyWeights()
        keep = []
        no_attn = not getattr(self, "_self_attn", True)
        # use_self_attn = False: the state dict has no spatial_attn.* tensors and spatial_linear is [.0: 128 x D, .2: 256 x 128]
        remap = {"emb0_w": "base.spatial_linear.0.weight", "emb0_b": "base.spatial_linear.0.bias",
                 "spatial_linear_w": "base.spatial_linear.2.weight", "spatial_linear_b": "base.spatial_linear.2.bias"} if no_attn else {}
        for field, key in A.POLICY_WEIGHT_KEYS:
            if no_attn and field not in remap and key.startswith("base.spatial_attn."):
                setattr(w, field, None)
                continue
            key = remap.get(field, key)
            t = state_dict[key].detach()
            if t.dtype != torch.float32 or not t.is_cuda:
                t = t.to(device=self.device, dtype=torch.float32)
            t = t.contiguous()
            keep.append(t)
            setattr(w, field, t.data_ptr())
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_policy_set_weights(self._h, C.byref(w), A.stream_ptr()), "cn_policy_set_weights")
        self._keep = keep  # keep sources alive until the async copies are ordered behind later work on the stream

    def act(self, obs, hxs, masks, eps=None, out=None, row_plan=None):
        """row_plan: HipEnvBatch.row_plan of the batch that produced `obs` with its last reset() / step() (optional; see there)."""
        E = obs["robot_node"].shape[0]
        dev = self.device
        if out is None:
            out = dict(value=torch.empty(E, 1, device=dev), action=torch.empty(E, 2, device=dev),
                       logp=torch.empty(E, 1, device=dev), hxs=torch.empty(E, 1, 128, device=dev))
        o = A.obs_struct(obs, row_plan)
        with torch.cuda.device(dev):
            A.check(A.lib().cn_policy_act(self._h, E, C.byref(o), A.ptr(hxs.contiguous()), A.ptr(masks.contiguous()),
                                          A.ptr(None if eps is None else eps.contiguous()), A.ptr(out["value"]), A.ptr(out["action"]),
                                          A.ptr(out["logp"]), A.ptr(out["hxs"]), A.stream_ptr()), "cn_policy_act")
        return out

    def get_value(self, obs, hxs, masks, out=None):
        E = obs["robot_node"].shape[0]
        if out is None:
            out = torch.empty(E, 1, device=self.device)
        o = A.obs_struct(obs)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_policy_get_value(self._h, E, C.byref(o), A.ptr(hxs.contiguous()), A.ptr(masks.contiguous()), A.ptr(out),
                                                A.stream_ptr()), "cn_policy_get_value")
        return out

    def taps(self, E):
        dev, H = self.device, self.H
        t = dict(spatial_lin=torch.empty(E, H, 256, device=dev), hr_attn=torch.empty(E, H, device=dev),
                 hr_out=torch.empty(E, 256, device=dev), robot_emb=torch.empty(E, 256, device=dev), actor_feat=torch.empty(E, 256, device=dev))
        with torch.cuda.device(dev):
            A.check(A.lib().cn_policy_get_taps(self._h, E, A.ptr(t["spatial_lin"]), A.ptr(t["hr_attn"]), A.ptr(t["hr_out"]),
                                               A.ptr(t["robot_emb"]), A.ptr(t["actor_feat"]), A.stream_ptr()), "cn_policy_get_taps")
        return t

    def set_gemm_mode(self, mode):
        """'fused' (default: the whole human-human block as one persistent kernel, bf16x3 split-precision MFMA, ~2e-5 of fp32),
        'bf16x3' (the same arithmetic as separate launches: embedding / q|k|v GEMMs, attention, out_proj) or 'fp32' (exact fp32 MFMA)."""
        A.check(A.lib().cn_policy_set_gemm_mode(self._h, {"fp32": 0, "bf16x3": 1, "fused": 2}[mode]), "cn_policy_set_gemm_mode")

    def set_taps(self, enabled):
        """Fused mode: keep / drop the test taps (robot_emb, hr_attn, hr_out, actor_feat) the robot-node kernel writes per forward."""
        A.check(A.lib().cn_policy_set_taps(self._h, int(bool(enabled))), "cn_policy_set_taps")

    def set_profiling(self, every):
        """0 / False: off; n >= 1 (True = 1): every n-th forward's dominant kernel is timed with a pair of events on its stream."""
        A.check(A.lib().cn_policy_set_profiling(self._h, int(every)), "cn_policy_set_profiling")

    def attach_env_tail(self, env):
        """env: a HipEnvBatch in tail-deferral mode (or None to detach): every forward releases that batch's held-back side work right after
        its human-human kernel is enqueued (cn_policy_set_post_hh_hook with cn_env_launch_tail).  Detach before closing the env."""
        if env is None:
            if getattr(self, "_h", None) is not None and self._h:
                A.check(A.lib().cn_policy_set_post_hh_hook(self._h, None, None), "cn_policy_set_post_hh_hook")
            old = getattr(self, "_tail_env", None)
            if old is not None:      # forget this policy in the batch's list (repeated attach / detach must not grow it)
                old._tail_policies[:] = [r for r in old._tail_policies if r() is not None and r() is not self]
            self._tail_env = None
            return
        if getattr(env, "_h", None) is None or not env._h:
            raise A.CnError("attach_env_tail: the env batch is closed")
        fn = C.cast(A.lib().cn_env_launch_tail, C.c_void_p)
        A.check(A.lib().cn_policy_set_post_hh_hook(self._h, fn, env._h), "cn_policy_set_post_hh_hook")
        self._tail_env = env          # keeps the batch alive as long as the hook points at it
        import weakref
        env._tail_policies[:] = [r for r in env._tail_policies if r() is not None and r() is not self]
        env._tail_policies.append(weakref.ref(self))   # ... and env.close() detaches the hook before the handle is freed

    def get_profile(self):
        ms = (C.c_double * 8)()
        n = (C.c_int64 * 8)()
        A.check(A.lib().cn_policy_get_profile(self._h, ms, n), "cn_policy_get_profile")
        return list(ms), list(n)

    def get_profile_samples(self):
        """The event brackets of the dominant kernel one by one [ms], in launch order (cn_policy_get_profile_samples)."""
        n = A.lib().cn_policy_get_profile_samples(self._h, None, 0)
        if n < 0:
            A.check(n, "cn_policy_get_profile_samples")
        buf = (C.c_float * max(n, 1))()
        n = A.lib().cn_policy_get_profile_samples(self._h, buf, n)
        return [float(buf[i]) for i in range(n)]

    def reset_profile(self):
        """Drop the samples and sums collected so far, keep the stride (the events stay warm)."""
        A.check(A.lib().cn_policy_reset_profile(self._h), "cn_policy_reset_profile")


def compact_visible(spatial_edges, visible_masks):
    """args.sort_humans = False: (spatial_edges [B,H,D] with the visible humans moved to the front, detected [B,1] = max(1, visible)) --
    cn_obs_compact_visible; see include/crowdnav_hip.h for why this equals the reference's mask-based attention."""
    B, H, D = spatial_edges.shape
    se = spatial_edges.detach().to(torch.float32).contiguous()
    vm = visible_masks.detach().reshape(B, H).to(torch.uint8).contiguous()
    out = torch.empty_like(se)
    det = torch.empty(B, 1, device=se.device)
    with torch.cuda.device(se.device):
        A.check(A.lib().cn_obs_compact_visible(B, H, D, A.ptr(se), A.ptr(vm), A.ptr(out), A.ptr(det), A.stream_ptr()), "cn_obs_compact_visible")
    return out, det


class StepStamps:
    """Device-side launch stamps of the rollout step's kernels (cn_prof_set_stamps / cn_prof_next_step): a ring of `steps` rows, one slot per
    kernel; a kernel stamps the 100 MHz device clock when it starts and when its last wavefront ends.  Process-wide, one at a time.

        st = StepStamps(steps, ("hh_fused", "rn_fused")); for ...: st.next(); <enqueue one step>; ...; torch.cuda.synchronize(); st.close()
        st.durations_us("hh_fused") -> per-step kernel durations;  st.table() -> start / end of every stamped kernel relative to row 0
    """

    def __init__(self, steps, kernels=tuple(A.PROF_KERNEL_IDS), device=None):
        _need_cuda()
        self.steps = int(steps)
        self.kernels = tuple(kernels)
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        ring = torch.zeros(self.steps, A.PROF_KERNELS, A.PROF_SLOT_WORDS, dtype=torch.int64, device=dev)
        ring[:, :, 0:1024:16] = -1      # start candidates (word 16 b of the first half): unsigned minimum counts
        self.ring = ring
        mask = 0
        for k in self.kernels:
            mask |= 1 << A.PROF_KERNEL_IDS[k]
        torch.cuda.synchronize(dev)
        A.check(A.lib().cn_prof_set_stamps(A.ptr(ring), self.steps, mask), "cn_prof_set_stamps")
        self._open = True

    def next(self):
        return A.lib().cn_prof_next_step()

    def close(self):
        if self._open:
            A.check(A.lib().cn_prof_set_stamps(None, 0, 0), "cn_prof_set_stamps")
            self._open = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _host(self):
        r = self.ring.cpu().numpy()
        import numpy as np
        t0 = r[:, :, 0:1024:16].astype(np.uint64).min(axis=2)          # all ones = never stamped
        t1 = r[:, :, 1024:2048:16].astype(np.uint64).max(axis=2)
        return t0, t1, r[:, :, 1]

    def durations_us(self, kernel):
        """Per stamped step: (last wavefront's end - first workgroup's start) of `kernel` in microseconds (10 ns resolution)."""
        import numpy as np
        t0, t1, _ = self._host()
        k = A.PROF_KERNEL_IDS[kernel]
        ok = (t1[:, k] > 0) & (t0[:, k] != np.uint64(0xFFFFFFFFFFFFFFFF))
        return ((t1[ok, k] - t0[ok, k]).astype(np.float64) * 0.01).tolist()

    def counts(self, kernel):
        import numpy as np
        t0, t1, c = self._host()
        k = A.PROF_KERNEL_IDS[kernel]
        ok = (t1[:, k] > 0) & (t0[:, k] != np.uint64(0xFFFFFFFFFFFFFFFF))
        return c[ok, k].tolist()

    def table(self):
        """[(step, kernel, start_us, end_us)] of every stamped launch, relative to the earliest stamp, sorted by start."""
        import numpy as np
        t0, t1, _ = self._host()
        rows = []
        valid = (t1 > 0) & (t0 != np.uint64(0xFFFFFFFFFFFFFFFF))
        if not valid.any():
            return rows
        base = t0[valid].min()
        names = {v: k for k, v in A.PROF_KERNEL_IDS.items()}
        for s_ in range(self.steps):
            for k in range(A.PROF_KERNELS):
                if valid[s_, k]:
                    rows.append((s_, names[k], float(t0[s_, k] - base) * 0.01, float(t1[s_, k] - base) * 0.01))
        rows.sort(key=lambda r: r[2])
        return rows


class HipGST:
    """cn_gst handle: GST predictor (cn_gst_predict) and the VecPretextNormalize processing (cn_gst_wrapper_*)."""

    def __init__(self, human_num, max_envs, device=None):
        _need_cuda()
        self.H, self.maxE = int(human_num), int(max_envs)
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_create(self.H, self.maxE, C.byref(h)), "cn_gst_create")
        self._h = h
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            A.lib().cn_gst_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, state_dict):
        w = A.GstWeights()
        keep = []
        for field, key in A.GST_WEIGHT_KEYS:
            t = state_dict[key].detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            setattr(w, field, t.data_ptr())
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_set_weights(self._h, C.byref(w), A.stream_ptr()), "cn_gst_set_weights")
        self._keep = keep

    def predict(self, in_traj, in_mask):
        """in_traj [E,H,5,2], in_mask [E,H,5] or [E,H,5,1] float -> (out_traj [E,H,5,5], out_mask [E,H,1])."""
        E = in_traj.shape[0]
        out = torch.empty(E, self.H, 5, 5, device=self.device)
        om = torch.empty(E, self.H, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_predict(self._h, E, A.ptr(in_traj.float().contiguous()), A.ptr(in_mask.float().reshape(E, self.H, 5).contiguous()),
                                           A.ptr(out), A.ptr(om), A.stream_ptr()), "cn_gst_predict")
        return out, om.unsqueeze(-1)

    def wrapper_reset(self, E):
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_wrapper_reset(self._h, int(E), A.stream_ptr()), "cn_gst_wrapper_reset")
        self._wrap_E = int(E)

    def wrapper_set_interval(self, pred_interval):
        """int(data.pred_timestep // env.time_step): the history keeps 4 * pred_interval + 1 observations, every pred_interval-th is fed."""
        A.check(A.lib().cn_gst_wrapper_set_interval(self._h, int(pred_interval)), "cn_gst_wrapper_set_interval")

    def wrapper_state(self):
        """(traj [len,E,H,2] float32, mask [len,E,H] uint8) = the observation history in time order, oldest first (cn_gst_wrapper_save)."""
        L, E = int(A.lib().cn_gst_wrapper_history_len(self._h)), self._wrap_E
        traj = torch.empty(L, E, self.H, 2, device=self.device)
        mask = torch.empty(L, E, self.H, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_wrapper_save(self._h, A.ptr(traj), A.ptr(mask), A.stream_ptr()), "cn_gst_wrapper_save")
        return traj, mask

    def wrapper_load_state(self, traj, mask):
        L = int(A.lib().cn_gst_wrapper_history_len(self._h))
        if traj.shape[0] != L or tuple(traj.shape[2:]) != (self.H, 2) or tuple(mask.shape) != tuple(traj.shape[:3]):
            raise A.CnError("history of shape %s / %s does not fit this wrapper (length %d, %d humans)" % (tuple(traj.shape), tuple(mask.shape), L, self.H))
        E = int(traj.shape[1])
        traj = traj.to(device=self.device, dtype=torch.float32).contiguous()
        mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_wrapper_load(self._h, E, A.ptr(traj), A.ptr(mask), A.stream_ptr()), "cn_gst_wrapper_load")
        torch.cuda.current_stream(self.device).synchronize()   # the sources are temporaries
        self._wrap_E = E

    def wrapper_step(self, obs, rewards, dist, collision_penalty, out=None):
        """obs: raw env observation (spatial_edges [E,H,12] by human id, visible_masks u8/bool); rewards [E] float32 updated in place."""
        E = obs["robot_node"].shape[0]
        if out is None:
            out = torch.empty(E, self.H, 12, device=self.device)
        o = A.Obs()
        o.robot_node = A.ptr(obs["robot_node"])
        o.spatial_edges = A.ptr(obs["spatial_edges"])
        vm = obs["visible_masks"]
        vm = vm.view(torch.uint8) if vm.dtype == torch.bool else vm
        o.visible_masks = A.ptr(vm.contiguous())
        with torch.cuda.device(self.device):
            A.check(A.lib().cn_gst_wrapper_step(self._h, E, C.byref(o), float(dist), float(collision_penalty), A.ptr(rewards), A.ptr(out), A.stream_ptr()),
                    "cn_gst_wrapper_step")
        return out


class HHAttention(torch.autograd.Function):
    """softmax(scale * q k^T) v per (sample, head) on compacted rows, forward and backward as HIP kernels
    (cn_hh_attention_fwd / cn_hh_attention_bwd).  qkv [R,1536] float32 cuda, row_off [B+1] int32 cuda."""

    @staticmethod
    def forward(ctx, qkv, row_off, B, H, scale):
        qkv = qkv.contiguous()
        out = torch.empty(qkv.shape[0], 512, device=qkv.device)
        cls = torch.empty(int(A.lib().cn_hh_attention_workspace_ints(int(B))), dtype=torch.int32, device=qkv.device)   # size-class lists
        A.check(A.lib().cn_hh_attention_fwd(int(B), int(H), A.ptr(qkv), A.ptr(row_off), float(scale), A.ptr(out), A.ptr(cls), A.stream_ptr()),
                "cn_hh_attention_fwd")
        ctx.save_for_backward(qkv, row_off, cls)
        ctx.meta = (int(B), int(H), float(scale))
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, row_off, cls = ctx.saved_tensors
        B, H, scale = ctx.meta
        d_qkv = torch.empty_like(qkv)
        A.check(A.lib().cn_hh_attention_bwd(B, H, A.ptr(qkv), A.ptr(row_off), A.ptr(d_out.contiguous()), scale, A.ptr(d_qkv), A.ptr(cls), 1,
                                            A.stream_ptr()), "cn_hh_attention_bwd")
        return d_qkv, None, None, None, None


class HRAttention(torch.autograd.Function):
    """Robot-human attention on compacted rows (cn_hr_attention_fwd / cn_hr_attention_bwd): u [B,256] = Ws^T t (the
    spatial_edge_layer projection moved to the robot side, see include/crowdnav_hip.h), o [R,256], row_off [B+1] int32
    -> hr [B,256]."""

    @staticmethod
    def forward(ctx, u, o, row_off, H):
        u, o = u.contiguous(), o.contiguous()
        B = u.shape[0]
        hr = torch.empty(B, 256, device=u.device)
        attn = torch.empty(B, H, device=u.device)
        A.check(A.lib().cn_hr_attention_fwd(B, int(H), A.ptr(u), A.ptr(o), A.ptr(row_off), A.ptr(hr), A.ptr(attn), A.stream_ptr()), "cn_hr_attention_fwd")
        ctx.save_for_backward(u, o, row_off, attn)
        ctx.H = int(H)
        return hr

    @staticmethod
    def backward(ctx, d_hr):
        u, o, row_off, attn = ctx.saved_tensors
        d_u, d_o = torch.empty_like(u), torch.empty_like(o)
        A.check(A.lib().cn_hr_attention_bwd(u.shape[0], ctx.H, A.ptr(u), A.ptr(o), A.ptr(row_off), A.ptr(attn), A.ptr(d_hr.contiguous()), A.ptr(d_u), A.ptr(d_o),
                                            A.stream_ptr()), "cn_hr_attention_bwd")
        return d_u, d_o, None, None


class GRUSequence(torch.autograd.Function):
    """The human-node GRU over a [T,N] rollout slice with the done mask applied to h before every step (the reference's
    split-at-done trick, rl/networks/srnn_model.py:52-104, is arithmetically this).  gi [T,N,384] = x W_ih^T + b_ih,
    h0 [N,128], m [T,N,1] -> hs [T,N,128].  ONE launch for the whole forward sequence and one for the backward
    (cn_gru_seq_fwd / cn_gru_seq_bwd: W_hh resident in registers as bf16 hi / lo fragments, bf16x3 MFMA); the weight gradient of W_hh is one
    product over all T*N rows at the end."""

    @staticmethod
    def forward(ctx, gi, h0, m, w_hh, b_hh):
        T, N = gi.shape[0], gi.shape[1]
        gi, h0, m = gi.contiguous(), h0.contiguous(), m.contiguous()
        w, b = w_hh.detach().contiguous(), b_hh.detach().contiguous()
        hs = torch.empty(T, N, 128, device=gi.device)
        hms = torch.empty(T, N, 128, device=gi.device)
        gates = torch.empty(T, N, 512, device=gi.device)
        A.check(A.lib().cn_gru_seq_fwd(T, N, A.ptr(gi), A.ptr(h0), A.ptr(m), A.ptr(w), A.ptr(b), A.ptr(hs), A.ptr(hms), A.ptr(gates), A.stream_ptr()),
                "cn_gru_seq_fwd")
        ctx.save_for_backward(hms, gates, m, w)
        return hs

    @staticmethod
    def backward(ctx, d_hs):
        hms, gates, m, w = ctx.saved_tensors
        T, N = hms.shape[0], hms.shape[1]
        d_hs = d_hs.contiguous()
        dgi = torch.empty(T, N, 384, device=hms.device)
        dgh = torch.empty(T, N, 384, device=hms.device)
        dh0 = torch.empty(N, 128, device=hms.device)
        A.check(A.lib().cn_gru_seq_bwd(T, N, A.ptr(gates), A.ptr(hms), A.ptr(m), A.ptr(w), A.ptr(d_hs), A.ptr(dgi), A.ptr(dgh), A.ptr(dh0), A.stream_ptr()),
                "cn_gru_seq_bwd")
        # d(W_hh) = d(gh)^T hms and d(b_hh) = column sums of d(gh): a reduction over all T*N rows into a 384 x 128 matrix, which the
        # library product leaves on a dozen workgroups (287 us) -- the split-K weight-gradient kernel does both in one pass
        dw, db = wgrad(dgh.view(T * N, 384), hms.view(T * N, 128))
        return dgi, dh0, None, dw, db


def split_bf16(w, transpose=False):
    """fp32 matrix -> (hi, lo) bf16 planes (int16 storage) of w, or of w^T when transpose is set."""
    w = w.contiguous()
    rows, cols = w.shape
    shape = (cols, rows) if transpose else (rows, cols)
    hi = torch.empty(shape, dtype=torch.int16, device=w.device)
    lo = torch.empty(shape, dtype=torch.int16, device=w.device)
    A.check(A.lib().cn_split_bf16(A.ptr(w), rows, cols, int(bool(transpose)), A.ptr(hi), A.ptr(lo), A.stream_ptr()), "cn_split_bf16")
    return hi, lo


def linear_act(x, w, b, act, aux=None, relu_from=None, pad_to=0):
    """act(x w^T + b) on the split-precision kernel with the robot-node sequence's epilogues (cn_linear_fwd_act): act 0 none, 1 ReLU, 2 tanh,
    3 = times [aux > 0], 4 = times (1 - aux^2); columns >= relu_from also get a ReLU; pad_to = the weight's row count padded with zero rows
    to a multiple of 128 (the result then has pad_to columns).  x [M,K] (row stride may exceed K), w [N,K]."""
    M, K = x.shape
    N = w.shape[0]
    Np = pad_to or N

    def rows_ptr(t):   # a column slice of a wider row-major buffer: unit column stride, any row stride
        if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
            raise A.CnError("linear_act: operands must be fp32 GPU matrices with unit column stride")
        return C.c_void_p(t.data_ptr())
    w = w.detach().contiguous()
    hi = torch.empty(Np, K, dtype=torch.int16, device=x.device)
    lo = torch.empty(Np, K, dtype=torch.int16, device=x.device)
    A.check(A.lib().cn_split_bf16_padded(A.ptr(w), N, K, 0, Np if pad_to else 0, A.ptr(hi), A.ptr(lo), A.stream_ptr()), "cn_split_bf16_padded")
    bias = None
    if b is not None:
        bias = torch.zeros(Np, device=x.device)
        bias[:N] = b.detach()
    y = torch.empty(M, Np, device=x.device)
    A.check(A.lib().cn_linear_fwd_act(M, Np, K, rows_ptr(x), x.stride(0), A.ptr(hi), A.ptr(lo), A.ptr(bias) if bias is not None else None, int(act),
                                      rows_ptr(aux) if aux is not None else None, aux.stride(0) if aux is not None else 0,
                                      int(relu_from) if relu_from is not None else 1 << 30, A.ptr(y), Np, A.stream_ptr()), "cn_linear_fwd_act")
    return y


def linear_supported(x, w):
    """Shapes the split-precision training kernels cover (the three large human-human Linear layers)."""
    N, K = w.shape
    return x.is_cuda and x.dtype == torch.float32 and N % 128 == 0 and K % 128 == 0


class Embed0(torch.autograd.Function):
    """relu(x w^T + b) for the D -> 128 input embedding of the human-human block (cn_embed0_fwd / cn_embed0_bwd)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        R, D = x.shape
        y = torch.empty(R, 128, device=x.device)
        if R:
            A.check(A.lib().cn_embed0_fwd(R, D, A.ptr(x), A.ptr(w.detach().contiguous()), A.ptr(b.detach().contiguous()), A.ptr(y), A.stream_ptr()), "cn_embed0_fwd")
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        R, D = x.shape
        if R == 0:
            return None, x.new_zeros(128, D), x.new_zeros(128)
        blocks = min(R, 4096)   # 16 resident blocks of 128 threads per CU: each walks its rows 8 at a time (one round trip per trip)
        part = torch.empty(blocks, 128, D + 1, device=x.device)
        dwb = torch.empty(128, D + 1, device=x.device)
        A.check(A.lib().cn_embed0_bwd(R, D, A.ptr(x), A.ptr(y), A.ptr(dy.contiguous()), blocks, A.ptr(part), A.ptr(dwb), A.stream_ptr()), "cn_embed0_bwd")
        return None, dwb[:, :D].contiguous(), dwb[:, D].contiguous()


def wgrad(dy, x):
    """(dy^T x [N,K], column sums of dy [N]) over all M rows on the split-K TN kernel (cn_linear_wgrad); N % 64 == 0, K % 128 == 0."""
    dy, x = dy.contiguous(), x.contiguous()
    M, N = dy.shape
    K = x.shape[1]
    splits = A.lib().cn_linear_wgrad_splits(M, N, K)
    if splits <= 0:
        return dy.t() @ x, dy.sum(0)
    part = torch.empty(splits, N, K, device=x.device)
    dbp = torch.empty(splits, N, device=x.device)
    dw = torch.empty(N, K, device=x.device)
    db = torch.empty(N, device=x.device)
    A.check(A.lib().cn_linear_wgrad(M, N, K, A.ptr(dy), N, None, A.ptr(x), K, splits, A.ptr(part), A.ptr(dbp), A.ptr(dw), A.ptr(db), A.stream_ptr()),
            "cn_linear_wgrad")
    return dw, db


def _small_mm(a, b):
    """a [M,K] @ b [K,N] (any strides, fp32, on the GPU) through cn_small_mm -> contiguous [M,N]."""
    M, K = a.shape
    K2, N = b.shape
    assert K == K2
    if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32):
        raise A.CnError("cn_small_mm: float32 tensors on the GPU expected")
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        # (raw pointers + element strides: the operands are views -- transposes, row slices of in_proj_weight -- by design)
        A.check(A.lib().cn_small_mm(M, N, K, C.c_void_p(a.data_ptr()), a.stride(0), a.stride(1), C.c_void_p(b.data_ptr()), b.stride(0), b.stride(1),
                                    A.ptr(out), A.stream_ptr()), "cn_small_mm")
    return out


class SmallMM(torch.autograd.Function):
    """a @ b for WEIGHT-sized operands (the affine folds of the update and their backward): forward and both gradients are cn_small_mm
    launches -- transposes are strides, a vector operand is an [n, 1] matrix.  Replaces the library's GEMM / GEMV kernels on the training path."""

    @staticmethod
    def forward(ctx, a, b):
        vec = b.dim() == 1
        b2 = b.unsqueeze(1) if vec else b
        a32, b32 = a.detach().float(), b2.detach().float()
        ctx.save_for_backward(a32, b32)
        ctx.vec = vec
        out = _small_mm(a32, b32)
        return out.squeeze(1) if vec else out

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = (dc.unsqueeze(1) if ctx.vec else dc).float()
        da = _small_mm(dc, b.t()) if ctx.needs_input_grad[0] else None
        db = _small_mm(a.t(), dc) if ctx.needs_input_grad[1] else None
        if db is not None and ctx.vec:
            db = db.squeeze(1)
        return da, db


def weight_mm(a, b):
    """a @ b between parameters / their slices: cn_small_mm on the GPU (fp32), torch on the CPU."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32:
        return SmallMM.apply(a, b)
    return a @ b


class RightMatmul(torch.autograd.Function):
    """t @ w for a tall t [M,N] and a small w [N,K]: the weight gradient t^T d(out) (a reduction over all M rows) on the split-K
    TN kernel; forward and d(t) are ordinary library products."""

    @staticmethod
    def forward(ctx, t, w):
        ctx.save_for_backward(t, w)
        return t @ w

    @staticmethod
    def backward(ctx, du):
        t, w = ctx.saved_tensors
        dt = du @ w.t() if ctx.needs_input_grad[0] else None
        dw = wgrad(t, du)[0] if ctx.needs_input_grad[1] else None
        return dt, dw


def wgrad_supported(x, w):
    N, K = w.shape
    return x.is_cuda and x.dtype == torch.float32 and N % 64 == 0 and K % 128 == 0


class WgradLinear(torch.autograd.Function):
    """y = x w^T + b for the per-sample layers (a few hundred output columns, tens of thousands of rows): forward and input
    gradient are ordinary library products; the weight/bias gradient, a reduction over all rows into a tiny matrix that the
    BLAS heuristics leave on a handful of workgroups, runs on the split-K TN kernel (cn_linear_wgrad)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        x, dy = x.contiguous(), dy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = dy @ w if ctx.needs_input_grad[0] else None
        splits = A.lib().cn_linear_wgrad_splits(M, N, K)
        part = torch.empty(splits, N, K, device=x.device)
        dbp = torch.empty(splits, N, device=x.device)
        dw = torch.empty(N, K, device=x.device)
        db = torch.empty(N, device=x.device)
        A.check(A.lib().cn_linear_wgrad(M, N, K, A.ptr(dy), N, None, A.ptr(x), K, splits, A.ptr(part), A.ptr(dbp), A.ptr(dw), A.ptr(db), A.stream_ptr()),
                "cn_linear_wgrad")
        return dx, dw, db


class HipLinear(torch.autograd.Function):
    """y = [relu](x w^T + b) with forward, input gradient and weight/bias gradient on the bf16x3 MFMA kernels
    (cn_linear_fwd / cn_linear_wgrad).  x [M,K], w [N,K], b [N]; N, K multiples of 128."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=x.device)
        if M:
            hi, lo = split_bf16(w.detach())
            A.check(A.lib().cn_linear_fwd(M, N, K, A.ptr(x), K, None, A.ptr(hi), A.ptr(lo), A.ptr(b.detach().contiguous()), int(bool(relu)), A.ptr(y), N,
                                          A.stream_ptr()), "cn_linear_fwd")
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dy = dy.contiguous()
        gate = A.ptr(y) if ctx.relu else None          # ReLU backward is applied while dY is loaded (no masking pass)
        dx = dw = db = None
        if M == 0:
            return torch.zeros_like(x), torch.zeros_like(w), x.new_zeros(N), None
        if ctx.needs_input_grad[0]:
            hi_t, lo_t = split_bf16(w.detach(), transpose=True)              # [K,N]: dX = dY W as an NT product with W^T
            dx = torch.empty(M, K, device=x.device)
            A.check(A.lib().cn_linear_fwd(M, K, N, A.ptr(dy), N, gate, A.ptr(hi_t), A.ptr(lo_t), None, 0, A.ptr(dx), K, A.stream_ptr()), "cn_linear_fwd(dX)")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            splits = A.lib().cn_linear_wgrad_splits(M, N, K)
            part = torch.empty(splits, N, K, device=x.device)
            dbp = torch.empty(splits, N, device=x.device)
            dw = torch.empty(N, K, device=x.device)
            db = torch.empty(N, device=x.device)
            A.check(A.lib().cn_linear_wgrad(M, N, K, A.ptr(dy), N, gate, A.ptr(x), K, splits, A.ptr(part), A.ptr(dbp), A.ptr(dw), A.ptr(db), A.stream_ptr()),
                    "cn_linear_wgrad")
        return dx, dw, db, None


class HHBlockFused(torch.autograd.Function):
    """The whole human-human block of evaluate_actions (embedding_layer -> folded q|k|v -> 8-head attention -> folded
    out_proj∘spatial_linear) on the compacted live rows: forward = ONE launch of the rollout's fused kernel on the training weights
    (cn_hh_block_fwd), which writes every activation the backward needs exactly once; backward = the per-layer kernels of HipLinear /
    HHAttention / Embed0 on those activations, in reverse order.  spatial_edges [B,H,D], x_live [R,D] (its live rows, for the input
    layer's weight gradient), row_off [B+1] int32; weights as the mirror composes them (qkv / os folded, q unscaled)."""

    @staticmethod
    def forward(ctx, spatial_edges, x_live, row_off, emb0_w, emb0_b, emb2_w, emb2_b, qkv_w, qkv_b, os_w, os_b):
        B, H, D = spatial_edges.shape
        R = x_live.shape[0]
        dev = spatial_edges.device
        se = spatial_edges.contiguous()
        ws = torch.empty(int(A.lib().cn_hh_block_workspace_bytes()), dtype=torch.uint8, device=dev)
        e0, x, qkv = torch.empty(R, 128, device=dev), torch.empty(R, 512, device=dev), torch.empty(R, 1536, device=dev)
        attn, out = torch.empty(R, 512, device=dev), torch.empty(R, 256, device=dev)
        ws_w = [t.detach().contiguous() for t in (emb0_w, emb0_b, emb2_w, emb2_b, qkv_w, qkv_b, os_w, os_b)]
        A.check(A.lib().cn_hh_block_fwd(B, H, D, A.ptr(se), A.ptr(row_off), *[A.ptr(t) for t in ws_w], 0.125, A.ptr(ws), A.ptr(e0), A.ptr(x), A.ptr(qkv),
                                        A.ptr(attn), A.ptr(out), A.stream_ptr()), "cn_hh_block_fwd")
        ctx.save_for_backward(x_live, row_off, e0, x, qkv, attn, out, emb2_w, qkv_w, os_w)
        ctx.meta = (B, H, D)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x_live, row_off, e0, x, qkv, attn, out, emb2_w, qkv_w, os_w = ctx.saved_tensors
        B, H, D = ctx.meta
        R = x_live.shape[0]
        L = A.lib()
        dev = d_out.device

        def layer_bwd(dy, gate, w, inp):
            """Linear (+ReLU when `gate` = its output) backward: dX = (dY * [gate > 0]) W, dW = (dY * [gate > 0])^T X, db = column sums."""
            M, N = dy.shape
            K = w.shape[1]
            g = A.ptr(gate) if gate is not None else None
            hi_t, lo_t = split_bf16(w.detach(), transpose=True)
            dx = torch.empty(M, K, device=dev)
            A.check(L.cn_linear_fwd(M, K, N, A.ptr(dy), N, g, A.ptr(hi_t), A.ptr(lo_t), None, 0, A.ptr(dx), K, A.stream_ptr()), "cn_linear_fwd(dX)")
            splits = L.cn_linear_wgrad_splits(M, N, K)
            part, dbp = torch.empty(splits, N, K, device=dev), torch.empty(splits, N, device=dev)
            dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            A.check(L.cn_linear_wgrad(M, N, K, A.ptr(dy), N, g, A.ptr(inp), K, splits, A.ptr(part), A.ptr(dbp), A.ptr(dw), A.ptr(db), A.stream_ptr()), "cn_linear_wgrad")
            return dx, dw, db

        d_out = d_out.contiguous()
        if R == 0:
            z = lambda t: torch.zeros_like(t)   # noqa: E731
            return (None, None, None, x_live.new_zeros(128, D), x_live.new_zeros(128), z(emb2_w), x_live.new_zeros(512), z(qkv_w), x_live.new_zeros(1536),
                    z(os_w), x_live.new_zeros(256))
        d_attn, d_os_w, d_os_b = layer_bwd(d_out, out, os_w, attn)                       # out = relu(attn Wos^T + b)
        d_qkv = torch.empty_like(qkv)
        cls = torch.empty(int(L.cn_hh_attention_workspace_ints(int(B))), dtype=torch.int32, device=dev)
        A.check(L.cn_hh_attention_bwd(B, H, A.ptr(qkv), A.ptr(row_off), A.ptr(d_attn), 0.125, A.ptr(d_qkv), A.ptr(cls), 0, A.stream_ptr()), "cn_hh_attention_bwd")
        d_x, d_qkv_w, d_qkv_b = layer_bwd(d_qkv, None, qkv_w, x)                         # qkv = x Wc^T + bc
        d_e0, d_emb2_w, d_emb2_b = layer_bwd(d_x, x, emb2_w, e0)                         # x = relu(e0 W2^T + b2)
        blocks = min(R, 4096)
        part = torch.empty(blocks, 128, D + 1, device=dev)
        dwb = torch.empty(128, D + 1, device=dev)
        A.check(L.cn_embed0_bwd(R, D, A.ptr(x_live), A.ptr(e0), A.ptr(d_e0), blocks, A.ptr(part), A.ptr(dwb), A.stream_ptr()), "cn_embed0_bwd")
        return (None, None, None, dwb[:, :D].contiguous(), dwb[:, D].contiguous(), d_emb2_w, d_emb2_b, d_qkv_w, d_qkv_b, d_os_w, d_os_b)


class RnSequence(torch.autograd.Function):
    """Everything behind the human-human block in evaluate_actions for a [T, N] rollout slice -- robot_linear, [u | encoder_linear], robot-human
    attention, edge_attention_embed, the GRU over the T steps with the done mask, the actor / critic trunks, critic_linear and the log-probability
    of the given actions -- as ONE call forward (cn_rn_seq_fwd) and ONE backward (cn_rn_seq_bwd).  robot_node [B,7], temporal [B,2], out_sp [R,256]
    (compacted rows of the human-human block), row_off [B+1] int32, h0 [N,128], masks [B], actions [B,2]; weights in the order of
    _abi.RN_WEIGHT_FIELDS as the mirror composes them (te = [Ws^T Wt ; encoder_linear], ac0 = (actor.0 ; critic.0) o output_linear).
    Returns value [B,1], logp [B,1] and the final hidden state [N,128] (not differentiable)."""

    @staticmethod
    def forward(ctx, robot_node, temporal, out_sp, row_off, h0, masks, actions, T, N, H, *weights):
        B, dev = T * N, out_sp.device
        f = lambda t: t.detach().to(torch.float32).contiguous()   # noqa: E731
        rn, te, osp, h0c, m, act = f(robot_node).view(B, 7), f(temporal).view(B, 2), f(out_sp), f(h0).view(N, 128), f(masks).view(B), f(actions).view(B, 2)
        ws = [f(w) for w in weights]
        for w, shp, name in zip(ws, A.RN_WEIGHT_SHAPES, A.RN_WEIGHT_FIELDS):
            if tuple(w.shape) != shp:
                raise A.CnError("RnSequence: weight %s has shape %s, expected %s" % (name, tuple(w.shape), shp))
        wst = A.RnWeights(*[w.data_ptr() for w in ws])
        widths = dict(rs=256, z=384, hr=256, attn=H, gi=384, hs=128, hms=128, gates=512, a1=512, a2=512)
        saved = {k: torch.empty(B, widths[k], device=dev) for k in A.RN_SAVED_FIELDS}
        sst = A.RnSaved(*[saved[k].data_ptr() for k in A.RN_SAVED_FIELDS])
        value, logp = torch.empty(B, 1, device=dev), torch.empty(B, 1, device=dev)
        work = torch.empty(int(A.lib().cn_rn_seq_fwd_workspace_floats()), device=dev)
        A.check(A.lib().cn_rn_seq_fwd(T, N, H, A.ptr(rn), A.ptr(te), A.ptr(osp), A.ptr(row_off), A.ptr(h0c), A.ptr(m), A.ptr(act), C.byref(wst), C.byref(sst),
                                      A.ptr(work), A.ptr(value), A.ptr(logp), A.stream_ptr()), "cn_rn_seq_fwd")
        ctx.save_for_backward(rn, te, osp, row_off, m, act, *ws, *[saved[k] for k in A.RN_SAVED_FIELDS])
        ctx.meta = (T, N, H, len(ws))
        h_last = saved["hs"][(T - 1) * N:].clone()
        ctx.mark_non_differentiable(h_last)
        return value, logp, h_last

    @staticmethod
    def backward(ctx, d_value, d_logp, _d_h):
        T, N, H, nw = ctx.meta
        t = ctx.saved_tensors
        rn, te, osp, row_off, m, act = t[:6]
        ws, sv = t[6:6 + nw], t[6 + nw:]
        B, dev = T * N, osp.device
        wst = A.RnWeights(*[w.data_ptr() for w in ws])
        sst = A.RnSaved(*[x.data_ptr() for x in sv])
        grads = [torch.empty_like(w) for w in ws]
        gst = A.RnWeights(*[g.data_ptr() for g in grads])
        work = torch.empty(int(A.lib().cn_rn_seq_workspace_floats(T, N)), device=dev)
        d_osp, d_h0 = torch.empty_like(osp), torch.empty(N, 128, device=dev)
        dv = d_value.reshape(B).to(torch.float32).contiguous()
        dl = d_logp.reshape(B).to(torch.float32).contiguous()
        A.check(A.lib().cn_rn_seq_bwd(T, N, H, A.ptr(rn), A.ptr(te), A.ptr(osp), A.ptr(row_off), A.ptr(m), A.ptr(act), C.byref(wst), C.byref(sst), A.ptr(dv), A.ptr(dl),
                                      A.ptr(work), A.ptr(d_osp), A.ptr(d_h0), C.byref(gst), A.stream_ptr()), "cn_rn_seq_bwd")
        return (None, None, d_osp, None, None, None, None, None, None, None, *grads)


class PPOLoss(torch.autograd.Function):
    """(value_loss, action_loss) of rl/ppo/ppo.py:66-84 as one tensor [2]: forward = cn_ppo_loss_fwd, backward =
    cn_ppo_loss_bwd (gradients w.r.t. `values` and `logp` only -- everything else is rollout data)."""

    @staticmethod
    def forward(ctx, values, logp, old_logp, adv, value_preds, returns, clip_param, use_clipped_value_loss):
        ts = [t.reshape(-1).contiguous() for t in (values, logp, old_logp, adv, value_preds, returns)]
        n = ts[0].numel()
        if any(t.numel() != n or t.dtype != torch.float32 for t in ts):
            raise A.CnError("PPOLoss: all inputs must be float32 tensors of the same number of elements")
        ws = torch.empty(A.lib().cn_ppo_loss_workspace_doubles(), dtype=torch.float64, device=ts[0].device)
        losses = torch.empty(2, device=ts[0].device)
        A.check(A.lib().cn_ppo_loss_fwd(n, *[A.ptr(t) for t in ts], float(clip_param), int(bool(use_clipped_value_loss)), A.ptr(ws), A.ptr(losses),
                                        A.stream_ptr()), "cn_ppo_loss_fwd")
        ctx.save_for_backward(*ts)
        ctx.meta = (n, float(clip_param), int(bool(use_clipped_value_loss)), values.shape, logp.shape)
        return losses

    @staticmethod
    def backward(ctx, g):
        ts = ctx.saved_tensors
        n, clip, ucv, vshape, lshape = ctx.meta
        dv, dlp = torch.empty(n, device=g.device), torch.empty(n, device=g.device)
        A.check(A.lib().cn_ppo_loss_bwd(n, *[A.ptr(t) for t in ts], clip, ucv, A.ptr(g.contiguous()), A.ptr(dv), A.ptr(dlp), A.stream_ptr()),
                "cn_ppo_loss_bwd")
        return dv.view(vshape), dlp.view(lshape), None, None, None, None, None, None


def adam_clip_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas, eps, max_grad_norm, grad_scale=1.0, workspace=None, norm_out=None):
    """nn.utils.clip_grad_norm_ + torch.optim.Adam.step over one flat fp32 bucket (cn_adam_clip_step), in place."""
    n = param.numel()
    if workspace is None:
        workspace = torch.empty(A.lib().cn_adam_workspace_doubles(), dtype=torch.float64, device=param.device)
    A.check(A.lib().cn_adam_clip_step(n, A.ptr(param), A.ptr(grad), A.ptr(exp_avg), A.ptr(exp_avg_sq), float(grad_scale),
                                      float(max_grad_norm if max_grad_norm is not None else 0.0), float(lr), float(betas[0]), float(betas[1]),
                                      float(eps), int(step), A.ptr(workspace), A.ptr(norm_out), A.stream_ptr()), "cn_adam_clip_step")


def gae(rewards, values, masks, gamma, lam, returns):
    """rewards [T,N,1], values/masks/returns [T+1,N,1] contiguous float32 device tensors; fills returns[:T]."""
    T, N = rewards.shape[0], rewards.shape[1]
    A.check(A.lib().cn_gae(T, N, A.ptr(rewards), A.ptr(values), A.ptr(masks), float(gamma), float(lam), A.ptr(returns), A.stream_ptr()), "cn_gae")
    return returns


def adv_stats(returns, values, n):
    stats = torch.zeros(3, dtype=torch.float64, device=returns.device)
    A.check(A.lib().cn_adv_stats(int(n), A.ptr(returns), A.ptr(values), A.ptr(stats), A.stream_ptr()), "cn_adv_stats")
    return stats


def adv_normalize(returns, values, stats, n, out):
    A.check(A.lib().cn_adv_normalize(int(n), A.ptr(returns), A.ptr(values), A.ptr(stats), A.ptr(out), A.stream_ptr()), "cn_adv_normalize")
    return out


class MinibatchStepper:
    """One PPO minibatch -- gather from the rollout storage, train-mode forward, losses, backward, every parameter gradient written into the
    caller's flat bucket -- as ONE boundary call (cn_ppo_minibatch_step; rl/networks/storage.py:184-253 + rl/networks/model.py:82-90 +
    rl/ppo/ppo.py:66-88).  Built by ppo.PPO for the default network; the workspace is one cached byte tensor that grows with the row count."""

    def __init__(self, policy):
        self.policy = policy
        self._ws = None
        self._ptrs = None

    @staticmethod
    def supported(policy, rollouts):
        base = policy.base
        if not (base.use_self_attn and base.sort_humans and base.train_gemm_mode == "bf16x3" and base.train_fused_hh and base.train_fused_rn
                and base.fused_rn_shapes_ok() and base.human_num <= 48 and base.edge_width <= 16):
            return False
        if tuple(base.spatial_attn.embedding_layer[0].weight.shape) != (128, base.edge_width):
            return False
        need = ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")
        ts = [rollouts.obs.get(k) for k in need] + [rollouts.recurrent_hidden_states["human_node_rnn"], rollouts.masks, rollouts.actions,
                                                    rollouts.value_preds, rollouts.returns, rollouts.action_log_probs]
        return all(t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)

    def _structs(self):
        """Parameter / gradient pointer structs (views of the flat buckets ppo.PPO binds; re-read whenever a pointer moved)."""
        named = dict(self.policy.named_parameters())
        key = tuple((named[k].data_ptr(), named[k].grad.data_ptr()) for _, k in A.POLICY_WEIGHT_KEYS)
        if self._ptrs is None or self._ptrs[0] != key:
            w, g = A.PolicyWeights(), A.PolicyWeights()
            for field, k in A.POLICY_WEIGHT_KEYS:
                p = named[k]
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad is None:
                    raise A.CnError("MinibatchStepper: parameter %s must be a contiguous float32 GPU tensor with a bound gradient" % k)
                setattr(w, field, p.data_ptr())
                setattr(g, field, p.grad.data_ptr())
            self._ptrs = (key, w, g)
        return self._ptrs[1], self._ptrs[2]

    def row_totals(self, rollouts):
        """[E] int64 on the HOST: the compacted rows every env contributes to a minibatch (one readback per update())."""
        det = rollouts.obs["detected_human_num"]
        T, E = rollouts.rewards.shape[0], rollouts.rewards.shape[1]
        tot = torch.empty(E, dtype=torch.int32, device=det.device)
        with torch.cuda.device(det.device):
            A.check(A.lib().cn_ppo_row_totals(T, E, self.policy.base.human_num, A.ptr(det), A.ptr(tot), A.stream_ptr()), "cn_ppo_row_totals")
        return tot.cpu().to(torch.int64)

    def step(self, rollouts, advantages, env_idx, rows, hyper, losses_out, value_logp_out=None):
        """env_idx [N] int32 on the device, rows = their row total, hyper = (clip, value_loss_coef, entropy_coef, use_clipped_value_loss),
        losses_out [3] float32 on the device; value_logp_out (optional, tests): [2, T * N] float32 on the device."""
        base = self.policy.base
        T, E = rollouts.rewards.shape[0], rollouts.rewards.shape[1]
        N, H, D = int(env_idx.numel()), base.human_num, base.edge_width
        dev = rollouts.rewards.device
        L = A.lib()
        need = int(L.cn_ppo_minibatch_workspace_bytes(T, N, H, D, int(rows)))
        if need <= 0:
            raise A.CnError("cn_ppo_minibatch_workspace_bytes: unsupported shape T=%d N=%d H=%d D=%d rows=%d" % (T, N, H, D, rows))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = None                       # free the old block first: the two do not need to coexist
            self._ws = torch.empty(int(need * 1.1) + 4096, dtype=torch.uint8, device=dev)
        w, g = self._structs()
        b = A.PpoBatch(T, N, E, H, D)
        ts = dict(env_idx=env_idx, robot_node=rollouts.obs["robot_node"], temporal_edges=rollouts.obs["temporal_edges"],
                  spatial_edges=rollouts.obs["spatial_edges"], detected_human_num=rollouts.obs["detected_human_num"],
                  h0=rollouts.recurrent_hidden_states["human_node_rnn"], masks=rollouts.masks, actions=rollouts.actions,
                  value_preds=rollouts.value_preds, returns=rollouts.returns, old_logp=rollouts.action_log_probs, adv=advantages)
        for k in A.PPO_BATCH_TENSORS:
            setattr(b, k, ts[k].data_ptr())
        hy = A.PpoHyper(float(hyper[0]), float(hyper[1]), float(hyper[2]), int(bool(hyper[3])))
        with torch.cuda.device(dev):
            A.check(L.cn_ppo_minibatch_step(C.byref(b), int(rows), C.byref(w), C.byref(g), C.byref(hy), C.c_void_p(self._ws.data_ptr()), int(self._ws.numel()),
                                            A.ptr(losses_out), A.ptr(value_logp_out), A.stream_ptr()), "cn_ppo_minibatch_step")
