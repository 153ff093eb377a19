"""Policy -- host-side mirror of the reference's `rl.networks.model.Policy` (selfAttn_merge_srnn base + DiagGaussian).

Same constructor signature, `act` / `get_value` / `evaluate_actions` contracts and state-dict keys as
/root/reference/rl/networks/model.py:14-90 and rl/networks/selfAttn_srnn_temp_node.py:287-449, so reference
checkpoints load and `train.py` runs unchanged.  Two execution paths:

  * rollout (`act`, `get_value`, no autograd) on a GPU -> hand-written HIP kernels through the C ABI
    (cn_policy_act / cn_policy_get_value).  There is no fallback: on CUDA tensors a missing extension raises.
  * training (`evaluate_actions`, needs gradients) -> the same math expressed in torch ops so autograd provides the
    backward (hand-written backward kernels are the next step).  CPU tensors also take this path; it
    exists for the CPU unit tests and the reference's `--no-cuda` plumbing mode, never silently on a GPU box.

Parameter construction order follows the reference so that `torch.manual_seed(s); Policy(...)` yields bit-identical
initial weights (tests/test_host_policy.py checks this against checksums captured from the reference).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class Box:
    """Minimal gym.spaces.Box stand-in (gym is not a dependency of the hot path)."""

    def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Dict:
    def __init__(self, spaces):
        self.spaces = OrderedDict(sorted(spaces.items()))


def make_spaces(human_num, edge_width, with_masks=True):
    """Observation / action spaces of CrowdSimVarNum-v0 (crowd_sim_var_num.py:37-58) and the Pred variants."""
    d = {"robot_node": Box(shape=(1, 7)), "temporal_edges": Box(shape=(1, 2)), "spatial_edges": Box(shape=(human_num, edge_width)),
         "detected_human_num": Box(shape=(1,))}
    if with_masks:
        d["visible_masks"] = Box(shape=(human_num,), dtype=np.bool_)
    return Dict(d), Box(shape=(2,))


def _arg(args, name, default):
    if args is None:
        return default
    if isinstance(args, dict):
        return args.get(name, default)
    return getattr(args, name, default)


def _skinny_linear(x, w, b):
    """nn.Linear with one or two output columns over tens of thousands of rows (critic_linear, dist.fc_mean in the update).  On the GPU
    the library runs these skinny products, and above all their weight gradients, on a handful of workgroups (0.1-0.2 ms each); as
    row-wise multiply + reduce they are memory-bound passes over x, and autograd's backward of them is too."""
    if x.is_cuda and x.dim() == 2 and w.shape[0] <= 4 and x.shape[0] >= 4096:
        return torch.stack([(x * w[n]).sum(-1) for n in range(w.shape[0])], -1) + b
    return F.linear(x, w, b)


def _ortho(module, gain=1.0):
    nn.init.orthogonal_(module.weight.data, gain=gain)
    nn.init.constant_(module.bias.data, 0)
    return module


class _AddBias(nn.Module):
    """rl/networks/network_utils.py:31-42 -- keeps the checkpoint key `dist.logstd._bias` ([2,1])."""

    def __init__(self, bias):
        super().__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        return x + self._bias.t().view(1, -1)


class _DiagGaussian(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.fc_mean = _ortho(nn.Linear(num_inputs, num_outputs))
        self.logstd = _AddBias(torch.zeros(num_outputs))


class _EndRNN(nn.Module):
    def __init__(self, rnn_size, embedding_size, edge_rnn_size, output_size):
        super().__init__()
        self.gru = nn.GRU(embedding_size * 2, rnn_size)
        for name, param in self.gru.named_parameters():
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                nn.init.orthogonal_(param)
        self.encoder_linear = nn.Linear(256, embedding_size)
        self.edge_attention_embed = nn.Linear(edge_rnn_size, embedding_size)
        self.output_linear = nn.Linear(rnn_size, output_size)


class _EdgeAttention(nn.Module):
    def __init__(self, edge_rnn_size, attention_size):
        super().__init__()
        self.temporal_edge_layer = nn.ModuleList()
        self.spatial_edge_layer = nn.ModuleList()
        self.temporal_edge_layer.append(nn.Linear(edge_rnn_size, attention_size))
        self.spatial_edge_layer.append(nn.Linear(edge_rnn_size, attention_size))


class _SpatialSelfAttn(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.embedding_layer = nn.Sequential(nn.Linear(input_size, 128), nn.ReLU(), nn.Linear(128, 512), nn.ReLU())
        self.q_linear = nn.Linear(512, 512)
        self.v_linear = nn.Linear(512, 512)
        self.k_linear = nn.Linear(512, 512)
        self.multihead_attn = nn.MultiheadAttention(512, 8)


class AttnGraphBase(nn.Module):
    """The network body (`base`).  Hyper-parameters as in arguments.py:155-206."""

    def __init__(self, obs_space_dict, args):
        super().__init__()
        self.is_recurrent = True
        self.args = args
        self.human_num = obs_space_dict["spatial_edges"].shape[0]
        self.edge_width = obs_space_dict["spatial_edges"].shape[1]
        self.seq_length = _arg(args, "seq_length", 30)
        self.nenv = _arg(args, "num_processes", 16)
        self.nminibatch = _arg(args, "num_mini_batch", 2)
        self.human_node_rnn_size = _arg(args, "human_node_rnn_size", 128)
        self.human_human_edge_rnn_size = _arg(args, "human_human_edge_rnn_size", 256)
        self.output_size = _arg(args, "human_node_output_size", 256)
        emb = _arg(args, "human_node_embedding_size", 64)
        attention_size = _arg(args, "attention_size", 64)
        if (self.human_node_rnn_size, self.human_human_edge_rnn_size, self.output_size, emb, attention_size) != (128, 256, 256, 64, 64):
            raise NotImplementedError("the HIP kernels are specialised to the reference's network sizes (128/256/256/64/64)")
        # arguments.py:189, :206.  use_self_attn = False: no human-human attention, spatial_linear embeds the raw edges
        # (selfAttn_srnn_temp_node.py:340-345); sort_humans = False: attention masks from `visible_masks` instead of the detected count (:378-383)
        self.use_self_attn = bool(_arg(args, "use_self_attn", True))
        self.sort_humans = bool(_arg(args, "sort_humans", True))
        env_name = _arg(args, "env_name", None)
        expect = {"CrowdSimVarNum-v0": 2, "CrowdSimPred-v0": 12, "CrowdSimPredRealGST-v0": 12}.get(env_name)
        if expect is not None and _arg(args, "predict_steps", 5) == 5 and expect != self.edge_width:
            raise ValueError("env_name %s expects spatial edge width %d, observation space has %d" % (env_name, expect, self.edge_width))
        gain = math.sqrt(2)
        # construction order == reference (selfAttn_srnn_temp_node.py:309-343) so seeded inits are bit-identical
        self.humanNodeRNN = _EndRNN(self.human_node_rnn_size, emb, self.human_human_edge_rnn_size, self.output_size)
        self.attn = _EdgeAttention(self.human_human_edge_rnn_size, attention_size)
        h = self.output_size
        self.actor = nn.Sequential(_ortho(nn.Linear(h, h), gain), nn.Tanh(), _ortho(nn.Linear(h, h), gain), nn.Tanh())
        self.critic = nn.Sequential(_ortho(nn.Linear(h, h), gain), nn.Tanh(), _ortho(nn.Linear(h, h), gain), nn.Tanh())
        self.critic_linear = _ortho(nn.Linear(h, 1), gain)
        self.robot_linear = nn.Sequential(_ortho(nn.Linear(9, 256), gain), nn.ReLU())
        self.human_node_final_linear = _ortho(nn.Linear(self.output_size, 2), gain)
        if self.use_self_attn:
            self.spatial_attn = _SpatialSelfAttn(self.edge_width)
            self.spatial_linear = nn.Sequential(_ortho(nn.Linear(512, 256), gain), nn.ReLU())
        else:
            self.spatial_linear = nn.Sequential(_ortho(nn.Linear(self.edge_width, 128), gain), nn.ReLU(), _ortho(nn.Linear(128, 256), gain), nn.ReLU())

    # ---- training-time forward in torch ops (autograd) ----
    # arithmetic of the three large human-human Linear layers in the PPO update on the GPU: 'bf16x3' = split-precision MFMA
    # kernels (forward, dX, dW; same arithmetic as the rollout forward), 'fp32' = torch / rocBLAS fp32
    train_gemm_mode = "bf16x3"
    # the human-human block of the update's forward as one fused launch (hip.HHBlockFused; crowds of <= 48 humans) instead of five
    train_fused_hh = True
    # everything behind the human-human block (robot node, robot-human attention, GRU sequence, trunks, heads, log-prob) as ONE call forward and
    # ONE backward (hip.RnSequence: cn_rn_seq_fwd / cn_rn_seq_bwd) instead of torch modules with HIP Functions spliced in
    train_fused_rn = os.environ.get("CN_TRAIN_FUSED_RN", "1") != "0"   # (the environment switch is for A/B timing: bench.py's PPO leg)

    def counted_inputs(self, inputs):
        """sort_humans = True: the inputs as they are.  sort_humans = False (selfAttn_srnn_temp_node.py:378-383, :410-414): both attention modules
        mask by `visible_masks` (an all-invisible sample keeps human 0).  They are permutation-equivariant over the humans, and masked humans
        contribute exactly nothing (their keys are excluded, their rows meet an exactly-zero robot-human weight), so the masked form equals
        the counted form on the observation with the visible humans moved to the front -- stable, so that equal inputs give equal outputs."""
        if self.sort_humans:
            return inputs
        se = inputs["spatial_edges"]
        B = se.shape[0]
        se = se.reshape(B, self.human_num, self.edge_width)
        vm = inputs["visible_masks"].reshape(B, self.human_num)
        out = dict(inputs)
        if se.is_cuda:
            from .hip import compact_visible
            out["spatial_edges"], out["detected_human_num"] = compact_visible(se, vm)
        else:
            m = vm.to(torch.bool).clone()
            m[~m.any(1), 0] = True
            order = torch.argsort((~m).to(torch.int8), dim=1, stable=True)
            out["spatial_edges"] = torch.gather(se, 1, order.unsqueeze(-1).expand(-1, -1, self.edge_width))
            out["detected_human_num"] = m.sum(1, keepdim=True).to(se.dtype)
        return out

    def fused_rn_shapes_ok(self):
        """hip.RnSequence hard-codes every layer shape of the robot-node sequence (_abi.RN_WEIGHT_SHAPES) and computes its products as bf16x3
        splits: evaluate_actions takes it only when the modules have exactly those shapes and train_gemm_mode asks for that arithmetic
        ('fp32' keeps the torch / exact-fp32 path of forward_sequence)."""
        rnn, g = self.humanNodeRNN, self.humanNodeRNN.gru
        return (tuple(self.robot_linear[0].weight.shape) == (256, 9) and tuple(rnn.edge_attention_embed.weight.shape) == (64, 256)
                and tuple(rnn.encoder_linear.weight.shape) == (64, 256) and tuple(self.attn.temporal_edge_layer[0].weight.shape) == (64, 256)
                and tuple(self.attn.spatial_edge_layer[0].weight.shape) == (64, 256)
                and tuple(g.weight_ih_l0.shape) == (384, 128) and tuple(g.weight_hh_l0.shape) == (384, 128)
                and tuple(rnn.output_linear.weight.shape) == (256, 128) and tuple(self.actor[2].weight.shape) == (256, 256)
                and tuple(self.critic_linear.weight.shape) == (1, 256))

    def rn_sequence(self, inputs, out_sp, row_off, h0, masks, actions, T, N, dist):
        """(value [B,1], logp [B,1], h_T [N,128]) through hip.RnSequence.  The two affine pairs without a nonlinearity in between are composed
        here in torch ops (tiny products, autograd carries the folded gradients back to both factors): u = Ws^T (Wt r + bt) -- the
        spatial_edge_layer projection moved to the robot side, its bias keeps an exactly-zero gradient -- and (actor.0 ; critic.0) o output_linear."""
        from .hip import RnSequence
        B = T * N
        sl, tl, rnn = self.attn.spatial_edge_layer[0], self.attn.temporal_edge_layer[0], self.humanNodeRNN
        from .hip import weight_mm as mm     # weight-by-weight products: cn_small_mm forward and backward (no library GEMM / GEMV)
        te_w = torch.cat([mm(sl.weight.t(), tl.weight), rnn.encoder_linear.weight], 0)
        te_b = torch.cat([mm(sl.weight.t(), tl.bias) + 0.0 * sl.bias.sum(), rnn.encoder_linear.bias], 0)
        w0 = torch.cat([self.actor[0].weight, self.critic[0].weight], 0)
        ac0_w = mm(w0, rnn.output_linear.weight)
        ac0_b = mm(w0, rnn.output_linear.bias) + torch.cat([self.actor[0].bias, self.critic[0].bias], 0)
        g = rnn.gru
        return RnSequence.apply(inputs["robot_node"].reshape(B, 7), inputs["temporal_edges"].reshape(B, 2), out_sp, row_off, h0.reshape(N, -1), masks.reshape(B),
                                actions.reshape(B, 2), T, N, self.human_num,
                                self.robot_linear[0].weight, self.robot_linear[0].bias, te_w, te_b, rnn.edge_attention_embed.weight, rnn.edge_attention_embed.bias,
                                g.weight_ih_l0, g.bias_ih_l0, g.weight_hh_l0, g.bias_hh_l0, ac0_w, ac0_b, self.actor[2].weight, self.actor[2].bias,
                                self.critic[2].weight, self.critic[2].bias, self.critic_linear.weight, self.critic_linear.bias, dist.fc_mean.weight, dist.fc_mean.bias,
                                dist.logstd._bias)

    def _big_linear(self, x, w, b, relu=False):
        if x.is_cuda and self.train_gemm_mode == "bf16x3":
            from . import hip
            if hip.linear_supported(x, w):
                return hip.HipLinear.apply(x, w, b, relu)
        y = F.linear(x, w, b)
        return F.relu(y) if relu else y

    def _lin(self, x, w, b):
        """Per-sample Linear layer of the update: on the GPU the weight gradient (a reduction over all T*N rows into a small
        matrix) goes to the split-K TN kernel, everything else stays a library product."""
        if x.is_cuda and self.train_gemm_mode == "bf16x3":
            from . import hip
            if hip.linear_supported(x, w) and x.shape[0] >= 16384:
                return hip.HipLinear.apply(x, w, b, False)   # [128 a, 128 b] weights: forward, dX and dW on the pipelined bf16x3 kernels
            if hip.wgrad_supported(x, w):
                return hip.WgradLinear.apply(x, w, b)
        return F.linear(x, w, b)

    def _mlp2(self, seq, x):
        """Sequential(Linear, Tanh, Linear, Tanh) (actor / critic trunks)."""
        return torch.tanh(self._lin(torch.tanh(self._lin(x, seq[0].weight, seq[0].bias)), seq[2].weight, seq[2].bias))

    def _hh_block(self, spatial_edges, det):
        """[B,H,D] -> [B,H,256].  SpatialEdgeSelfAttn.forward + spatial_linear (selfAttn_srnn_temp_node.py:63-91,:408).

        Only the detected humans (index < det) are pushed through the linear layers: padded humans are masked as keys
        and their own outputs only ever meet an exactly-zero robot-human attention weight, so values and gradients are
        identical to the dense computation while the dominant GEMMs shrink by H / mean(det) (~3.4x at 20 humans).
        The tiny per-env attention itself runs on zero-padded [B,8,H,64] tensors."""
        B, H, D = spatial_edges.shape
        sa = self.spatial_attn if self.use_self_attn else None
        det = det.clamp(1, H)   # every sample has 1..H rows, as the env guarantees (crowd_sim_var_num.py:290-292) and the kernels assume
        valid = torch.arange(H, device=spatial_edges.device).view(1, H) < det.view(B, 1)     # key padding mask
        idx = valid.reshape(-1).nonzero(as_tuple=False).squeeze(1)                            # live (sample, human) rows
        x_live = spatial_edges.reshape(B * H, D).index_select(0, idx)
        if not self.use_self_attn:
            # selfAttn_srnn_temp_node.py:404-408 with use_self_attn = False: output_spatial = spatial_linear(spatial_edges), a two-layer MLP per
            # human; only the live rows are computed (the others only ever meet an exactly-zero robot-human weight)
            l0, l2 = self.spatial_linear[0], self.spatial_linear[2]
            if x_live.is_cuda and D <= 16:
                from .hip import Embed0
                e0 = Embed0.apply(x_live, l0.weight, l0.bias)
            else:
                e0 = F.relu(l0(x_live))
            o = self._big_linear(e0, l2.weight, l2.bias, relu=True)
            if o.is_cuda:
                nd = det.to(torch.int32)
                return o, torch.cat([nd.new_zeros(1), nd.cumsum(0, dtype=torch.int32)])
            return o.new_zeros(B * H, o.shape[1]).index_copy(0, idx, o).view(B, H, -1), valid
        emb0, emb2 = sa.embedding_layer[0], sa.embedding_layer[2]
        if x_live.is_cuda and self.train_gemm_mode == "bf16x3" and self.train_fused_hh and H <= 48 and D <= 16 and emb0.weight.shape[0] == 128:
            # ONE launch for the whole block (the rollout's fused kernel on the training weights, writing the activations the backward
            # needs); the affine pairs are composed exactly as below, so both factors still receive their exact gradients
            from .hip import HHBlockFused
            W, b = sa.multihead_attn.in_proj_weight, sa.multihead_attn.in_proj_bias
            lins = (sa.q_linear, sa.k_linear, sa.v_linear)
            from .hip import weight_mm as mm     # the folds and their backward as cn_small_mm launches
            Wc = torch.cat([mm(W[i * 512:(i + 1) * 512], lins[i].weight) for i in range(3)], 0)
            bc = torch.cat([mm(W[i * 512:(i + 1) * 512], lins[i].bias) + b[i * 512:(i + 1) * 512] for i in range(3)], 0)
            op, sl = sa.multihead_attn.out_proj, self.spatial_linear[0]
            nd = det.clamp(1, H).to(torch.int32)   # 1..H rows per sample, like the rollout kernels (crowd_sim_var_num.py:290-292)
            row_off = torch.cat([nd.new_zeros(1), nd.cumsum(0, dtype=torch.int32)])
            o = HHBlockFused.apply(spatial_edges, x_live, row_off, emb0.weight, emb0.bias, emb2.weight, emb2.bias, Wc, bc,
                                   mm(sl.weight, op.weight), mm(sl.weight, op.bias) + sl.bias)
            return o, row_off
        if x_live.is_cuda and D <= 16 and emb0.weight.shape[0] == 128:
            from .hip import Embed0
            e0 = Embed0.apply(x_live, emb0.weight, emb0.bias)
        else:
            e0 = F.relu(emb0(x_live))
        e = self._big_linear(e0, emb2.weight, emb2.bias, relu=True)
        # (q|k|v)_linear followed by in_proj is an affine pair with no nonlinearity in between: compose the two weight
        # matrices first (a 512^3 product, differentiable, so both factors still receive their exact gradients) and run ONE
        # [rows,512]x[512,1536] GEMM instead of six [rows,512]x[512,512] ones, in the forward and in the backward pass.
        W, b = sa.multihead_attn.in_proj_weight, sa.multihead_attn.in_proj_bias
        lins = (sa.q_linear, sa.k_linear, sa.v_linear)
        Wc = torch.cat([W[i * 512:(i + 1) * 512] @ lins[i].weight for i in range(3)], 0)
        bc = torch.cat([W[i * 512:(i + 1) * 512] @ lins[i].bias + b[i * 512:(i + 1) * 512] for i in range(3)], 0)
        qkv = self._big_linear(e, Wc, bc)                                                   # [rows, 1536] = [q | k | v]
        if qkv.is_cuda:
            # attention core on the compacted rows, forward AND backward as hand-written HIP kernels
            from .hip import HHAttention
            nd = det.clamp(1, H).to(torch.int32)   # 1..H rows per sample, like the rollout kernels (crowd_sim_var_num.py:290-292)
            row_off = torch.cat([nd.new_zeros(1), nd.cumsum(0, dtype=torch.int32)])
            o_live = HHAttention.apply(qkv, row_off, B, H, 0.125)
        else:
            # CPU tensors (unit tests): the same math on zero-padded [B,8,H,64] tensors in torch ops
            q, k, v = qkv.split(512, dim=-1)

            def pad(x):
                return x.new_zeros(B * H, 512).index_copy(0, idx, x).view(B, H, 8, 64).transpose(1, 2)

            scores = torch.matmul(pad(q), pad(k).transpose(-1, -2)) * 0.125
            scores = scores.masked_fill(~valid.view(B, 1, 1, H), float("-inf"))
            o = torch.matmul(torch.softmax(scores, dim=-1), pad(v)).transpose(1, 2).reshape(B * H, 512)
            o_live = o.index_select(0, idx)
        # same composition for out_proj followed by spatial_linear (Linear -> Linear -> ReLU)
        op, sl = sa.multihead_attn.out_proj, self.spatial_linear[0]
        o = self._big_linear(o_live, sl.weight @ op.weight, sl.weight @ op.bias + sl.bias, relu=True)
        if o.is_cuda:
            return o, row_off                     # stays compacted: the robot-human attention kernels index rows through row_off
        out_sp = o.new_zeros(B * H, o.shape[1]).index_copy(0, idx, o).view(B, H, -1)
        return out_sp, valid

    def _hr_attention(self, robot_states, out_sp, valid):
        """EdgeAttention_M (selfAttn_srnn_temp_node.py:145-223): [B,256],[B,H,256] -> [B,256]."""
        H = out_sp.shape[1]
        t = self.attn.temporal_edge_layer[0](robot_states)
        s = self.attn.spatial_edge_layer[0](out_sp)
        a = (t.unsqueeze(1) * s).sum(-1) * (H / math.sqrt(64.0))
        a = torch.softmax(a.masked_fill(~valid, -1e9), dim=-1)
        return torch.bmm(a.unsqueeze(1), out_sp).squeeze(1), a

    def _gru_cell(self, gi, h):
        g = self.humanNodeRNN.gru
        gh = F.linear(h, g.weight_hh_l0, g.bias_hh_l0)
        i_r, i_z, i_n = gi.chunk(3, -1)
        h_r, h_z, h_n = gh.chunk(3, -1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        return (1.0 - z) * n + z * h

    def forward_sequence(self, inputs, h0, masks, T, N):
        """inputs: dict of [T*N, ...] (T-major), h0 [N,1,128] or [N,128], masks [T*N,1].
        Returns value [T*N,1], actor features [T*N,256], h_T [N,128].  Masking h at every step is arithmetically the
        reference's split-at-done trick (rl/networks/srnn_model.py:52-104)."""
        B = T * N
        inputs = self.counted_inputs(inputs)
        robot_in = torch.cat((inputs["temporal_edges"].reshape(B, 2), inputs["robot_node"].reshape(B, 7)), dim=-1)
        robot_states = self.robot_linear(robot_in)
        det = inputs["detected_human_num"].reshape(B).to(torch.int64).clamp(min=1)
        out_sp, valid = self._hh_block(inputs["spatial_edges"].reshape(B, self.human_num, self.edge_width), det)
        if out_sp.is_cuda:
            # compacted rows [R,256] + row offsets: HIP robot-human attention forward/backward (no dense [B,H,256] tensors)
            # t . (Ws o_j + bs) = (Ws^T t) . o_j + const: the spatial_edge_layer projection moves to the robot side (B rows
            # instead of ~6B) and its bias, which the softmax cannot see, keeps an exactly-zero gradient
            from .hip import HRAttention
            sl = self.attn.spatial_edge_layer[0]
            tl = self.attn.temporal_edge_layer[0]
            from .hip import RightMatmul
            t_emb = self._lin(robot_states, tl.weight, tl.bias)
            u = (RightMatmul.apply(t_emb, sl.weight) if t_emb.shape[0] >= 4096 else t_emb @ sl.weight) + 0.0 * sl.bias.sum()
            hr = HRAttention.apply(u, out_sp, valid, self.human_num)
        else:
            hr, _ = self._hr_attention(robot_states, out_sp, valid)
        rnn = self.humanNodeRNN
        x = torch.cat((F.relu(self._lin(robot_states, rnn.encoder_linear.weight, rnn.encoder_linear.bias)),
                       F.relu(self._lin(hr, rnn.edge_attention_embed.weight, rnn.edge_attention_embed.bias))), dim=-1)
        gi = self._lin(x, rnn.gru.weight_ih_l0, rnn.gru.bias_ih_l0).view(T, N, -1)
        m = masks.reshape(T, N, 1)
        h = h0.reshape(N, -1)
        if gi.is_cuda:
            from .hip import GRUSequence
            hs_all = GRUSequence.apply(gi, h, m, rnn.gru.weight_hh_l0, rnn.gru.bias_hh_l0)
            h = hs_all[-1]
        else:
            hs = []
            for gi_t, m_t in zip(gi.unbind(0), m.unbind(0)):
                h = self._gru_cell(gi_t, h * m_t)
                hs.append(h)
            hs_all = torch.stack(hs, 0)
        out = self._lin(hs_all.view(B, -1), rnn.output_linear.weight, rnn.output_linear.bias)
        value = _skinny_linear(self._mlp2(self.critic, out), self.critic_linear.weight, self.critic_linear.bias)
        return value, self._mlp2(self.actor, out), h


class Policy(nn.Module):
    """Drop-in for rl.networks.model.Policy."""

    def __init__(self, obs_shape, action_space, base=None, base_kwargs=None):
        super().__init__()
        if base not in ("selfAttn_merge_srnn", None):
            raise NotImplementedError("only base='selfAttn_merge_srnn' is implemented (the DS-RNN baseline 'srnn' is out of scope)")
        if action_space.__class__.__name__ != "Box":
            raise NotImplementedError("only Box(2) action spaces (holonomic robot) are implemented")
        self.base = AttnGraphBase(obs_shape, base_kwargs)
        self.srnn = True
        self.dist = _DiagGaussian(self.base.output_size, action_space.shape[0])
        self._hip = None
        self._hip_version = None
        self._zero_edge = {}

    @property
    def is_recurrent(self):
        return self.base.is_recurrent

    # ---- HIP rollout path ----
    def _weights_version(self):
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    # rollout arithmetic / launch structure (cn_policy_set_gemm_mode): 'fused' = two persistent kernels (default, fastest; an env's
    # result depends on its tile neighbours at the 1e-7 level through the softmax summation order), 'bf16x3' = the same arithmetic as
    # separate launches whose per-env results are independent of the batch composition, 'fp32' = exact fp32 MFMA
    rollout_gemm_mode = "fused"

    def _hip_policy(self, E, device):
        from .hip import HipPolicy
        if self._hip is None or self._hip.maxE < E or self._hip.device != device:
            self._hip = HipPolicy(self.base.human_num, self.base.edge_width, E, device=device)
            self._hip.set_self_attention(self.base.use_self_attn)
            self._hip.set_taps(False)          # rollout path: nobody reads the test taps
            self._hip_version = None
            self._hip_mode = None
        if getattr(self, "_hip_mode", None) != self.rollout_gemm_mode:
            self._hip.set_gemm_mode(self.rollout_gemm_mode)
            self._hip_mode = self.rollout_gemm_mode
        ver = self._weights_version()
        if ver != self._hip_version:
            self._hip.set_weights(self.state_dict())
            self._hip_version = ver
        return self._hip

    def weights_changed(self):
        """Tell the rollout path that parameter storage was written through raw pointers (the fused Adam kernel): the next
        act / get_value re-snapshots the weights (cn_policy_set_weights)."""
        self._hip_version = None

    def _edge_zeros(self, E, device):
        key = (E, str(device))
        if key not in self._zero_edge:
            self._zero_edge = {key: torch.zeros(1, 1, 1, device=device).expand(E, self.base.human_num + 1, self.base.human_human_edge_rnn_size)}
        return self._zero_edge[key]

    def _obs32(self, inputs):
        inputs = self.base.counted_inputs(inputs)      # sort_humans = False: visible humans first + their count (cn_obs_compact_visible)
        return {k: (v if v.dtype == torch.float32 else v.float()).contiguous() for k, v in inputs.items() if k != "visible_masks"}

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        E = inputs["robot_node"].shape[0]
        hx = rnn_hxs["human_node_rnn"]
        if hx.is_cuda:
            pol = self._hip_policy(E, hx.device)
            eps = None if deterministic else torch.randn(E, 2, device=hx.device)
            out = pol.act(self._obs32(inputs), hx.reshape(E, 1, 128), masks.reshape(E, 1).float(), eps=eps)
            value, action, logp, h = out["value"], out["action"], out["logp"], out["hxs"]
        else:
            with torch.no_grad():
                value, feat, h = self.base.forward_sequence(inputs, hx, masks, 1, E)
                mean = self.dist.fc_mean(feat)
                std = self.dist.logstd(torch.zeros_like(mean)).exp()
                action = mean if deterministic else mean + std * torch.randn_like(mean)
                logp = self._log_prob(mean, std, action)
            h = h.view(E, 1, 128)
        return value, action, logp, {"human_node_rnn": h, "human_human_edge_rnn": self._edge_zeros(E, hx.device)}

    def get_value(self, inputs, rnn_hxs, masks):
        E = inputs["robot_node"].shape[0]
        hx = rnn_hxs["human_node_rnn"]
        if hx.is_cuda:
            return self._hip_policy(E, hx.device).get_value(self._obs32(inputs), hx.reshape(E, 1, 128), masks.reshape(E, 1).float())
        with torch.no_grad():
            value, _, _ = self.base.forward_sequence(inputs, hx, masks, 1, E)
        return value

    @staticmethod
    def _log_prob(mean, std, action):
        var = std * std
        return (-((action - mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(-1, keepdim=True)

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        """inputs [T*N,...] (T = seq_length, N = num_processes / num_mini_batch), rnn_hxs at t=0 ([N,...])."""
        B = inputs["robot_node"].shape[0]
        N = rnn_hxs["human_node_rnn"].shape[0]
        T = B // N
        base = self.base
        if inputs["robot_node"].is_cuda and base.train_fused_rn and base.train_gemm_mode == "bf16x3" and base.fused_rn_shapes_ok():
            # train-mode forward as two boundary calls: the human-human block (cn_hh_block_fwd behind _hh_block) and the robot-node sequence
            # (cn_rn_seq_fwd), each with ONE backward entry
            inputs = base.counted_inputs(inputs)
            det = inputs["detected_human_num"].reshape(B).to(torch.int64).clamp(min=1)
            out_sp, row_off = base._hh_block(inputs["spatial_edges"].reshape(B, base.human_num, base.edge_width), det)
            value, logp, h = base.rn_sequence(inputs, out_sp, row_off, rnn_hxs["human_node_rnn"], masks, action, T, N, self.dist)
            logstd = self.dist.logstd._bias.t().view(1, -1)
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).mean()   # FixedNormal.entropy().mean(): the same for every sample
            return value, logp, entropy, {"human_node_rnn": h.view(N, 1, -1), "human_human_edge_rnn": self._edge_zeros(N, h.device)}
        value, feat, h = self.base.forward_sequence(inputs, rnn_hxs["human_node_rnn"], masks, T, N)
        mean = _skinny_linear(feat, self.dist.fc_mean.weight, self.dist.fc_mean.bias)
        logstd = self.dist.logstd(torch.zeros_like(mean))
        logp = self._log_prob(mean, logstd.exp(), action)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).mean()   # FixedNormal.entropy().mean(): over batch and dims
        return value, logp, entropy, {"human_node_rnn": h.view(N, 1, -1), "human_human_edge_rnn": self._edge_zeros(N, h.device)}
