"""numpy restatement of the attention-graph policy forward + rollout math.  TEST INFRASTRUCTURE ONLY.

Follows (all under /root/reference):
  rl/networks/selfAttn_srnn_temp_node.py:360-449  selfAttn_merge_SRNN.forward
      :63-91   SpatialEdgeSelfAttn.forward (HH multi-head attention, 512-d, 8 heads, key padding mask)
      :145-223 EdgeAttention_M (HR attention, temperature H/sqrt(64), masked_fill -1e9)
      :262-285 EndRNN.forward + rl/networks/srnn_model.py:35-105 RNNBase._forward_gru (h * mask, GRU)
  rl/networks/model.py:56-90      Policy.act / get_value / evaluate_actions
  rl/networks/distributions.py:36-44,76-95  FixedNormal / DiagGaussian
  rl/networks/storage.py:123-132  GAE;  rl/ppo/ppo.py:37-39 advantage normalisation; :64-78 PPO losses
torch.nn.MultiheadAttention is restated from its documented algorithm (in_proj -> per-head scaled QK^T ->
-inf on padded keys -> softmax -> PV -> out_proj).

Computation is float64 on float32 parameters: the reference is torch fp32, so agreement is ~1e-6; the tests pin
this module against golden outputs of the reference's own torch code (tests/golden/policy_*.npz, rollout_*.npz).
"""
import numpy as np

NUM_HEADS = 8
ATTN = 512


def _lin(x, sd, name):
    return x @ sd[name + ".weight"].astype(np.float64).T + sd[name + ".bias"].astype(np.float64)


def _relu(x):
    return np.maximum(x, 0.0)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def hh_attention(sd, spatial_edges, det):
    """spatial_edges [B,H,D], det [B] int -> [B,H,512]  (SpatialEdgeSelfAttn.forward)."""
    B, H, _ = spatial_edges.shape
    x = spatial_edges.astype(np.float64)
    e = _relu(_lin(x, sd, "base.spatial_attn.embedding_layer.0"))
    e = _relu(_lin(e, sd, "base.spatial_attn.embedding_layer.2"))
    q = _lin(e, sd, "base.spatial_attn.q_linear")
    k = _lin(e, sd, "base.spatial_attn.k_linear")
    v = _lin(e, sd, "base.spatial_attn.v_linear")
    W = sd["base.spatial_attn.multihead_attn.in_proj_weight"].astype(np.float64)
    b = sd["base.spatial_attn.multihead_attn.in_proj_bias"].astype(np.float64)
    q = q @ W[:ATTN].T + b[:ATTN]
    k = k @ W[ATTN:2 * ATTN].T + b[ATTN:2 * ATTN]
    v = v @ W[2 * ATTN:].T + b[2 * ATTN:]
    hd = ATTN // NUM_HEADS
    qh = q.reshape(B, H, NUM_HEADS, hd).transpose(0, 2, 1, 3) / np.sqrt(hd)
    kh = k.reshape(B, H, NUM_HEADS, hd).transpose(0, 2, 1, 3)
    vh = v.reshape(B, H, NUM_HEADS, hd).transpose(0, 2, 1, 3)
    s = qh @ kh.transpose(0, 1, 3, 2)                       # [B,heads,H,H]
    pad = np.arange(H)[None, :] >= np.asarray(det).reshape(B, 1)  # True = padded key
    s = np.where(pad[:, None, None, :], -np.inf, s)
    p = _softmax(s, axis=-1)
    o = (p @ vh).transpose(0, 2, 1, 3).reshape(B, H, ATTN)
    return _lin(o, sd, "base.spatial_attn.multihead_attn.out_proj")


def hr_attention(sd, robot_states, output_spatial, det):
    """robot_states [B,256], output_spatial [B,H,256] -> (weighted [B,256], attn [B,H])  (EdgeAttention_M)."""
    B, H, _ = output_spatial.shape
    t = _lin(robot_states, sd, "base.attn.temporal_edge_layer.0")           # [B,64]
    s = _lin(output_spatial, sd, "base.attn.spatial_edge_layer.0")          # [B,H,64]
    a = np.sum(t[:, None, :] * s, axis=-1) * (H / np.sqrt(64.0))
    valid = np.arange(H)[None, :] < np.asarray(det).reshape(B, 1)
    a = np.where(valid, a, -1e9)
    a = _softmax(a, axis=-1)
    return np.einsum("bh,bhc->bc", a, output_spatial), a


def gru_cell(sd, x, h):
    Wi = sd["base.humanNodeRNN.gru.weight_ih_l0"].astype(np.float64)
    Wh = sd["base.humanNodeRNN.gru.weight_hh_l0"].astype(np.float64)
    bi = sd["base.humanNodeRNN.gru.bias_ih_l0"].astype(np.float64)
    bh = sd["base.humanNodeRNN.gru.bias_hh_l0"].astype(np.float64)
    gi = x @ Wi.T + bi
    gh = h @ Wh.T + bh
    n = h.shape[-1]
    r = _sigmoid(gi[:, :n] + gh[:, :n])
    z = _sigmoid(gi[:, n:2 * n] + gh[:, n:2 * n])
    nn = np.tanh(gi[:, 2 * n:] + r * gh[:, 2 * n:])
    return (1.0 - z) * nn + z * h


def base_step(sd, obs, h, masks, taps=None):
    """One timestep for a batch: obs dict ([B,1,7],[B,1,2],[B,H,D],[B,1]), h [B,128], masks [B,1]."""
    B = obs["robot_node"].shape[0]
    robot_in = np.concatenate([obs["temporal_edges"].reshape(B, 2), obs["robot_node"].reshape(B, 7)], axis=-1).astype(np.float64)
    robot_states = _relu(_lin(robot_in, sd, "base.robot_linear.0"))
    det = obs["detected_human_num"].reshape(B).astype(np.int64)
    hh = hh_attention(sd, obs["spatial_edges"], det)
    out_sp = _relu(_lin(hh, sd, "base.spatial_linear.0"))
    hr, attn = hr_attention(sd, robot_states, out_sp, det)
    enc = _relu(_lin(robot_states, sd, "base.humanNodeRNN.encoder_linear"))
    edge = _relu(_lin(hr, sd, "base.humanNodeRNN.edge_attention_embed"))
    x = np.concatenate([enc, edge], axis=-1)
    h_new = gru_cell(sd, x, h.astype(np.float64) * masks.reshape(B, 1))
    out = _lin(h_new, sd, "base.humanNodeRNN.output_linear")
    hc = np.tanh(_lin(np.tanh(_lin(out, sd, "base.critic.0")), sd, "base.critic.2"))
    ha = np.tanh(_lin(np.tanh(_lin(out, sd, "base.actor.0")), sd, "base.actor.2"))
    value = _lin(hc, sd, "base.critic_linear")
    if taps is not None:
        taps.update(hh_out=hh, spatial_lin=out_sp, hr_out=hr, hr_attn=attn, robot_emb=robot_states)
    return value, ha, h_new


def dist_params(sd, actor_feat):
    mean = _lin(actor_feat, sd, "dist.fc_mean")
    logstd = sd["dist.logstd._bias"].astype(np.float64).reshape(1, -1)
    return mean, np.broadcast_to(logstd, mean.shape)


def log_prob(mean, logstd, action):
    var = np.exp(logstd) ** 2
    return np.sum(-((action - mean) ** 2) / (2 * var) - logstd - 0.5 * np.log(2 * np.pi), axis=-1, keepdims=True)


def entropy_mean(logstd):
    return float(np.mean(0.5 + 0.5 * np.log(2 * np.pi) + logstd))


def act(sd, obs, h, masks, action=None, taps=None):
    """deterministic act (mode) unless `action` is given; returns value, action, logp, h_new, actor_feat."""
    value, feat, h_new = base_step(sd, obs, h, masks, taps)
    mean, logstd = dist_params(sd, feat)
    a = mean if action is None else action
    return value, a, log_prob(mean, logstd, a), h_new, feat


def evaluate_actions(sd, obs_seq, h0, masks_seq, actions_seq):
    """obs_seq: list over T of obs dicts for N envs; masks_seq [T,N,1]; actions [T,N,2].
    Returns values [T*N,1], logp [T*N,1], entropy (scalar) in the reference's (T-major) flattening."""
    T = len(obs_seq)
    h = h0.astype(np.float64)
    vals, lps = [], []
    logstd = None
    for t in range(T):
        value, feat, h = base_step(sd, obs_seq[t], h, masks_seq[t])
        mean, logstd = dist_params(sd, feat)
        vals.append(value)
        lps.append(log_prob(mean, logstd, actions_seq[t].astype(np.float64)))
    return np.concatenate(vals, 0), np.concatenate(lps, 0), entropy_mean(logstd)


def gae(rewards, values, masks, gamma, lam):
    """rewards [T,N,1], values [T+1,N,1], masks [T+1,N,1] -> returns [T,N,1] (storage.py:123-132)."""
    T = rewards.shape[0]
    ret = np.zeros_like(rewards, dtype=np.float64)
    g = 0.0
    for t in reversed(range(T)):
        delta = rewards[t] + gamma * values[t + 1] * masks[t + 1] - values[t]
        g = delta + gamma * lam * masks[t + 1] * g
        ret[t] = g + values[t]
    return ret


def adv_normalize(returns, values):
    """ppo.py:37-39: unbiased std over all T*N."""
    adv = returns - values
    return (adv - adv.mean()) / (adv.std(ddof=1) + 1e-5)


def ppo_losses(values, logp, old_values, old_logp, returns, adv, clip=0.2):
    """ppo.py:64-78 -> (value_loss, action_loss)."""
    ratio = np.exp(logp - old_logp)
    s1 = ratio * adv
    s2 = np.clip(ratio, 1.0 - clip, 1.0 + clip) * adv
    action_loss = -np.mean(np.minimum(s1, s2))
    vclip = old_values + np.clip(values - old_values, -clip, clip)
    value_loss = 0.5 * np.mean(np.maximum((values - returns) ** 2, (vclip - returns) ** 2))
    return float(value_loss), float(action_loss)
