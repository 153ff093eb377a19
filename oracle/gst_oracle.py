"""numpy restatement of the GST trajectory-predictor path (BASELINE configs[3]).  TEST INFRASTRUCTURE ONLY.

Follows (under /root/reference):
  gst_updated/scripts/wrapper/crowd_nav_interface_parallel.py:45-114   CrowdNavPredInterfaceMultiEnv.forward
  gst_updated/src/gumbel_social_transformer/st_model.py:271-455         st_model.forward (faster_lstm, recursive decode,
                                                                         spatial_num_heads_edges = 0 -> full connectivity)
  gst_updated/src/gumbel_social_transformer/gumbel_social_transformer.py:43-96, node_encoder_layer_no_ghost.py:25-66
  gst_updated/src/gumbel_social_transformer/mha.py:236-242              float attention mask: multiply after softmax, renormalise
  rl/vec_env/vec_pretext_normalize.py:112-191                           VecPretextNormalize.process_obs_rew
Pinned by tests/golden/gst_e4_h20.npz (outputs of the reference's own torch code with formula weights).
"""
import numpy as np

INVALID = -999.0


def _lin(x, sd, name):
    return x @ sd[name + ".weight"].astype(np.float64).T + sd[name + ".bias"].astype(np.float64)


def _ln(x, sd, name, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * sd[name + ".weight"].astype(np.float64) + sd[name + ".bias"].astype(np.float64)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def transformer(sd, x, attn_mask):
    """x [B,H,2], attn_mask [B,H(target),H(neighbor)] float -> [B,H,64]  (one NodeEncoderLayer, 8 heads of 8)."""
    L = "gumbel_social_transformer.node_encoder_layers.0."
    B, H, _ = x.shape
    x = _lin(x.astype(np.float64), sd, "gumbel_social_transformer.node_embedding")
    ped = (attn_mask.sum(-1) > 0).astype(np.float64)[..., None]
    x = _ln(x, sd, L + "norm_node") * ped
    W, b = sd[L + "self_attn.in_proj_weight"].astype(np.float64), sd[L + "self_attn.in_proj_bias"].astype(np.float64)
    qkv = x @ W.T + b
    q, k, v = [t.reshape(B, H, 8, 8).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, axis=-1)]
    s = (q * 8 ** -0.5) @ k.transpose(0, 1, 3, 2)
    s = np.exp(s - s.max(-1, keepdims=True))
    p = s / s.sum(-1, keepdims=True)
    p = p * attn_mask[:, None]
    p = p / (p.sum(-1, keepdims=True) + 1e-10)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, H, 64)
    x = x + _lin(o, sd, L + "self_attn.out_proj")
    x2 = _ln(x, sd, L + "norm1_node")
    x2 = _lin(np.maximum(_lin(x2, sd, L + "linear1"), 0.0), sd, L + "linear2")
    return x + x2


def lstm_cell(sd, x, h, c):
    g = x @ sd["lstm.weight_ih_l0"].astype(np.float64).T + sd["lstm.bias_ih_l0"] + h @ sd["lstm.weight_hh_l0"].astype(np.float64).T + sd["lstm.bias_hh_l0"]
    i, f, gg, o = np.split(g, 4, axis=-1)
    c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
    return _sigmoid(o) * np.tanh(c), c


def interface_forward(sd, in_traj, in_mask, pred_len=5):
    """in_traj [E,H,T,2], in_mask [E,H,T,1] (0/1) -> out_traj [E,H,P,5] (mu_x, mu_y, sx, sy, corr), out_mask [E,H,1]."""
    E, H, T, _ = in_traj.shape
    m = in_mask[..., 0].astype(np.float64)
    # crowd_nav_interface_parallel.py:76-78 (note: every step is gated by the LAST step's mask, as written there)
    m_rel = np.concatenate([m[:, :, :1], m[:, :, :-1] * m[:, :, -1:]], axis=2)
    lm_fp = m_rel[:, :, -1]                                          # loss_mask_rel_full_partial [E,H]
    traj = in_traj.astype(np.float64)
    rel = np.concatenate([np.zeros((E, H, 1, 2)), traj[:, :, 1:] - traj[:, :, :-1]], axis=2)
    rel = INVALID * (1 - m_rel[..., None]) + rel * m_rel[..., None]
    v_obs = rel.transpose(0, 2, 1, 3).reshape(E * T, H, 2)
    mt = m_rel.transpose(0, 2, 1).reshape(E * T, H)
    attn_obs = mt[:, :, None] * mt[:, None, :]
    xs = transformer(sd, v_obs, attn_obs).reshape(E, T, H, 64)
    xs = xs * m_rel.transpose(0, 2, 1)[..., None]
    h = np.zeros((E * H, 64)); c = np.zeros((E * H, 64))
    for t in range(T):
        h, c = lstm_cell(sd, xs[:, t].reshape(E * H, 64), h, c)
    mk = lm_fp.reshape(E * H, 1)
    h, c = h * mk, c * mk
    attn_pred = lm_fp[:, :, None] * lm_fp[:, None, :]
    mus, sxs, sys_, corrs = [], [], [], []
    for tt in range(pred_len):
        if tt > 0:
            xt = transformer(sd, x_sample, attn_pred).reshape(E * H, 64) * mk
            hp, cp = lstm_cell(sd, xt, h, c)
            h = hp * mk + h * (1 - mk)
            c = cp * mk + c * (1 - mk)
        raw = _lin(h, sd, "hidden2pos").reshape(E, H, 5)
        mu, sx, sy, corr = raw[..., :2], np.exp(raw[..., 2:3]), np.exp(raw[..., 3:4]), np.tanh(raw[..., 4:5])
        x_sample = mu * lm_fp[..., None]
        mus.append(mu); sxs.append(sx); sys_.append(sy); corrs.append(corr)
    mu = np.cumsum(np.stack(mus, 1), 1)                                # [E,P,H,2]
    sx, sy, corr = np.stack(sxs, 1), np.stack(sys_, 1), np.stack(corrs, 1)
    sxc, syc = np.sqrt(np.cumsum(sx ** 2, 1)), np.sqrt(np.cumsum(sy ** 2, 1))
    corrc = np.cumsum(corr * sx * sy, 1) / (sxc * syc)
    lm_pred = lm_fp[:, None, :, None]
    mu = (mu + traj[:, :, -1][:, None]) * lm_pred + INVALID * (1 - lm_pred)
    out = np.concatenate([mu, sxc, syc, corrc], axis=3).transpose(0, 2, 1, 3)
    return out, lm_fp[..., None]


class PretextWrapper:
    """VecPretextNormalize state + process_obs_rew (buffers are NOT cleared when a single env auto-resets, like the reference)."""

    def __init__(self, sd, E, H, predict_steps=5, robot_radius=0.3, human_radius=0.3, collision_penalty=-20.0):
        self.sd, self.E, self.H, self.P = sd, E, H, predict_steps
        self.traj = [np.full((E, H, 2), -999.0) for _ in range(5)]
        self.mask = [np.zeros((E, H, 1), dtype=bool) for _ in range(5)]
        self.rr, self.hr, self.pen = robot_radius, human_radius, collision_penalty

    def process(self, robot_node, spatial_edges, visible_masks, rews):
        E, H, P = self.E, self.H, self.P
        robot_xy = robot_node.reshape(E, 1, 7)[:, :, :2].astype(np.float32)
        se = spatial_edges.astype(np.float32).copy()
        human_pos = robot_xy + se[:, :, :2]
        self.traj = self.traj[1:] + [human_pos]
        self.mask = self.mask[1:] + [visible_masks.reshape(E, H, 1).astype(bool)]
        in_traj = np.stack(self.traj).transpose(1, 2, 0, 3)
        in_mask = np.stack(self.mask).transpose(1, 2, 0, 3).astype(np.float32)
        out_traj, out_mask = interface_forward(self.sd, in_traj, in_mask, P)
        out_traj = out_traj.astype(np.float32)
        out_mask = out_mask.astype(bool)                                        # [E,H,1]
        d = np.linalg.norm(out_traj[..., :2] - robot_xy[:, :, None, :], axis=-1)      # [E,H,P]
        coll = (d < self.rr + self.hr) & out_mask
        pen = self.pen / 2.0 ** np.arange(2, P + 2).reshape(1, 1, P)
        rf = (coll.astype(np.float32) * pen.astype(np.float32)).reshape(E, -1).min(1)
        rews = rews + rf.reshape(E, 1)
        new_edges = (out_traj[..., :2] - robot_xy[:, :, None, :]).reshape(E, H, -1)
        mk = np.repeat(out_mask, 2 * P, axis=2)
        se[:, :, 2:] = np.where(mk, new_edges, se[:, :, 2:])
        order = np.argsort(np.linalg.norm(se[:, :, :2], axis=-1), axis=1, kind="stable")
        se = np.take_along_axis(se, order[:, :, None], axis=1)
        return se, rews.astype(np.float32)
