"""GST trajectory predictor + the VecPretextNormalize wrapper logic (BASELINE configs[3]), batched
over all envs on the device with no host synchronisation and no per-env Python loop.

On a GPU the predictor and the wrapper processing run as hand-written HIP kernels (csrc/gst.hip through cn_gst_predict /
cn_gst_wrapper_step); the torch-op expression of the same math below is what CPU tensors use (unit tests) and doubles
as an independent cross-check of the kernels.  Same state-dict keys as the shipped checkpoints (`epoch_100.pt` loads unchanged):
  gst_updated/src/gumbel_social_transformer/st_model.py:271-455 (faster_lstm, recursive decode, fully connected edges)
  gst_updated/scripts/wrapper/crowd_nav_interface_parallel.py:45-114
  rl/vec_env/vec_pretext_normalize.py:85-191
"""
import argparse
import glob
import json
import os
import pickle
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

INVALID = -999.0


class _SelfAttn(nn.Module):
    def __init__(self, d=64):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _NodeEncoderLayer(nn.Module):
    def __init__(self, d=64, ff=128):
        super().__init__()
        self.self_attn = _SelfAttn(d)
        self.norm_node = nn.LayerNorm(d)
        self.norm1_node = nn.LayerNorm(d)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)


class _GST(nn.Module):
    def __init__(self):
        super().__init__()
        self.node_embedding = nn.Linear(2, 64)
        self.node_encoder_layers = nn.ModuleList([_NodeEncoderLayer()])


class GSTPredictor(nn.Module):
    """Inference-only Gumbel Social Transformer with the shipped hyper-parameters (embedding 64, 8 heads, 1 layer,
    spatial_num_heads_edges = 0, ghost = False, LSTM 64, obs 5 / pred 5, output_dim 5)."""

    def __init__(self, obs_len=5, pred_len=5):
        super().__init__()
        self.gumbel_social_transformer = _GST()
        self.lstm = nn.LSTM(64, 64)
        self.hidden2pos = nn.Linear(64, 5)
        self.obs_len, self.pred_len = obs_len, pred_len
        self.eval()

    @staticmethod
    def from_checkpoint(path, device):
        """path: <model_dir>/checkpoint/epoch_100.pt as shipped with the reference (config.pred.model_dir)."""
        import numpy
        safe = [(numpy.core.multiarray.scalar, "numpy.core.multiarray.scalar"), (numpy.dtype, "numpy.dtype")]
        safe += [getattr(numpy.dtypes, n) for n in dir(numpy.dtypes) if n.endswith("DType")]
        with torch.serialization.safe_globals(safe):
            ck = torch.load(path, map_location=device, weights_only=True)
        m = GSTPredictor().to(device)
        m.load_state_dict(ck["model_state_dict"] if "model_state_dict" in ck else ck)
        return m

    def _transformer(self, x, attn_mask):
        """x [B,H,2], attn_mask [B,H,H] float (target, neighbor) -> [B,H,64]."""
        g = self.gumbel_social_transformer
        L = g.node_encoder_layers[0]
        B, H, _ = x.shape
        x = g.node_embedding(x)
        ped = (attn_mask.sum(-1) > 0).to(x.dtype).unsqueeze(-1)
        x = L.norm_node(x) * ped
        q, k, v = [t.view(B, H, 8, 8).transpose(1, 2) for t in F.linear(x, L.self_attn.in_proj_weight, L.self_attn.in_proj_bias).chunk(3, dim=-1)]
        p = torch.softmax((q * 8 ** -0.5) @ k.transpose(-1, -2), dim=-1)
        p = p * attn_mask.unsqueeze(1)                      # float mask: multiply after the softmax, then renormalise (mha.py:236-242)
        p = p / (p.sum(-1, keepdim=True) + 1e-10)
        o = (p @ v).transpose(1, 2).reshape(B, H, 64)
        x = x + L.self_attn.out_proj(o)
        return x + L.linear2(F.relu(L.linear1(L.norm1_node(x))))

    def _lstm_cell(self, x, h, c):
        g = F.linear(x, self.lstm.weight_ih_l0, self.lstm.bias_ih_l0) + F.linear(h, self.lstm.weight_hh_l0, self.lstm.bias_hh_l0)
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        return torch.sigmoid(o) * torch.tanh(c), c

    @torch.no_grad()
    def forward(self, in_traj, in_mask):
        """in_traj [E,H,T,2] world positions (-999 where unseen), in_mask [E,H,T,1] 0/1 float ->
        out_traj [E,H,P,5] (cumulative mu_x, mu_y, sigma_x, sigma_y, corr; positions -999 where not predicted), out_mask [E,H,1]."""
        E, H, T, _ = in_traj.shape
        P = self.pred_len
        m = in_mask[..., 0]
        m_rel = torch.cat([m[:, :, :1], m[:, :, :-1] * m[:, :, -1:]], dim=2)       # crowd_nav_interface_parallel.py:76-78
        lm_fp = m_rel[:, :, -1]
        rel = torch.cat([torch.zeros(E, H, 1, 2, device=in_traj.device), in_traj[:, :, 1:] - in_traj[:, :, :-1]], dim=2)
        rel = INVALID * (1 - m_rel.unsqueeze(-1)) + rel * m_rel.unsqueeze(-1)
        mt = m_rel.permute(0, 2, 1).reshape(E * T, H)
        xs = self._transformer(rel.permute(0, 2, 1, 3).reshape(E * T, H, 2), mt.unsqueeze(2) * mt.unsqueeze(1)).view(E, T, H, 64)
        xs = xs * m_rel.permute(0, 2, 1).unsqueeze(-1)
        h = torch.zeros(E * H, 64, device=in_traj.device)
        c = torch.zeros_like(h)
        for t in range(T):
            h, c = self._lstm_cell(xs[:, t].reshape(E * H, 64), h, c)
        mk = lm_fp.reshape(E * H, 1)
        h, c = h * mk, c * mk
        attn_pred = lm_fp.unsqueeze(2) * lm_fp.unsqueeze(1)
        mus, sxs, sys_, cors = [], [], [], []
        x_sample = None
        for tt in range(P):
            if tt > 0:
                xt = self._transformer(x_sample, attn_pred).reshape(E * H, 64) * mk
                hp, cp = self._lstm_cell(xt, h, c)
                h = hp * mk + h * (1 - mk)
                c = cp * mk + c * (1 - mk)
            raw = self.hidden2pos(h).view(E, H, 5)
            mu = raw[..., :2]
            mus.append(mu); sxs.append(raw[..., 2:3].exp()); sys_.append(raw[..., 3:4].exp()); cors.append(raw[..., 4:5].tanh())
            x_sample = mu * lm_fp.unsqueeze(-1)
        mu = torch.stack(mus, 1).cumsum(1)
        sx, sy, corr = torch.stack(sxs, 1), torch.stack(sys_, 1), torch.stack(cors, 1)
        sxc, syc = (sx ** 2).cumsum(1).sqrt(), (sy ** 2).cumsum(1).sqrt()
        corrc = (corr * sx * sy).cumsum(1) / (sxc * syc)
        lm_pred = lm_fp[:, None, :, None]
        mu = (mu + in_traj[:, :, -1].unsqueeze(1)) * lm_pred + INVALID * (1 - lm_pred)
        return torch.cat([mu, sxc, syc, corrc], dim=3).permute(0, 2, 1, 3).contiguous(), lm_fp.unsqueeze(-1)


class PretextProcessor:
    """State and per-step processing of VecPretextNormalize (rl/vec_env/vec_pretext_normalize.py:85-191)."""

    def __init__(self, predictor, num_envs, human_num, predict_steps, robot_radius, human_radius, collision_penalty, device, use_hip=None,
                 pred_interval=1):
        self.pred, self.E, self.H, self.P = predictor, num_envs, human_num, predict_steps
        self.dist, self.device = robot_radius + human_radius, torch.device(device)
        self.collision_penalty = float(collision_penalty)
        self.pen = (collision_penalty / 2.0 ** torch.arange(2, predict_steps + 2, device=device, dtype=torch.float32)).view(1, 1, predict_steps)
        # prediction stride (vec_pretext_normalize.py:56-57): the history holds (obs_seq_len - 1) * interval + 1 observations, every
        # interval-th of them is the predictor's input (:133-134)
        self.interval = int(pred_interval)
        if self.interval < 1:
            raise ValueError("pred_interval = int(data.pred_timestep // env.time_step) must be >= 1")
        self.buffer_len = 4 * self.interval + 1
        self.hip = None
        if use_hip if use_hip is not None else self.device.type == "cuda":
            if predict_steps != 5:
                raise NotImplementedError("the GST kernels are specialised to the shipped predictor (5 observed / 5 predicted steps)")
            from .hip import HipGST
            self.hip = HipGST(human_num, num_envs, device=self.device)     # raises if the extension is missing: no fallback on a GPU
            self.hip.set_weights(predictor.state_dict())
            self.hip.wrapper_set_interval(self.interval)
        self.reset_buffers()

    def reset_buffers(self):
        """VecPretextNormalize.reset(): dummy history.  (NOT called when a single env auto-resets: the reference keeps the
        stale history of the finished episode, :112 `done` is unused.)"""
        if self.hip is not None:
            self.hip.wrapper_reset(self.E)
            return
        self.traj = torch.full((self.buffer_len, self.E, self.H, 2), INVALID, device=self.device)
        self.mask = torch.zeros(self.buffer_len, self.E, self.H, 1, dtype=torch.bool, device=self.device)

    def state_dict(self):
        """The observation history (traj_buffer / mask_buffer of vec_pretext_normalize.py:85-101) in time order, oldest first, as CPU tensors."""
        if self.hip is not None:
            traj, mask = self.hip.wrapper_state()
            return {"traj": traj.cpu(), "mask": mask.cpu(), "interval": self.interval}
        return {"traj": self.traj.cpu().clone(), "mask": self.mask.reshape(self.buffer_len, self.E, self.H).to(torch.uint8).cpu(), "interval": self.interval}

    def load_state_dict(self, sd):
        if int(sd.get("interval", 1)) != self.interval or tuple(sd["traj"].shape) != (self.buffer_len, self.E, self.H, 2):
            raise ValueError("history of shape %s / stride %s does not fit this wrapper (%d x %d envs x %d humans, stride %d)"
                             % (tuple(sd["traj"].shape), sd.get("interval"), self.buffer_len, self.E, self.H, self.interval))
        if self.hip is not None:
            self.hip.wrapper_load_state(sd["traj"], sd["mask"])
            return
        self.traj = sd["traj"].to(self.device, torch.float32).clone()
        self.mask = sd["mask"].to(self.device).to(torch.bool).view(self.buffer_len, self.E, self.H, 1).clone()

    @torch.no_grad()
    def process(self, obs, rews):
        """obs: dict with robot_node [E,1,7], spatial_edges [E,H,2(P+1)] (unsorted, by human id), visible_masks [E,H] bool.
        rews [E] or [E,1] device tensor.  Returns (new spatial_edges [E,H,2(P+1)] sorted by distance, rews + social penalty)."""
        E, H, P = self.E, self.H, self.P
        if self.hip is not None:
            rews = rews.reshape(E).float().contiguous()
            se = self.hip.wrapper_step(obs, rews, self.dist, self.collision_penalty)
            return se, rews
        robot_xy = obs["robot_node"][:, :, :2]
        se = obs["spatial_edges"].clone()
        human_pos = robot_xy + se[:, :, :2]
        self.traj = torch.cat([self.traj[1:], human_pos.unsqueeze(0)], 0)
        self.mask = torch.cat([self.mask[1:], obs["visible_masks"].to(torch.bool).view(1, E, H, 1)], 0)
        out_traj, out_mask = self.pred(self.traj[::self.interval].permute(1, 2, 0, 3), self.mask[::self.interval].permute(1, 2, 0, 3).float())
        out_mask = out_mask.bool()
        rel = out_traj[..., :2] - robot_xy.unsqueeze(1)                       # robot-frame predictions [E,H,P,2]
        coll = (rel.norm(dim=-1) < self.dist) & out_mask
        rf = (coll.float() * self.pen).reshape(E, -1).min(dim=1).values
        rews = rews.reshape(E) + rf
        se[:, :, 2:] = torch.where(out_mask.expand(E, H, 2 * P), rel.reshape(E, H, 2 * P), se[:, :, 2:])
        order = torch.argsort(se[:, :, :2].norm(dim=-1), dim=1, stable=True)
        se = torch.gather(se, 1, order.unsqueeze(-1).expand(E, H, se.shape[2]))
        return se, rews


def load_predictor(config, device):
    model_dir = getattr(getattr(config, "pred", None), "model_dir", None)
    if model_dir is None:
        raise ValueError("config.pred.model_dir is required for CrowdSimPredRealGST-v0 with the prediction wrapper")
    return GSTPredictor.from_checkpoint(find_checkpoint(model_dir), device)


class _NamespaceOnlyUnpickler(pickle.Unpickler):
    """args.pickle holds an argparse.Namespace of plain values (gst_updated/scripts/experiments/train.py:88-89); nothing else may load."""

    def find_class(self, module, name):
        if (module, name) == ("argparse", "Namespace"):
            return argparse.Namespace
        raise pickle.UnpicklingError("args.pickle may only contain an argparse.Namespace of plain values, found %s.%s" % (module, name))


def find_checkpoint(model_dir):
    """<model_dir>/checkpoint/epoch_<num_epochs>.pt, num_epochs as the run recorded it -- the rule of the reference's loader
    (crowd_nav_interface_multi_env_parallel.py:21-28: args.pickle -> 'epoch_' + str(args.num_epochs) + '.pt').  args.json (written by
    gst_train.train) is preferred, then args.pickle through an unpickler restricted to argparse.Namespace, then the highest epoch_*.pt
    present (the shipped gst_updated/results/*/sj directories: epoch_100.pt)."""
    ckpt_dir = os.path.join(model_dir, "checkpoint")
    num_epochs = None
    if os.path.exists(os.path.join(ckpt_dir, "args.json")):
        with open(os.path.join(ckpt_dir, "args.json")) as f:
            num_epochs = json.load(f).get("num_epochs")
    elif os.path.exists(os.path.join(ckpt_dir, "args.pickle")):
        try:
            with open(os.path.join(ckpt_dir, "args.pickle"), "rb") as f:
                num_epochs = getattr(_NamespaceOnlyUnpickler(f).load(), "num_epochs", None)
        except Exception:
            num_epochs = None            # e.g. a Namespace holding numpy scalars: fall through to the directory listing
    if num_epochs is not None:
        path = os.path.join(ckpt_dir, "epoch_%d.pt" % int(num_epochs))
        if os.path.exists(path):
            return path
    found = sorted(glob.glob(os.path.join(ckpt_dir, "epoch_*.pt")), key=lambda p: int(re.sub(r"\D", "", os.path.basename(p)) or 0))
    if not found:
        raise FileNotFoundError("no GST checkpoint under %s (config.pred.model_dir must point at a gst_updated/results/.../sj style "
                                "directory or at the out_dir of gst_train.train)" % ckpt_dir)
    return found[-1]
