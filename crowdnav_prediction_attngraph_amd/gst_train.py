"""Training of the GST trajectory predictor on the data-collection env's files -- the counterpart of the reference's
gst_updated/src/mgnn/trajectories.py (dataset), gst_updated/src/gumbel_social_transformer/st_model.py (training-time forward and
loss, :62-112, :271-455) and gst_updated/scripts/experiments/train.py (loop), for the shipped hyper-parameters (SURVEY.md 8a-G3:
embedding 64, 8 heads, 1 layer, spatial_num_heads_edges = 0, no ghost, faster_lstm, obs 5 / pred 5, recursive decoding).

Two execution paths of the training step (train.py:121-146: forward, negative log-likelihood, backward, clip, Adam):
  * on a GPU (backend 'hip', the default there): HipGstTrainer -- cn_gst_train_step (csrc/gst_train.hip: forward + loss + hand-derived
    reverse pass, one workgroup per sequence, the reference's four dropout sites) and cn_adam_clip_step, through the C ABI;
  * the torch-op graph below under autograd (backend 'torch'): what CPU tensors use (unit tests, pinned to the reference's numbers) and the
    independent cross-check of the kernels (tests/test_gpu_gst_train.py holds the two against each other).
The dataset, the rotation augmentation, the learning-rate schedule, evaluation and the checkpoint format are host code either way, and the
producer of the data is the batched simulator (collect.py: thousands of simulated crowds per GPU).

Scope note: the reference trains on `<dataset>_dset_<split>_batch_trajectories.pt` files produced by scripts/data/create_*datasets*.py,
which are NOT part of the reference checkout (only the shell wrappers that call them are).  This module therefore feeds the loop with
TrajectoriesDataset items directly -- one sequence (all pedestrians of a 10-frame window) per optimiser step, which is what a
BatchTrajectoriesDataset item of one sequence is.
"""
import argparse
import json
import math
import os
import pickle
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

from .gst import GSTPredictor

INVALID = -999.0


def read_file(path, delim="\t"):
    """trajectories.py:163-174: rows of (frame id, pedestrian id, x, y)."""
    delim = {"tab": "\t", "space": " "}.get(delim, delim)
    rows = []
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(v) for v in line.split(delim)])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 4)


def seq_to_graph(seq, seq_rel):
    """mgnn/utils.py:44-77 with attn_mech 'rel_conv': V[t, h] = displacement of pedestrian h at step t; A[t, i, j] = pos_i - pos_j."""
    V = seq_rel.permute(2, 0, 1).contiguous().float()                  # [T, N, 2]
    x = seq.permute(2, 0, 1).float()                                  # [T, N, 2]
    A = x.unsqueeze(2) - x.unsqueeze(1)                               # [T, N, N, 2]
    return V, A


class TrajectoriesDataset(Dataset):
    """gst_updated/src/mgnn/trajectories.py:9-160.  Every window of obs + pred consecutive frames of a file in which at least one
    pedestrian is present throughout becomes a sequence: positions / displacements [N, 2, T] (-999 where missing), loss masks [N, T],
    the graph tensors of seq_to_graph and the per-step attention masks (outer product of the displacement mask)."""

    def __init__(self, data_dir, obs_seq_len=5, pred_seq_len=5, skip=1, delim="\t", invalid_value=INVALID, mode=None, frame_diff=1.0, verbose=False):
        super().__init__()
        self.data_dir, self.obs_seq_len, self.pred_seq_len, self.skip = data_dir, obs_seq_len, pred_seq_len, skip
        self.seq_len = T = obs_seq_len + pred_seq_len
        files = [os.path.join(data_dir, p) for p in os.listdir(data_dir)]
        num_peds, seqs, seqs_rel, masks, masks_rel, self.frame_id_seq = [], [], [], [], [], []
        for path in files:
            if verbose:
                print(path)
            data = read_file(path, delim)
            frames = np.unique(data[:, 0]).tolist()
            frame_data = [data[data[:, 0] == fr, :] for fr in frames]
            num_sequences = math.floor((len(frames) - T) / skip) + 1
            stop = num_sequences * skip + 1
            if mode is None:
                idx_range = range(0, stop, skip)
            elif mode == "train":
                idx_range = range(0, int(stop * 0.8), skip)
            elif mode in ("val", "test"):
                idx_range = range(int(stop * 0.8), stop, skip)
            else:
                raise RuntimeError("Wrong mode for TrajectoriesDataset.")
            for idx in idx_range:
                chunk = frame_data[idx:idx + T]
                if not chunk:
                    continue
                cur = np.concatenate(chunk, axis=0)
                start = cur[0, 0]
                peds = np.unique(cur[:, 1])
                # slot of every row inside the window (frame id -> step), rows on other frame ids are ignored like in the reference
                step_f = (cur[:, 0] - start) / frame_diff
                step = np.rint(step_f).astype(np.int64)
                on_grid = (step_f == step) & (step >= 0) & (step < T)
                col = np.searchsorted(peds, cur[:, 1])
                present = np.zeros((len(peds), T), dtype=np.int64)
                np.add.at(present, (col[on_grid], step[on_grid]), 1)
                if present.max() > 1:
                    raise RuntimeError("The same pedestrian has multiple locations in the same frame.")
                # :60-68 a pedestrian with a row in EVERY one of the window's frames, those frames spaced by frame_diff
                survive = False
                for k in range(len(peds)):
                    fr_k = np.unique(cur[col == k, 0])
                    if len(fr_k) == T and np.all(fr_k[1:] - fr_k[:-1] == frame_diff):
                        survive = True
                        break
                if not survive:
                    continue
                seq = np.ones((len(peds), 2, T)) * invalid_value
                seq_rel = np.ones((len(peds), 2, T)) * invalid_value
                seq[col[on_grid], :, step[on_grid]] = cur[on_grid, 2:]
                m = present.astype(np.float64)
                m_rel = np.zeros_like(m)
                m_rel[:, 0] = m[:, 0]
                m_rel[:, 1:] = m[:, 1:] * m[:, :-1]
                rel = np.zeros_like(seq)
                rel[:, :, 1:] = seq[:, :, 1:] - seq[:, :, :-1]
                sel = m_rel.astype(bool)[:, None, :].repeat(2, axis=1)
                seq_rel[sel] = rel[sel]
                num_peds.append(len(peds)); seqs.append(seq); seqs_rel.append(seq_rel); masks.append(m); masks_rel.append(m_rel)
                self.frame_id_seq.append(start)
        self.num_seq = len(seqs)
        if self.num_seq == 0:
            raise RuntimeError("no sequence of %d frames with a pedestrian present throughout in %s" % (T, data_dir))
        seq_all, rel_all = np.concatenate(seqs, axis=0), np.concatenate(seqs_rel, axis=0)
        self.obs_traj = torch.from_numpy(seq_all[:, :, :obs_seq_len]).type(torch.float)
        self.pred_traj = torch.from_numpy(seq_all[:, :, obs_seq_len:]).type(torch.float)
        self.obs_traj_rel = torch.from_numpy(rel_all[:, :, :obs_seq_len]).type(torch.float)
        self.pred_traj_rel = torch.from_numpy(rel_all[:, :, obs_seq_len:]).type(torch.float)
        self.loss_mask = torch.from_numpy(np.concatenate(masks, axis=0)).type(torch.float)
        self.loss_mask_rel = torch.from_numpy(np.concatenate(masks_rel, axis=0)).type(torch.float)
        cum = [0] + np.cumsum(num_peds).tolist()
        self.seq_start_end = list(zip(cum[:-1], cum[1:]))
        self.v_obs, self.A_obs, self.v_pred, self.A_pred, self.attn_mask_obs, self.attn_mask_pred = [], [], [], [], [], []
        for s, e in self.seq_start_end:
            v, a = seq_to_graph(self.obs_traj[s:e], self.obs_traj_rel[s:e])
            self.v_obs.append(v); self.A_obs.append(a)
            v, a = seq_to_graph(self.pred_traj[s:e], self.pred_traj_rel[s:e])
            self.v_pred.append(v); self.A_pred.append(a)
            lm = self.loss_mask_rel[s:e]                                              # [N, T]
            am = (lm.t().unsqueeze(2) * lm.t().unsqueeze(1)).float()                  # [T, N, N]
            self.attn_mask_obs.append(am[:obs_seq_len]); self.attn_mask_pred.append(am[obs_seq_len:])

    def __len__(self):
        return self.num_seq

    def __getitem__(self, index):
        s, e = self.seq_start_end[index]
        return [self.obs_traj[s:e], self.pred_traj[s:e], self.obs_traj_rel[s:e], self.pred_traj_rel[s:e], self.loss_mask_rel[s:e],
                self.loss_mask[s:e], self.v_obs[index], self.A_obs[index], self.v_pred[index], self.A_pred[index],
                self.attn_mask_obs[index], self.attn_mask_pred[index]]


# ---- training-time forward (st_model.forward with sampling = False) on the checkpoint-compatible GSTPredictor ----
def _transformer_train(model, x, attn_mask, p_drop):
    """GSTPredictor._transformer with the reference's four dropout sites (mha.py:243, node_encoder_layer_no_ghost.py:57,61,62)."""
    g = model.gumbel_social_transformer
    L = g.node_encoder_layers[0]
    B, H, _ = x.shape
    tr = model.training and p_drop > 0
    x = g.node_embedding(x)
    ped = (attn_mask.sum(-1) > 0).to(x.dtype).unsqueeze(-1)
    x = L.norm_node(x) * ped
    q, k, v = [t.view(B, H, 8, 8).transpose(1, 2) for t in F.linear(x, L.self_attn.in_proj_weight, L.self_attn.in_proj_bias).chunk(3, dim=-1)]
    p = torch.softmax((q * 8 ** -0.5) @ k.transpose(-1, -2), dim=-1)
    p = p * attn_mask.unsqueeze(1)
    p = p / (p.sum(-1, keepdim=True) + 1e-10)
    p = F.dropout(p, p_drop, tr)
    o = (p @ v).transpose(1, 2).reshape(B, H, 64)
    x = x + F.dropout(L.self_attn.out_proj(o), p_drop, tr)
    x2 = F.dropout(F.relu(L.linear1(L.norm1_node(x))), p_drop, tr)
    return x + F.dropout(L.linear2(x2), p_drop, tr)


def forward_train(model, v_obs, attn_mask_obs, loss_mask_rel, p_drop=0.1):
    """st_model.py:271-455 (faster_lstm, recursive, only_observe_full_period = False, sampling = False).
    v_obs [1,T,N,2], attn_mask_obs [1,T,N,N] (neighbour, target), loss_mask_rel [1,N,T+P] ->
    (mu [1,P,N,2], sx, sy, corr [1,P,N,1]), x_sample_pred [1,P,N,2], info{'loss_mask_rel_full_partial', 'loss_mask_per_pedestrian'}."""
    B, T, N, _ = v_obs.shape
    P = model.pred_len
    dev = v_obs.device
    lm_pp = (loss_mask_rel.sum(2) == loss_mask_rel.shape[2]).float()
    am = attn_mask_obs.permute(0, 1, 3, 2).reshape(B * T, N, N)                       # (target, neighbour)
    xs = _transformer_train(model, v_obs.reshape(B * T, N, 2), am, p_drop).view(B, T, N, 64)
    xs = xs * loss_mask_rel[:, :, :T].permute(0, 2, 1).unsqueeze(-1)
    h = torch.zeros(B * N, 64, device=dev)
    c = torch.zeros_like(h)
    for t in range(T):
        h, c = model._lstm_cell(xs[:, t].reshape(B * N, 64), h, c)
    lm_fp = loss_mask_rel[:, :, T - 1]                                                # [B, N]
    mk = lm_fp.reshape(B * N, 1)
    h, c = h * mk, c * mk
    attn_pred = (lm_fp.unsqueeze(2) * lm_fp.unsqueeze(1)).permute(0, 2, 1)
    mus, sxs, sys_, cors, samples = [], [], [], [], []
    x_sample = None
    for tt in range(P):
        if tt > 0:
            xt = _transformer_train(model, x_sample.reshape(B, N, 2), attn_pred, p_drop).reshape(B * N, 64) * mk
            hp, cp = model._lstm_cell(xt, h, c)
            h = hp * mk + h * (1 - mk)
            c = cp * mk + c * (1 - mk)
        raw = model.hidden2pos(h).view(B, N, 5).unsqueeze(1)
        mu = raw[..., :2]
        mus.append(mu); sxs.append(raw[..., 2:3].exp()); sys_.append(raw[..., 3:4].exp()); cors.append(raw[..., 4:5].tanh())
        x_sample = mu * lm_fp.unsqueeze(1).unsqueeze(-1)
        samples.append(x_sample)
    gp = (torch.cat(mus, 1), torch.cat(sxs, 1), torch.cat(sys_, 1), torch.cat(cors, 1))
    return gp, torch.cat(samples, 1), {"loss_mask_rel_full_partial": lm_fp, "loss_mask_per_pedestrian": lm_pp}


def negative_log_likelihood_full_partial(gaussian_params, x_target, loss_mask_ped, loss_mask_pred_seq):
    """st_model.py:62-112 -> (prob_loss [P,N] already masked, eventual_loss_mask [P,N])."""
    mu, sx, sy, corr = gaussian_params
    m_t = loss_mask_pred_seq.permute(0, 2, 1).unsqueeze(-1)
    m_p = loss_mask_ped.unsqueeze(1).unsqueeze(-1)
    mu = mu * m_t * m_p
    corr = corr * m_t * m_p
    x_target = x_target * m_t * m_p
    sx = (sx * m_t + (1. - m_t)) * m_p + (1. - m_p)
    sy = (sy * m_t + (1. - m_t)) * m_p + (1. - m_p)
    sigma = torch.cat((sx, sy), dim=3)
    xn = (x_target - mu) / sigma
    nx, ny = xn[..., 0:1], xn[..., 1:2]
    t1 = torch.log(1. - corr ** 2.) / 2. + torch.log(sx) + torch.log(sy)
    t2 = (nx ** 2. - 2. * corr * nx * ny + ny ** 2.) / (2. * (1. - corr ** 2.))
    prob_loss = (t1 + t2).squeeze(3).squeeze(0)
    elm = m_t[0, :, :, 0] * loss_mask_ped[0]
    return prob_loss * elm, elm


def average_offset_error(x_pred, x_target, loss_mask=None):
    """mgnn/utils.py:8-17."""
    err = torch.sqrt(((torch.cumsum(x_pred, 1) - torch.cumsum(x_target, 1)) ** 2.).sum(3))[0]
    aoe = err.mean(0)
    return aoe * loss_mask[0] if loss_mask is not None else aoe


def final_offset_error(x_pred, x_target, loss_mask=None):
    """mgnn/utils.py:19-28."""
    err = torch.sqrt(((torch.cumsum(x_pred, 1) - torch.cumsum(x_target, 1)) ** 2.).sum(3))[0]
    foe = err[-1]
    return foe * loss_mask[0] if loss_mask is not None else foe


def rotate_graph(vtx, theta):
    """mgnn/utils.py:80-90 (vertices only: the edge tensor is unused with spatial_num_heads_edges = 0)."""
    c, s = np.cos(theta), np.sin(theta)
    return torch.cat((vtx[..., 0:1] * c - vtx[..., 1:2] * s, vtx[..., 0:1] * s + vtx[..., 1:2] * c), dim=-1)


def sequence_loss(model, item, device, p_drop=0.1):
    """One step's loss exactly as train.py:113-137 computes it (non-deterministic branch: NLL / number of valid (step, pedestrian))."""
    obs_traj, pred_gt, obs_rel, pred_rel_gt, lm_rel, lm, v_obs, A_obs, v_pred_gt, A_pred_gt, am_obs, am_pred = item
    v_obs, v_pred_gt, am_obs, lm_rel = v_obs.to(device), v_pred_gt.to(device), am_obs.to(device), lm_rel.to(device)
    gp, xs, info = forward_train(model, v_obs, am_obs, lm_rel, p_drop)
    prob_loss, elm = negative_log_likelihood_full_partial(gp, v_pred_gt, info["loss_mask_rel_full_partial"], lm_rel[:, :, -model.pred_len:])
    return prob_loss.sum() / elm.sum(), gp, xs, info, v_pred_gt


class HipGstTrainer:
    """The training step of the predictor on the MI355X through the C ABI: forward + negative log-likelihood + backward as ONE boundary call
    (cn_gst_train_step, csrc/gst_train.hip: one workgroup per sequence, hand-derived reverse pass, the reference's four dropout sites with the
    library's own counter-based masks) and gradient-norm clip + Adam as another (cn_adam_clip_step over one flat bucket).  Replaces, per
    optimiser step, train.py:121-146: model(...) -> negative_log_likelihood_full_partial -> loss.backward() -> clip_grad_norm_ -> optimizer.step().
    The model's parameters become views of the flat bucket, so state_dict() / checkpoints are those of the torch path."""

    def __init__(self, model, lr=1e-3, clip_grad=10.0, betas=(0.9, 0.999), eps=1e-8, seed=1000, optimizer=None):
        from . import _abi as A
        self.A = A
        self.model = model
        params = [dict(model.named_parameters())[k] for _, k in A.GST_WEIGHT_KEYS]
        if not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            raise A.CnError("HipGstTrainer: the predictor must live on the GPU in float32 (there is no CPU fallback of the HIP path)")
        dev = params[0].device
        n = sum((p.numel() + 3) // 4 * 4 for p in params)         # every tensor 16-byte aligned inside the bucket
        self.flat = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k in ("p", "g", "m", "v")}
        self.w, self.g = A.GstWeights(), A.GstWeights()
        off = 0
        for (field, _), p in zip(A.GST_WEIGHT_KEYS, params):
            k = p.numel()
            pv, gv = self.flat["p"][off:off + k].view_as(p), self.flat["g"][off:off + k].view_as(p)
            pv.copy_(p.data)
            p.data = pv
            p.grad = gv
            setattr(self.w, field, pv.data_ptr())
            setattr(self.g, field, gv.data_ptr())
            if optimizer is not None:      # the torch optimiser object stays the owner of the moments (its state_dict() goes into the checkpoints)
                optimizer.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.flat["m"][off:off + k].view_as(p), "exp_avg_sq": self.flat["v"][off:off + k].view_as(p)}
            off += (k + 3) // 4 * 4
        self.optimizer = optimizer
        self.lr, self.clip_grad, self.betas, self.eps = float(lr), clip_grad, betas, float(eps)
        self.step_no, self.seed = 0, int(seed)
        self.adam_ws = torch.empty(A.lib().cn_adam_workspace_doubles(), dtype=torch.float64, device=dev)
        self.ws = None
        self.dev = dev

    def loss_and_grads(self, v_obs, v_pred, loss_mask_rel, p_drop=0.1, seed=None):
        """v_obs [B,5,N,2], v_pred [B,5,N,2], loss_mask_rel [B,N,10] (any device) -> (loss_and_count [2] on the device, gauss [B,5,N,5]: mu_x, mu_y,
        sigma_x, sigma_y, corr); the gradients land in the flat bucket (every parameter's .grad).  Crowds of fewer than 4 pedestrians are padded
        with absent ones."""
        A = self.A
        C = A.C
        B, T, N, _ = v_obs.shape
        if T != 5 or v_pred.shape[1] != 5 or N > 64:
            raise A.CnError("HipGstTrainer: 5 observed + 5 predicted steps and at most 64 pedestrians per sequence (got %d + %d steps, %d pedestrians)" % (T, v_pred.shape[1], N))
        Np = max(N, 4)
        f = lambda t: t.to(self.dev, torch.float32)   # noqa: E731
        vo, vp, lm = f(v_obs), f(v_pred), f(loss_mask_rel)
        if Np != N:
            vo = torch.nn.functional.pad(vo, (0, 0, 0, Np - N)); vp = torch.nn.functional.pad(vp, (0, 0, 0, Np - N)); lm = torch.nn.functional.pad(lm, (0, 0, 0, Np - N))
        vo, vp, lm = vo.contiguous(), vp.contiguous(), lm.contiguous()
        need = int(A.lib().cn_gst_train_workspace_bytes(B, Np))
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        out = torch.empty(2, device=self.dev)
        gauss = torch.empty(B, 5, Np, 5, device=self.dev)
        sd = self.seed + 7919 * self.step_no if seed is None else int(seed)
        with torch.cuda.device(self.dev):
            A.check(A.lib().cn_gst_train_step(B, Np, A.ptr(vo), A.ptr(vp), A.ptr(lm), C.byref(self.w), C.byref(self.g), float(p_drop), C.c_uint64(sd & (2 ** 64 - 1)),
                                              C.c_void_p(self.ws.data_ptr()), int(self.ws.numel()), A.ptr(out), A.ptr(gauss), A.stream_ptr()), "cn_gst_train_step")
        return out, gauss[:, :, :N]

    def optimizer_step(self):
        """clip_grad_norm_(parameters, clip_grad) + Adam.step() (train.py:143-146) over the flat bucket."""
        from . import hip
        self.step_no += 1
        if self.optimizer is not None:
            for st in self.optimizer.state.values():
                st["step"] = torch.tensor(float(self.step_no))
        hip.adam_clip_step(self.flat["p"], self.flat["g"], self.flat["m"], self.flat["v"], self.step_no, self.lr, self.betas, self.eps, self.clip_grad,
                           workspace=self.adam_ws)


def temperature(epoch, total_epochs, base_temp, temp_min=0.03):
    """temperature_scheduler.py (kept for the checkpoint / log; without edge heads the Gumbel temperature is never read)."""
    return max((1 - epoch / total_epochs) * (base_temp - temp_min) + temp_min, temp_min)


def evaluate(model, loader, device):
    """eval.py's `inference` in 'val' mode: mean loss over the sequences, aoe / foe over the fully observed pedestrians."""
    model.eval()
    losses, aoes, foes, ms = [], [], [], []
    with torch.no_grad():
        for item in loader:
            if item[6].shape[2] > 128:
                continue
            loss, gp, xs, info, v_pred_gt = sequence_loss(model, item, device, 0.0)
            lm = info["loss_mask_per_pedestrian"]
            losses.append(loss.item())
            aoes.append(average_offset_error(xs, v_pred_gt, lm).cpu().numpy()); foes.append(final_offset_error(xs, v_pred_gt, lm).cpu().numpy())
            ms.append(lm[0].cpu().numpy())
    m = max(float(np.concatenate(ms).sum()), 1.0)
    return float(np.mean(losses)), float(np.concatenate(aoes).sum() / m), float(np.concatenate(foes).sum() / m)


def train(data_dir, out_dir, num_epochs=100, temp_epochs=100, lr=1e-3, clip_grad=10.0, rotation_pattern="random", save_epochs=10, init_temp=0.5,
          random_seed=1000, device=None, num_workers=0, log=print, backend=None):
    """gst_updated/scripts/experiments/train.py:49-195 for the shipped configuration.  data_dir holds the text files of collect.py /
    collect_data.py; the first 80 % of every file's windows train, the rest validate (TrajectoriesDataset modes).  Writes
    <out_dir>/checkpoint/{epoch_<n>.pt, args.pickle, train_hist.pickle} in the reference's format (+ args.json / train_hist.json): the
    directory is a valid config.pred.model_dir for load_predictor here and for the reference's CrowdNavPredInterfaceMultiEnv."""
    torch.manual_seed(random_seed)
    np.random.seed(random_seed)
    device = torch.device(device if device is not None else ("cuda:0" if torch.cuda.is_available() else "cpu"))
    ds_train = TrajectoriesDataset(data_dir, mode="train")
    ds_val = TrajectoriesDataset(data_dir, mode="val")
    loader_train = DataLoader(ds_train, batch_size=1, shuffle=True, num_workers=num_workers)
    loader_val = DataLoader(ds_val, batch_size=1, shuffle=False, num_workers=num_workers)
    model = GSTPredictor().to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=max(int(temp_epochs / 4), 1), gamma=0.3)
    # backend: 'hip' = the training step as hand-written kernels through the C ABI (HipGstTrainer; the default on a GPU), 'torch' = the op graph
    # above under autograd (CPU tests, and the cross-check of the kernels)
    if backend is None:
        backend = "hip" if device.type == "cuda" else "torch"
    if backend == "hip" and clip_grad is None:
        raise ValueError("backend='hip' clips the gradient norm in its fused Adam step: pass a clip_grad (the reference's default is 10)")
    hip_tr = HipGstTrainer(model, lr=lr, clip_grad=clip_grad, seed=random_seed, optimizer=optimizer) if backend == "hip" else None
    ckpt_dir = os.path.join(out_dir, "checkpoint")
    os.makedirs(ckpt_dir, exist_ok=True)
    run_args = dict(spatial="gumbel_social_transformer", temporal="faster_lstm", output_dim=5, embedding_size=64, spatial_num_heads=8,
                    spatial_num_heads_edges=0, spatial_num_layers=1, ghost=False, lstm_hidden_size=64, lstm_num_layers=1, decode_style="recursive",
                    detach_sample=False, motion_dim=2, only_observe_full_period=False, dataset="sj", obs_seq_len=5, pred_seq_len=5, batch_size=1,
                    lr=lr, clip_grad=clip_grad, rotation_pattern=rotation_pattern, num_epochs=num_epochs, temp_epochs=temp_epochs,
                    save_epochs=save_epochs, init_temp=init_temp, random_seed=random_seed, deterministic=False, resume_training=False,
                    resume_epoch=None)
    with open(os.path.join(ckpt_dir, "args.json"), "w") as f:
        json.dump(run_args, f)
    # the reference's loaders (crowd_nav_interface_multi_env_parallel.py:21-28, scripts/experiments/eval.py:30) unpickle an
    # argparse.Namespace from args.pickle and open 'epoch_<args.num_epochs>.pt': write that too (train.py:88-89)
    with open(os.path.join(ckpt_dir, "args.pickle"), "wb") as f:
        pickle.dump(argparse.Namespace(**run_args), f)
    hist = {"epoch": 0, "train_loss_task": [], "val_loss_task": [], "train_aoe_task": [], "val_aoe_task": [], "train_foe_task": [], "val_foe_task": []}
    for epoch in range(1, num_epochs + 1):
        model.train()
        t0 = time.time()
        tau = temperature(epoch, temp_epochs, init_temp)
        losses, aoes, foes, ms = [], [], [], []
        for item in loader_train:
            if item[6].shape[2] > 128:            # train.py:118-119
                continue
            if rotation_pattern is not None:
                theta = (torch.randint(0, 4, ()).float() / 2. * np.pi).item() if rotation_pattern == "right_angle" else (torch.rand(()) * 2. * np.pi).item()
                item = list(item)
                item[6], item[8] = rotate_graph(item[6], theta), rotate_graph(item[8], theta)
            if hip_tr is not None:
                # one boundary call: forward, loss, backward (dropout 0.1 like model.train()); then the fused clip + Adam step
                lm_rel = item[4].to(device)
                hip_tr.lr = optimizer.param_groups[0]["lr"]                       # the StepLR schedule below drives the fused step too
                out, gauss = hip_tr.loss_and_grads(item[6], item[8], lm_rel, p_drop=0.1)
                hip_tr.optimizer_step()
                losses.append(out[0].item())
                lm_fp = lm_rel[:, :, model.obs_len - 1]
                xs = gauss[..., :2] * lm_fp.unsqueeze(1).unsqueeze(-1)
                lm = (lm_rel.sum(2) == lm_rel.shape[2]).float()
                v_pred_gt = item[8].to(device)
                aoes.append(average_offset_error(xs, v_pred_gt, lm).cpu().numpy()); foes.append(final_offset_error(xs, v_pred_gt, lm).cpu().numpy())
                ms.append(lm[0].cpu().numpy())
                continue
            loss, gp, xs, info, v_pred_gt = sequence_loss(model, item, device)
            losses.append(loss.item())
            loss.backward()
            lm = info["loss_mask_per_pedestrian"]
            aoes.append(average_offset_error(xs.detach(), v_pred_gt, lm).cpu().numpy()); foes.append(final_offset_error(xs.detach(), v_pred_gt, lm).cpu().numpy())
            ms.append(lm[0].cpu().numpy())
            if clip_grad is not None:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
            optimizer.step()
            optimizer.zero_grad()
        scheduler.step()
        m = max(float(np.concatenate(ms).sum()), 1.0)
        tr = (float(np.mean(losses)), float(np.concatenate(aoes).sum() / m), float(np.concatenate(foes).sum() / m))
        va = evaluate(model, loader_val, device)
        for k, a, b in (("loss", tr[0], va[0]), ("aoe", tr[1], va[1]), ("foe", tr[2], va[2])):
            hist["train_%s_task" % k].append(a); hist["val_%s_task" % k].append(b)
        hist["epoch"] = epoch
        log("Epoch: %d | train loss: %.4f | val loss: %.4f | train aoe: %.4f | val aoe: %.4f | train foe: %.4f | val foe: %.4f | tau %.3f | period: %.2f sec"
            % (epoch, tr[0], va[0], tr[1], va[1], tr[2], va[2], tau, time.time() - t0))
        if epoch % save_epochs == 0 or epoch == num_epochs:
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                        "lr_scheduler_state_dict": scheduler.state_dict(), "train_loss_epoch": tr[0], "val_loss_epoch": va[0], "train_aoe_epoch": tr[1],
                        "val_aoe_epoch": va[1], "train_foe_epoch": tr[2], "val_foe_epoch": va[2]}, os.path.join(ckpt_dir, "epoch_%d.pt" % epoch))
            with open(os.path.join(ckpt_dir, "train_hist.json"), "w") as f:
                json.dump(hist, f)
            with open(os.path.join(ckpt_dir, "train_hist.pickle"), "wb") as f:      # train.py:189-190 (resume reads it back)
                pickle.dump(hist, f)
    model.eval()
    return model, hist
