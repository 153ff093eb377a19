#!/usr/bin/env python3
"""Golden vectors for the GST predictor path (BASELINE configs[3]) from the reference's own code (build container only):
  * CrowdNavPredInterfaceMultiEnv.forward (gst_updated/scripts/wrapper/crowd_nav_interface_parallel.py:45-114) over
    st_model.forward (gst_updated/src/gumbel_social_transformer/st_model.py:271-455) with formula weights;
  * VecPretextNormalize.process_obs_rew (rl/vec_env/vec_pretext_normalize.py:112-191) over a short observation sequence.
The shipped checkpoint's hyper-parameters are rebuilt as a Namespace; no pickle is loaded."""
import argparse
import json
import os
import sys
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import as R  # noqa: E402
import policy_util as PU  # noqa: E402

GST_ARGS = dict(spatial="gumbel_social_transformer", temporal="faster_lstm", motion_dim=2, output_dim=5, embedding_size=64,
                spatial_num_heads=8, spatial_num_heads_edges=0, spatial_num_layers=1, ghost=False, lstm_hidden_size=64,
                lstm_num_layers=1, obs_seq_len=5, pred_seq_len=5, only_observe_full_period=False, decode_style="recursive",
                detach_sample=False, num_epochs=100, init_temp=0.5)


def gst_formula_state_dict(shapes):
    """Formula weights with magnitudes that keep LayerNorm / exp / tanh in a sane range."""
    sd = {}
    for t, (k, shp) in enumerate(shapes.items()):
        w = PU.formula_tensor(100 + t, tuple(shp))
        if k.endswith("norm_node.weight") or k.endswith("norm1_node.weight"):
            w = (1.0 + 0.5 * w / 0.1).astype(np.float32)   # around 1
        if k.startswith("hidden2pos"):
            w = (0.3 * w).astype(np.float32)
        sd[k] = w
    return sd


def synth_traj(E, H, seed):
    rs = np.random.RandomState(seed)
    pos0 = rs.uniform(-6, 6, (E, H, 1, 2))
    vel = rs.uniform(-0.3, 0.3, (E, H, 1, 2))
    traj = (pos0 + vel * np.arange(5).reshape(1, 1, 5, 1) + 0.02 * rs.standard_normal((E, H, 5, 2))).astype(np.float32)
    mask = (rs.uniform(size=(E, H, 5, 1)) > 0.25)
    mask[:, 0] = True            # always-visible human
    mask[:, 1] = False           # never-visible human
    mask[:, 2, :4] = True; mask[:, 2, 4] = False    # disappears at the last step -> not predicted
    mask[:, 3, :3] = False; mask[:, 3, 3:] = True   # appears late
    traj = np.where(mask, traj, -999.0).astype(np.float32)
    return traj, mask.astype(np.float32)


REAL_CKPT = ("/root/reference/gst_updated/results/100-gumbel_social_transformer-faster_lstm-lr_0.001-init_temp_0.5-edge_head_0-ebd_64-snl_1-snh_8-"
             "seed_1000_rand/sj/checkpoint/epoch_100.pt")


def real_state_dict():
    """The shipped predictor weights (config.pred.model_dir of the reference, 67 269 parameters) as numpy arrays."""
    from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
    m = GSTPredictor.from_checkpoint(REAL_CKPT, "cpu")      # weights_only load with the numpy allow-list (no pickle code runs)
    return {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}


def build_predictor(E, real=False):
    import torch
    from gst_updated.scripts.wrapper.crowd_nav_interface_parallel import CrowdNavPredInterfaceMultiEnv
    from gst_updated.src.gumbel_social_transformer.st_model import st_model
    args = argparse.Namespace(**GST_ARGS)
    torch.manual_seed(0)
    model = st_model(args, device="cpu")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = real_state_dict() if real else gst_formula_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    pred = CrowdNavPredInterfaceMultiEnv.__new__(CrowdNavPredInterfaceMultiEnv)
    pred.args = pred.args_eval = args
    pred.device = torch.device("cpu")
    pred.nenv = E
    pred.model = model
    return pred, shapes


def main_real():
    """Second fixture: the SHIPPED weights (stored in the fixture, 270 KB) and the reference predictor's outputs on three input
    sets -- pins the kernels / oracle at the weight magnitudes of the real model, not only at formula weights."""
    R.install()
    import torch
    E, H = 4, 20
    pred, shapes = build_predictor(E, real=True)
    out = {"w/" + k: v for k, v in real_state_dict().items()}
    for case, seed in (("a", 11), ("b", 12), ("c", 13)):
        traj, mask = synth_traj(E, H, seed)
        with torch.no_grad():
            o_traj, o_mask = pred.forward(input_traj=torch.from_numpy(traj), input_binary_mask=torch.from_numpy(mask))
        out["in_traj_" + case], out["in_mask_" + case] = traj, mask
        out["out_traj_" + case], out["out_mask_" + case] = o_traj.numpy(), o_mask.numpy()
    out["meta"] = np.array(json.dumps(dict(E=E, H=H, args=GST_ARGS, shapes={k: list(v) for k, v in shapes.items()}, source=REAL_CKPT)))
    path = os.path.join(HERE, "gst_real_e4_h20.npz")
    np.savez_compressed(path, **out)
    print("gst real-weights golden -> %s (%.0f KB); out_traj_a[0,0,0]=%s" % (os.path.basename(path), os.path.getsize(path) / 1024, out["out_traj_a"][0, 0, 0]))


def main_wide():
    """Third fixture (round 4): VecPretextNormalize.process_obs_rew of the reference at 64 envs x 20 humans over 8 steps, with the shipped
    prediction stride (pred_interval 1) AND with data.pred_timestep = 2 x env.time_step (pred_interval 2: a 9-deep history read every second
    entry, vec_pretext_normalize.py:56-57, :133-134).  Inputs come from tests/policy_util.gst_wrapper_stream (regenerated by the consumer),
    the file holds the reference's outputs only."""
    R.install()
    import torch
    from rl.vec_env.vec_pretext_normalize import VecPretextNormalize
    E, H, T = 64, 20, 8
    pred, shapes = build_predictor(E)
    cfg = R.make_config(**{"sim.human_num": H, "sim.predict_method": "inferred", "env.use_wrapper": True})
    out = {}
    for interval in (1, 2):
        L = 4 * interval + 1
        w = VecPretextNormalize.__new__(VecPretextNormalize)
        w.config, w.device, w.num_envs, w.max_human_num, w.predictor = cfg, torch.device("cpu"), E, H, pred
        w.pred_interval, w.buffer_len = interval, L                  # what __init__ derives at :56-57
        w.traj_buffer = deque(list(-torch.ones((L, E, H, 2)) * 999), maxlen=L)      # reset(), :88-91
        w.mask_buffer = deque(list(torch.zeros((L, E, H, 1), dtype=torch.bool)), maxlen=L)
        w.step_counter = 0
        w.last_pos = torch.zeros(E, H, 2)
        se_out, rew_out = [], []
        for t, o in enumerate(PU.gst_wrapper_stream(E, H, T + 2 * interval, 31)):
            O = {"robot_node": torch.from_numpy(o["robot_node"]), "temporal_edges": torch.zeros(E, 1, 2), "spatial_edges": torch.from_numpy(o["spatial_edges"].copy()),
                 "visible_masks": torch.from_numpy(o["visible_masks"]), "detected_human_num": torch.from_numpy(np.maximum(o["visible_masks"].sum(1), 1).astype(np.float32).reshape(E, 1))}
            obs, rews = w.process_obs_rew(O, np.zeros(E), rews=o["rews_in"].copy())
            se_out.append(obs["spatial_edges"].numpy().copy())
            rew_out.append(np.asarray(rews, dtype=np.float32))
        out["se_i%d" % interval] = np.stack(se_out)
        out["rews_i%d" % interval] = np.stack(rew_out)
    out["meta"] = np.array(json.dumps(dict(E=E, H=H, T=T, seed=31, intervals=[1, 2], steps={"1": T + 2, "2": T + 4}, args=GST_ARGS,
                                           shapes={k: list(v) for k, v in shapes.items()})))
    path = os.path.join(HERE, "gst_wrapper_e64_h20.npz")
    np.savez_compressed(path, **out)
    print("gst wrapper golden -> %s (%.0f KB); rews_i2[-1][:4]=%s" % (os.path.basename(path), os.path.getsize(path) / 1024, out["rews_i2"][-1].ravel()[:4]))


def main():
    if "--real" in sys.argv:
        return main_real()
    if "--wide" in sys.argv:
        return main_wide()
    R.install()
    import torch
    E, H = 4, 20
    pred, shapes = build_predictor(E)
    out = {}
    for case, seed in (("a", 1), ("b", 2)):
        traj, mask = synth_traj(E, H, seed)
        with torch.no_grad():
            o_traj, o_mask = pred.forward(input_traj=torch.from_numpy(traj), input_binary_mask=torch.from_numpy(mask))
        out["in_traj_" + case] = traj
        out["in_mask_" + case] = mask
        out["out_traj_" + case] = o_traj.numpy()
        out["out_mask_" + case] = o_mask.numpy()
    # ---- wrapper: VecPretextNormalize.process_obs_rew over a 7-step observation sequence ----
    from rl.vec_env.vec_pretext_normalize import VecPretextNormalize
    cfg = R.make_config(**{"sim.human_num": H, "sim.predict_method": "inferred", "env.use_wrapper": True})
    w = VecPretextNormalize.__new__(VecPretextNormalize)
    w.config = cfg
    w.device = torch.device("cpu")
    w.num_envs = E
    w.max_human_num = H
    w.predictor = pred
    w.pred_interval = 1
    w.buffer_len = 5
    w.traj_buffer = deque(list(-torch.ones((5, E, H, 2)) * 999), maxlen=5)
    w.mask_buffer = deque(list(torch.zeros((5, E, H, 1), dtype=torch.bool)), maxlen=5)
    w.step_counter = 0
    w.last_pos = torch.zeros(E, H, 2)
    rs = np.random.RandomState(9)
    pos = rs.uniform(-5, 5, (E, H, 2))
    vel = rs.uniform(-0.25, 0.25, (E, H, 2))
    robot = rs.uniform(-3, 3, (E, 2))
    T = 7
    for t in range(T):
        pos = pos + vel
        robot = robot + rs.uniform(-0.2, 0.2, (E, 2))
        rel = pos - robot[:, None, :]
        vis = np.linalg.norm(rel, axis=-1) - 0.6 <= 5.0
        vis[:, 5] = t % 3 != 0   # flickering human
        se = np.where(vis[..., None], rel, 15.0)
        O = {"robot_node": torch.from_numpy(np.concatenate([robot, np.full((E, 1), 0.3), robot * 0, np.ones((E, 1)), np.full((E, 1), 1.57)], 1)
                                            .astype(np.float32).reshape(E, 1, 7)),
             "temporal_edges": torch.zeros(E, 1, 2),
             "spatial_edges": torch.from_numpy(np.tile(se, (1, 1, 6)).astype(np.float32)),
             "visible_masks": torch.from_numpy(vis),
             "detected_human_num": torch.from_numpy(np.maximum(vis.sum(1), 1).astype(np.float32).reshape(E, 1))}
        out["w_in_robot_node_%d" % t] = O["robot_node"].numpy().copy()
        out["w_in_spatial_edges_%d" % t] = O["spatial_edges"].numpy().copy()
        out["w_in_visible_masks_%d" % t] = O["visible_masks"].numpy().copy()
        rews_in = rs.uniform(-1, 1, (E, 1)).astype(np.float32)
        obs, rews = w.process_obs_rew(O, np.zeros(E), rews=rews_in.copy())
        out["w_in_rews_%d" % t] = rews_in
        out["w_out_spatial_edges_%d" % t] = obs["spatial_edges"].numpy().copy()
        out["w_out_rews_%d" % t] = np.asarray(rews, dtype=np.float32)
    out["meta"] = np.array(json.dumps(dict(E=E, H=H, T=T, args=GST_ARGS, shapes={k: list(v) for k, v in shapes.items()})))
    path = os.path.join(HERE, "gst_e4_h20.npz")
    np.savez_compressed(path, **out)
    print("gst golden -> %s (%.0f KB); out_traj[0,0]=%s mask sum=%s; wrapper rews[-1]=%s" % (
        os.path.basename(path), os.path.getsize(path) / 1024, out["out_traj_a"][0, 0, 0], out["out_mask_a"].sum(), out["w_out_rews_%d" % (T - 1)].ravel()))
    for k, v in shapes.items():
        print("  ", k, v)


if __name__ == "__main__":
    main()
