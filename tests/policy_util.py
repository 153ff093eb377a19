"""Formula-filled policy weights so parity fixtures need no 10 MB weight file.

weight(t, k) for the t-th state-dict tensor (key order) and flat index k is a pure integer hash -> exact in
float64 on every machine, so the GPU box can rebuild exactly the weights the golden outputs were produced with.
"""
import numpy as np


def _hash_u32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x = (x ^ (x >> 16)) * np.uint64(0x7FEB352D) & 0xFFFFFFFF
    x = (x ^ (x >> 15)) * np.uint64(0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x


def formula_tensor(t, shape, fan_in=None):
    n = int(np.prod(shape))
    k = np.arange(n, dtype=np.uint64) + np.uint64(0x9E3779B9) * np.uint64(t + 1)
    u = (_hash_u32(k) >> np.uint64(8)).astype(np.float64) / float(1 << 24) * 2.0 - 1.0  # uniform (-1, 1)
    if len(shape) >= 2 and shape[-1] > 1 and shape[0] > 1:
        amp = 1.6 / np.sqrt(shape[-1] if fan_in is None else fan_in)
    else:
        amp = 0.1
    return (amp * u).astype(np.float32).reshape(shape)


def formula_state_dict(shapes):
    """shapes: ordered {key: shape} -> {key: float32 ndarray}."""
    return {k: formula_tensor(t, tuple(s)) for t, (k, s) in enumerate(shapes.items())}


def synth_obs(E, H, D, seed):
    """Plausible observation batch (float32) with sorted humans, variable detected counts."""
    rs = np.random.RandomState(seed)
    robot_node = np.concatenate([rs.uniform(-6, 6, (E, 2)), np.full((E, 1), 0.3), rs.uniform(-6, 6, (E, 2)),
                                 np.ones((E, 1)), np.full((E, 1), np.pi / 2)], axis=1).astype(np.float32).reshape(E, 1, 7)
    temporal = rs.uniform(-1, 1, (E, 1, 2)).astype(np.float32)
    det = rs.randint(1, H + 1, size=E)
    det[0] = 1
    det[-1] = H
    spatial = np.full((E, H, D), 15.0, dtype=np.float32)
    for e in range(E):
        p = rs.uniform(-4, 4, (det[e], 2))
        p = p[np.argsort(np.linalg.norm(p, axis=1))]
        v = rs.uniform(-1, 1, (det[e], 2))
        for k in range(D // 2):
            spatial[e, :det[e], 2 * k:2 * k + 2] = p + 0.25 * k * v
    return dict(robot_node=robot_node, temporal_edges=temporal, spatial_edges=spatial,
                detected_human_num=det.astype(np.float32).reshape(E, 1))


def gst_wrapper_stream(E, H, T, seed):
    """Deterministic observation stream for the VecPretextNormalize tests (tests/golden/make_golden_gst.py --wide and its consumers):
    humans drift with constant velocities, the robot random-walks, visibility = sensor range with one flickering human.  Returns T dicts of
    numpy arrays (robot_node [E,1,7], spatial_edges [E,H,12] = the relative position tiled, visible_masks [E,H] bool, rews_in [E,1])."""
    rs = np.random.RandomState(seed)
    pos = rs.uniform(-5, 5, (E, H, 2))
    vel = rs.uniform(-0.25, 0.25, (E, H, 2))
    robot = rs.uniform(-3, 3, (E, 2))
    out = []
    for t in range(T):
        pos = pos + vel
        robot = robot + rs.uniform(-0.2, 0.2, (E, 2))
        rel = pos - robot[:, None, :]
        vis = np.linalg.norm(rel, axis=-1) - 0.6 <= 5.0
        vis[:, 5 % H] = t % 3 != 0
        se = np.where(vis[..., None], rel, 15.0)
        out.append(dict(
            robot_node=np.concatenate([robot, np.full((E, 1), 0.3), robot * 0, np.ones((E, 1)), np.full((E, 1), 1.57)], 1).astype(np.float32).reshape(E, 1, 7),
            spatial_edges=np.tile(se, (1, 1, 6)).astype(np.float32), visible_masks=vis.copy(),
            rews_in=rs.uniform(-1, 1, (E, 1)).astype(np.float32)))
    return out
