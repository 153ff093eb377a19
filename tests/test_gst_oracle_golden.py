"""Pin the numpy GST oracle against golden outputs of the reference's own torch code (tests/golden/gst_e4_h20.npz)."""
import json
import os
import sys

import numpy as np

from oracle import gst_oracle as G
from tests.golden_util import GOLDEN

sys.path.insert(0, GOLDEN)
from make_golden_gst import gst_formula_state_dict  # noqa: E402  (pure numpy helper: formula weights)


def _load():
    z = np.load(os.path.join(GOLDEN, "gst_e4_h20.npz"))
    meta = json.loads(str(z["meta"]))
    sd = gst_formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()})
    return z, meta, sd


def test_interface_forward_matches_reference():
    z, meta, sd = _load()
    for case in ("a", "b"):
        out, mask = G.interface_forward(sd, z["in_traj_" + case], z["in_mask_" + case])
        np.testing.assert_array_equal(mask, z["out_mask_" + case])
        valid = z["out_mask_" + case][..., 0] > 0
        assert 0 < valid.sum() < valid.size
        np.testing.assert_allclose(out[valid], z["out_traj_" + case][valid], rtol=1e-4, atol=1e-4)
        # unpredicted humans carry the -999 marker in the position slots
        assert np.all(z["out_traj_" + case][~valid][..., :2] == -999.0) and np.all(out[~valid][..., :2] == -999.0)


def test_wrapper_process_obs_rew_matches_reference():
    z, meta, sd = _load()
    E, H, T = meta["E"], meta["H"], meta["T"]
    w = G.PretextWrapper(sd, E, H)
    for t in range(T):
        se, rews = w.process(z["w_in_robot_node_%d" % t], z["w_in_spatial_edges_%d" % t], z["w_in_visible_masks_%d" % t], z["w_in_rews_%d" % t])
        np.testing.assert_allclose(se, z["w_out_spatial_edges_%d" % t], rtol=1e-4, atol=1e-4, err_msg="spatial_edges @%d" % t)
        np.testing.assert_allclose(rews, z["w_out_rews_%d" % t], atol=1e-5, err_msg="rews @%d" % t)


def test_interface_forward_matches_reference_with_the_shipped_weights():
    """tests/golden/gst_real_e4_h20.npz carries the reference's shipped predictor weights (epoch_100.pt, 67 269 parameters) and
    its outputs: the oracle at the REAL weight magnitudes."""
    z = np.load(os.path.join(GOLDEN, "gst_real_e4_h20.npz"))
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w/")}
    assert sum(v.size for v in sd.values()) == 67269
    for case in ("a", "b", "c"):
        out, mask = G.interface_forward(sd, z["in_traj_" + case], z["in_mask_" + case])
        np.testing.assert_array_equal(mask, z["out_mask_" + case])
        valid = z["out_mask_" + case][..., 0] > 0
        np.testing.assert_allclose(out[valid], z["out_traj_" + case][valid], rtol=1e-4, atol=1e-4)
