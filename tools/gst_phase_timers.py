"""Per-phase shader-clock sums of the GST kernels (run on an MI355X against a -DGST_TIMING build of the library):
    make -C crowdnav_prediction_attngraph_amd/csrc GSTFLAGS=-DGST_TIMING OUT=... ; CN_HIP_LIB=<that build> python tools/gst_phase_timers.py
Thread 0 of every workgroup adds clock64() differences into [workgroup][16]: slots 0..7 the layer kernel's phases, 8..12 the LSTM kernel's
(CN_GST_REUSE=0: every launch encodes its whole window)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from crowdnav_prediction_attngraph_amd import _abi as A
from crowdnav_prediction_attngraph_amd.hip import HipGST
from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
E, H = 2048, 20
lib = A.lib()
lib.cn_gst_set_timing.argtypes = [C.c_void_p]
g = HipGST(H, E)
torch.manual_seed(0)
g.set_weights(GSTPredictor().state_dict())
traj = torch.randn(E, H, 5, 2, device="cuda").cumsum(2) * 0.3
mask = (torch.rand(E, H, 5, device="cuda") > 0.1).float()
buf = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
for _ in range(3): g.predict(traj, mask)
torch.cuda.synchronize()
lib.cn_gst_set_timing(buf.data_ptr())
N = 10
for _ in range(N): g.predict(traj, mask)
torch.cuda.synchronize()
lib.cn_gst_set_timing(None)
t = buf.view(256, 16).double().mean(0).cpu().numpy() / N
print("mean shader-clock cycles per workgroup and forward, by slot (layer: 0 stage inputs + embedding + LayerNorm, 1 in_proj, 2 attention core, 3 out_proj, 4 LayerNorm,")
print("5 linear1, 7 linear2 + store; LSTM: 8 tile load, 9 stage x, 10 gates (MFMA), 11 cell, 12 state store + head):")
print(np.round(t).astype(np.int64).tolist(), "sum", int(t.sum()))
