"""MI355X-native hot path of CrowdNav++ (Shuijing725/CrowdNav_Prediction_AttnGraph).

Host side (Python on PyTorch-ROCm) mirrors the reference's interfaces for this path -- the gym-style env ids,
`make_vec_envs`, `Policy`, `RolloutStorage`, `PPO.update` -- over hand-written HIP kernels reached through the C ABI in
include/crowdnav_hip.h (crowdnav_prediction_attngraph_amd/libcrowdnav_hip.so).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
