"""The C-ABI library loads on a CPU-only box and exports every symbol include/crowdnav_hip.h declares; device entry
points fail loudly (no CPU fallback) when there is no GPU.  No compute is launched here."""
import ctypes as C
import os
import re

import pytest

from crowdnav_prediction_attngraph_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "crowdnav_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    if not os.path.exists(A.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(A.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libcrowdnav_hip.so does not export %s" % name
    assert sorted(A.ABI_SYMBOLS) == declared, "python binding and header disagree"


def test_struct_layouts_match_the_header():
    lib = A.lib()
    cfg = A.default_env_config()
    assert cfg.human_num == 20 and cfg.predict_steps == 5 and cfg.time_step == 0.25 and cfg.orca_time_horizon_obst == 5.0
    assert abs(cfg.circle_radius - 6 * 2 ** 0.5) < 1e-12 and cfg.test_size == 500
    assert lib.cn_env_obs_width(C.byref(cfg)) == 2
    cfg.env_kind = 1
    assert lib.cn_env_obs_width(C.byref(cfg)) == 12
    assert C.sizeof(A.PolicyWeights) == 8 * len(A.POLICY_WEIGHT_KEYS) == 8 * 45
    # the binding refuses a library of another ABI version (A.lib() above raised otherwise); header, library and binding agree
    hdr = open(os.path.join(ROOT, "include", "crowdnav_hip.h")).read()
    assert lib.cn_version() == A.ABI_VERSION == int(re.search(r"#define CN_ABI_VERSION (\d+)", hdr).group(1))
    assert (A.PROF_KERNELS, A.PROF_SLOT_WORDS) == (int(re.search(r"CN_PROF_KERNELS = (\d+)", hdr).group(1)), int(re.search(r"CN_PROF_SLOT_WORDS = (\d+)", hdr).group(1)))


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = A.lib()
    h = C.c_void_p()
    cfg = A.default_env_config()
    rc = lib.cn_env_create(C.byref(cfg), 4, 425, 0, C.byref(h))
    assert rc == -3 and b"no HIP device" in lib.cn_last_error()
    rc = lib.cn_policy_create(20, 2, 4, C.byref(h))
    assert rc == -3
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    with pytest.raises(A.CnError):
        make_vec_envs("CrowdSimVarNum-v0", 425, 4, 0.99, None, "cuda", False)
