"""The attempt-parallel placement loop of csrc/env_sim.hip (place_by_rejection: 64 candidates of a rejection-sampling loop per pass, read
straight out of the staged MT19937 block) must consume the stream exactly like the reference's one-candidate-at-a-time loop
(crowd_sim_var_num.py:116-146, crowd_sim.py:415-485): same accepted candidate, same stream position, same regenerated state -- for any
stream offset, for loops that straddle the 624-word block, and when the attempt bound ends the loop.  tests/native/placement_batch_check.cpp
restates both control flows on the host (the lanes as a loop) over 120 000 placements; the device code itself is pinned by the bit-exact
simulator tests of test_gpu_env.py (-m gpu)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_batched_rejection_sampling_consumes_the_stream_like_the_serial_loop(tmp_path):
    exe = str(tmp_path / "placement_batch_check")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(HERE, "native", "placement_batch_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout
