#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_base.so" CN_TN_NB=2 CN_TN_NB=4 "CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_base.so" CN_TN_NB=2 CN_TN_NB=4; do
  echo "== $v"; env $v PYTHONPATH=$GRAFT_REPO_ROOT timeout 200 python tools/tn_bench.py 2>&1 | grep -v amdgpu.ids
done
