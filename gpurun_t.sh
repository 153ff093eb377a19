cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gst_host.py -x -q -m gpu --timeout 300 > gpurun_out/t_gst.log 2>&1; tail -25 gpurun_out/t_gst.log
