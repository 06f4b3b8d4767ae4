cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_text.py tests/test_metrics.py tests/test_gpu_swin.py -q -m gpu -x 2>&1 | tail -15 | tee $O/pytest.log
