cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_swin.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -3
bash tools/ab_opts.sh "--config 2" "--config 2 --no-overlap-semantic" 3
