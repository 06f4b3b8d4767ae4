cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -2
bash tools/ab_opts.sh "--option wgrad_big_tiles=1" "--option wgrad_big_tiles=0" 3
bash tools/ab_opts.sh "--config 3 --option wgrad_big_tiles=1" "--config 3 --option wgrad_big_tiles=0" 2
