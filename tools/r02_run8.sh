cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
bash tools/ab_opts.sh "--option merged_rel_reduce=1" "--option merged_rel_reduce=0" 3
bash tools/ab_opts.sh "--config 3 --option merged_rel_reduce=1" "--config 3 --option merged_rel_reduce=0" 2
