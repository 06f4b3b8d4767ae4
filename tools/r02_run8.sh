cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
bash tools/ab_opts.sh "--option norm_single_stage=0" "--option norm_single_stage=4" 3
bash tools/ab_opts.sh "--config 3 --option norm_single_stage=0" "--config 3 --option norm_single_stage=4" 2
