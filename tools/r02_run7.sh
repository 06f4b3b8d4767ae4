cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "single_stage" 2>&1 | tail -5
bash tools/ab_opts.sh "--option norm_single_stage=1" "--option norm_single_stage=0" 2 2>&1 | tee $O/ab_norm1.log
bash tools/ab_opts.sh "--option norm_single_stage=2" "--option norm_single_stage=0" 2 2>&1 | tee $O/ab_norm2.log
bash tools/ab_opts.sh "--config 3 --option norm_single_stage=1" "--config 3 --option norm_single_stage=0" 2 2>&1 | tee $O/ab_norm_c3.log
