cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -15
bash tools/ab_opts.sh "--option fused_c16_fwd=1" "--option fused_c16_fwd=0" 3 2>&1 | tee $O/ab_c16.log
