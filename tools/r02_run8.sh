cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "fused_projection or bitwise" 2>&1 | tail -2
bash tools/ab_opts.sh "--option fused_c16_dgrad=1" "--option fused_c16_dgrad=0" 3 2>&1 | tee $O/ab_dgrad16.log
bash tools/ab_opts.sh "--config 3 --option fused_c16_dgrad=1" "--config 3 --option fused_c16_dgrad=0" 2 2>&1 | tee $O/ab_dgrad16_c3.log
