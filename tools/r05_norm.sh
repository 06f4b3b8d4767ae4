cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "instnorm_backward_reduction or bitwise_reproducible or c16_gather" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -x -k "config1" 2>&1 | tail -5
( echo "A = fused_norm_red 0, B = default; config 1"; bash tools/ab_opts.sh "--option fused_norm_red=0" "" 3
  echo "config 3"; bash tools/ab_opts.sh "--config 3 --option fused_norm_red=0" "--config 3" 2
  echo "config 4"; bash tools/ab_opts.sh "--config 4 --option fused_norm_red=0" "--config 4" 2 ) 2>&1 | tee $O/ab_norm.txt
bash tools/kstat.sh "--config 1" "c16_dgrad\|instnorm" 2>&1 | tail -8
