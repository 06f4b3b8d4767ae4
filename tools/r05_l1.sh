cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "l1_seed or golden or config1 or storage_option or two_steps or graph" 2>&1 | tail -8
( echo "A = fused_l1 0, B = default (1); config 1"; bash tools/ab_opts.sh "--option fused_l1=0" "" 3
  echo "config 3"; bash tools/ab_opts.sh "--config 3 --option fused_l1=0" "--config 3" 2 ) 2>&1 | tee $O/ab_l1.txt
bash tools/kstat.sh "--config 1" "tail_bwd\|clamp_l1\|loss_finish" 2>&1 | tail -5
