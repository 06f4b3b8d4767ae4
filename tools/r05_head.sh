cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "golden or intermediates or config1 or fp32_every or pad" 2>&1 | tail -5
( echo "A = round-4 tree (f313e5b), B = this tree; config 1"; bash tools/ab_rounds.sh "--config 1 --steps 30" 3
  echo "config 3"; bash tools/ab_rounds.sh "--config 3 --steps 20" 3
  echo "config 4"; bash tools/ab_rounds.sh "--config 4 --steps 20" 3
  echo "config 2"; bash tools/ab_rounds.sh "--config 2 --steps 10" 2 ) 2>&1 | tee $O/ab_rounds.txt
bash tools/kstat.sh "--config 1" "head_conv" 2>&1 | tail -3
