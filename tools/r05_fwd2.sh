cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "two_windows_per_cu" 2>&1 | tail -15
for v in 1 2; do ./scratch/bench_fwd2 32 $v; ./scratch/bench_fwd2 64 $v; done
./scratch/bench_fwd2_st 32 2
./scratch/bench_fused 256 32 0 1
( echo "config 3: A = fused_attn_fwd2 0, B = auto (variant 2)"; bash tools/ab_opts.sh "--config 3 --option fused_attn_fwd2=0" "--config 3" 3
  echo "config 3: A = variant 1, B = variant 2"; bash tools/ab_opts.sh "--config 3 --option fused_attn_fwd2=1" "--config 3 --option fused_attn_fwd2=2" 2
  echo "config 4: A = off, B = auto"; bash tools/ab_opts.sh "--config 4 --option fused_attn_fwd2=0" "--config 4" 2 ) 2>&1 | tee $O/ab_fwd2.txt
