# round 2, GPU call 2: parity of the new default path, fused vs unfused A/B, single-stream rocprof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -40 > $O/pytest2.log
tail -12 $O/pytest2.log
bash tools/ab_opts.sh "--option fused_attn_fwd=1" "--option fused_attn_fwd=0" 3 2>&1 | tee $O/ab_fused.log
rm -rf $O/prof_single
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-side-stream > $O/prof_single.log 2>&1
find $O/prof_single -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_single_fused.csv
find $O/prof_single -type f -delete 2>/dev/null
head -30 $O/kernel_stats_single_fused.csv | cut -c1-140
