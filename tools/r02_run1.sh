# round 2, first GPU call: whole -m gpu suite (no -x: see every failure), bench lines of configs 1-4,
# rocprofv3 kernel stats of configs 2 and 4, two DP ranks on one GPU
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -150 > $O/pytest1.log
tail -25 $O/pytest1.log
for c in 1 2 3 4; do
  extra="--no-cpu-baseline"; [ $c = 1 ] && extra=""
  timeout 600 python bench.py --config $c $extra > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-600 $O/bench_c$c.json
done
for c in 2 4; do
  rm -rf $O/prof_c$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$c -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/prof_c$c.log 2>&1
  find $O/prof_c$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c$c.csv
  find $O/prof_c$c -type f -delete 2>/dev/null
  head -6 $O/kernel_stats_c$c.csv | cut -c1-160
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp2_check.py 2>&1 | tail -3 | tee $O/dp2.log
