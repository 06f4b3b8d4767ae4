# final build of round 3: HBM traffic counters (build-stamped), the four bench lines, the all-events line, the two config-1 rocprofv3
# summaries, the one-rank RCCL path and the smoke check
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f2; mkdir -p $O
timeout 1200 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -18 $O/pmc_traffic.log
for c in 1 2 3 4; do
  extra="--no-cpu-baseline"; [ $c = 1 ] && extra=""
  timeout 600 python bench.py --config $c $extra > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-260 $O/bench_c$c.json
done
timeout 300 python bench.py --config 1 --no-cpu-baseline --all-kernel-events > $O/bench_c1_all_events.json 2>/dev/null
prof() {  # name, bench args...
  n=$1; shift
  rm -rf $O/prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 bench.py "$@" --no-cpu-baseline > $O/prof_$n.log 2>&1
  find $O/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$n.csv
  find $O/prof_$n -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_$n.csv
  find $O/prof_$n -type f -delete 2>/dev/null
  head -3 $O/kernel_stats_$n.csv | cut -c1-150
}
prof c1_default --config 1
python tools/timeline.py <(head -1 $O/kernel_trace_c1_default.csv) $O/kernel_trace_c1_default.csv 0 10 > $O/timeline_c1_default.txt 2>&1
prof c1_single --config 1 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
rm -f $O/kernel_trace_*.csv
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-comm-path --no-cpu-baseline > $O/bench_c1_rccl_one_rank.json 2> $O/bench_rccl.err; cut -c1-200 $O/bench_c1_rccl_one_rank.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
