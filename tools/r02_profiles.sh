# round 2 evidence run: whole -m gpu suite, bench lines of configs 1-4, rocprofv3 kernel stats (config 1 default and
# single stream, configs 2 and 4), SQ counters incl. SQ_VALU_MFMA_BUSY_CYCLES, HBM traffic counters.
# Everything lands in gpurun_out/r02p/; the files to keep are copied into profiles/ by hand afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p
mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -60 > $O/pytest.log
tail -8 $O/pytest.log
# HBM traffic counters first: bench.py then finds profiles/pmc_traffic.json with this build's stamp and reports roofline.traffic
timeout 1200 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -16 $O/pmc_traffic.log
for c in 1 2 3 4; do
  extra="--no-cpu-baseline"; [ $c = 1 ] && extra=""
  timeout 600 python bench.py --config $c $extra > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-400 $O/bench_c$c.json
done
prof() {  # name, bench args...
  n=$1; shift
  rm -rf $O/prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 bench.py "$@" --no-cpu-baseline > $O/prof_$n.log 2>&1
  find $O/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$n.csv
  find $O/prof_$n -type f -delete 2>/dev/null
  head -4 $O/kernel_stats_$n.csv | cut -c1-150
}
prof c1_default --config 1
prof c1_single --config 1 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c2_default --config 2 --steps 5 --warmup 2 --no-kernel-events
prof c4_default --config 4 --steps 5 --warmup 2 --no-kernel-events
timeout 900 python tools/pmc_sq.py > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq.txt; head -12 $O/pmc_sq.txt | cut -c1-260
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp2_check.py 2>&1 | tail -3 | tee $O/dp2.log
