# round 3 evidence run: whole -m gpu suite, HBM traffic counters, bench lines of configs 1-4, rocprofv3 kernel stats (config 1
# default and single stream, configs 2-4), SQ counters, option A/B on one box, the conv microbenchmarks, the bf16 quality probe.
# Everything lands in gpurun_out/r03e/; the files to keep are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -40 > $O/pytest.log
tail -3 $O/pytest.log
# HBM traffic counters first: bench.py then finds profiles/pmc_traffic.json with this build's stamp and reports roofline.traffic
timeout 1200 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -16 $O/pmc_traffic.log
for c in 1 2 3 4; do
  extra="--no-cpu-baseline"; [ $c = 1 ] && extra=""
  timeout 600 python bench.py --config $c $extra > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-300 $O/bench_c$c.json
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
prof c2_default --config 2 --steps 5 --warmup 2 --no-kernel-events
prof c3_default --config 3 --steps 5 --warmup 2 --no-kernel-events
prof c4_default --config 4 --steps 5 --warmup 2 --no-kernel-events
rm -f $O/kernel_trace_*.csv
timeout 900 python tools/pmc_sq.py > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq.txt; head -12 $O/pmc_sq.txt | cut -c1-260
# same-box option A/B (config 1, 20 steps each, three rounds)
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  ab default ""
  ab single_stream "--no-side-stream"
  ab gate1 "--option gate_branch=1"
  ab ungated "--option gate_branch=-1"
  ab conv_bwd_separate "--option fused_conv_bwd=0"
  ab attn_bwd2 "--option attn_bwd=2"
  ab attn_bwd1 "--option attn_bwd=1"
  ab fused_tail1 "--option fused_tail=1"
  ab qkv_stored "--option fused_attn_fwd=1 --option fused_c16_fwd=1"
  ab conv_tile_kernel "--option conv_rows=0"
done > $O/ab_config1.txt 2>&1
cat $O/ab_config1.txt | sort | awk '{a[$1]+=$3; n[$1]++} END {for (k in a) printf "%-22s %.3f ms\n", k, a[k]/n[k]}' | sort -k2n | tee $O/ab_config1_mean.txt
timeout 120 ./scratch/bench_conv_bwd > $O/bench_conv_bwd.txt 2>&1; tail -12 $O/bench_conv_bwd.txt
timeout 120 ./scratch/bench_conv_rows > $O/bench_conv_rows.txt 2>&1; tail -8 $O/bench_conv_rows.txt
timeout 1500 python tools/bf16_quality.py --json $O/bf16_quality.json > $O/bf16_quality.log 2>&1; tail -6 $O/bf16_quality.log
