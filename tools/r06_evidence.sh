# round 6 evidence run (ONE gpurun call, one MI355X box, the FINAL build): whole -m gpu suite, HBM traffic counters of configs 1 / 3 / 4
# (build-stamped), the DEFAULT bench line (with `also` and `also_summary`), full bench lines of configs 2-4 and fp32, the all-events line,
# rocprofv3 kernel summaries (config 1 with / without events + timelines, single stream; configs 2-4; fp32), traffic tables, SQ counters,
# the one-rank RCCL line, the tail-backward micro-benchmark with stamps, same-box A/B of the round's option and of the round-5 library
# (scratch/lib_base_r6.so) against this one, soak, smoke.  Everything lands in gpurun_out/r06e/; tools/r06_collect.sh copies into profiles/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e
mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -30 > $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -3 $O/pmc_traffic.log
timeout 900 python tools/pmc_traffic.py --config 4 > $O/pmc_traffic_config4.log 2>&1; cp profiles/pmc_traffic_config4.json $O/pmc_traffic_config4.json
timeout 900 python tools/pmc_traffic.py --config 3 > $O/pmc_traffic_config3.log 2>&1; cp profiles/pmc_traffic_config3.json $O/pmc_traffic_config3.json
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time; cut -c1-300 $O/bench_default.json; tail -c 900 $O/bench_default.json
for c in 2 3 4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-200 $O/bench_c$c.json
done
timeout 600 python bench.py --config 1 --dtype fp32 --no-cpu-baseline > $O/bench_c1_fp32.json 2> $O/bench_c1_fp32.err; cut -c1-200 $O/bench_c1_fp32.json
timeout 300 python bench.py --config 1 --no-cpu-baseline --all-kernel-events > $O/bench_c1_all_events.json 2>/dev/null
prof() {  # name, bench args...
  n=$1; shift
  rm -rf $O/prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 bench.py "$@" --no-cpu-baseline --no-also > $O/prof_$n.log 2>&1
  find $O/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$n.csv
  find $O/prof_$n -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_$n.csv
  find $O/prof_$n -type f -delete 2>/dev/null
  head -3 $O/kernel_stats_$n.csv | cut -c1-150
}
prof c1_default --config 1
python tools/timeline.py <(head -1 $O/kernel_trace_c1_default.csv) $O/kernel_trace_c1_default.csv 0 10 > $O/timeline_c1_default.txt 2>&1
prof c1_default_no_events --config 1 --no-kernel-events
python tools/timeline.py <(head -1 $O/kernel_trace_c1_default_no_events.csv) $O/kernel_trace_c1_default_no_events.csv 0 10 > $O/timeline_c1_default_no_events.txt 2>&1
prof c1_single --config 1 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c2_default --config 2 --steps 5 --warmup 2 --no-kernel-events
prof c3_default --config 3 --steps 5 --warmup 2 --no-kernel-events
prof c3_single --config 3 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c4_default --config 4 --steps 5 --warmup 2 --no-kernel-events
prof c4_single --config 4 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c1_fp32_single --config 1 --dtype fp32 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
rm -f $O/kernel_trace_*.csv
python tools/traffic_table.py $O/pmc_traffic_config4.json $O/kernel_stats_c4_single.csv > $O/traffic_table_config4.txt 2>&1; head -12 $O/traffic_table_config4.txt | cut -c1-170
python tools/traffic_table.py $O/pmc_traffic.json $O/kernel_stats_c1_single.csv > $O/traffic_table_config1.txt 2>&1
python tools/traffic_table.py $O/pmc_traffic_config3.json $O/kernel_stats_c3_single.csv > $O/traffic_table_config3.txt 2>&1
timeout 900 python tools/pmc_sq.py > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq.txt; head -6 $O/pmc_sq.txt | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-comm-path --no-cpu-baseline > $O/bench_c1_rccl_one_rank.json 2> $O/bench_rccl.err; cut -c1-330 $O/bench_c1_rccl_one_rank.json
( cd scratch && ./bench_tail_bwd16; for w in 0 5; do ./bench_tail_bwd16_st$w | grep -A4 "B=16 LR 128x128 (L1" | grep "variant\|kernel"; done ) > $O/tail_bwd_final.txt 2>&1; cat $O/tail_bwd_final.txt | grep kernel
( echo "A = tail_bwd_mfma32 0 (the 16x16x32 tile kernel of rounds 2-5), B = default"; bash tools/ab_opts.sh "--option tail_bwd_mfma32=0" "" 3; bash tools/ab_opts.sh "--config 3 --option tail_bwd_mfma32=0" "--config 3" 2 ) > $O/ab_options.txt 2>&1
cp scratch/lib_base_r6.so scratch/libA.so; cp m2trans_amd/libm2t.so scratch/libB.so
( echo "A = the round-5 kernels (scratch/lib_base_r6.so: round-5 tree + the deferred-seed ordering fix), B = this build; config 1"; bash tools/ab_libs.sh "--steps 30" 3
  echo "config 3"; bash tools/ab_libs.sh "--config 3 --steps 20" 3
  echo "config 4"; bash tools/ab_libs.sh "--config 4 --steps 20" 3
  echo "config 2"; bash tools/ab_libs.sh "--config 2 --steps 20" 2 ) > $O/ab_libs.txt 2>&1
cat $O/ab_libs.txt
timeout 600 python tools/soak.py 8 12 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
