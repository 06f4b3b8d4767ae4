# round 4 evidence run (ONE gpurun call, one MI355X box, the FINAL build): whole -m gpu suite, HBM traffic counters of configs 1 and 4
# (build-stamped), the bench lines of configs 1-4 in bf16 and the fp32 line of config 1, the all-events line, rocprofv3 kernel
# summaries of configs 1 (two-stream + timeline, single stream) and 2-4, SQ counters, the one-rank RCCL line, the tail micro-benchmark
# and the VALU rate probe.  Everything lands in gpurun_out/r04e/; the files to keep are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e
mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -30 > $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -20 $O/pmc_traffic.log
timeout 900 python tools/pmc_traffic.py --config 4 > $O/pmc_traffic_config4.log 2>&1; cp profiles/pmc_traffic_config4.json $O/pmc_traffic_config4.json; tail -8 $O/pmc_traffic_config4.log
timeout 900 python tools/pmc_traffic.py --config 3 > $O/pmc_traffic_config3.log 2>&1; cp profiles/pmc_traffic_config3.json $O/pmc_traffic_config3.json; tail -4 $O/pmc_traffic_config3.log
for c in 1 2 3 4; do
  extra="--no-cpu-baseline"; [ $c = 1 ] && extra=""
  timeout 600 python bench.py --config $c $extra > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-260 $O/bench_c$c.json
done
timeout 600 python bench.py --config 1 --dtype fp32 --no-cpu-baseline > $O/bench_c1_fp32.json 2> $O/bench_c1_fp32.err; cut -c1-260 $O/bench_c1_fp32.json
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
prof c1_default_no_events --config 1 --no-kernel-events
python tools/timeline.py <(head -1 $O/kernel_trace_c1_default_no_events.csv) $O/kernel_trace_c1_default_no_events.csv 0 10 > $O/timeline_c1_default_no_events.txt 2>&1
prof c1_single --config 1 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c2_default --config 2 --steps 5 --warmup 2 --no-kernel-events
prof c3_default --config 3 --steps 5 --warmup 2 --no-kernel-events
prof c3_single --config 3 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c4_default --config 4 --steps 5 --warmup 2 --no-kernel-events
prof c4_single --config 4 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
rm -f $O/kernel_trace_*.csv
python tools/traffic_table.py $O/pmc_traffic_config4.json $O/kernel_stats_c4_single.csv > $O/traffic_table_config4.txt 2>&1; head -14 $O/traffic_table_config4.txt | cut -c1-170
python tools/traffic_table.py $O/pmc_traffic.json $O/kernel_stats_c1_single.csv > $O/traffic_table_config1.txt 2>&1
timeout 900 python tools/pmc_sq.py > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq.txt; head -10 $O/pmc_sq.txt | cut -c1-260
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-comm-path --no-cpu-baseline > $O/bench_c1_rccl_one_rank.json 2> $O/bench_rccl.err; cut -c1-200 $O/bench_c1_rccl_one_rank.json
timeout 200 ./scratch/bench_tail_new > $O/bench_tail.txt 2>&1; grep "tail_\|stream" $O/bench_tail.txt | tail -12
# per-phase stamps of the C = 256 / C = 64 attention kernels (stand-alone harnesses): backward, forward without / with branch_prep inside
( for c in 256 64; do ./scratch/bench_res_st $c 16; ./scratch/bench_res $c 16 | head -1; done; ./scratch/bench_res 256 32 | head -1
  echo "--- with branch_prep_bwd of the next branch inside"; ./scratch/bench_res_st 256 16 1; ./scratch/bench_res 256 16 1 | head -1; ./scratch/bench_res 256 32 1 | head -1 ) > $O/attn_bwd_stamps.txt 2>&1
( for c in 256 64; do for p in 0 1; do ./scratch/bench_fused_st $c 16 0 $p; ./scratch/bench_fused $c 16 0 $p; done; done; ./scratch/bench_fused 256 32 0 1 ) > $O/attn_fwd_stamps.txt 2>&1
# same-box A/B of the round's two schedule / fusion options (three alternations each, configs 1 and 3)
( echo "A = fork_on_kernel 0, B = default"; bash tools/ab_opts.sh "--option fork_on_kernel=0" "" 3; bash tools/ab_opts.sh "--config 3 --option fork_on_kernel=0" "--config 3" 2
  echo "A = fused_prep_fwd 0, B = default"; bash tools/ab_opts.sh "--option fused_prep_fwd=0" "" 3; bash tools/ab_opts.sh "--config 3 --option fused_prep_fwd=0" "--config 3" 2
  echo "A = fused_prep_bwd 0, B = default"; bash tools/ab_opts.sh "--option fused_prep_bwd=0" "" 3; bash tools/ab_opts.sh "--config 3 --option fused_prep_bwd=0" "--config 3" 2
  echo "A = fused_tail 4 (x4 row-streaming backward), B = default"; bash tools/ab_opts.sh "--option fused_tail=4" "" 2
  echo "config 4: A = fused_tail 0 (plain x3 tail kernels), B = default"; bash tools/ab_opts.sh "--config 4 --option fused_tail=0" "--config 4" 2 ) > $O/ab_options.txt 2>&1
timeout 100 ./scratch/bench_valu_rate > $O/valu_rate.txt 2>&1
timeout 600 python tools/soak.py 8 12 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
