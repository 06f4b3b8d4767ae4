# round 5 evidence run (ONE gpurun call, one MI355X box, the FINAL build): whole -m gpu suite, HBM traffic counters of configs 1 / 3 / 4
# (build-stamped), the DEFAULT bench line (with its `also` list), full bench lines of configs 2-4 and fp32, the all-events line,
# rocprofv3 kernel summaries (config 1 with / without events + timelines, single stream; configs 2-4), traffic tables, SQ counters,
# the one-rank RCCL line, attention stamps (backward, 8-wave forward, the two-windows-per-CU forward with its knock-out / stagger
# series and SQ counters), same-box A/B of the round's options and of the round-4 library against this one, soak, smoke.
# Everything lands in gpurun_out/r05e/; tools/r05_collect.sh copies the files to keep into profiles/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e
mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -30 > $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -3 $O/pmc_traffic.log
timeout 900 python tools/pmc_traffic.py --config 4 > $O/pmc_traffic_config4.log 2>&1; cp profiles/pmc_traffic_config4.json $O/pmc_traffic_config4.json
timeout 900 python tools/pmc_traffic.py --config 3 > $O/pmc_traffic_config3.log 2>&1; cp profiles/pmc_traffic_config3.json $O/pmc_traffic_config3.json
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time; cut -c1-300 $O/bench_default.json
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
timeout 900 python tools/pmc_sq.py --config 3 > $O/pmc_sq_c3.log 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq_config3.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-comm-path --no-cpu-baseline > $O/bench_c1_rccl_one_rank.json 2> $O/bench_rccl.err; cut -c1-330 $O/bench_c1_rccl_one_rank.json
# per-phase stamps: backward (8 waves, resident), forward 8-wave kernel, forward two-windows-per-CU kernels
( for c in 256 64; do ./scratch/bench_res_st $c 16; ./scratch/bench_res $c 16 | head -1; done; ./scratch/bench_res 256 32 | head -1
  echo "--- with branch_prep_bwd of the next branch inside"; ./scratch/bench_res_st 256 16 1; ./scratch/bench_res 256 16 1 | head -1; ./scratch/bench_res 256 32 1 | head -1 ) > $O/attn_bwd_stamps.txt 2>&1
( for c in 256 64; do for p in 0 1; do ./scratch/bench_fused_st $c 16 0 $p; ./scratch/bench_fused $c 16 0 $p; done; done; ./scratch/bench_fused 256 32 0 1; ./scratch/bench_fused_st 256 32 0 1 ) > $O/attn_fwd_stamps.txt 2>&1
( echo "== k_attn_fwd2.hip: event-bracketed launches (windows per workgroup, stagger)"; for v in 1 2; do for b in 16 32 64; do ./scratch/bench_fwd2 $b $v; done; done
  echo "== the 8-wave kernel (k_attn_fused.hip, branch_prep inside) on the same box"; ./scratch/bench_fused 256 32 0 1; ./scratch/bench_fused 256 16 0 1
  echo "== stamps, one window per 4-wave workgroup (two workgroups per CU), batch 32"; ./scratch/bench_fwd2_st 32 1
  echo "== stamps, two windows per 8-wave workgroup, batch 32"; ./scratch/bench_fwd2_st 32 2
  echo "== knock-out: no weight-fragment reloads (wrong results, timing only)"; ./scratch/bench_fwd2_dbg_NO_WEIGHT_RELOAD 32 1 | head -10
  echo "== the second workgroup of each CU started late by n x 8 k cycles (s_sleep)"; for sg in 0 1 2 3 4 6; do ./scratch/bench_fwd2 32 1 $sg; done; ./scratch/bench_fwd2_st 32 1 3 | head -10
  echo "== SQ counters (tools/pmc_bin.py): fwd2 (one window per workgroup), then the 8-wave kernel"; python tools/pmc_bin.py $GRAFT_REPO_ROOT/scratch/bench_fwd2 32 1 2>&1 | grep -v copyBuffer; python tools/pmc_bin.py $GRAFT_REPO_ROOT/scratch/bench_fused 256 32 0 1 2>&1 | grep -v copyBuffer ) > $O/attn_fwd2_stamps.txt 2>&1
# same-box A/B of the round's options
( echo "A = fused_l1 0, B = default"; bash tools/ab_opts.sh "--option fused_l1=0" "" 3; bash tools/ab_opts.sh "--config 3 --option fused_l1=0" "--config 3" 2
  echo "A = default, B = fused_norm_red 1"; bash tools/ab_opts.sh "" "--option fused_norm_red=1" 2; bash tools/ab_opts.sh "--config 3" "--config 3 --option fused_norm_red=1" 2
  echo "config 3: A = default (8-wave forward), B = fused_attn_fwd2 1"; bash tools/ab_opts.sh "--config 3" "--config 3 --option fused_attn_fwd2=1" 3
  echo "config 3: A = default, B = fused_attn_fwd2 2"; bash tools/ab_opts.sh "--config 3" "--config 3 --option fused_attn_fwd2=2" 2
  echo "config 4: A = default, B = fused_attn_fwd2 1"; bash tools/ab_opts.sh "--config 4" "--config 4 --option fused_attn_fwd2=1" 2
  echo "config 2: A = main priority 0 (default), B = -1"; bash tools/ab_opts.sh "--config 2 --main-priority 0" "--config 2 --main-priority -1" 2
  echo "config 2: A = default (encoder beside the backward), B = --no-overlap-semantic"; bash tools/ab_opts.sh "--config 2" "--config 2 --no-overlap-semantic" 2 ) > $O/ab_options.txt 2>&1
( echo "A = round-4 final tree (f313e5b: scratch/r4tree, its own bench.py + library), B = this tree; config 1"; bash tools/ab_rounds.sh "--config 1 --steps 30" 3
  echo "config 3"; bash tools/ab_rounds.sh "--config 3 --steps 20" 3
  echo "config 4"; bash tools/ab_rounds.sh "--config 4 --steps 20" 3
  echo "config 2"; bash tools/ab_rounds.sh "--config 2 --steps 20" 2 ) > $O/ab_libs.txt 2>&1
( echo "fp32 parity mode, config 1: A = fp32_fast 0 (16x16x4 GEMMs of rounds 1-4), B = default (32x32x2)"; bash tools/ab_opts.sh "--dtype fp32 --steps 8 --warmup 2 --option fp32_fast=0" "--dtype fp32 --steps 8 --warmup 2" 2
  echo "== scratch/bench_gemm_f32 64 (the step's fp32 GEMM shapes at batch 16, 128x128 LR)"; ./scratch/bench_gemm_f32 64 ) > $O/fp32_gemm.txt 2>&1
cat $O/ab_libs.txt
timeout 600 python tools/soak.py 8 12 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
