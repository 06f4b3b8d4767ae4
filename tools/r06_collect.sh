#!/bin/bash
# copies the files of gpurun_out/r06e/ (written by r06_evidence.sh on the GPU box) into profiles/ under their round-6 names
O=gpurun_out/r06e
cp $O/pytest.log profiles/r06_pytest_gpu.log
cp $O/bench_default.json profiles/r06_bench_default_with_also.json
cp $O/bench_c2.json profiles/r06_bench_config2_bf16_b32_semantic.json
cp $O/bench_c3.json profiles/r06_bench_config3_bf16_b32.json; cp $O/bench_c4.json profiles/r06_bench_config4_x3_256_b8.json
cp $O/bench_c1_fp32.json profiles/r06_bench_config1_fp32_b16.json; cp $O/bench_c1_all_events.json profiles/r06_bench_config1_all_kernel_events.json
cp $O/bench_c1_rccl_one_rank.json profiles/r06_bench_config1_rccl_one_rank.json
for n in c1_default c1_default_no_events c1_single c2_default c3_default c3_single c4_default c4_single c1_fp32_single; do
  m=$(echo $n | sed 's/c\([0-9]\)_default_no_events/config\1_default_no_events/; s/c\([0-9]\)_default$/config\1_default/; s/c\([0-9]\)_single/config\1_single_stream/; s/c1_fp32_single/config1_fp32_single_stream/')
  cp $O/kernel_stats_$n.csv profiles/r06_kernel_stats_$m.csv
done
cp $O/timeline_c1_default.txt profiles/r06_timeline_config1_default.txt; cp $O/timeline_c1_default_no_events.txt profiles/r06_timeline_config1_default_no_events.txt
cp $O/pmc_sq.txt profiles/r06_pmc_sq_config1_single_stream.txt
cp $O/traffic_table_config4.txt profiles/r06_traffic_table_config4.txt; cp $O/traffic_table_config1.txt profiles/r06_traffic_table_config1.txt; cp $O/traffic_table_config3.txt profiles/r06_traffic_table_config3.txt
cp $O/tail_bwd_final.txt profiles/r06_tail_bwd_final.txt
cp $O/ab_options.txt profiles/r06_ab_options.txt; cp $O/ab_libs.txt profiles/r06_ab_libs_round5_vs_round6.txt; cp $O/soak.txt profiles/r06_soak.txt
cp $O/pmc_traffic.json profiles/pmc_traffic.json; cp $O/pmc_traffic_config4.json profiles/pmc_traffic_config4.json
cp $O/pmc_traffic_config3.json profiles/pmc_traffic_config3.json
