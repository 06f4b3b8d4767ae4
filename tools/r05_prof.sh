# round 5: rocprofv3 kernel summaries: fp32 mode (config 1), bf16 config 1 / 3 single stream (current build)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p
mkdir -p $O
prof() {  # name, bench args...
  n=$1; shift
  rm -rf $O/prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 bench.py "$@" --no-cpu-baseline --no-also > $O/prof_$n.log 2>&1
  find $O/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$n.csv
  find $O/prof_$n -type f -delete 2>/dev/null
  head -4 $O/kernel_stats_$n.csv | cut -c1-150
}
prof c1_fp32 --config 1 --dtype fp32 --steps 5 --warmup 2 --no-kernel-events
prof c1_fp32_single --config 1 --dtype fp32 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c1_single --config 1 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c3_single --config 3 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
prof c4_single --config 4 --steps 5 --warmup 2 --no-kernel-events --no-side-stream
