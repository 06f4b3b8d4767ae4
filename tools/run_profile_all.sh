# refresh everything under profiles/ from one box: tests, PMC traffic (two passes), the bench line, rocprofv3 stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 900 python tools/pmc_traffic.py 2>&1 | tail -3
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json
cut -c1-900 gpurun_out/bench_default.json
rm -rf gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --no-cpu-baseline --no-also > gpurun_out/prof_default.log 2>&1
find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_default.csv
rm -rf gpurun_out/prof_single
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-side-stream > gpurun_out/prof_single.log 2>&1
find gpurun_out/prof_single -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_single.csv
find gpurun_out/prof_default gpurun_out/prof_single gpurun_out/pmc -type f -delete 2>/dev/null
head -8 gpurun_out/kernel_stats_default.csv | cut -c1-150
