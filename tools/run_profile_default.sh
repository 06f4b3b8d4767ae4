cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json
cat gpurun_out/bench_default.json | cut -c1-1200
rm -rf gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --no-cpu-baseline --no-also > gpurun_out/prof_default.log 2>&1
tail -1 gpurun_out/prof_default.log | cut -c1-300
find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_default.csv
head -12 gpurun_out/kernel_stats_default.csv | cut -c1-160
find gpurun_out/prof_default -type f ! -name "*kernel_stats.csv" -delete
