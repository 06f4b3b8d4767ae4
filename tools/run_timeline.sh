# kernel trace of the default bench + tools/timeline.py summary (and the kernels of a time window, ms: $1 $2)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --no-cpu-baseline --no-kernel-events --steps 4 --warmup 2 > gpurun_out/trace.log 2>&1
F=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
head -1 $F > gpurun_out/trace_head.txt
python tools/timeline.py gpurun_out/trace_head.txt $F $1 $2 > gpurun_out/timeline.txt
head -5 gpurun_out/timeline.txt
find gpurun_out/trace -type f -delete
