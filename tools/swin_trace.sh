# kernel trace of one configs[2] step: the Swin-T forward kernels in launch order with durations
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/trace2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace2 -- python3 bench.py --config 2 --no-cpu-baseline --no-kernel-events --steps 2 --warmup 1 > gpurun_out/trace2.log 2>&1
F=$(find gpurun_out/trace2 -name "*kernel_trace.csv" | head -1)
python - "$F" > gpurun_out/swin_trace.txt <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if 'swin_patchify' in r['Kernel_Name']]
heads = [i for i, r in enumerate(rows) if 'swin_head' in r['Kernel_Name']]
a = idx[-1]; b = [i for i in heads if i > a][0]
t0 = rows[a - 6]['s']
for r in rows[a - 6:b + 4]:
    n = re.sub(r'^void ', '', r['Kernel_Name']); n = re.sub(r'^_Z\d+', '', n)
    print("%9.1f %8.1fus st=%s %-60s grid=%s wg=%s" % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r['Stream_Id'], n[:60], r['Grid_Size_X'], r['Workgroup_Size_X']))
P
find gpurun_out/trace2 -type f -delete
head -3 gpurun_out/swin_trace.txt
