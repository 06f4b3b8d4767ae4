cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for opt in -2 -3 -4; do
rm -rf gpurun_out/r03h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h -- python3 bench.py --steps 5 --warmup 2 --no-kernel-events --no-side-stream --no-cpu-baseline --option wgrad_big_tiles=$opt > /dev/null 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('gpurun_out/r03h/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wgrad_tn_dma' in r['Name']: print('mode $opt', r['Name'][:44], r['Calls'], '%.1f us avg  min %.1f max %.1f'%(float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
rm -rf gpurun_out/r03h
