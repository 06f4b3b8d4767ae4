#!/usr/bin/env python3
"""Timeline analysis of one training step from a rocprofv3 kernel trace (csv):
   per-stream busy time, main-stream gaps, and the list of kernels of the last step in start order."""
import csv, sys, re
head = open(sys.argv[1]).read().strip().replace('"', '').split(',')
rows = [dict(zip(head, r)) for r in csv.reader(open(sys.argv[2])) if r and r[0] != 'Kind']
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# last step = from the last pack_kernel to the last adam_kernel
packs = [i for i, r in enumerate(rows) if 'pack_kernel' in r['Kernel_Name']]
adams = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
a = packs[-1]; b = [i for i in adams if i > a][0]
step = rows[a:b + 1]
t0 = step[0]['s']; t1 = max(r['e'] for r in step)
print("step wall (first start .. last end): %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'^_Z\d+', '', n)
    return n[:44]
streams = {}
for r in step: streams.setdefault(r['Stream_Id'] + '/' + r['Queue_Id'], []).append(r)
for k, v in streams.items():
    busy = sum(r['e'] - r['s'] for r in v)
    gaps = sum(max(0, v[i + 1]['s'] - v[i]['e']) for i in range(len(v) - 1))
    print("stream/queue %s: %d kernels busy %.3f ms, gaps %.3f ms, span %.3f..%.3f" % (k, len(v), busy / 1e6, gaps / 1e6, (v[0]['s'] - t0) / 1e6, (v[-1]['e'] - t0) / 1e6))
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[3]), float(sys.argv[4])
    for r in step:
        ts = (r['s'] - t0) / 1e6
        if lo <= ts <= hi:
            print("%8.3f %8.3f %6.1fus  st=%s  %s grid=%s" % (ts, (r['e'] - t0) / 1e6, (r['e'] - r['s']) / 1e3, r['Stream_Id'], short(r['Kernel_Name']), r['Grid_Size_X']))
