#!/usr/bin/env python3
"""Join a pmc_traffic file (bytes per launch per kernel) with a rocprofv3 kernel_stats CSV (average duration per kernel) into the
table BASELINE configs[4] asks for: per kernel launches, us, MB per launch, achieved GB/s and fraction of the 8 TB/s HBM peak.
    python tools/traffic_table.py profiles/pmc_traffic_config4.json profiles/r04_kernel_stats_config4_single_stream.csv [filter]"""
import csv, json, sys
blob = json.load(open(sys.argv[1]))
flt = sys.argv[3] if len(sys.argv) > 3 else ""
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Name"][:120]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
rows = []
for name, d in blob["kernels"].items():
    if flt and flt not in name:
        continue
    best = None
    for n2, v in dur.items():
        if n2[:100] == name[:100] or n2.startswith(name[:80]) or name.startswith(n2[:80]):
            best = v
            break
    if not best:
        continue
    by = d["fetch_bytes"] + d["write_bytes"]
    rows.append((best[0] * best[1], name[:86], best[1], best[0], d["fetch_bytes"] / 1e6, d["write_bytes"] / 1e6, by / best[0] / 1e3))
rows.sort(reverse=True)
print(f"# {blob.get('workload')} stamp {blob.get('source_stamp')}: HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB units) / average launch duration of the single-stream run")
print(f"{'kernel':86s} {'calls':>6s} {'us':>8s} {'read MB':>9s} {'write MB':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
for _, n, c, us, fm, wm, gbs in rows[:40]:
    print(f"{n:86s} {c:6d} {us:8.1f} {fm:9.2f} {wm:9.2f} {gbs:8.0f} {gbs / 8000:9.3f}")
