"""Prints the figures the round-5 docs quote, from gpurun_out/r05e/ (or profiles/r05_*): one place to read them off after an evidence run."""
import csv, json, os, re, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05e"
def last_json(f):
    return json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
d = last_json("bench_default.json")
r = d["roofline"]
print("default line:", d["value"], d["ms_per_step"], "frac", r["frac"], "avg_us", r.get("avg_launch_us"), "traffic", r.get("traffic"), "cpu", d["cpu_baseline"]["value"])
print("  also:", [(a["workload"], a["value"], a["ms_per_step"], (a.get("dominant_kernel") or {}).get("frac")) for a in d["also"]])
print("  others:", [(o["category"], o["est_ms_per_step"]) for o in r.get("others", [])])
print("  wall:", open(os.path.join(O, "bench_default.time")).read().split()[1] if os.path.exists(os.path.join(O, "bench_default.time")) else None)
for f in ("bench_c2", "bench_c3", "bench_c4", "bench_c1_fp32", "bench_c1_all_events", "bench_c1_rccl_one_rank"):
    x = last_json(f + ".json"); print(f, x["value"], x["ms_per_step"], (x.get("roofline") or {}).get("frac"), x.get("exposed_comm_ms_per_step"))
def ab(path):
    cur, out = None, {}
    for line in open(os.path.join(O, path)):
        line = line.strip()
        m = re.match(r"^([AB]) ([\d.]+) ([\d.]+)$", line)
        if m:
            out.setdefault(cur, {}).setdefault(m.group(1), []).append((float(m.group(2)), float(m.group(3))))
        elif line:
            cur = line
    for k, v in out.items():
        a = v.get("A", []); b = v.get("B", [])
        ma = sum(x[0] for x in a) / max(1, len(a)); mb = sum(x[0] for x in b) / max(1, len(b))
        ta = sum(x[1] for x in a) / max(1, len(a)); tb = sum(x[1] for x in b) / max(1, len(b))
        print(f"  [{k[:90]}] A {ma:.1f} ({ta:.3f} ms)  B {mb:.1f} ({tb:.3f} ms)  B/A {mb / ma - 1:+.1%}")
print("ab_libs:"); ab("ab_libs.txt")
print("ab_options:"); ab("ab_options.txt")
print("fp32:"); ab("fp32_gemm.txt")
for line in open(os.path.join(O, "timeline_c1_default_no_events.txt")).read().splitlines()[:3]: print(line[:160])
rows = list(csv.DictReader(open(os.path.join(O, "kernel_stats_c1_single.csv"))))
print("single-stream kernel sum per step (7 steps):", round(sum(float(x["TotalDurationNs"]) for x in rows) / 7e6, 3), "ms")
want = ("tail_bwd_fused", "fused_fwd_kernelILi256", "conv3x3_c64_bwd", "bwd_res_kernelILi256ELi2ELi8ELb1ELb0ELb1", "bwd_res_kernelILi256ELi2ELi8ELb1ELb0ELb0", "bwd_res_kernelILi64", "bwd_c16",
        "instnorm_bwd_red1", "instnorm_bwd_red2", "instnorm_bwd_apply_kernelIDF16bLb0", "conv3x3_c64_rows_kernel<1", "tail_fwd_stream", "tail_expand", "pack_kernel", "head_conv_fwd")
for x in rows:
    if any(w in x["Name"] for w in want): print("  ", x["Name"][:80], int(x["Calls"]) // 7, round(float(x["AverageNs"]) / 1e3, 1))
tot = 0
for line in open(os.path.join(O, "traffic_table_config1.txt")):
    m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s*$", line)
    if m: tot += int(m.group(2)) / 7 * (float(m.group(4)) + float(m.group(5)))
print("HBM GB per step (config 1):", round(tot / 1e3, 2))
print(open(os.path.join(O, "pytest.log")).read().strip().splitlines()[-1])
print("stamp:", json.load(open(os.path.join(O, "pmc_traffic.json"))).get("source_stamp"))
