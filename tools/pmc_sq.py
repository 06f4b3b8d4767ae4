#!/usr/bin/env python3
"""Per-kernel SQ counters (wave cycles, waits, LDS bank conflicts) via rocprofv3 --pmc (GPU box only).
    python tools/pmc_sq.py [extra bench args]      -> gpurun_out/pmc_sq.txt"""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_sq")
COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
            "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"]
# second set (M2T_PMC_SET=2): where the waiting goes -- LDS waits, outstanding VMEM / LDS instructions, LDS FIFO pressure
if os.environ.get("M2T_PMC_SET") == "2":
    COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_INST_LEVEL_VMEM",
                "SQ_INST_LEVEL_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL"]
os.makedirs(OUT, exist_ok=True)
cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", OUT, "--",
       sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also",
       "--no-kernel-events", "--no-side-stream"] + sys.argv[1:]
subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(OUT, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == COUNTERS[0]:
            cnt[k] += 1
lines = []
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    if os.environ.get("M2T_PMC_SET") == "2":
        lines.append(f"{k:70s} n={cnt[k]:5d} " + " ".join(f"{n[3:].lower()}={c[n]/wc:6.3f}" for n in COUNTERS[1:]))
        continue
    lines.append(f"{k:70s} n={cnt[k]:5d} wavecyc/launch={wc/cnt[k]:12.0f} wait_any={c['SQ_WAIT_ANY']/wc:5.2f} wait_inst={c['SQ_WAIT_INST_ANY']/wc:5.2f} "
                 f"active={c['SQ_ACTIVE_INST_ANY']/wc:5.2f} lds_act={c['SQ_ACTIVE_INST_LDS']/wc:5.2f} bankconf/ldsidx={c['SQ_LDS_BANK_CONFLICT']/(c['SQ_LDS_IDX_ACTIVE'] or 1):5.2f} "
                 f"mfma_busy/launch={c['SQ_VALU_MFMA_BUSY_CYCLES']/cnt[k]:10.0f}")
open(os.path.join(ROOT, "gpurun_out", "pmc_sq.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
