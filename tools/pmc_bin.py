#!/usr/bin/env python3
"""SQ counters of a stand-alone benchmark binary (GPU box only): python tools/pmc_bin.py <binary> [args]  ->  per-kernel table.
Two passes (8 SQ counters each): waits / activity, then instruction counts."""
import csv, glob, os, subprocess, sys, collections, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"],
        ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]]
for si, counters in enumerate(SETS):
    out = f"/tmp/pmc_bin_{si}"
    shutil.rmtree(out, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--"] + sys.argv[1:]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == counters[0]:
                cnt[k] += 1
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:6]:
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        print(f"{k:60s} n={cnt[k]:4d} wavecyc/launch={wc/cnt[k]:.4g} " + " ".join(f"{n[3:].lower()}={c[n]/wc:.4f}" for n in counters[1:]))
