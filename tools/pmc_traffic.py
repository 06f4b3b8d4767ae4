#!/usr/bin/env python3
"""Collect per-kernel HBM traffic with rocprofv3 PMC counters (GPU box only) and write
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC
slots'), each with --kernel-trace only.  Units/corrections per MI355X_MICROARCH.md section HBM:
counters are in KiB-like units of 1024 B... FETCH_SIZE under-reports wide coalesced reads by exactly 2x on
gfx950, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for 16-B streaming stores.

    python tools/pmc_traffic.py            # run from the repo root on the GPU box
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc")

CATEGORY_OF = [   # (substring of the kernel name, extra substring, category)
    ("window_attn_fused_fwd_kernel", "ILi256E", "attn_fused_c256"), ("window_attn_fused_fwd_kernel", "ILi64E", "attn_fused_c64"),
    ("tail_fwd_stream_kernel", "", "tail_fwd_fused"), ("tail_fwd_fused_kernel", "", "tail_fwd_fused"), ("window_attn_fused_c16_fwd_kernel", "", "attn_fused_c16"),
    ("window_attn_bwd_res_kernel", "ILi256E", "attn_bwd_c256"), ("window_attn_bwd_res_kernel", "ILi64E", "attn_bwd_c64"),
    ("window_attn_fwd_res_kernel", "ILi256E", "attn_fwd_c256"), ("window_attn_fwd_res_kernel", "ILi64E", "attn_fwd_c64"),
    ("tail_bwd_fused_kernel", "", "final_conv_dgrad"), ("tail_bwd32_kernel", "", "final_conv_dgrad"),
    ("window_attn_bwd_c16_kernel", "", "attn_bwd_c16"), ("window_attn_fwd_c16_kernel", "", "attn_fwd_c16"),
    ("window_attn_fwd_kernel", "Li16E", "attn_fwd_c16"), ("window_attn_fwd_kernel", "Li64E", "attn_fwd_c64"),
    ("window_attn_fwd_kernel", "Li256E", "attn_fwd_c256"), ("window_attn_bwd_kernel", "Li16E", "attn_bwd_c16"),
    ("window_attn_bwd_kernel", "Li64E", "attn_bwd_c64"), ("window_attn_bwd_kernel", "Li256E", "attn_bwd_c256"),
    ("conv3x3_c64_bwd_rows_kernel", "", "conv3x3_bwd"), ("conv3x3_c64_rows_kernel", "", "conv3x3_fwd"),
    ("wgrad_tn_fast_kernel", "", "wgrad_tn_fast"), ("wgrad_tn_big_kernel", "", "wgrad_tn_big"), ("c16_dgrad_prep_kernel", "", "c16_dgrad_prep"),
    ("conv3x3_c64_wgrad_kernel", "", "conv3x3_wgrad"), ("conv3x3_c64_kernel", "", "conv3x3_fwd+dgrad"), ("conv3x3_c64_pipe_kernel", "", "conv3x3_fwd+dgrad"),
    ("final_conv_fwd_kernel", "", "final_conv_fwd"), ("final_conv_dgrad_kernel", "", "final_conv_dgrad"),
    ("final_conv_wgrad_kernel", "", "final_conv_wgrad"), ("wgrad_tn_kernel", "", "wgrad_tn(all)"),
    ("gemm_nt_kernel", "", "gemm_nt(all)"),
]


CONFIG = "1"


def run(counter):
    d = os.path.join(OUT + ("" if CONFIG == "1" else "_config" + CONFIG), counter)
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--config", CONFIG, "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-also", "--no-kernel-events", "--no-side-stream"]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv produced"
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            per.setdefault(name, []).append(float(r["Counter_Value"]))
    return per


def main():
    global CONFIG
    if len(sys.argv) > 2 and sys.argv[1] == "--config":
        CONFIG = sys.argv[2]
    sys.path.insert(0, ROOT)
    from bench import source_stamp
    fetch = run("FETCH_SIZE")
    write = run("WRITE_SIZE")
    traffic, detail = {}, {}
    for name in set(fetch) | set(write):
        f = fetch.get(name, [])
        w = write.get(name, [])
        fb = 2.0 * 1024.0 * (sum(f) / len(f)) if f else 0.0      # gfx950: FETCH_SIZE counts half of a wide coalesced read
        wb = 1024.0 * (sum(w) / len(w)) if w else 0.0
        cat = None
        for sub, extra, c in CATEGORY_OF:
            if sub in name and extra in name:
                cat = c
                break
        detail[name[:120]] = {"launches": max(len(f), len(w)), "fetch_bytes": fb, "write_bytes": wb, "category": cat}
        if cat and cat not in traffic:
            traffic[cat] = fb + wb
        elif cat:
            traffic[cat] = max(traffic[cat], fb + wb) if "(all)" in cat else (traffic[cat] + fb + wb) / 2
    out = {"note": "bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, averaged over launches (bench.py --config " + CONFIG + ", single stream)",
           "source_stamp": source_stamp(), "workload": "config" + CONFIG, "dtype": "bf16",
           "traffic_bytes_per_launch": traffic, "kernels": detail}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    name = "pmc_traffic.json" if CONFIG == "1" else f"pmc_traffic_config{CONFIG}.json"
    with open(os.path.join(ROOT, "profiles", name), "w") as fo:
        json.dump(out, fo, indent=1, sort_keys=True)
    for k, v in sorted(traffic.items()):
        print(f"{k:22s} {v/1e6:10.2f} MB / launch")


if __name__ == "__main__":
    main()
