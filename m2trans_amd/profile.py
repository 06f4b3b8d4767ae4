"""Kernel-category timing (HIP events inside libm2t.so) and the roofline arithmetic used by
bench.py.  Category ids mirror `enum m2t_prof_cat` in csrc/m2t_kernels.h.

Algorithmic work per launch (per-unit figure x units per launch):
  window attention fwd : 2 products x 2*64*100*C FLOP per window           = 25 600 C FLOP/window
                         bytes: read q|k|v (3C) + write out (C) per pixel   = 4 C es B/pixel
  fused qkv + attention: the projection's 2*C*3C FLOP per pixel (each pixel once: the halo rows a window projects
                         again are not counted) + 25 600 C FLOP per window
                         bytes: read d (C) + residual (C), write q|k|v (3C) + out (C) per pixel = 6 C es B/pixel
  window attention bwd : 5 products (S, dP, dq, dK, dV)                     = 64 000 C FLOP/window
                         bytes: read qkv (3C) + gO (C), write gqkv (3C)     = 7 C es B/pixel
  conv3x3 64->64       : 2*64*576 FLOP/pixel; bytes: in + out (+ residual)  = 128..192 es B/pixel
  1x1 GEMMs            : 2*K*N FLOP/row; bytes (K + N) es B/row
  tail conv 64->3      : 2*64*27 FLOP/HR pixel; bytes 64 es (+ 3*4 out) B/HR pixel
Peaks (MI355X_MICROARCH.md): HBM 8.0 TB/s; dense MFMA 2.5 PFLOP/s bf16, 157.3 TFLOP/s fp32.
"""
from __future__ import annotations

import ctypes as C
import json
import os

from . import _lib

CATS = ["attn_fwd_c16", "attn_fwd_c64", "attn_fwd_c256", "attn_bwd_c16", "attn_bwd_c64", "attn_bwd_c256",
        "conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "gemm_qkv", "gemm_qkv_dgrad", "wgrad_qkv",
        "tail_gemm", "tail_wgrad", "final_conv_fwd", "final_conv_dgrad", "final_conv_wgrad",
        "attn_fused_c64", "attn_fused_c256", "tail_fwd_fused", "attn_fused_c16", "conv3x3_bwd"]
KERNEL_OF = {
    "attn_fwd_c16": "window_attn_fwd_kernel<C=16>", "attn_fwd_c64": "window_attn_fwd_kernel<C=64>",
    "attn_fwd_c256": "window_attn_fwd_kernel<C=256>", "attn_bwd_c16": "window_attn_bwd_kernel<C=16>",
    "attn_bwd_c64": "window_attn_bwd_kernel<C=64>", "attn_bwd_c256": "window_attn_bwd_kernel<C=256>",
    "conv3x3_fwd": "conv3x3_c64_kernel (forward)", "conv3x3_dgrad": "conv3x3_c64_kernel (data gradient)",
    "conv3x3_wgrad": "conv3x3_c64_wgrad_kernel", "gemm_qkv": "gemm_nt_kernel (qkv projections)",
    "gemm_qkv_dgrad": "gemm_nt_kernel (qkv data gradients)", "wgrad_qkv": "wgrad_tn_kernel (qkv weight gradients)",
    "tail_gemm": "gemm_nt_kernel (tail 1x1 + pixel shuffle, fwd+dgrad)", "tail_wgrad": "wgrad_tn_kernel (tail)",
    "final_conv_fwd": "final_conv_fwd_kernel", "final_conv_dgrad": "final_conv_dgrad_kernel",
    "final_conv_wgrad": "final_conv_wgrad_kernel",
    "attn_fused_c64": "window_attn_fused_fwd_kernel<C=64,L=1> (qkv projection + window attention + IWT/residual)",
    "attn_fused_c256": "window_attn_fused_fwd_kernel<C=256,L=2> (qkv projection + window attention + IWT^2/residual)",
    "attn_fused_c16": "window_attn_fused_c16_fwd_kernel (InstanceNorm apply + qkv projection + window attention + residual, wave per window)",
    "tail_fwd_fused": "tail_fwd_stream_kernel (1x1 expansion + PixelShuffle + GELU + tail conv, row-streaming)",
    "conv3x3_bwd": "conv3x3_c64_bwd_rows_kernel (64->64 3x3 conv: data gradient + weight / bias gradient in one row-streaming pass)",
}
KERNEL_OF_BF16 = {   # bf16 mode launches the specialised kernels for these categories
    "attn_fwd_c16": "window_attn_fwd_c16_kernel (wave per window)",
    "attn_bwd_c16": "window_attn_bwd_c16_kernel (wave per window)",
    "attn_bwd_c64": "window_attn_bwd_res_kernel<C=64,L=1> (window resident in LDS)",
    "attn_bwd_c256": "window_attn_bwd_res_kernel<C=256,L=2> (window resident in LDS)",
    "gemm_qkv": "gemm_nt_kernel / gemm_nt_wide_kernel (qkv projections)",
    "gemm_qkv_dgrad": "gemm_nt_kernel / gemm_nt_wide_kernel (qkv data gradients)",
    "tail_gemm": "tail_expand_kernel (fwd) + gemm_nt_kernel (data gradients)",
    "final_conv_dgrad": "tail_bwd32_kernel (x4, round 6; tail_bwd_fused_kernel with tail_bwd_mfma32 = 0) / tail_bwd_stream_kernel (x2, x3): tail conv dgrad+wgrad, GELU', expansion dgrad+wgrad, recomputing",
}
MERGED = {"conv3x3_fwd+dgrad": ("conv3x3_fwd", "conv3x3_dgrad")}
MERGED_KERNEL = {"conv3x3_fwd+dgrad": "conv3x3_c64_rows_kernel (64->64 3x3 conv, row-streaming LDS-DMA kernel: forward and data gradient are the same kernel)"}
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TF = {"bf16": 2500.0, "fp32": 157.3}
ALL_MASK = (1 << len(CATS)) - 1


def enable(mask: int = ALL_MASK, sample_every: int = 1):
    """sample_every = n > 1: the dispatch-timed (single-kernel) categories take their events on every n-th launch only."""
    _lib.check(_lib.load().m2t_profile_sample_every(int(sample_every)), "m2t_profile_sample_every")
    _lib.check(_lib.load().m2t_profile_enable(C.c_ulonglong(mask)), "m2t_profile_enable")


def read_all():
    lib = _lib.load()
    out = {}
    for i, name in enumerate(CATS):
        ms, n = C.c_double(0.0), C.c_longlong(0)
        _lib.check(lib.m2t_profile_read(i, C.byref(ms), C.byref(n)), "m2t_profile_read")
        out[name] = (ms.value, n.value)
    return out


def algorithmic_work(B: int, lr: int, scale: int, dtype: str, n_blocks: int = 8, fused_attn_fwd: bool | None = None,
                     fused_tail_fwd: bool = False, fused_qkv_dgrad: bool | None = None, fused_c16_fwd: bool | None = None,
                     c16_recompute: bool | None = None, c64_recompute: bool | None = None, fused_conv_bwd: bool | None = None,
                     c16_prep: bool | None = None, fused_prep_fwd: bool | None = None, fused_prep_bwd: bool | None = None):
    """Per STEP totals {category: (flops, bytes, launches)} for the x4-style model at padded LR size lr.
    fused_attn_fwd (default: bf16 mode): the C = 64 / 256 branches run qkv projection + attention as one kernel, so
    the forward `gemm_qkv` / `attn_fwd_*` categories are then empty (the C = 16 branch runs InstanceNorm apply +
    projection + attention as `attn_fused_c16`).  fused_tail_fwd: option "fused_tail" = 2 of the plan (the bf16 default)."""
    if fused_attn_fwd is None:
        fused_attn_fwd = dtype == "bf16"
    if fused_c16_fwd is None:             # plan option "fused_c16_fwd": the C = 16 branch has its own fused kernel
        fused_c16_fwd = dtype == "bf16"
    if fused_qkv_dgrad is None:          # plan option "fused_qkv_dgrad" (default on in bf16 mode): C = 64 / 256 branches
        fused_qkv_dgrad = dtype == "bf16"
    if c16_recompute is None:             # plan option "fused_c16_fwd" = 2 (default in bf16 mode): qkv1 is not stored, the backward recomputes it
        c16_recompute = dtype == "bf16"
    if c64_recompute is None:             # plan option "fused_attn_fwd" = 2 (default in bf16 mode): qkv2 is not stored
        c64_recompute = dtype == "bf16"
    if fused_prep_fwd is None:            # plan option "fused_prep_fwd" (bf16 default): branch_prep inside the fused forward attention kernels
        fused_prep_fwd = dtype == "bf16" and bool(fused_attn_fwd)
    if fused_prep_bwd is None:            # plan option "fused_prep_bwd" (bf16 default): branch 4's branch_prep_bwd inside branch 3's attention backward
        fused_prep_bwd = dtype == "bf16" and bool(fused_qkv_dgrad)
    if c16_prep is None:                  # plan option "attn_bwd" = 3 (default in bf16 mode): no data-gradient GEMM for the C = 16 branch either
        c16_prep = dtype == "bf16" and bool(fused_qkv_dgrad)
    if fused_conv_bwd is None:            # plan option "fused_conv_bwd" (default on in bf16 mode)
        fused_conv_bwd = dtype == "bf16"
    es = 2 if dtype == "bf16" else 4
    H = W = (lr + 31) // 32 * 32
    P = H * W
    nb = n_blocks
    w = {}
    br = [(16, 0), (64, 1), (256, 2), (256, 2)]

    def add(cat, fl, by, n):
        f0, b0, n0 = w.get(cat, (0.0, 0.0, 0))
        w[cat] = (f0 + fl, b0 + by, n0 + n)

    for bi, (C_, L) in enumerate(br):
        M = B * P // (4 ** L)
        win = M // 64
        dg = fused_qkv_dgrad and C_ >= 64
        # reads q|k|v (3C) and the output gradient (C), writes dq|dK|dV (3C); with the projection data gradient inside the
        # kernel also g_d (C) and 2 M 3C C more FLOPs
        rc = (c16_recompute and C_ == 16 and fused_c16_fwd) or (c64_recompute and C_ == 64 and fused_attn_fwd and dg)
        # with recompute: reads d (C) instead of q|k|v (3C), + the projection's FLOPs again
        # branch 3 (index 2) with branch 4's branch_prep_bwd inside: also reads branch 4's g_d rows and g_xc[chunk 3], writes g_n[chunk 3] and
        # the updated g_xc[chunk 2] (C per low-resolution pixel each)
        pbk = 4 if (fused_prep_bwd and dg and bi == 2) else 0
        add(f"attn_bwd_c{C_}", nb * (win * 64000.0 * C_ + (2.0 * M * C_ * 3 * C_ if dg else 0.0) + (2.0 * M * C_ * 3 * C_ if rc else 0.0)),
            nb * M * ((8 if dg else 7) - (2 if rc else 0) + pbk) * C_ * es, nb)
        if (fused_c16_fwd if C_ == 16 else fused_attn_fwd):
            # reads x (+ the residual rows for C >= 64; for C = 16 the residual IS x), writes qkv + out (+ d1 for C = 16;
            # with recompute the C = 16 kernel writes d1 and out only)
            # with branch_prep inside (C >= 64): reads the block-input plane and the previous branch's output plane (C per low-resolution
            # pixel each) instead of d and the residual, and writes d as well (xin, its own residual, is a temporary: not priced)
            pf = 1 if (fused_prep_fwd and C_ >= 64) else 0
            add(f"attn_fused_c{C_}", nb * (win * 25600.0 * C_ + 2.0 * M * C_ * 3 * C_), nb * M * ((3 if rc else 6) + pf) * C_ * es, nb)   # (C = 64 with recompute: x, residual in, out written = 3 C too)
        else:
            add(f"attn_fwd_c{C_}", nb * win * 25600.0 * C_, nb * M * (4 * C_ + (C_ if C_ == 16 else 0)) * es, nb)
            add("gemm_qkv", nb * 2.0 * M * C_ * 3 * C_, nb * M * 4 * C_ * es, nb)
        if not dg and not (c16_prep and C_ == 16):
            add("gemm_qkv_dgrad", nb * 2.0 * M * C_ * 3 * C_, nb * M * 4 * C_ * es, nb)
        add("wgrad_qkv", nb * 2.0 * M * C_ * 3 * C_, nb * M * 4 * C_ * es, nb)
    conv_fl = 2.0 * B * P * 64 * 576
    add("conv3x3_fwd", nb * conv_fl, nb * B * P * 64 * es * 3, nb)
    if fused_conv_bwd:
        # one pass: reads the output gradient and the conv input, writes the input gradient (the fp32 partial slabs of the weight
        # gradient, 37.7 MB per launch at 256 workgroups, are bookkeeping of the split, not algorithmic bytes)
        add("conv3x3_bwd", nb * 2 * conv_fl, nb * B * P * 64 * es * 3, nb)
    else:
        add("conv3x3_dgrad", nb * conv_fl, nb * B * P * 64 * es * 2, nb)
        add("conv3x3_wgrad", nb * conv_fl, nb * B * P * 64 * es * 2, nb)
    stream_x23 = scale != 4 and dtype == "bf16" and fused_tail_fwd       # x2 / x3: the whole tail as two row-streaming kernels (round 4)
    if scale == 4:
        # tail.0: M=BP, K=64, N=256 ; tail.3: M=4BP ; each: fwd + dgrad GEMM and a wgrad
        for M in (B * P, 4 * B * P):
            add("tail_gemm", 2 * 2.0 * M * 64 * 256, 2 * M * (64 + 256) * es, 2)
            add("tail_wgrad", 2.0 * M * 64 * 256, M * (64 + 256) * es, 1)
        HR = 16 * B * P
    else:
        r2 = scale * scale
        M = B * P
        if not stream_x23:
            add("tail_gemm", 2 * 2.0 * M * 64 * 64 * r2, 2 * M * (64 + 64 * r2) * es, 2)
            add("tail_wgrad", 2.0 * M * 64 * 64 * r2, M * (64 + 64 * r2) * es, 1)
        HR = r2 * B * P
    fin = 2.0 * HR * 64 * 27
    if stream_x23:
        # k_tail_stream.hip / k_tail_bwd_stream.hip: forward reads the body output (M x 64), writes the fp32 image; backward reads
        # g(sr) and the body output, writes g(body output); the expansion's FLOPs once forward, three times backward (recompute, data
        # gradient, weight gradient), the tail conv's once / twice
        M = B * P
        ex = 2.0 * M * 64 * 64 * scale * scale
        w["tail_fwd_fused"] = (fin + ex, M * 64 * es + HR * 12, 1)
        w["final_conv_dgrad"] = (2 * fin + 3 * ex, HR * 12 + 2 * M * 64 * es, 1)
    elif scale == 4 and dtype == "bf16" and fused_tail_fwd:
        # fused forward tail (k_tail_stream.hip / k_tail_fwd.hip): reads gelu(t1) (HR/4 pixels x 64), writes the fp32 output; tail.3 expansion + tail conv
        w["tail_fwd_fused"] = (fin + 2.0 * (HR // 4) * 64 * 256, (HR // 4) * 64 * es + HR * 12, 1)
        f0, b0, n0 = w["tail_gemm"]
        w["tail_gemm"] = (f0 - 2.0 * (HR // 4) * 64 * 256, b0 - (HR // 4) * (64 + 256) * es, n0 - 1)
    else:
        add("final_conv_fwd", fin, HR * (64 * es + 12), 1)
    if stream_x23:
        pass
    elif scale == 4 and dtype == "bf16":
        # fused tail backward (k_tail_bwd.hip): tail conv dgrad + wgrad, GELU', tail.3 dgrad + wgrad in one pass:
        # reads g(sr) (HR x 12 B), gelu(t1), gelu'(t1) (HR/4); writes g(t1) (HR/4).  The stored variant (option "fused_tail" = 1)
        # also reads gelu(t2), gelu'(t2) (HR x 2 x 64 es); the default recomputing variant (>= 2) does not -- it pays the tail.3
        # expansion's FLOPs a second time instead
        mid = HR // 4
        if fused_tail_fwd:
            w["final_conv_dgrad"] = (2 * fin + 3 * 2.0 * mid * 64 * 256, HR * 12 + mid * 3 * 64 * es, 1)
        else:
            w["final_conv_dgrad"] = (2 * fin + 2 * 2.0 * mid * 64 * 256, HR * (2 * 64 * es + 12) + mid * 3 * 64 * es, 1)
    else:
        add("final_conv_dgrad", fin, HR * (2 * 64 * es + 12), 1)
        add("final_conv_wgrad", fin, HR * (64 * es + 12), 1)
    return w


def plan_options(plan) -> dict:
    """The kernel-selection options in force on a plan (m2t_plan_query("opt:<key>")): which kernels ran decides which
    rows of `algorithmic_work` apply."""
    o = {k: bool(plan.query("opt:" + k)) for k in ("fused_attn_fwd", "fused_c16_fwd", "conv_rows", "fused_conv_bwd", "fused_prep_fwd", "fused_prep_bwd")}
    o["fused_tail_fwd"] = plan.query("opt:fused_tail") >= 2
    o["c16_recompute"] = plan.query("opt:fused_c16_fwd") == 2
    o["c64_recompute"] = plan.query("opt:fused_attn_fwd") == 2
    o["fused_qkv_dgrad"] = plan.query("opt:attn_bwd") >= 2
    o["c16_prep"] = plan.query("opt:attn_bwd") == 3
    return o


def roofline_report(B: int, lr: int, scale: int, dtype: str, steps: int, pmc_file: str | None = None,
                    source_stamp: str | None = None, workload: str | None = None, plan=None):
    """Roofline object for the dominant (largest total time) kernel category measured in the
    timed region, plus a compact table of the others.  `traffic` (PMC bytes per launch) comes from a separate
    rocprofv3 --pmc collection (tools/pmc_traffic.py -> profiles/pmc_traffic.json); it is reported only when that
    file was collected on THIS build (same kernel-source stamp), workload and compute dtype, otherwise null."""
    t = read_all()
    opts = plan_options(plan) if plan is not None else {}
    work = algorithmic_work(B, lr, scale, dtype, fused_attn_fwd=opts.get("fused_attn_fwd"), fused_tail_fwd=bool(opts.get("fused_tail_fwd", False)),
                            fused_qkv_dgrad=opts.get("fused_qkv_dgrad"), fused_c16_fwd=opts.get("fused_c16_fwd"),
                            c16_recompute=opts.get("c16_recompute"), c64_recompute=opts.get("c64_recompute"),
                            fused_conv_bwd=opts.get("fused_conv_bwd"), c16_prep=opts.get("c16_prep"), fused_prep_fwd=opts.get("fused_prep_fwd"), fused_prep_bwd=opts.get("fused_prep_bwd"))
    traffic = {}
    if pmc_file and os.path.exists(pmc_file):
        try:
            blob = json.load(open(pmc_file))
            if source_stamp is not None and blob.get("source_stamp") == source_stamp and blob.get("workload", "config1") == workload \
                    and blob.get("dtype", "bf16") == dtype:
                traffic = blob.get("traffic_bytes_per_launch", {})
        except Exception:
            traffic = {}
    rows = []
    for name, (ms, n) in t.items():
        if n == 0 or name not in work:
            continue
        fl, by, nl = work[name]
        fl_l, by_l = fl / nl, by / nl                  # per launch
        avg_s = ms / n * 1e-3
        t_hbm = by_l / (HBM_PEAK_GBS * 1e9)
        t_mfma = fl_l / (MFMA_PEAK_TF[dtype] * 1e12)
        bound = "hbm" if t_hbm >= t_mfma else "mfma"
        if bound == "hbm":
            ach, peak, unit = by_l / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = fl_l / avg_s / 1e12, MFMA_PEAK_TF[dtype], "TFLOP/s"
        kname = KERNEL_OF_BF16.get(name, KERNEL_OF[name]) if dtype == "bf16" else KERNEL_OF[name]
        rows.append({"kernel": kname, "category": name, "bound": bound, "achieved": round(ach, 2),
                     "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": traffic.get(name),
                     "avg_launch_us": round(avg_s * 1e6, 2), "sampled_launches": n, "sampled_total_ms": round(ms, 3),
                     "launches_per_step": nl, "est_ms_per_step": round(avg_s * 1e3 * nl, 4),
                     "hbm_GBs": round(by_l / avg_s / 1e9, 1), "mfma_TFs": round(fl_l / avg_s / 1e12, 2)})
    if not rows:
        return None
    for r in rows:
        if r["traffic"] is None:
            alias = {"conv3x3_fwd": "conv3x3_fwd+dgrad", "conv3x3_dgrad": "conv3x3_fwd+dgrad"}.get(r["category"])
            if alias:
                r["traffic"] = traffic.get(alias)
    # one kernel, one tile shape, launched under two categories: rocprofv3 lists it as ONE row, so does the headline
    for merged, parts in MERGED.items():
        sub = [r for r in rows if r["category"] in parts]
        if len(sub) != len(parts):
            continue
        n = sum(r["sampled_launches"] for r in sub)
        ms = sum(r["sampled_total_ms"] for r in sub)
        by = sum(work[r["category"]][1] / work[r["category"]][2] * r["sampled_launches"] for r in sub)
        fl = sum(work[r["category"]][0] / work[r["category"]][2] * r["sampled_launches"] for r in sub)
        nl = sum(r["launches_per_step"] for r in sub)
        sec = ms * 1e-3
        bound = "hbm" if by / (HBM_PEAK_GBS * 1e9) >= fl / (MFMA_PEAK_TF[dtype] * 1e12) else "mfma"
        ach, peak, unit = (by / sec / 1e9, HBM_PEAK_GBS, "GB/s") if bound == "hbm" else (fl / sec / 1e12, MFMA_PEAK_TF[dtype], "TFLOP/s")
        rows.append({"kernel": MERGED_KERNEL[merged], "category": merged, "bound": bound, "achieved": round(ach, 2), "peak": peak,
                     "unit": unit, "frac": round(ach / peak, 4), "traffic": traffic.get(merged),
                     "avg_launch_us": round(ms / n * 1e3, 2), "sampled_launches": n, "sampled_total_ms": round(ms, 3),
                     "launches_per_step": nl, "est_ms_per_step": round(ms / n * nl, 4),
                     "hbm_GBs": round(by / sec / 1e9, 1), "mfma_TFs": round(fl / sec / 1e12, 2)})
    # (events may ride on a 1-in-n sample of a category's launches: rank by average x launches per step, never by the sampled totals)
    rows.sort(key=lambda r: -r["est_ms_per_step"])
    # the headline object is the dominant SINGLE kernel (one shape per launch); the qkv / tail GEMM
    # categories aggregate four different shapes each and are listed under "others"
    parts_of_merged = {c for m, parts in MERGED.items() if any(r["category"] == m for r in rows) for c in parts}
    single = [r for r in rows if r["category"].startswith(("attn_", "conv3x3_", "final_conv_")) and r["category"] not in parts_of_merged]
    first = single[0] if single else rows[0]
    rows.remove(first)
    rows.insert(0, first)
    top = dict(rows[0])
    top["others"] = [{k: r[k] for k in ("category", "bound", "frac", "avg_launch_us", "launches_per_step", "est_ms_per_step", "hbm_GBs", "mfma_TFs")}
                     for r in rows[1:] if r["category"] not in MERGED]
    return top


def dominant_mask(totals) -> int:
    """Bit mask of the categories of the dominant single-shape kernel, from `read_all()` totals (forward and data
    gradient of the 3x3 conv are one kernel)."""
    cand = {}
    merged_parts = set()
    for m, parts in MERGED.items():
        if all(totals.get(c, (0.0, 0))[1] > 0 for c in parts):
            cand[m] = (sum(totals[c][0] for c in parts), sum(1 << CATS.index(c) for c in parts))
            merged_parts.update(parts)
    for i, c in enumerate(CATS):
        if c.startswith(("attn_", "conv3x3_", "final_conv_")) and c not in merged_parts and totals.get(c, (0.0, 0))[1] > 0:
            cand[c] = (totals[c][0], 1 << i)
    if not cand:
        return 0
    return max(cand.values(), key=lambda v: v[0])[1]
