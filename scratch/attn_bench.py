"""Micro-benchmark of the standalone window-attention backward (bf16) at the plan's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m2trans_amd import _lib as lib
L = lib.load()
def run(C, h, w, B=16, iters=30):
    tdt = torch.bfloat16
    qkv = torch.randn(B, h, w, 3 * C, device="cuda").to(tdt)
    go = torch.randn(B, h, w, C, device="cuda").to(tdt)
    rh = torch.randn(10 * C // 2, device="cuda"); rw = torch.randn(10 * C // 2, device="cuda")
    gq = torch.empty_like(qkv); grh = torch.empty_like(rh); grw = torch.empty_like(rw)
    nb = L.m2t_window_attention_bwd_scratch_bytes(lib.BF16, B, h, w, C)
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = lib.stream_ptr()
    def call():
        lib.check(L.m2t_window_attention_bwd(lib.BF16, lib.ptr(qkv), lib.ptr(rh), lib.ptr(rw), lib.ptr(go), lib.ptr(gq),
                                             lib.ptr(grh), lib.ptr(grw), lib.ptr(scratch), B, h, w, C, st), "bwd")
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for dbg in (0,):
  os.environ["M2T_ATTN_DBG"] = str(dbg)
  for C, h in ((256, 32), (64, 64)):
    print(f"dbg={dbg}", f"C={C} h={h}: {run(C, h, h):.1f} us per call (attn bwd + halo gather + rel reduce x2)")

import ctypes
os.environ["M2T_ATTN_DBG"] = "128"
for C, h in ((256, 32), (64, 64)):
    B = 16
    tdt = torch.bfloat16
    qkv = torch.randn(B, h, h, 3 * C, device="cuda").to(tdt)
    go = torch.randn(B, h, h, C, device="cuda").to(tdt)
    rh = torch.randn(10 * C // 2, device="cuda"); rw = torch.randn(10 * C // 2, device="cuda")
    gq = torch.empty_like(qkv); grh = torch.empty_like(rh); grw = torch.empty_like(rw)
    nb = L.m2t_window_attention_bwd_scratch_bytes(lib.BF16, B, h, h, C)
    scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        lib.check(L.m2t_window_attention_bwd(lib.BF16, lib.ptr(qkv), lib.ptr(rh), lib.ptr(rw), lib.ptr(go), lib.ptr(gq),
                                             lib.ptr(grh), lib.ptr(grw), lib.ptr(scratch), B, h, h, C, lib.stream_ptr()), "bwd")
    torch.cuda.synchronize()
    nwin = B * (h // 8) ** 2
    woff = (nwin * 100 * 2 * C * 2 + 255) & ~255
    ts = scratch[woff:woff + 32 * 8].view(torch.int64).cpu().tolist()
    for blk in (0, 1):
        t = ts[16 * blk:16 * blk + 11]
        print(f"C={C} block {blk*100}:", [t[i + 1] - t[i] for i in range(10)], "total", t[10] - t[0])
