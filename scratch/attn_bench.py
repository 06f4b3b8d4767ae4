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
  for C, h in ((256, 32), (64, 64), (16, 128)):
    print(f"dbg={dbg}", f"C={C} h={h}: {run(C, h, h):.1f} us per call (attn bwd + halo gather + rel reduce x2)")

