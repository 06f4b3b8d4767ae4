"""Build libm2t.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m m2trans_amd.build            # incremental
    python -m m2trans_amd.build --force

No torch involvement: the library is plain HIP behind `extern "C"`.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libm2t.so")
SOURCES = ["k_pointwise.hip", "k_gemm.hip", "k_conv.hip", "k_attn.hip", "k_attn_res.hip", "k_attn_c16.hip", "k_attn_fused.hip", "k_tail_bwd.hip", "k_tail_fwd.hip", "k_tail_stream.hip", "k_tail_bwd_stream.hip", "k_swin.hip", "k_metrics.hip", "k_fsim.hip", "k_datas.hip", "m2t_api.hip", "m2t_swin.hip", "m2t_rlutrans.hip", "m2t_text.hip"]
HEADERS = ["m2t_common.h", "m2t_kernels.h", "m2t_gemm_load.h", "m2t_haar.h", "m2t_window.h", os.path.join("..", "..", "include", "m2t.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=off", "-fvisibility=hidden",
         "-Rpass-analysis=kernel-resource-usage"]
# kernels where register spills are known and accepted (fp32 parity-mode instantiations: speed is not their job)
SPILL_OK = ("If", "IfL", "float")


def _spill_report(src: str, remarks: str):
    """Scan -Rpass-analysis=kernel-resource-usage output: a spilled prefetch register turns an asynchronous
    global load into a synchronous one (the fused tail backward lost 37 % to a 4-register spill)."""
    import re
    name, out = None, []
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and int(m.group(1)) > 0:
            out.append((name, int(m.group(1))))
    return out


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def run(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        import re as _re
        other = "\n".join(l for l in r.stderr.splitlines()
                          if "kernel-resource-usage" not in l and not _re.match(r"^\s*(\d+\s*)?\|", l) and l.strip())
        if verbose and other.strip():
            print(other)
        for name, nbytes in _spill_report(src, r.stderr):
            print(f"note: {os.path.basename(src)}: kernel {name[:90]} uses {nbytes} B/lane of scratch", flush=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
