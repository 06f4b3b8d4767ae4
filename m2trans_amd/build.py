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
SOURCES = ["k_pointwise.hip", "k_gemm.hip", "k_conv.hip", "k_attn.hip", "k_attn_res.hip", "k_attn_c16.hip", "k_attn_fused.hip", "k_attn_fwd2.hip", "k_tail_bwd.hip", "k_tail_fwd.hip", "k_tail_stream.hip", "k_tail_bwd_stream.hip", "k_swin.hip", "k_metrics.hip", "k_fsim.hip", "k_datas.hip", "m2t_api.hip", "m2t_swin.hip", "m2t_rlutrans.hip", "m2t_text.hip"]
HEADERS = ["m2t_common.h", "m2t_kernels.h", "m2t_gemm_load.h", "m2t_haar.h", "m2t_window.h", os.path.join("..", "..", "include", "m2t.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=off", "-fvisibility=hidden",
         "-Rpass-analysis=kernel-resource-usage"]
# Scratch policy (round 5): a kernel that uses scratch fails the build unless it is an fp32 parity-mode instantiation (template
# argument `float`: speed is not their job) or is listed in SPILL_JUSTIFIED with a MEASURED justification.  A spilled value is
# reloaded by a VMEM operation that retires in order with the kernel's prefetches and waits for its stores in flight.
SPILL_JUSTIFIED: dict = {}      # mangled-name substring -> justification (none needed at the moment)


class ScratchError(RuntimeError):
    pass


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


def _spill_allowed(name: str) -> bool:
    # fp32 instantiations carry `f` as the first template TYPE argument of the kernel: ..._kernelIfLi16E...
    return "kernelIf" in name or any(k in name for k in SPILL_JUSTIFIED)


def spill_report() -> dict:
    """{source: [(kernel, bytes per lane)]} of the objects currently built (sidecar files written at compile time, so an
    incremental build reports the same as a clean one)."""
    import json
    out = {}
    objdir = os.path.join(HERE, "build")
    for s in SOURCES:
        side = os.path.join(objdir, s.replace(".hip", ".spills.json"))
        if os.path.exists(side):
            rows = json.load(open(side))
            if rows:
                out[s] = [tuple(r) for r in rows]
    return out


def scratch_violations() -> list:
    return [(src, name, nbytes) for src, rows in spill_report().items() for name, nbytes in rows if not _spill_allowed(name)]


def build(force: bool = False, verbose: bool = True, enforce_scratch: bool = True) -> str:
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
        import json as _json
        spills = _spill_report(src, r.stderr)
        with open(obj.replace(".o", ".spills.json"), "w") as f:
            _json.dump(spills, f)
        for name, nbytes in spills:
            tag = "note (fp32 parity instantiation / justified)" if _spill_allowed(name) else "ERROR"
            print(f"{tag}: {os.path.basename(src)}: kernel {name[:90]} uses {nbytes} B/lane of scratch", flush=True)

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
    bad = scratch_violations()
    if bad and enforce_scratch:
        raise ScratchError("kernels outside the fp32 whitelist use scratch (fix them or list them in SPILL_JUSTIFIED with a measurement):\n"
                           + "\n".join(f"  {src}: {name[:100]}: {n} B/lane" for src, name, n in bad))
    return LIB


if __name__ == "__main__":
    try:
        print(build(force="--force" in sys.argv))
    except ScratchError as e:
        print(e, file=sys.stderr)
        sys.exit(3)
