"""Process-per-GPU data parallelism for the M2Trans step (replaces nn.DataParallel, train.py:73).

The reference scatters a batch over GPUs inside ONE process and reduces gradients onto GPU 0
every step (parameter broadcast + output gather + grad reduce).  Here every rank owns a
persistent replica and a shard of the global batch; nothing couples samples (InstanceNorm is
per (b,c), attention per window, L1 is a mean), so the only exchange is ONE all-reduce (RCCL
over xGMI on the GPU box, gloo in the CPU tests) of the flat gradient buffer:

    loss_r  = lambda * sum_{shard r} |sr - hr| / N_global          (each rank)
    grad    = SUM_r grad_r   (all-reduce)  ==  gradient of the full-batch mean
    Adam runs redundantly on every rank (3.6 M parameters: no sharded optimiser needed).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_size(global_batch: int, world: int) -> int:
    """Equal shards only: the global-mean scaling below assumes every rank holds B_global / world samples."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    return global_batch // world


def global_divisor(local_hr_numel: int, world: int) -> float:
    """Denominator of the L1 mean over the GLOBAL batch, given this rank's (equal) shard."""
    return float(local_hr_numel) * world


class GradBucket:
    """The single flat gradient bucket (3 629 760 fp32 values = 14.5 MB at x4).

    comm_dtype=torch.bfloat16 halves the bytes on the wire (7.3 MB); the sum is then carried out
    in bf16 by the collective, so fp32 is the default and the parity path."""

    def __init__(self, flat: torch.Tensor, process_group=None, comm_dtype: torch.dtype = torch.float32, force: bool = False,
                 expect_world: Optional[int] = None):
        if flat.dim() != 1 or not flat.is_contiguous():
            raise ValueError("GradBucket needs a contiguous 1-D buffer")
        self.flat = flat
        self.pg = process_group
        self.comm_dtype = comm_dtype
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if expect_world is not None and int(expect_world) != self.world:
            # TrainStep divides the loss by world_size: without a matching process group every rank would train on
            # gradients scaled by 1/N with no exchange and no error
            raise RuntimeError(f"GradBucket: world_size {expect_world} was requested but the process group has {self.world} "
                               "rank(s) (torch.distributed not initialised?)")
        self._wire = None if comm_dtype == flat.dtype else torch.empty_like(flat, dtype=comm_dtype)
        # force: issue the collectives even in a one-rank group (exercises the RCCL / stream path on a single GPU)
        self.force = bool(force) and dist.is_available() and dist.is_initialized()

    def all_reduce(self, async_op: bool = False):
        if self.world == 1 and not self.force:
            return None
        if self._wire is None:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        self._wire.copy_(self.flat)
        dist.all_reduce(self._wire, op=dist.ReduceOp.SUM, group=self.pg)
        self.flat.copy_(self._wire)
        return None

    def all_reduce_range(self, lo: int, hi: int):
        """SUM-all-reduce flat[lo:hi] in place on the CURRENT stream (used by the overlapped exchange: every rank
        issues the same ranges in the same order).  With a bf16 wire only THIS range is staged through the wire
        buffer (cast, collective, cast back, all on the current stream), so the overlap with the backward pass is
        kept and no full-buffer copy is made."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return None
        if self._wire is None:
            return dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
        w = self._wire[lo:hi]
        w.copy_(self.flat[lo:hi])
        dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.pg)
        self.flat[lo:hi].copy_(w)
        return None


def broadcast_params(flat_params: torch.Tensor, src: int = 0, process_group=None):
    """Make every replica start from rank `src`'s weights (the reference re-broadcasts every
    forward; persistent replicas need it once)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_params, src=src, group=process_group)
