"""The training step of the reference (train.py:173-214, setup :76-82,358), driven on MI355X.

``TrainStep`` restates the hot loop body

    optimizer.zero_grad(); sr = model(lr); loss = L1Loss()(sr, hr) * lambda_l1 (+ const clip term);
    loss.backward(); optimizer.step()

as five C-ABI calls on one stream (forward, loss+seed, backward, fused Adam; plus the RCCL
all-reduce of the flat gradient buffer when world_size > 1 -- in two contiguous pieces, the first of
which (tail + 3/4 of the body) starts as soon as m2t_backward has finished it, on a communication
stream that runs under the rest of the backward pass), with no host synchronisation:
the loss stays on the device (the reference's three ``float(loss)`` syncs per step,
train.py:212-214, are left to the caller's logging cadence).

Data parallelism (replaces nn.DataParallel, train.py:73): one process per GPU, persistent
replicas, equal shards of the global batch; each rank divides its L1 sum by the GLOBAL element
count, so SUM-all-reduced gradients equal the full-batch gradient.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib
from .dist import GradBucket, global_divisor
from .M2Trans_network import M2Trans


def cosine_lr(epoch: int, lr0: float = 1e-4, eta_min: float = 1e-6, t_max: float = 200.0) -> float:
    """CosineAnnealingLR(optimizer, float(epochs), eta_min) evaluated at `epoch` scheduler steps
    (train.py:82,358; the scheduler is stepped once per epoch)."""
    return eta_min + 0.5 * (lr0 - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max))


_STREAMS: dict = {}


def _shared_stream(device, role: str):
    """One extra stream per (device, role) for the whole process.  Every HIP stream takes one of the runtime's few hardware queues
    (4 by default); a process that builds several TrainStep objects one after the other (bench.py's `also` workloads) would otherwise
    leave a trail of streams behind, and a later plan's main and side stream can end up on the same queue (measured: -10 ... -40 %)."""
    key = (str(torch.device(device)), role)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=device)
    return _STREAMS[key]


class TrainStep:
    def __init__(self, model: M2Trans, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 lambda_l1: float = 1.0, process_group=None, world_size: Optional[int] = None,
                 grad_bucket_dtype: torch.dtype = torch.float32, semantic_loss=None, lambda_clip: float = 0.0,
                 overlap_comm: bool = True, force_comm_path: bool = False, overlap_semantic: bool = True):
        self.model = model
        # the SemanticLoss forward needs only sr (final after m2t_forward): it runs on its own stream under the backward pass
        self.overlap_semantic = bool(overlap_semantic)
        self.sem_stream = None
        self.lr = float(lr)
        self.betas = (float(betas[0]), float(betas[1]))
        self.eps = float(eps)
        self.lambda_l1 = float(lambda_l1)
        # the MedCLIP regulariser (train.py:78,203-205): a no-grad constant added to the logged loss
        self.semantic_loss = semantic_loss
        self.lambda_clip = float(lambda_clip)
        self.clip_loss = None
        self.pg = process_group
        if world_size is None:
            world_size = torch.distributed.get_world_size(process_group) if (
                torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        self.world_size = int(world_size)
        self.step_count = 0
        self.scheduler_last_epoch = 0          # CosineAnnealingLR.last_epoch of the run (checkpoint.py keeps it across save / resume)
        flat = model.flat_params
        if not flat.is_cuda:
            raise _lib.M2TError("TrainStep needs the model on a HIP device (model.to('cuda'))")
        self.grads = model.attach_flat_grads()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.l1_loss = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.loss = self.l1_loss
        # force_comm_path: build the exchange machinery even for one rank (tests exercise the stream / event logic)
        self.bucket = GradBucket(self.grads, process_group, grad_bucket_dtype, force=force_comm_path,
                                 expect_world=self.world_size) if (self.world_size > 1 or force_comm_path) else None
        # overlapped exchange for either wire format (a bf16 wire stages each range through a slice of the wire buffer)
        self.overlap_comm = bool(overlap_comm) and self.bucket is not None
        self.comm_stream = _shared_stream(flat.device, "comm") if self.overlap_comm else None
        self._last_plan = None
        # bench.py / audits: with measure_exposed_comm the compute stream's wait for the gradient exchange is bracketed by two
        # timing events per step (exposed_comm_events: [(before, after)]) -- what the exchange costs the step after the overlap
        self.measure_exposed_comm = False
        self.exposed_comm_events = []

    def set_lr(self, lr: float):
        self.lr = float(lr)

    # -- pieces (also used by tests) -------------------------------------------------------
    def forward_backward(self, lr_img: torch.Tensor, hr_img: torch.Tensor, captions=None) -> torch.Tensor:
        """forward + L1 (+ the constant SemanticLoss term) + backward into model.flat_grads; returns the
        device loss tensor (this rank's share of the global mean)."""
        m = self.model
        lib = _lib.load()
        plan = m._plan_for(lr_img)
        lr_img = lr_img.contiguous().float()
        hr_img = hr_img.contiguous().float()
        B = lr_img.shape[0]
        if tuple(hr_img.shape) != (B, 3, lr_img.shape[2] * m.scale, lr_img.shape[3] * m.scale):
            raise _lib.M2TError("hr shape must be [B,3,H*scale,W*scale]")
        divisor = global_divisor(hr_img.numel(), self.world_size)      # global mean (equal shards)
        use_clip = self.semantic_loss is not None and self.lambda_clip > 0 and captions is not None
        sr = torch.empty_like(hr_img) if use_clip else None
        plan.gen += 1
        plan.trained = True              # (the plan LRU of the model keeps training plans while forward-only ones remain)
        self._last_plan = plan
        with torch.cuda.device(lr_img.device):
            st = _lib.stream_ptr()
            ws = _lib.ptr(plan.workspace)
            _lib.check(lib.m2t_forward(plan.handle, _lib.ptr(m.flat_params), _lib.ptr(lr_img), _lib.ptr(sr),
                                       float(m.rgb_range), 1, ws, st), "m2t_forward")
            # (deferred: the loss and the backward seed are produced inside m2t_backward, which follows at once -- on the bf16 x4
            #  path by the fused tail backward itself; hr_img stays alive until then)
            _lib.check(lib.m2t_l1_loss_deferred(plan.handle, _lib.ptr(hr_img), self.lambda_l1, divisor, float(m.rgb_range),
                                                _lib.ptr(self.l1_loss), ws, st), "m2t_l1_loss_deferred")
            fwd_done = torch.cuda.current_stream(lr_img.device).record_event() if (use_clip and self.overlap_semantic) else None
            _lib.check(lib.m2t_backward(plan.handle, _lib.ptr(m.flat_params), _lib.ptr(lr_img), _lib.ptr(self.grads),
                                        ws, st), "m2t_backward")
        if use_clip:
            # clip_loss += loss_clip(sr[i], hr[i], caption_i) * lambda_clip  (train.py:203-205); no gradient
            if self.overlap_semantic:
                # the backward pass is already enqueued on the caller's stream; the encoder (which synchronises its own
                # stream once for the crop table) follows the forward pass on a second stream and is joined afterwards
                main = torch.cuda.current_stream(lr_img.device)
                if self.sem_stream is None:
                    # (normal priority: PyTorch-ROCm exposes no priority below the default one)
                    self.sem_stream = _shared_stream(lr_img.device, "semantic")
                self.sem_stream.wait_event(fwd_done)
                with torch.cuda.stream(self.sem_stream):
                    self.clip_loss = self.semantic_loss.batch(sr, hr_img, captions) * self.lambda_clip
                main.wait_stream(self.sem_stream)
            else:
                self.clip_loss = self.semantic_loss.batch(sr, hr_img, captions) * self.lambda_clip
            self.loss = self.l1_loss + self.clip_loss
        else:
            self.loss = self.l1_loss
        return self.loss

    def all_reduce_grads(self):
        """SUM the gradients over the ranks.  Overlapped mode: m2t_backward (already enqueued) completes the flat
        buffer in contiguous buckets (tail, block pairs from last to first, head); each bucket's all-reduce is
        enqueued on the communication stream behind that bucket's event, so it runs under the remaining backward
        kernels; the compute stream then waits for the communication stream (before Adam)."""
        if self.bucket is None:
            return
        ev0 = None
        if self.measure_exposed_comm:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(self.grads.device))
        self._all_reduce_grads()
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(torch.cuda.current_stream(self.grads.device))
            self.exposed_comm_events.append((ev0, ev1))

    def _all_reduce_grads(self):
        if not self.overlap_comm or self._last_plan is None:
            self.bucket.all_reduce()
            return
        lib = _lib.load()
        plan = self._last_plan
        main = torch.cuda.current_stream(self.grads.device)
        buckets = plan.grad_buckets()
        # two collectives: everything that is final once the third-last block pair has been reduced (the tail and
        # 3/4 of the body: its exchange runs under the last quarter of the backward pass), then the rest.  More,
        # smaller collectives cost more in launch latency than they hide (measured with one rank: +25 us each).
        cut = max(0, len(buckets) - 3)
        groups = [(cut, buckets[cut][0], buckets[0][1]), (len(buckets) - 1, 0, buckets[cut][0])] if cut > 0 else \
                 [(len(buckets) - 1, 0, buckets[0][1])]
        with torch.cuda.device(self.grads.device), torch.cuda.stream(self.comm_stream):
            for last, lo, hi in groups:
                _lib.check(lib.m2t_stream_wait_bucket(plan.handle, last, self.comm_stream.cuda_stream), "m2t_stream_wait_bucket")
                self.bucket.all_reduce_range(lo, hi)
        main.wait_stream(self.comm_stream)

    def optimizer_step(self):
        self.step_count += 1
        lib = _lib.load()
        with torch.cuda.device(self.grads.device):
            _lib.check(lib.m2t_adam_step(_lib.ptr(self.model.flat_params), _lib.ptr(self.grads), _lib.ptr(self.exp_avg),
                                         _lib.ptr(self.exp_avg_sq), self.grads.numel(), self.lr, self.betas[0],
                                         self.betas[1], self.eps, self.step_count, 1.0, _lib.stream_ptr()),
                       "m2t_adam_step")

    # -- the step ----------------------------------------------------------------------------
    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor, captions=None) -> torch.Tensor:
        loss = self.forward_backward(lr_img, hr_img, captions)
        self.all_reduce_grads()
        self.optimizer_step()
        return loss
