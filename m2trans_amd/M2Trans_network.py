"""Drop-in replacement for the reference's ``models/M2Trans_network.py`` on MI355X.

Same plugin surface as the reference (``create_model(args)`` resolved by module name,
train.py:70; class ``M2Trans`` imported by test.py:16,66), same ``state_dict`` (123 entries,
names/shapes/order of models/M2Trans_network.py:17-56), same ``forward`` contract
(float32 NCHW in [0, rgb_range] -> float32 NCHW x scale, :58-76) -- but every device
operation is a hand-written gfx950 kernel in ``libm2t.so`` reached through the C ABI of
``include/m2t.h``.  PyTorch only owns the memory, the stream and (optionally) the autograd
edge around the whole model.  There is no CPU / eager fallback: without the HIP library
or a GPU tensor, ``forward`` raises.

Differences that are deliberate (and invisible through the reference's own call sites):
  * all trainable parameters are views into ONE flat float32 buffer (``model.flat_params``),
    gradients likewise (``model.flat_grads``) -> fused Adam, single RCCL all-reduce;
  * ``args.compute_dtype`` (optional, default "fp32"; or env M2T_COMPUTE_DTYPE) selects exact-fp32
    MFMA or bf16 MFMA with fp32 accumulation/statistics/master weights.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
import warnings
import weakref
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import M2TError

_BRANCH_C = (16, 64, 256, 256)
# plans (+ their workspaces, ~0.25 GB per 128x128 image in bf16) kept per model: an eval loop over images of arbitrary sizes
# (test.py:77-122, metrics.evaluate) would otherwise grow HBM by one workspace per distinct shape, forever
PLAN_CACHE_SIZE = int(os.environ.get("M2T_PLAN_CACHE", "8"))


def create_model(args):
    """Reference hook: models/M2Trans_network.py:12-13."""
    return M2Trans(args)


class _Holder(nn.Module):
    """A parameter container that only exists to reproduce the reference's state_dict names."""


class Plan:
    """m2t_plan + its workspace (one per (B, H0, W0, dtype))."""

    def __init__(self, B: int, H0: int, W0: int, scale: int, n_blocks: int, dtype: int, device):
        lib = _lib.load()
        h = C.c_void_p()
        self.handle = None
        # everything the plan may create on a device (events, the side stream) belongs to `device`, not to whatever is current
        with torch.cuda.device(device):
            _lib.check(lib.m2t_plan_create(C.byref(h), B, H0, W0, scale, n_blocks, dtype), "m2t_plan_create")
            self.handle = h
            self.B, self.H0, self.W0, self.scale, self.n_blocks, self.dtype = B, H0, W0, scale, n_blocks, dtype
            self.device = device
            self.gen = 0
            self.trained = False         # a backward pass has run on this plan: the LRU keeps it while forward-only plans remain
            nbytes = self.query("workspace_bytes")
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
            _lib.check(lib.m2t_plan_init_workspace(self.handle, _lib.ptr(self.workspace), _lib.stream_ptr()),
                       "m2t_plan_init_workspace")

    def query(self, key: str) -> int:
        v = _lib.load().m2t_plan_query(self.handle, key.encode())
        if v < 0:
            raise KeyError(key)
        return int(v)

    def grad_buckets(self):
        """[(lo, hi)] float ranges of the flat gradient buffer in the order m2t_backward completes them."""
        n = self.query("grad_buckets")
        return [(self.query(f"grad_bucket_lo:{i}"), self.query(f"grad_bucket_hi:{i}")) for i in range(n)]

    def ws_tensor(self, name: str, shape=None, dtype=None) -> torch.Tensor:
        """View of a named workspace tensor (tests / introspection)."""
        off, n = self.query("ws:" + name), self.query("wsn:" + name)
        if dtype is None:
            dtype = torch.float32 if self.dtype == _lib.F32 else torch.bfloat16
        es = torch.empty((), dtype=dtype).element_size()
        t = self.workspace[off: off + n * es].view(dtype)
        return t.view(shape) if shape is not None else t

    def __del__(self):
        try:
            if self.handle:
                _lib.load().m2t_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _M2TransFunction(torch.autograd.Function):
    """One autograd node around the whole network: forward = m2t_forward, backward =
    m2t_set_output_grad + m2t_backward (hand-written kernels, no autograd graph inside)."""

    @staticmethod
    def forward(ctx, x, model, *params):
        plan = model._plan_for(x)
        # the kernels (forward AND the head weight gradient of backward) read raw contiguous fp32 NCHW memory:
        # convert once and save the CONVERTED tensor (a channels_last / sliced / half-precision input would
        # otherwise hand m2t_backward a pointer with the wrong layout or element size)
        xc = x.detach().contiguous().float()
        sr = model._run_forward(plan, xc, keep=True)
        ctx.model, ctx.plan, ctx.gen = model, plan, plan.gen
        ctx.save_for_backward(xc)
        return sr

    @staticmethod
    def backward(ctx, g_sr):
        model, plan = ctx.model, ctx.plan
        if ctx.gen != plan.gen:
            raise M2TError("M2Trans.backward: the workspace of this shape was overwritten by a later forward; "
                           "call backward before the next forward of the same shape")
        (x,) = ctx.saved_tensors
        lib = _lib.load()
        g_sr = g_sr.contiguous().float()
        with torch.cuda.device(x.device):
            st = _lib.stream_ptr()
            _lib.check(lib.m2t_set_output_grad(plan.handle, _lib.ptr(g_sr), float(model.rgb_range),
                                               _lib.ptr(plan.workspace), st), "m2t_set_output_grad")
            grads = torch.empty_like(model.flat_params)
            _lib.check(lib.m2t_backward(plan.handle, _lib.ptr(model.flat_params), _lib.ptr(x), _lib.ptr(grads),
                                        _lib.ptr(plan.workspace), st), "m2t_backward")
        plan.trained = True
        if model._dp_master is not None:
            model._dp_release()
        outs = [grads[o: o + n].view(s) for (o, n, s) in model._slots]
        return (None, None, *outs)


class _ReplicaState:
    """What one nn.DataParallel replica needs on ITS device and cannot share: the flat fp32 copy of the parameters the
    kernels read and the plans (workspaces with the saved activations).  Owned by the master module, handed to the
    short-lived replica objects DataParallel builds every forward, returned when the replica (and the autograd node that
    holds it) is gone."""

    def __init__(self, device, numel: int):
        self.device = device
        self.flat = torch.empty(numel, dtype=torch.float32, device=device)
        self.plans: "OrderedDict[Tuple, Plan]" = OrderedDict()
        self.busy = False
        self.owner = None                # token of the replica that holds the state (a late finalizer must not free a re-issued state)


def _release_replica_state(state: _ReplicaState, lock: threading.Lock, token) -> None:
    with lock:
        if state.owner is token:
            state.owner = None
            state.busy = False


_DP_WARNED = False


class M2Trans(nn.Module):
    def __init__(self, args):
        super().__init__()
        n_feats = int(args.n_feats)
        if n_feats != 64 or int(getattr(args, "colors", 3)) != 3:
            raise M2TError("the gfx950 kernels are specialised for n_feats=64, colors=3 (every shipped config)")
        self.scale = int(args.scale)
        if self.scale not in (2, 3, 4):
            raise M2TError("scale must be 2, 3 or 4")
        self.window_sizes = [8, 16, 32]
        self.rgb_range = float(args.rgb_range)
        self.n_blocks = int(args.n_blocks)
        cd = getattr(args, "compute_dtype", None) or os.environ.get("M2T_COMPUTE_DTYPE", "fp32")
        self.compute_dtype = str(cd)
        self._plans: "OrderedDict[Tuple, Plan]" = OrderedDict()
        self._dp_master = None                       # set on the replicas nn.DataParallel makes of this module
        self._dp_params = None
        self._dp_release = None
        self._dp_pool: Dict[int, List[_ReplicaState]] = {}
        self._dp_lock = threading.Lock()
        self._build_parameters(n_feats)
        self._flatten()

    # ------------------------------------------------------------------ parameters
    def _build_parameters(self, nf: int):
        """Create the parameters in the reference's construction order with the reference's
        initialisers (models/M2Trans_network.py:30-56,119-126,277-288,342-345,370-379), so the
        same torch seed yields the same initial weights."""
        def conv(cin, cout, k, bias=True):
            c = nn.Conv2d(cin, cout, kernel_size=k, bias=bias)      # default kaiming-uniform init
            return c.weight.detach().clone(), (c.bias.detach().clone() if bias else None)

        def mean_shift(sign):
            conv(3, 3, 1)                                            # nn.Conv2d.__init__ consumes RNG first
            h = _Holder()
            std = torch.ones(3)
            mean = torch.tensor([0.4488, 0.4371, 0.4040])
            h.weight = nn.Parameter(torch.eye(3).view(3, 3, 1, 1) / std.view(3, 1, 1, 1), requires_grad=False)
            h.bias = nn.Parameter(sign * self.rgb_range * mean / std, requires_grad=False)
            return h

        self.sub_mean = mean_shift(-1)
        self.add_mean = mean_shift(1)
        w, b = conv(3, nf, 3)
        self.head = _Holder()
        self.head.weight, self.head.bias = nn.Parameter(w), nn.Parameter(b)
        self.body = nn.ModuleList()
        for _ in range(self.n_blocks):
            blk = _Holder()
            for i, c in enumerate(_BRANCH_C, start=1):
                a = _Holder()
                a.rel_h = nn.Parameter(torch.randn(1, 10, 1, c // 2))
                a.rel_w = nn.Parameter(torch.randn(1, 1, 10, c // 2))
                wq, _ = conv(c, 3 * c, 1, bias=False)
                a.qkv_conv = _Holder()
                a.qkv_conv.weight = nn.Parameter(wq)
                nn.init.kaiming_normal_(a.qkv_conv.weight, mode="fan_out", nonlinearity="relu")
                nn.init.normal_(a.rel_h, 0, 1)
                nn.init.normal_(a.rel_w, 0, 1)
                setattr(blk, f"attn{i}", a)
            w, b = conv(nf, nf, 3)
            ff = _Holder()
            ff.weight, ff.bias = nn.Parameter(w), nn.Parameter(b)
            blk.feed_forward = nn.ModuleDict({"0": ff})
            self.body.append(blk)
        tail = OrderedDict()
        if self.scale == 4:
            for key in ("0", "3"):
                w, b = conv(nf, nf * 4, 1)
                h = _Holder()
                h.weight, h.bias = nn.Parameter(w), nn.Parameter(b)
                tail[key] = h
            w, _ = conv(nf, 3, 3, bias=False)
            h = _Holder()
            h.weight = nn.Parameter(w)
            tail["6"] = h
        else:
            w, b = conv(nf, nf * self.scale * self.scale, 1)
            h = _Holder()
            h.weight, h.bias = nn.Parameter(w), nn.Parameter(b)
            tail["0"] = h
            w, _ = conv(nf, 3, 3, bias=False)
            h = _Holder()
            h.weight = nn.Parameter(w)
            tail["3"] = h
        self.tail = nn.ModuleDict(tail)

    def _trainable(self) -> List[Tuple[str, nn.Parameter]]:
        return [(n, p) for n, p in self.named_parameters() if not n.startswith(("sub_mean", "add_mean"))]

    def _flatten(self):
        """(Re)build the flat fp32 parameter buffer and make every trainable parameter a view of it."""
        named = self._trainable()
        dev = named[0][1].device
        for n, p in named:
            if p.dtype != torch.float32:
                raise M2TError("master parameters must stay float32 (use compute_dtype='bf16' for bf16 math)")
        total = sum(p.numel() for _, p in named)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        slots, off = [], 0
        for n, p in named:
            k = p.numel()
            flat[off: off + k].copy_(p.data.reshape(-1))
            p.data = flat[off: off + k].view(p.shape)
            slots.append((off, k, tuple(p.shape)))
            off += k
        self.flat_params = flat
        self._slots = slots
        self._names = [n for n, _ in named]
        self.flat_grads = None
        self._plans = OrderedDict()
        self._dp_pool = {}

    # ------------------------------------------------------------------ nn.DataParallel over more than one device
    def _replicate_for_data_parallel(self):
        """``nn.DataParallel(model)`` (train.py:73, test.py:68) on a node with SEVERAL visible GPUs replicates the module every
        forward (``torch.nn.parallel.replicate``): the replica is a shallow copy whose parameters are broadcast copies set as
        plain attributes.  A replica of THIS module additionally needs its own flat parameter buffer and plans on its device
        (the master's live on ``device_ids[0]``): they are bound in ``forward`` from a per-device pool the master owns, so the
        unchanged reference script runs on all devices -- one Python thread per replica, parameters broadcast and gradients
        reduced to device 0 by DataParallel every step, exactly the reference's mechanism and its cost.  The fast path on a
        multi-GPU node is one process per GPU (``python -m torch.distributed.run --nproc-per-node N``; ``TrainStep`` /
        ``m2trans_amd.dist``): a one-time warning says so.  ``M2T_DATA_PARALLEL=error`` turns the warning into an M2TError."""
        global _DP_WARNED
        mode = os.environ.get("M2T_DATA_PARALLEL", "replicate")
        msg = ("nn.DataParallel is replicating M2Trans over several devices: every step broadcasts the parameters and "
               "reduces the gradients through device 0 from Python threads of ONE process (the reference's train.py:73 "
               "mechanism).  For full speed launch one process per GPU instead: python -m torch.distributed.run "
               "--nproc-per-node <N> with m2trans_amd.train_step.TrainStep (RCCL all-reduce of the flat gradient buffer), or "
               "restrict this process to one device (nn.DataParallel(model, device_ids=[0]) / HIP_VISIBLE_DEVICES).")
        if mode == "error":
            raise M2TError(msg)
        if not _DP_WARNED:
            _DP_WARNED = True
            warnings.warn(msg, stacklevel=3)
        replica = super()._replicate_for_data_parallel()
        replica._dp_master = self
        replica._dp_params = None
        replica._dp_release = None
        replica.flat_params = None          # bound to this replica's device in forward()
        replica.flat_grads = None
        replica._plans = None
        return replica

    def _bind_replica(self, x: torch.Tensor) -> None:
        """First use of a DataParallel replica (its device is only known now): take a free state of x.device from the master's
        pool, gather the broadcast parameter copies into that state's flat buffer (ONE cat: 14.5 MB), serve plans from it."""
        if self.flat_params is not None:
            return
        master = self._dp_master
        if not x.is_cuda:
            raise M2TError("M2Trans (MI355X build) runs only on a HIP device tensor; there is no CPU fallback")
        params = []
        for n in master._names:
            obj = self
            for part in n.split("."):
                obj = getattr(obj, part)
            if obj.device != x.device:
                raise M2TError(f"DataParallel replica: parameter {n} is on {obj.device}, the replica's input on {x.device}")
            params.append(obj)
        dev = x.device.index
        with master._dp_lock:
            pool = master._dp_pool.setdefault(dev, [])
            state = next((s for s in pool if not s.busy), None)
            if state is None:
                state = _ReplicaState(x.device, master.flat_params.numel())
                pool.append(state)
            state.busy = True
            token = state.owner = object()
        # The state returns to the pool as soon as nothing needs its workspace any more: behind a no-grad forward, or behind the backward
        # pass of this replica's autograd node (work enqueued later on the stream is ordered behind it, as for the master's own plan; a
        # second backward through a retained graph then fails the plan-generation check instead of reading overwritten activations).
        # The finalizer covers replicas whose graph is dropped without a backward.
        self._dp_release = lambda: _release_replica_state(state, master._dp_lock, token)
        weakref.finalize(self, _release_replica_state, state, master._dp_lock, token)
        with torch.no_grad():
            torch.cat([p.detach().reshape(-1) for p in params], out=state.flat)
        self.flat_params = state.flat
        self._plans = state.plans
        self._dp_params = params

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def attach_flat_grads(self) -> torch.Tensor:
        """Make every p.grad a view of one flat buffer (what the fused step driver uses)."""
        if self.flat_grads is None or self.flat_grads.device != self.flat_params.device:
            self.flat_grads = torch.zeros_like(self.flat_params)
            for (n, p), (o, k, s) in zip(self._trainable(), self._slots):
                p.grad = self.flat_grads[o: o + k].view(s)
        return self.flat_grads

    def param_offsets(self) -> Dict[str, Tuple[int, int]]:
        return {n: (o, k) for n, (o, k, _) in zip(self._names, self._slots)}

    # ------------------------------------------------------------------ state dict
    def load_state_dict(self, state_dict, strict=False):
        """Behaves like the reference's lenient loader (models/M2Trans_network.py:88-112):
        copies matching names, tolerates a mismatching ``tail`` (different scale), raises on
        any other shape mismatch; additionally strips DataParallel's ``module.`` prefix that
        the reference's checkpoints carry (train.py:73,345)."""
        own = self.state_dict()
        seen = set()
        for name, value in state_dict.items():
            if name.startswith("module."):
                name = name[len("module."):]
            seen.add(name)
            if name not in own:
                if strict and "tail" not in name:
                    raise KeyError('unexpected key "{}" in state_dict'.format(name))
                continue
            value = value.data if isinstance(value, nn.Parameter) else value
            if tuple(own[name].shape) != tuple(value.shape):
                if "tail" in name:
                    print("Replace pre-trained upsampler to new one...")
                    continue
                raise RuntimeError("While copying the parameter named {}, whose dimensions in the model are {} "
                                   "and whose dimensions in the checkpoint are {}.".format(
                                       name, tuple(own[name].shape), tuple(value.shape)))
            own[name].copy_(value)
        if strict:
            missing = set(own.keys()) - seen
            if missing:
                raise KeyError('missing keys in state_dict: "{}"'.format(missing))

    # ------------------------------------------------------------------ forward
    def _dtype_code(self) -> int:
        if self.compute_dtype in ("fp32", "float32", "f32"):
            return _lib.F32
        if self.compute_dtype in ("bf16", "bfloat16"):
            return _lib.BF16
        raise M2TError(f"unknown compute_dtype {self.compute_dtype!r}")

    def _device_ok(self, x) -> None:
        if not x.is_cuda:
            raise M2TError("M2Trans (MI355X build) runs only on a HIP device tensor; there is no CPU fallback")
        if self.flat_params.device != x.device:
            raise M2TError(f"model parameters are on {self.flat_params.device}, input on {x.device}")

    def _check_plan(self, plan: Plan) -> None:
        """The library's parameter inventory must be this module's (names, offsets, sizes)."""
        if plan.query("num_params") != self.flat_params.numel():
            raise M2TError("parameter inventory of the library and of the module differ")
        for n, (o, k, _) in zip(self._names, self._slots):
            if plan.query("param:" + n) != o or plan.query("numel:" + n) != k:
                raise M2TError(f"parameter layout mismatch for {n}")

    def _plan_for(self, x: torch.Tensor) -> Plan:
        if x.dim() != 4 or x.shape[1] != 3:
            raise M2TError("expected input of shape [B,3,H,W]")
        self._device_ok(x)
        B, _, H0, W0 = x.shape
        key = (B, H0, W0, self._dtype_code(), x.device.index)
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(B, H0, W0, self.scale, self.n_blocks, key[3], x.device)
            self._check_plan(plan)
            self._plans[key] = plan
            # least-recently-used plans go first; a plan whose backward is still pending is kept alive by its autograd node
            # (ctx.plan), only the cache entry is dropped
            # Plans that have run a backward pass (the training shape: rebuilding it costs a workspace allocation, the descriptor
            # uploads and a first_backward pass) go only when nothing else is left: a validation sweep over many image sizes
            # (test.py:77-122 between epochs) evicts among ITS OWN forward-only plans.
            while len(self._plans) > max(1, PLAN_CACHE_SIZE):
                victim = next((k for k, pl in self._plans.items() if not getattr(pl, "trained", False) and k != key), None)
                if victim is None:
                    victim = next(k for k in self._plans if k != key)
                del self._plans[victim]
        else:
            self._plans.move_to_end(key)
        return plan

    def _run_forward(self, plan: Plan, x: torch.Tensor, keep: bool, want_sr: bool = True):
        lib = _lib.load()
        x = x.contiguous().float()
        B, _, H0, W0 = x.shape
        sr = torch.empty(B, 3, H0 * self.scale, W0 * self.scale, dtype=torch.float32, device=x.device) if want_sr else None
        plan.gen += 1
        with torch.cuda.device(x.device):
            _lib.check(lib.m2t_forward(plan.handle, _lib.ptr(self.flat_params), _lib.ptr(x), _lib.ptr(sr),
                                       float(self.rgb_range), 1 if keep else 0, _lib.ptr(plan.workspace),
                                       _lib.stream_ptr()), "m2t_forward")
        return sr

    def forward(self, x):
        if self._dp_master is not None:                      # a replica made by nn.DataParallel: bind its device state first
            self._bind_replica(x)
        if torch.is_grad_enabled():                          # (the parameter walk only where an autograd edge may be needed)
            params = self._dp_params if self._dp_master is not None else [p for _, p in self._trainable()]
            if any(p.requires_grad for p in params):
                return _M2TransFunction.apply(x, self, *params)
        sr = self._run_forward(self._plan_for(x), x, keep=False)
        if self._dp_master is not None:
            self._dp_release()
        return sr

    def check_image_size(self, x):
        """Kept for API parity (models/M2Trans_network.py:78-86); the kernels pad by index math."""
        _, _, h, w = x.size()
        wsize = 32
        ph, pw = (wsize - h % wsize) % wsize, (wsize - w % wsize) % wsize
        return torch.nn.functional.pad(x, (0, pw, 0, ph), "reflect")
