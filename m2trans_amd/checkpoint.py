"""Checkpoint wire format of the reference (train.py:341-349 save, :85-108 resume, test.py:64-70 load).

    torch.save({'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'stat_dict'}, path)

* model keys carry DataParallel's ``module.`` prefix (train.py:73);
* optimizer_state_dict is torch.optim.Adam's: per-parameter ``step``/``exp_avg``/``exp_avg_sq`` indexed in
  ``model.parameters()`` order with the 4 frozen MeanShift tensors included (the reference passes every
  parameter to Adam, train.py:81; frozen ones simply never get state);
* scheduler_state_dict is CosineAnnealingLR's.  The reference saves inside the epoch loop BEFORE ``scheduler.step()``
  (train.py:341-358), so the checkpoint of epoch E (1-based) holds ``last_epoch = E - 1``, ``_step_count = E`` and the
  learning rate epoch E trained with, cosine(E - 1).

Both dicts are produced by REAL ``torch.optim.Adam`` / ``CosineAnnealingLR`` objects built the way train.py:81-82
builds them (so every key the installed torch writes is present with torch's own value), then filled with the
fused step driver's flat moment buffers.  Pinned: the reference's model, optimiser and scheduler run through two
epochs and saved as train.py:341-349 does give the committed manifest ``tests/golden/checkpoint_manifest.json``
(written by the pinning script of the test infrastructure); ``tests/test_host_cpu.py`` requires ``export_checkpoint``
to match it key for key.
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import torch


def _param_index(model) -> Dict[str, int]:
    return {n: i for i, (n, _) in enumerate(model.named_parameters())}


def export_checkpoint(model, train_step=None, epoch: int = 1, stat_dict: Optional[dict] = None,
                      lr0: float = 1e-4, eta_min: float = 1e-6, t_max: float = 200.0) -> dict:
    """The dict train.py:341-349 saves at the end of (1-based) epoch `epoch`."""
    sd = {"module." + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    out = {"epoch": int(epoch), "model_state_dict": sd}
    if train_step is not None:
        params = [p for _, p in model.named_parameters()]                 # ALL parameters, like train.py:81
        opt = torch.optim.Adam(params, lr=lr0, weight_decay=0)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, float(t_max), eta_min=eta_min)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                               # "scheduler.step() before optimizer.step()"
            for _ in range(max(0, int(epoch) - 1)):                       # epochs 1 .. E-1 have stepped the scheduler
                sched.step()
        # the learning rate the step driver really used (equals the schedule's when the caller follows cosine_lr)
        opt.param_groups[0]["lr"] = float(train_step.lr)
        sched._last_lr = [float(train_step.lr)]
        if train_step.step_count > 0:
            idx = _param_index(model)
            for n, (o, k, shp) in zip(model._names, model._slots):
                p = params[idx[n]]
                opt.state[p] = {"step": torch.tensor(float(train_step.step_count)),
                                "exp_avg": train_step.exp_avg[o:o + k].view(shp).detach().cpu().clone(),
                                "exp_avg_sq": train_step.exp_avg_sq[o:o + k].view(shp).detach().cpu().clone()}
        out["optimizer_state_dict"] = opt.state_dict()
        out["scheduler_state_dict"] = sched.state_dict()
    out["stat_dict"] = stat_dict or {}
    return out


def import_checkpoint(ckpt: dict, model, train_step=None) -> int:
    """Load a reference-format checkpoint; returns the epoch to continue from (train.py:97-100).
    The learning rate comes from the optimizer's param_groups (what `optimizer.load_state_dict` restores); the
    scheduler's position is kept in ``train_step.scheduler_last_epoch`` so the caller can continue the cosine
    schedule with ``cosine_lr(train_step.scheduler_last_epoch + k)`` after k further scheduler steps -- like the
    reference, whose resumed run re-uses the saved epoch's rate for its first epoch (train.py:103,358)."""
    model.load_state_dict(ckpt["model_state_dict"], strict=True)
    if train_step is None:
        return int(ckpt.get("epoch", 0)) + 1
    opt = ckpt.get("optimizer_state_dict") or {}
    sch = ckpt.get("scheduler_state_dict")
    if not opt.get("state") and not opt.get("param_groups") and sch is None:
        # weights only = the reference's --pretrain load (train.py:85-88): model weights alone, fresh Adam moments, fresh
        # CosineAnnealingLR, start_epoch stays 1 (train.py:62)
        train_step.exp_avg.zero_()
        train_step.exp_avg_sq.zero_()
        train_step.step_count = 0
        train_step.scheduler_last_epoch = 0
        return 1
    if sch is None:
        # Adam state without the schedule it belongs to: the reference's --resume (train.py:97-103) always loads both, and
        # continuing epoch-N moments / learning rate on a schedule restarted at 0 would silently train something else
        raise ValueError("checkpoint carries optimizer_state_dict but no scheduler_state_dict: not a resume checkpoint of "
                         "train.py:341-349; pass the model weights alone (a --pretrain load) or a complete checkpoint")
    if opt.get("state"):
        idx = _param_index(model)
        st = opt["state"]
        step = 0
        for n, (o, k, shp) in zip(model._names, model._slots):
            s = st.get(idx[n]) or st.get(str(idx[n]))
            if s is None:
                continue
            train_step.exp_avg[o:o + k].copy_(s["exp_avg"].reshape(-1).to(train_step.exp_avg))
            train_step.exp_avg_sq[o:o + k].copy_(s["exp_avg_sq"].reshape(-1).to(train_step.exp_avg_sq))
            step = int(float(s["step"]))
        train_step.step_count = step
    else:
        # a resume checkpoint written before the first optimizer step: optimizer.load_state_dict (train.py:101) would leave an EMPTY
        # Adam state, i.e. zero moments and step 0 -- not whatever this TrainStep accumulated before the load
        train_step.exp_avg.zero_()
        train_step.exp_avg_sq.zero_()
        train_step.step_count = 0
    pg = opt.get("param_groups")
    if pg:
        train_step.set_lr(pg[0]["lr"])
    train_step.scheduler_last_epoch = int(sch.get("last_epoch", 0))
    return int(ckpt.get("epoch", 0)) + 1
