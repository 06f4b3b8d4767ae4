"""Checkpoint wire format of the reference (train.py:341-349 save, :85-108 resume, test.py:64-70 load).

    torch.save({'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'stat_dict'}, path)

* model keys carry DataParallel's ``module.`` prefix (train.py:73);
* optimizer_state_dict is torch.optim.Adam's: per-parameter ``step``/``exp_avg``/``exp_avg_sq`` indexed in
  ``model.parameters()`` order with the 4 frozen MeanShift tensors included (the reference passes every
  parameter to Adam, train.py:81; frozen ones simply never get state);
* scheduler_state_dict is CosineAnnealingLR's (only ``last_epoch`` matters for the closed form used here).

The fused step driver keeps Adam's moments in two flat buffers; these helpers convert both ways so a run can
be resumed by either implementation.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


def _param_index(model) -> Dict[str, int]:
    return {n: i for i, (n, _) in enumerate(model.named_parameters())}


def export_checkpoint(model, train_step=None, epoch: int = 0, stat_dict: Optional[dict] = None,
                      lr0: float = 1e-4, eta_min: float = 1e-6, t_max: float = 200.0) -> dict:
    sd = {"module." + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    out = {"epoch": int(epoch), "model_state_dict": sd, "stat_dict": stat_dict or {}}
    if train_step is not None:
        idx = _param_index(model)
        state = {}
        for n, (o, k, shp) in zip(model._names, model._slots):
            state[idx[n]] = {"step": torch.tensor(float(train_step.step_count)),
                             "exp_avg": train_step.exp_avg[o:o + k].view(shp).detach().cpu().clone(),
                             "exp_avg_sq": train_step.exp_avg_sq[o:o + k].view(shp).detach().cpu().clone()}
        group = {"lr": train_step.lr, "betas": train_step.betas, "eps": train_step.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "initial_lr": lr0, "params": list(range(len(idx)))}
        out["optimizer_state_dict"] = {"state": state if train_step.step_count > 0 else {}, "param_groups": [group]}
        out["scheduler_state_dict"] = {"T_max": t_max, "eta_min": eta_min, "base_lrs": [lr0], "last_epoch": int(epoch),
                                       "_step_count": int(epoch) + 1, "_last_lr": [train_step.lr]}
    return out


def import_checkpoint(ckpt: dict, model, train_step=None) -> int:
    """Load a reference-format checkpoint; returns the epoch to continue from (train.py:97-100)."""
    torch.nn.Module.load_state_dict  # noqa: B018  (the model's own lenient loader is used below)
    model.load_state_dict(ckpt["model_state_dict"], strict=True)
    if train_step is not None and ckpt.get("optimizer_state_dict", {}).get("state"):
        idx = _param_index(model)
        st = ckpt["optimizer_state_dict"]["state"]
        step = 0
        for n, (o, k, shp) in zip(model._names, model._slots):
            s = st.get(idx[n]) or st.get(str(idx[n]))
            if s is None:
                continue
            train_step.exp_avg[o:o + k].copy_(s["exp_avg"].reshape(-1).to(train_step.exp_avg))
            train_step.exp_avg_sq[o:o + k].copy_(s["exp_avg_sq"].reshape(-1).to(train_step.exp_avg_sq))
            step = int(float(s["step"]))
        train_step.step_count = step
        pg = ckpt["optimizer_state_dict"].get("param_groups")
        if pg:
            train_step.set_lr(pg[0]["lr"])
    return int(ckpt.get("epoch", 0)) + 1
