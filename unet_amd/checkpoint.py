"""Checkpoint, best-model / early-stop and results-JSON protocol of the reference's ``train.py``.

What a caller of the reference relies on:

* ``save_checkpoint`` / ``load_checkpoint``  -- train.py:498-568, 571-701 (the non-FSDP branch: the model
  in scope is data-parallel).  The file is a plain dict ``{"epoch", "model", "optimizer", "cond_mode",
  ["config"], ["split_info"], ["reverse_model"]}``; ``"model"`` is the state dict of the UNWRAPPED module,
  whose keys / shapes are the reference's (tests/golden/init_checksums.npz), so files move both ways.
* ``EarlyStopping``                          -- train.py:2355-2357, 2511-2543, 2744-2757: lower val loss is
  better, strict ``<``; patience counts epochs without improvement; the val loss is averaged over ranks and
  the stop decision is rank 0's, broadcast, so that all ranks leave the loop together.
* ``history_entry`` / ``results_json``        -- train.py:2580-2584, 3290-3299, 4571-4668: the per-epoch history
  rows and the JSON that ``LOSO/runner.py`` and ``phase_three`` read back (``test_avg_r2`` ... keys).

Loading uses ``torch.load(..., weights_only=True)``: a checkpoint holds tensors, numbers, strings, lists and
dicts only; anything else in a file is refused rather than unpickled.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist


def _is_primary(process_group=None) -> bool:
    from .dp import group_rank
    return group_rank(process_group) == 0        # NO_GROUP (a fold job of run_loso): every rank is its own primary


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def _tolist(v):
    return v.tolist() if hasattr(v, "tolist") else list(v)


def save_checkpoint(model: torch.nn.Module, optimizer: torch.optim.Optimizer, epoch: int, path,
                    reverse_model: Optional[torch.nn.Module] = None, config: Optional[Dict[str, Any]] = None,
                    data: Optional[Dict[str, Any]] = None, process_group=None) -> None:
    """train.py:652-701.  Only rank 0 writes.  ``config`` entries whose key starts with ``_`` (live objects
    such as ``_session_matcher``) are left out of the saved config: they are not data."""
    if not _is_primary(process_group):
        return
    checkpoint: Dict[str, Any] = {
        "epoch": epoch,
        "model": _unwrap(model).state_dict(),
        "optimizer": optimizer.state_dict(),
        "cond_mode": config.get("cond_mode", "unknown") if config else "unknown",
    }
    if config is not None:
        checkpoint["config"] = {k: v for k, v in config.items() if not str(k).startswith("_")}
    if data is not None:
        split_info = dict(data["split_info"]) if "split_info" in data else {}
        for key in ("val_idx", "test_idx", "train_idx"):
            if key in data:
                split_info[key] = _tolist(data[key])
        checkpoint["split_info"] = split_info
    if reverse_model is not None:
        checkpoint["reverse_model"] = _unwrap(reverse_model).state_dict()
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(checkpoint, path)


def load_checkpoint(path, model: torch.nn.Module, reverse_model: Optional[torch.nn.Module] = None,
                    expected_cond_mode: Optional[str] = None, optimizer: Optional[torch.optim.Optimizer] = None) -> str:
    """train.py:498-568: strict ``load_state_dict`` into the unwrapped module; returns the file's cond_mode.
    ``optimizer`` (not in the reference's signature) also restores the optimizer state for a resume."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    checkpoint = torch.load(path, map_location="cpu", weights_only=True)
    checkpoint_cond_mode = checkpoint.get("cond_mode", "unknown")
    if expected_cond_mode is not None and checkpoint_cond_mode != "unknown" and checkpoint_cond_mode != expected_cond_mode:
        raise ValueError(
            f"Checkpoint cond_mode mismatch!\n"
            f"  Checkpoint has: {checkpoint_cond_mode}\n"
            f"  Model expects:  {expected_cond_mode}\n"
            f"  The checkpoint was trained with a different conditioning architecture.\n"
            f"  For stage2 spectral shift experiments, use a checkpoint with matching cond_mode.\n"
            f"  Either:\n"
            f"    1. Set --cond-mode {checkpoint_cond_mode} to match the checkpoint, or\n"
            f"    2. Use a different checkpoint (set BEST_CHECKPOINT=/path/to/matching/checkpoint.pt)")
    _unwrap(model).load_state_dict(checkpoint["model"])
    if reverse_model is not None and "reverse_model" in checkpoint:
        _unwrap(reverse_model).load_state_dict(checkpoint["reverse_model"])
    if optimizer is not None and "optimizer" in checkpoint:
        optimizer.load_state_dict(checkpoint["optimizer"])
    return checkpoint_cond_mode


def checkpoint_name(config: Optional[Dict[str, Any]]) -> str:
    """train.py:2530-2531: ``{checkpoint_prefix}_best_model.pt`` (phase-three folds) or ``best_model.pt``."""
    prefix = (config or {}).get("checkpoint_prefix")
    return f"{prefix}_best_model.pt" if prefix else "best_model.pt"


class EarlyStopping:
    """Best-epoch tracking and the all-rank stop decision of the reference's epoch loop."""

    def __init__(self, patience: int = 8, process_group=None):           # train.py:2330 default 8
        self.patience = patience
        self.pg = process_group
        self.best_val_loss = float("inf")                                  # train.py:2355-2357
        self.best_epoch = 0
        self.patience_counter = 0

    def sync_val_loss(self, val_loss: float, device) -> float:
        """train.py:2516-2520: mean over ranks."""
        from .dp import group_active
        if group_active(self.pg):
            t = torch.tensor(float(val_loss), device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)         # gloo has no AVG; SUM / world is the same value
            val_loss = float(t.item()) / dist.get_world_size(self.pg)
        return float(val_loss)

    def update(self, val_loss: float, epoch: int) -> bool:
        """train.py:2522-2543: True when this epoch is the new best (the caller then saves the checkpoint)."""
        is_best = val_loss < self.best_val_loss
        if is_best:
            self.best_val_loss = val_loss
            self.best_epoch = epoch
            self.patience_counter = 0
        else:
            self.patience_counter += 1
        return is_best

    def should_stop(self, device=None) -> bool:
        """train.py:2744-2750: rank 0's decision, broadcast."""
        from .dp import group_active
        stop = self.patience_counter >= self.patience
        if group_active(self.pg):
            t = torch.tensor([1 if stop else 0], dtype=torch.int32, device=device)
            dist.broadcast(t, src=0, group=self.pg)
            stop = int(t.item()) == 1
        return stop


def history_entry(epoch: int, train_metrics: Dict[str, Any], val_metrics: Dict[str, Any],
                  per_session_metrics: Optional[Dict[str, Dict[str, Any]]] = None) -> Dict[str, Any]:
    """train.py:2580-2583."""
    entry = {"epoch": epoch, **train_metrics, **{f"val_{k}": v for k, v in val_metrics.items()}}
    for sess_name, sess_m in (per_session_metrics or {}).items():
        for k, v in sess_m.items():
            entry[f"val_{sess_name}_{k}"] = v
    return entry


def results_json(results: Dict[str, Any], arch: str = "condunet", fold: Optional[int] = None) -> Dict[str, Any]:
    """train.py:4571-4668: the ``--output-results-file`` payload from ``train()``'s return value
    (``best_val_loss, best_epoch, test_metrics, history, n_parameters, total_time``)."""
    history: List[Dict[str, Any]] = results.get("history", [])
    train_losses = [h.get("loss", 0.0) for h in history]
    val_losses = [h.get("val_loss", 0.0) for h in history]
    val_r2s = [h.get("val_r2", 0.0) for h in history]
    val_corrs = [h.get("val_corr", 0.0) for h in history]
    val_maes = [h.get("val_mae", 0.0) for h in history]
    best_epoch = results["best_epoch"]
    pick = lambda xs: xs[best_epoch - 1] if best_epoch > 0 and len(xs) >= best_epoch else 0.0
    per_session_r2, per_session_corr, per_session_loss = {}, {}, {}
    if history:
        best_entry = history[best_epoch - 1] if best_epoch <= len(history) else history[-1]
        for key, value in best_entry.items():
            if key.startswith("val_") and "_r2" in key and key != "val_r2":
                if key[4:-3]:
                    per_session_r2[key[4:-3]] = value
            elif key.startswith("val_") and "_corr" in key and key != "val_corr":
                if key[4:-5]:
                    per_session_corr[key[4:-5]] = value
            elif key.startswith("val_") and "_loss" in key and key != "val_loss":
                if key[4:-5]:
                    per_session_loss[key[4:-5]] = value
    test_metrics = results.get("test_metrics", {})
    return {
        "architecture": arch,
        "fold": fold if fold is not None else 0,
        "best_val_r2": pick(val_r2s),
        "best_val_mae": pick(val_maes),
        "best_val_corr": pick(val_corrs),
        "best_val_loss": results["best_val_loss"],
        "best_epoch": best_epoch,
        "train_losses": train_losses,
        "val_losses": val_losses,
        "val_r2s": val_r2s,
        "val_corrs": val_corrs,
        "per_session_r2": per_session_r2,
        "per_session_corr": per_session_corr,
        "per_session_loss": per_session_loss,
        "per_session_test_results": test_metrics.get("per_session_test_results", []),
        "test_avg_r2": test_metrics.get("test_avg_r2") or test_metrics.get("r2"),
        "test_avg_corr": test_metrics.get("test_avg_corr") or test_metrics.get("corr"),
        "test_avg_delta": test_metrics.get("test_avg_delta") or test_metrics.get("loss"),
        "total_time": results.get("total_time", 0.0),
        "epochs_trained": len(history),
        "n_parameters": results.get("n_parameters", 0),
        "completed_successfully": True,
        "gate_mean": history[-1].get("gate_mean", None) if history else None,
        "matched_corr": test_metrics.get("matched_corr"),
        "matched_r2": test_metrics.get("matched_r2"),
        "matched_nrmse": test_metrics.get("matched_nrmse"),
        "baseline_corr": test_metrics.get("baseline_corr"),
        "baseline_r2": test_metrics.get("baseline_r2"),
    }


def write_results_json(results: Dict[str, Any], path, arch: str = "condunet", fold: Optional[int] = None) -> None:
    """train.py:4669-4672 (``json.dump(..., indent=2, default=str)``)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        json.dump(results_json(results, arch, fold), f, indent=2, default=str)
