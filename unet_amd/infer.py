"""Sliding-window inference over long recordings (phase_five/sliding_window.py:127-278) on the GPU.

The reference walks the recording one window at a time: one host->device copy, one batch-1 forward
and one device->host copy per window, then overlap-adds in numpy (sliding_window.py:163-253).  Here
the recording is uploaded ONCE, windows are strided views of it (never gathered on the host), they
run through the model in batches, and the overlap-average is a gather kernel on the device
(``unet_overlap_add``).  With several ranks (one process per GPU) each takes a contiguous range of
windows (``windows.shard_windows``), accumulates raw sums for its span, and one all-reduce (SUM) of
``output`` merges the seams before the division by the per-sample window count -- the count is a pure
function of the geometry, so no counts are exchanged.

Same results as the reference's ``continuous_inference``: uncovered tail samples stay 0, counts are
clamped at 1, modes "average" | "last" | "first".
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .dp import group_rank, group_size
from .ops import _dt, _p, _stream
from .windows import n_windows_inference, shard_windows

_MODES = {"average": 0, "last": 1, "first": 2}


def overlap_add(pred: torch.Tensor, first_window: int, total_windows: int, stride: int, out: torch.Tensor,
                accumulate: bool, finalize: bool, mode: str = "average", counts: Optional[torch.Tensor] = None,
                window_size: Optional[int] = None):
    """pred [nw, C, W] (bf16/fp32, contiguous) -> out [C, n_samples] fp32 (see include/unet_hip.h).
    ``pred=None`` with ``window_size`` given is the finalize-only call."""
    if pred is None and window_size is None:
        raise ValueError("a finalize-only call needs window_size")
    nw, c, w = pred.shape if pred is not None else (0, out.shape[0], window_size)
    if pred is not None:
        pred = pred.contiguous()
    check(_lib.load().unet_overlap_add(_dt(pred) if pred is not None else _lib.F32, _p(pred), nw, first_window,
                                       total_windows, c, w, stride, out.shape[1], _p(out), _p(counts), int(accumulate),
                                       int(finalize), _MODES[mode], _stream()), "unet_overlap_add")


def plan_windows(n_samples: int, window_size: int, stride: int, process_group=None):
    """Host side of the sharding: (total windows of the recording, this rank's contiguous share of them, number of
    ranks sharing the recording).  ``dp.NO_GROUP`` (or no process group at all) = one rank owns every window."""
    total = n_windows_inference(n_samples, window_size, stride)
    world = group_size(process_group)              # 1 without torch.distributed and under dp.NO_GROUP
    rank = group_rank(process_group)
    return total, shard_windows(total, world, window_size, stride)[rank], world


def merge_seams(out: torch.Tensor, process_group=None):
    """One all-reduce (SUM) of the raw overlap sums across the ranks that share the recording; nothing to do for a
    rank that owns it alone."""
    if group_size(process_group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=process_group)
    return out


@torch.no_grad()
def continuous_inference(model: torch.nn.Module, signal, window_size: int, stride: int, device="cuda",
                         overlap_mode: str = "average", batch_windows: int = 64, process_group=None,
                         return_tensor: bool = False, model_kwargs: Optional[dict] = None, graphs=False):
    """Drop-in for phase_five/sliding_window.py:256-278: ``signal`` is ``[n_channels, n_samples]``
    (numpy or tensor); returns the continuous prediction ``[n_out_channels, n_samples]`` (numpy float32
    unless ``return_tensor``).  ``model(x, **model_kwargs)`` is called on ``[B, C, W]`` batches in eval mode; a tensor
    in ``model_kwargs`` whose first dimension is the recording's window count is sliced along with the batch.

    ``process_group``: the ranks that SHARE this recording (default: all of them, each takes a contiguous range of
    windows and the seams are merged by one all-reduce).  ``dp.NO_GROUP`` = this rank alone processes the whole
    recording although ``torch.distributed`` is up: inference inside a fold-parallel LOSO job, where the other ranks
    run other folds' models and must not be met in a collective.

    ``graphs``: True, or a ``graphs.GraphedInference`` built over ``model`` (pass one to reuse its captured graphs from
    recording to recording): the full batches are replayed as one HIP graph each instead of ~150 launches; a last,
    shorter batch runs eagerly."""
    if overlap_mode not in _MODES:
        raise ValueError(f"Unknown overlap_mode: {overlap_mode}. Available: average, last, first")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("unet_amd.infer runs on the GPU only; there is no CPU path")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if any(t.device != dev for t in list(model.parameters()) + list(model.buffers())):
        model = model.to(dev)          # (an unconditional .to() would drop the packed-weight cache on every call)
    model.eval()
    sig = torch.as_tensor(signal)
    if sig.dtype not in (torch.float32, torch.bfloat16):
        sig = sig.float()
    sig = sig.to(dev, non_blocking=True).contiguous()          # ONE upload of the whole recording
    n_ch, n_samples = sig.shape
    total, shard, world = plan_windows(n_samples, window_size, stride, process_group)
    out = None
    windows = sig.unfold(1, window_size, stride).permute(1, 0, 2) if total > 0 else None   # [total, C, W] view
    kw = model_kwargs or {}
    graphed = None
    if graphs:
        from .graphs import GraphedInference
        graphed = graphs if isinstance(graphs, GraphedInference) else GraphedInference(model)
        if graphed.model is not model:
            raise ValueError("the GraphedInference passed as graphs= was built over another model")
    done = 0
    while done < shard.n_windows:
        nb = min(batch_windows, shard.n_windows - done)
        first = shard.first_window + done
        # a tensor keyword with one entry per window of the recording (odor_ids, cond_emb) is sliced with the batch
        kwb = {k: (v[first:first + nb] if (torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == total) else v)
               for k, v in kw.items()}
        pred = (graphed if (graphed is not None and nb == batch_windows) else model)(windows[first:first + nb], **kwb)
        if isinstance(pred, tuple):
            pred = pred[0]
        if out is None:
            out = torch.zeros((pred.shape[1], n_samples), dtype=torch.float32, device=dev)
        overlap_add(pred, first, total, stride, out, accumulate=True, finalize=False, mode=overlap_mode)
        done += nb
    if out is None:
        n_out = getattr(model, "out_channels", n_ch)
        out = torch.zeros((n_out, n_samples), dtype=torch.float32, device=dev)
    merge_seams(out, process_group)
    if overlap_mode == "average" and total > 0:
        overlap_add(None, 0, total, stride, out, accumulate=True, finalize=True, window_size=window_size)
    return out if return_tensor else out.cpu().numpy()


def continuous_r2(pred, target) -> float:
    """phase_five/runner.py:281-285: 1 - SS_res / (SS_tot + 1e-8) over the whole recording (numpy)."""
    pred, target = np.asarray(pred, dtype=np.float64), np.asarray(target, dtype=np.float64)
    ss_res = ((target - pred) ** 2).sum()
    ss_tot = ((target - target.mean()) ** 2).sum()
    return float(1 - ss_res / (ss_tot + 1e-8))
