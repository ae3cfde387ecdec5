"""Data-parallel gradient exchange: bucketed all-reduce overlapped with backward.

Replaces the reference's ``DDP(model, broadcast_buffers=False, ...)`` wrap (train.py:2154-2202)
and its per-tensor ``sync_gradients_manual`` (train.py:738-758) for one-process-per-GPU runs over
RCCL/xGMI (backend "nccl" on ROCm) -- and over gloo on CPU for the tests.

Design (MI355X-first, not DDP's reducer re-typed):
  * gradients are exchanged in a few flat fp32 buckets filled in reverse parameter-registration
    order (the order backward produces them).  When the last gradient of a bucket has been produced
    (post-accumulate hook) the bucket is packed with ONE multi-tensor copy, every ``.grad`` is
    re-pointed at its slice of the flat buffer (so the optimizer reads communication memory directly,
    no scatter copy after the reduce), and the all-reduce is issued asynchronously: RCCL moves
    bucket k over xGMI while the HIP kernels compute the gradients of bucket k+1;
  * ``zero_grad()`` only drops the ``.grad`` references (set-to-none): autograd then *assigns* fresh
    gradients instead of launching one tiny accumulate kernel per parameter (145 of them at config B);
  * BatchNorm buffers are never synchronised (per-rank statistics, as in the reference:
    ``broadcast_buffers=False``); rank 0's buffers are what a checkpoint holds;
  * xGMI is point-to-point (7 links per GPU), so fewer, larger buckets are preferred: default
    64 MiB (the 168 MB of config-B gradients travel as 3 messages plus a <= 4 MiB tail, see ``tail_bytes``);
  * with one process there is nothing to exchange and the class is a no-op around autograd.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


NO_GROUP = False
"""Pass as ``process_group`` to say "this job is NOT data-parallel", even though ``torch.distributed`` is
initialised: fold-parallel LOSO runs a different (fold, seed) job on every rank (``trainer.run_loso``), and such a
job must not broadcast weights, average gradients or share early-stop decisions with the other ranks."""


def _pad4(n: int) -> int:
    return (n + 3) & ~3


def group_active(process_group=None) -> bool:
    return process_group is not NO_GROUP and dist.is_available() and dist.is_initialized()


def group_size(process_group=None) -> int:
    return dist.get_world_size(process_group) if group_active(process_group) else 1


def group_rank(process_group=None) -> int:
    return dist.get_rank(process_group) if group_active(process_group) else 0


class BucketedGradAllReduce:
    def __init__(self, params, bucket_bytes: int = 64 << 20, process_group=None, average: bool = True,
                 tail_bytes: int = 4 << 20, force: bool = False, late_params=None):
        """``tail_bytes``: the parameters registered first (embedding, ``inc``) produce their gradients last, and the
        bucket that holds them is launched when backward has nothing left to overlap it with; they get a bucket of
        their own of at most this size, so the exposed message is a few MiB instead of a full bucket.
        ``late_params``: parameters whose gradients are known to arrive at the very END of backward wherever they are
        registered (``CondUNet1D.late_grad_parameters()``: the FiLM embedding projections of every block, computed by
        one hoisted node in front of the first block).  They join the tail bucket; left in their blocks' buckets they
        would hold every bucket back until backward is over.  ``None``: ask the module (if ``params`` is one) or
        nobody."""
        if isinstance(params, torch.nn.Module):
            if late_params is None and hasattr(params, "late_grad_parameters"):
                late_params = params.late_grad_parameters()
            params = params.parameters()
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        late = {id(p) for p in (late_params or ())}
        self.pg = process_group
        self.average = average
        self.world = group_size(process_group)
        self.force = force and dist.is_available() and dist.is_initialized()   # issue the collectives in a one-rank group too
        self.buckets: List[torch.Tensor] = []
        self._groups: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._bucket_of = {}
        self._handles = []
        self._hooks = []
        for p in self.params:                   # all of them, the tail bucket's included
            if p.dtype != torch.float32:
                raise TypeError("gradient buckets are fp32 (master weights are fp32, as in the reference)")
        n_tail, tail_elems = 0, 0
        for p in self.params:
            if (tail_elems + p.numel()) * 4 > tail_bytes:
                break
            n_tail += 1
            tail_elems += p.numel()
        if n_tail == len(self.params):
            n_tail = 0                          # everything fits one small bucket: no separate tail
        tail = list(self.params[:n_tail]) + [p for p in self.params[n_tail:] if id(p) in late]
        in_tail = {id(p) for p in tail}
        cur, cur_elems = [], 0
        for p in reversed(self.params):
            if id(p) in in_tail:
                continue
            if cur and (cur_elems + p.numel()) * 4 > bucket_bytes:
                self._groups.append(cur)
                cur, cur_elems = [], 0
            cur.append(p)
            cur_elems += p.numel()
        if cur:
            self._groups.append(cur)
        if tail:
            self._groups.append(list(reversed(tail)))
        for b, group in enumerate(self._groups):
            # every view starts on a 16-byte boundary (the padding stays zero on every rank): the optimizer's float4
            # path (optim.ClipAdamW) and RCCL both see aligned tensors
            flat = torch.zeros(sum(_pad4(p.numel()) for p in group), dtype=torch.float32, device=group[0].device)
            views, off = [], 0
            for p in group:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += _pad4(p.numel())
                self._bucket_of[p] = b
            self.buckets.append(flat)
            self._views.append(views)
        self._expected = [len(g) for g in self._groups]
        self._pending = list(self._expected)
        self._launched = [False] * len(self._groups)
        if self.world > 1 or self.force:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- per-step protocol --------------------------------------------------------------------
    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self._pending = list(self._expected)
        self._launched = [False] * len(self._groups)
        self._handles = []

    def _on_grad(self, p: torch.nn.Parameter):
        b = self._bucket_of[p]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b: int):
        if (self.world == 1 and not self.force) or self._launched[b]:
            return
        self._launched[b] = True
        flat, group, views = self.buckets[b], self._groups[b], self._views[b]
        have = [(v, p.grad) for v, p in zip(views, group) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        with torch.no_grad():
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            for v, p in zip(views, group):
                if p.grad is None:
                    v.zero_()  # parameter unused this step: contributes zeros (find_unused_parameters semantics)
                p.grad = v
        if self.average and _has_avg(self.pg):
            h = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            self._handles.append((h, None))
        else:  # gloo has no AVG
            h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._handles.append((h, flat if self.average else None))

    def finish(self):
        """Call after backward: flushes buckets that did not fill (some parameter got no gradient this
        step), then waits for every bucket."""
        for b in range(len(self._groups)):
            self._launch(b)
        for h, scale in self._handles:
            h.wait()
            if scale is not None:
                scale.div_(self.world)
        self._handles = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def _has_avg(process_group=None) -> bool:
    """ReduceOp.AVG exists in RCCL (backend "nccl") only; gloo gets SUM followed by a division."""
    return str(dist.get_backend(process_group)).lower() == "nccl"


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """What DDP's constructor does once: make every rank start from rank `src`'s parameters and
    buffers (train.py seeds ranks differently, `seed + rank`, train.py:763-764)."""
    if group_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)


def coalesced_grad_mean(params, process_group=None):
    """Mean of the gradients of a SMALL module across ranks in ONE message -- the reference's
    ``sync_gradients_manual`` (train.py:738-758) issues one all-reduce per parameter (28 tiny collectives for the
    conditioning encoder); xGMI collectives are latency-bound at that size, so the gradients are flattened into
    one fp32 buffer, reduced once and copied back.  No-op without an initialised process group or with ``NO_GROUP``."""
    world = group_size(process_group)
    if world == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    # every rank sends the same layout: a parameter without a gradient on this rank contributes zeros (and gets the
    # mean of the others), so ranks whose gradients differ in presence cannot mismatch message sizes
    flat = torch.cat([(p.grad.detach().reshape(-1).float() if p.grad is not None
                       else torch.zeros(p.numel(), dtype=torch.float32, device=p.device)) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
    flat.div_(world)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).to(p.dtype).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
