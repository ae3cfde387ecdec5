"""HIP-graph capture of the training step.

One config-B training step launches ~1000 kernels; issued one by one from Python the host needs
longer to enqueue them than the MI355X needs to run them.  Windows have a fixed shape, so the whole
step is captured once into a HIP graph (``torch.cuda.CUDAGraph`` = ``hipGraph`` on ROCm) and replayed
with one launch per step.  The HIP kernels of libunet_hip.so are capture-safe by construction: they
never allocate, never synchronise, and launch only on the stream they are given (torch's current
stream, which is the capture stream during capture).

Single process:   [ zero-grad | forward | loss | backward | clip | optimizer ]      = one graph
Data parallel:    [ forward | loss | backward ] graph  ->  bucket pack + RCCL all-reduce (eager,
                  outside any graph)  ->  [ clip | optimizer ] graph
                  (``overlap=False``: no overlap of communication with backward)
Data parallel, ``overlap=True``: ONE graph that contains the collectives.  The step is captured with the
                  ``BucketedGradAllReduce`` hooks armed: when the last gradient of a bucket is produced inside the
                  captured backward, the hook packs the bucket and issues its all-reduce asynchronously (RCCL runs it
                  on the process group's own stream; the capture records that as a fork from the backward stream), and
                  the wait is captured only in front of the clip.  In the replayed graph the collective of bucket k
                  therefore depends on bucket k's gradients alone and runs beside the kernels of bucket k+1 -- graph
                  replay AND overlap.  RCCL collectives are capturable (torch.distributed keeps captured work off the
                  watchdog); if the capture is refused the constructor raises and the caller falls back to the eager
                  hook-driven path (``bench.py`` does).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .dp import _pad4, BucketedGradAllReduce, _has_avg, group_size


def _capture_mode() -> str:
    """With a process group up, its watchdog thread polls the events of earlier collectives (the warm-up steps', the
    start-up broadcast's) while this thread captures: under the default "global" mode such a call from ANOTHER thread
    invalidates the capture (seen as an intermittent abort inside the captured backward).  "thread_local" confines the
    capture's restrictions to the capturing thread, which is the only one that issues work here.
    ``UNET_CAPTURE_MODE`` overrides the choice (scripts/diag_capture_mode.py uses it to show the global-mode failure)."""
    forced = os.environ.get("UNET_CAPTURE_MODE")
    if forced in ("global", "thread_local", "relaxed"):
        return forced
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


class GraphedTrainStep:
    """Capture ``loss = loss_fn(model(*inputs), *targets); loss.backward(); clip; optimizer.step()``.

    ``inputs`` / ``targets`` given at construction define the static shapes; ``__call__`` copies new
    batches into the static buffers and replays.  Returns the (static) loss tensor.
    """

    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer: torch.optim.Optimizer,
                 inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor], clip_norm: Optional[float] = None,
                 warmup: int = 3, process_group=None, bucket_bytes: int = 64 << 20,
                 forward: Optional[Callable] = None, extra_modules: Sequence[torch.nn.Module] = (),
                 overlap: bool = False, force_collectives: bool = False):
        """``extra_modules``: small modules trained in the same step (the conditioning encoder).  As in the
        reference (train.py:1553-1558) each gets its own clip, applied per rank BEFORE its gradients are averaged,
        whereas the model's clip follows the all-reduce."""
        self.model, self.loss_fn, self.opt, self.clip = model, loss_fn, optimizer, clip_norm
        self.model_params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.extra_params = [[p for p in m.parameters() if p.requires_grad] for m in extra_modules]
        self.params: List[torch.nn.Parameter] = self.model_params + [p for grp in self.extra_params for p in grp]
        self.static_in = [t.clone() for t in inputs]
        self.static_tg = [t.clone() for t in targets]
        self.pg = process_group
        self.world = group_size(process_group)
        self.overlap = overlap and (self.world > 1 or force_collectives)
        if self.overlap:
            self._init_overlapped(forward, bucket_bytes, warmup, force_collectives)
            return

        def fwd_bwd():
            out = (forward or model)(*self.static_in)
            loss = loss_fn(out, *self.static_tg)
            loss.backward()
            if self.clip is not None:
                for grp in self.extra_params:
                    torch.nn.utils.clip_grad_norm_(grp, self.clip)
            return loss

        def update():
            if self.clip is not None and not getattr(self.opt, "clips_gradients", False):   # optim.ClipAdamW clips inside step()
                torch.nn.utils.clip_grad_norm_(self.model_params, self.clip)
            self.opt.step()

        # warm-up on a side stream (allocator pools, lazy kernel attributes, optimizer state)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self.opt.zero_grad(set_to_none=True)
                fwd_bwd()
                if self.world > 1:
                    self._reduce_eager_setup(bucket_bytes)
                    self._allreduce()
                update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        if self.world == 1:
            with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
                self.loss = fwd_bwd()
                update()
            self.graph_update = None
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
                self.loss = fwd_bwd()
            self.static_grads = [p.grad for p in self.params]
            for p, v in zip(self.params, self._views):
                p.grad = v                                   # the optimizer reads the reduced buckets
            self.graph_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_update, capture_error_mode=_capture_mode()):
                update()

    # -- data parallel, collectives inside the graph ----------------------------------------------
    def _init_overlapped(self, forward, bucket_bytes: int, warmup: int, force: bool):
        """``force``: arm the hooks and issue the collectives also in a one-rank group (how the capture path is
        exercised on a single GPU: tests/test_gpu_dp_graph.py)."""
        if self.extra_params:
            raise NotImplementedError("overlap=True with extra modules: their per-rank clip precedes their own small "
                                      "all-reduce (train.py:1553-1558); use overlap=False or the eager path")
        model, loss_fn = self.model, self.loss_fn
        late = self.model.late_grad_parameters() if hasattr(self.model, "late_grad_parameters") else None
        self.reducer = BucketedGradAllReduce(self.model_params, bucket_bytes, self.pg, force=force, late_params=late)

        def full_step():
            self.reducer.zero_grad()
            out = (forward or model)(*self.static_in)
            loss = loss_fn(out, *self.static_tg)
            loss.backward()                      # bucket hooks fire in here: pack + async all-reduce per bucket
            self.reducer.finish()                # the only wait on the collectives
            if self.clip is not None and not getattr(self.opt, "clips_gradients", False):
                torch.nn.utils.clip_grad_norm_(self.model_params, self.clip)
            self.opt.step()
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                full_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.loss = full_step()
        self.graph_update = None

    # -- data-parallel pieces (eager, outside the graphs) ---------------------------------------
    def _reduce_eager_setup(self, bucket_bytes: int):
        if hasattr(self, "_flat"):
            return
        self._flat, self._views = [], []
        cur, cur_elems, groups = [], 0, []
        for p in reversed(self.params):
            if cur and (cur_elems + p.numel()) * 4 > bucket_bytes:
                groups.append(cur)
                cur, cur_elems = [], 0
            cur.append(p)
            cur_elems += p.numel()
        if cur:
            groups.append(cur)
        view_of = {}
        for grp in groups:
            flat = torch.zeros(sum(_pad4(p.numel()) for p in grp), dtype=torch.float32, device=grp[0].device)
            off = 0
            for p in grp:
                view_of[p] = flat[off:off + p.numel()].view_as(p)     # 16-byte aligned views, zero padding
                off += _pad4(p.numel())
            self._flat.append(flat)
        self._views = [view_of[p] for p in self.params]

    def _allreduce(self, grads: Optional[List[torch.Tensor]] = None):
        grads = grads if grads is not None else [p.grad for p in self.params]
        src = [(v, g) for v, g in zip(self._views, grads) if g is not None]
        with torch.no_grad():
            torch._foreach_copy_([v for v, _ in src], [g for _, g in src])
            for v, g in zip(self._views, grads):
                if g is None:
                    v.zero_()
        handles = []
        for flat in self._flat:
            if _has_avg(self.pg):
                handles.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
            else:
                handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), flat))
        for h, scale in handles:
            h.wait()
            if scale is not None:
                scale.div_(self.world)
        if grads is not self.__dict__.get("static_grads"):
            for p, v in zip(self.params, self._views):
                p.grad = v

    # -- per step ---------------------------------------------------------------------------------
    def __call__(self, inputs: Optional[Sequence[torch.Tensor]] = None, targets: Optional[Sequence[torch.Tensor]] = None):
        for static, new in ((self.static_in, inputs), (self.static_tg, targets)):
            if new is None:
                continue
            if len(new) != len(static):
                raise ValueError(f"captured step takes {len(static)} tensors here, got {len(new)}")
            for s, t in zip(static, new):
                if t.shape != s.shape:       # copy_ would broadcast a batch of 1 into every row without a word
                    raise ValueError(f"captured step was recorded for batches of shape {tuple(s.shape)}, got "
                                     f"{tuple(t.shape)}: run other shapes (a last partial batch) through an eager step")
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        if self.graph_update is not None:
            self._allreduce(self.static_grads)
            self.graph_update.replay()
        return self.loss


class GraphedInference:
    """``model(x, **kwargs)`` in eval mode as ONE HIP graph per input shape, for serving loops and sliding-window
    inference at a fixed batch (phase_five/benchmark.py:189-278 measures exactly this call at batch 1 / 8 / 32, where
    the eager forward is launch-bound: ~150 launches whose enqueue costs the host more than the GPU needs to run them).

    The first call with a new shape signature runs the model eagerly twice (which also fills the packed-weight cache),
    captures one forward with the cached weight images in place (``ops.pack_cache_in_capture``: no pack launches in
    the graph) and keeps static input / output buffers; later calls copy the inputs in and replay.  The returned
    tensor is the STATIC output buffer: it is overwritten by the next call with the same shapes (clone it to keep it).
    The graphs read the weight images of capture time; every call checks the parameters' version counters and storage
    addresses and the pack cache's generation (``ops.clear_pack_cache()`` runs on ``model.train()``,
    ``load_state_dict`` and ``.to()``) and drops all graphs if one moved.  That covers training in between: the
    optimizer kernels bump no version counter, but a model cannot be trained without ``model.train()``.  Weights
    written behind all of these (``p.data`` arithmetic in eval mode) need an explicit ``reset()``."""

    def __init__(self, model: torch.nn.Module, warmup: int = 2, max_graphs: int = 8):
        self.model, self.warmup, self.max_graphs = model, warmup, max_graphs
        self._graphs = {}
        self._stamp = None
        self.captures = 0            # graphs captured so far (tests, diagnostics)

    def reset(self):
        self._graphs.clear()
        self._stamp = None

    def _param_stamp(self):
        from . import ops
        return (ops.pack_cache_generation(),) + tuple((p._version, p.data_ptr()) for p in self.model.parameters()) + \
            tuple((b._version, b.data_ptr()) for b in self.model.buffers())

    @staticmethod
    def _sig(x, kwargs):
        def one(v):
            return (tuple(v.shape), v.dtype, str(v.device)) if torch.is_tensor(v) else ("py", repr(v))
        return (one(x),) + tuple((k, one(v)) for k, v in sorted(kwargs.items()))

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, **kwargs):
        from . import ops
        if self.model.training:
            raise RuntimeError("GraphedInference replays an eval-mode forward: call model.eval() first")
        stamp = self._param_stamp()
        if stamp != self._stamp:
            self._graphs.clear()
            self._stamp = stamp
        sig = self._sig(x, kwargs)
        entry = self._graphs.get(sig)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            sx = x.clone()
            skw = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kwargs.items()}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, self.warmup)):
                    self.model(sx, **skw)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with ops.pack_cache_in_capture(), torch.cuda.graph(graph, capture_error_mode=_capture_mode()):
                sy = self.model(sx, **skw)
            entry = (graph, sx, skw, sy, ops.pack_cache_tensors())
            self._graphs[sig] = entry
            self.captures += 1
        graph, sx, skw, sy, _keep = entry
        sx.copy_(x)
        for k, v in kwargs.items():
            if torch.is_tensor(v):
                skw[k].copy_(v)
        graph.replay()
        return sy
