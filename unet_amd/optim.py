"""K17: ``clip_grad_norm_`` + ``AdamW.step`` of the reference's training loop as two HIP passes.

The reference does (train.py:1553-1559, optimizer arguments train.py:183-193, 2214-2241)::

    torch.nn.utils.clip_grad_norm_(model.parameters(), GRAD_CLIP)      # one norm per parameter GROUP of our optimizer
    optimizer.step()                                                    # torch.optim.AdamW(betas=(0.7596, 0.9203), wd=0)

which torch runs as ~12 launches per group (per-tensor norms, stack + norm, clamp, a multi-tensor scale of every
gradient, four multi-tensor AdamW launches).  ``ClipAdamW`` is a ``torch.optim.Optimizer`` with the same
``param_groups`` / ``state_dict`` layout as ``torch.optim.AdamW`` (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter,
so the reference's checkpoints load) that does the same arithmetic in three launches per group
(``csrc/optim.hip``): partial sums of g², one block for the norm and the clip factor, one pass that
updates p, m, v from the clipped gradient.  The clipped gradient is not written back (nothing reads it: the loop drops
the gradients right after the step); ``total_norm(group)`` returns the pre-clip norm like ``clip_grad_norm_`` does.
Nothing synchronises with the host, so the step can be captured in a HIP graph.

The moments live in two flat fp32 buffers per group; ``state[p]["exp_avg"]`` etc. are views into them.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import torch

from . import _lib
from .ops import check, _stream


class _Group:
    """Device tables of one parameter group (built at the first step, rebuilt if the parameter list changes)."""

    def __init__(self, params: List[torch.nn.Parameter]):
        dev = params[0].device
        self.params = params
        start4, off = [], 0
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise TypeError("ClipAdamW: parameters must be contiguous fp32 tensors on one device (fp32 master "
                                "weights, as in the reference)")
            start4.append(off)
            off += (p.numel() + 3) // 4
        start4.append(off)
        if off >= 1 << 31:
            raise ValueError("ClipAdamW: more than 2^33 parameters in one group")
        self.flat = 4 * off
        self.start4 = torch.tensor(start4, dtype=torch.int64).to(torch.int32).to(dev)        # values < 2^31
        self.numel = torch.tensor([p.numel() for p in params], dtype=torch.int64).to(torch.int32).to(dev)
        self.p_ptrs = torch.tensor([p.data_ptr() for p in params], dtype=torch.int64, device=dev)
        self.ptr_key = tuple(p.data_ptr() for p in params)
        self.m = torch.zeros(self.flat, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.flat, dtype=torch.float32, device=dev)
        self.step = torch.zeros(len(params), dtype=torch.float32, device=dev)     # one counter per tensor, as torch
        self.coef = torch.zeros(8, dtype=torch.float32, device=dev)
        self.lr = torch.zeros((), dtype=torch.float32, device=dev)
        self.lr_host = None
        lib = _lib.load()
        # sub-tables of at most OPT_MAX_TENSORS tensors: (first, count, flat elements, first partial, blocks)
        self.batches, nb = [], 0
        for a in range(0, len(params), _lib.OPT_MAX_TENSORS):
            b = min(len(params), a + _lib.OPT_MAX_TENSORS)
            flat = 4 * (start4[b] - start4[a])
            blocks = int(lib.unet_opt_blocks(flat))
            self.batches.append((a, b - a, flat, nb, blocks))
            nb += blocks
        self.partial = torch.zeros(max(nb, 1), dtype=torch.float32, device=dev)
        self.n_partial = nb


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None):
        """Arguments as ``torch.optim.AdamW`` (amsgrad / maximize are not offered: the reference never sets them).
        ``max_grad_norm``: clip the global L2 norm of each parameter group's gradients to this value inside ``step()``
        (``None`` or 0: no clipping); a group may override it with its own ``max_grad_norm`` entry."""
        if not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("ClipAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      max_grad_norm=max_grad_norm))
        self._groups = {}
        self.clips_gradients = True        # GraphedTrainStep / Trainer: do not clip a second time

    # -- tables -------------------------------------------------------------------------------
    def _tables(self, gi: int, group) -> _Group:
        params = [p for p in group["params"] if p.requires_grad]
        if not params:
            return None
        g = self._groups.get(gi)
        key = tuple(p.data_ptr() for p in params)
        if g is None or g.ptr_key != key:
            old = g
            g = _Group(params)
            starts = g.start4.cpu().tolist()
            for i, p in enumerate(params):
                a, n = 4 * starts[i], p.numel()
                st = self.state[p]
                # a state loaded from a checkpoint (or moments of a previous table) move into the flat buffers
                if "exp_avg" in st:
                    g.m[a:a + n].copy_(st["exp_avg"].reshape(-1))
                    g.v[a:a + n].copy_(st["exp_avg_sq"].reshape(-1))
                st["exp_avg"] = g.m[a:a + n].view_as(p)
                st["exp_avg_sq"] = g.v[a:a + n].view_as(p)
                prev = st.get("step")
                if prev is not None:
                    g.step[i] = float(prev)
                st["step"] = g.step[i]
            del old
            self._groups[gi] = g
        return g

    def load_state_dict(self, state_dict):
        """As ``torch.optim.Optimizer.load_state_dict``, except for two things torch's version loses: it replaces
        every param_group by the saved one, so a ``torch.optim.AdamW`` checkpoint of the reference (no
        ``max_grad_norm`` key) would silently switch the clip off, and a device-scalar ``lr`` (graph-captured steps
        read it through its pointer) would become a float no captured step sees."""
        before = [(g.get("max_grad_norm"), g.get("lr")) for g in self.param_groups]
        super().load_state_dict(state_dict)
        for group, (max_norm, lr) in zip(self.param_groups, before):
            if "max_grad_norm" not in group:
                group["max_grad_norm"] = max_norm
            if isinstance(lr, torch.Tensor):
                lr.fill_(float(group["lr"]))      # keep the tensor (and its address), take the loaded value
                group["lr"] = lr
        self._groups = {}                  # the loaded moments are copied into fresh flat buffers at the next step

    def total_norm(self, group_index: int = 0) -> torch.Tensor:
        """Global L2 norm of the group's gradients at the last ``step()`` (before clipping), a device scalar."""
        return self._groups[group_index].coef[3]

    # -- step -----------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        st = _stream()
        for gi, group in enumerate(self.param_groups):
            g = self._tables(gi, group)
            if g is None:
                continue
            lr = group["lr"]
            if isinstance(lr, torch.Tensor) and lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1:
                lr_ptr = lr.data_ptr()                               # a device scalar the caller updates in place
            else:
                lr = float(lr)
                if lr != g.lr_host:
                    g.lr.fill_(lr)
                    g.lr_host = lr
                lr_ptr = g.lr.data_ptr()
            grads = []
            for p in g.params:
                gr = p.grad
                if gr is None:
                    grads.append(None)
                    continue
                if gr.dtype != torch.float32 or not gr.is_contiguous() or gr.is_sparse:
                    raise TypeError("ClipAdamW: gradients must be contiguous dense fp32 tensors")
                grads.append(gr.data_ptr())
            if all(x is None for x in grads):
                continue
            beta1, beta2 = group["betas"]
            max_norm = group.get("max_grad_norm") or 0.0
            arrs = []
            for a, n, flat, p0, blocks in g.batches:
                arr = (C.c_void_p * n)(*grads[a:a + n])
                arrs.append(arr)
                check(lib.unet_grad_sqnorm(arr, n, g.start4[a:].data_ptr(), g.numel[a:].data_ptr(), flat,
                                           g.partial[p0:].data_ptr(), g.step[a:].data_ptr(), st), "unet_grad_sqnorm")
            check(lib.unet_clip_adamw_prepare(g.partial.data_ptr(), g.n_partial, float(max_norm), lr_ptr,
                                              g.coef.data_ptr(), st), "unet_clip_adamw_prepare")
            for (a, n, flat, p0, blocks), arr in zip(g.batches, arrs):
                check(lib.unet_clip_adamw_apply(arr, n, g.p_ptrs[a:].data_ptr(), g.start4[a:].data_ptr(),
                                                g.numel[a:].data_ptr(), flat, g.m.data_ptr(), g.v.data_ptr(),
                                                g.coef.data_ptr(), g.step[a:].data_ptr(), float(beta1), float(beta2),
                                                float(group["eps"]),
                                                float(group["weight_decay"]), st), "unet_clip_adamw_apply")
        return loss
