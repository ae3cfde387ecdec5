"""Autograd bindings of the HIP kernels (libunet_hip.so) for channels-last activations.

Every heavy tensor op of the CondUNet1D hot path is a ``torch.autograd.Function`` whose
forward and backward call the C ABI with raw device pointers on torch's current stream.
PyTorch provides memory, streams and the autograd tape only; tiny per-(window, channel)
algebra (norm statistics finalisation, SE/FiLM MLPs on [N, C] vectors) stays in torch.

All activations here are "NLC": contiguous ``[N, L, C]`` (channels last), dtype bf16 or fp32.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_IDS, ConvDesc, check

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------
def _dt(t: Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    if t.dtype == torch.float32:
        return _lib.F32
    raise TypeError(f"unet_amd kernels take bfloat16 or float32 activations, got {t.dtype}")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need_cuda(t: Tensor):
    if not t.is_cuda:
        raise RuntimeError("unet_amd kernels run on the GPU only (tensor is on %s); there is no CPU path" % t.device)


def _nlc_ok(t: Tensor) -> Tensor:
    """Rows may be a strided slice of a longer tensor; channels must be dense."""
    if t.dim() != 3:
        raise ValueError(f"expected [N, L, C], got {tuple(t.shape)}")
    if t.stride(2) != 1 or t.stride(1) < t.shape[2]:
        t = t.contiguous()
    return t


def _f32c(t: Tensor) -> Tensor:
    return t.detach().to(torch.float32).contiguous()


# When set to a list, every MFMA conv launch appends (kind, algorithmic FLOPs, executed FLOPs, start event, end event);
# bench.py uses it to time the dominant kernel live, on the stream the kernels are launched on.
PROFILE: Optional[list] = None


# include/unet_hip.h UNET_VARIANT_*: the kernel the dispatcher picked for the launch just made
_VARIANTS = {1: "conv_fwd_kernel", 2: "conv_fwd2_kernel", 3: "conv_stream_kernel", 11: "conv_wgrad_kernel",
             12: "conv_wgrad2_kernel", 13: "conv_wgrad3_kernel"}


class _Timed:
    def __init__(self, kind: str, flops: float, alg_scale: float = 1.0):
        """flops: the reference algorithm's count (28 taps for the merged dilated conv); flops / alg_scale is what
        the kernel executes (23 merged taps)."""
        self.on = PROFILE is not None
        if self.on:
            self.rec = (kind, flops, flops / alg_scale, torch.cuda.Event(enable_timing=True),
                        torch.cuda.Event(enable_timing=True))

    def __enter__(self):
        if self.on:
            self.rec[3].record()

    def __exit__(self, *exc):
        if self.on:
            self.rec[4].record()
            variant = _VARIANTS.get(int(_lib.load().unet_conv1d_last_variant()), self.rec[0])
            PROFILE.append((variant,) + self.rec[1:])


def parts(rows: int) -> int:
    return int(_lib.load().unet_parts(rows))


# ---------------------------------------------------------------------------------------------
# convolution geometry (tap lists)
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ConvGeom:
    """y[up*m + phase_j] += W_j x[stride*m + off_j]; taps grouped by phase."""
    offs: Tuple[int, ...]
    phases: Tuple[int, ...]
    stride: int = 1
    up: int = 1
    alg_scale: float = 1.0  # reference-algorithm taps / executed taps (28/23 for the merged dilated conv)

    @property
    def n_taps(self) -> int:
        return len(self.offs)

    def alg_flops(self, n: int, l_out: int, c_out: int, c_in: int) -> float:
        """FLOPs of the reference's algorithm for one launch: 2*Cin*Cout per (tap, output row it feeds)."""
        rows = sum((l_out - r + self.up - 1) // self.up for r in self.phases)
        return 2.0 * n * c_out * c_in * rows * self.alg_scale

    @functools.lru_cache(maxsize=None)
    def dgrad(self) -> Tuple["ConvGeom", List[int]]:
        """Geometry of the data gradient and, for each of ITS taps, the forward tap it uses."""
        taps = []
        for j, (o, r) in enumerate(zip(self.offs, self.phases)):
            if self.up == 1 and self.stride == 1:
                taps.append((0, -o, j))
            elif self.up == 1 and self.stride == 2:
                rp = o % 2             # dx row i = 2m + off_j, so this tap feeds rows of parity off_j mod 2
                taps.append((rp, (rp - o) // 2, j))
            elif self.up == 2 and self.stride == 1:
                taps.append((0, r - 2 * o, j))
            else:
                raise ValueError("unsupported geometry")
        taps.sort(key=lambda t: (t[0], t[1]))
        if self.up == 1 and self.stride == 1:
            g = ConvGeom(tuple(t[1] for t in taps), tuple(0 for _ in taps), 1, 1, self.alg_scale)
        elif self.up == 1 and self.stride == 2:
            g = ConvGeom(tuple(t[1] for t in taps), tuple(t[0] for t in taps), 1, 2, self.alg_scale)
        else:
            g = ConvGeom(tuple(t[1] for t in taps), tuple(0 for _ in taps), 2, 1, self.alg_scale)
        return g, [t[2] for t in taps]


def geom_conv(k: int, stride: int = 1, dilation: int = 1) -> ConvGeom:
    """nn.Conv1d(k, stride, padding=(k-1)*d//2, dilation=d) (models.py:558, 689-699)."""
    pad = (k - 1) * dilation // 2
    offs = tuple(i * dilation - pad for i in range(k))
    return ConvGeom(offs, tuple(0 for _ in offs), stride, 1)


def geom_convt_k4s2p1() -> Tuple[ConvGeom, Tuple[int, ...]]:
    """nn.ConvTranspose1d(k=4, s=2, p=1) (models.py:592, 881): o = 2i - 1 + k.
    Even outputs use k=1 (i=m) and k=3 (i=m-1); odd outputs k=0 (i=m+1) and k=2 (i=m).
    Returns the geometry and the kernel index each tap reads."""
    return ConvGeom((-1, 0, 0, 1), (0, 0, 1, 1), 1, 2), (3, 1, 2, 0)


def merged_dilated_taps(k: int, dilations: Sequence[int]) -> Tuple[ConvGeom, List[Tuple[int, int, int]]]:
    """The 4 parallel dilated convs of MultiScaleDilatedConv1d (models.py:739-759) are linear
    in x, so their softmax-weighted sum is ONE conv whose taps are the union of the branch
    taps (28 taps -> 23 distinct offsets for k=7, d=(1,4,16,32)).  Returns the merged geometry
    (stride filled in by the caller) and, for every (branch, k) pair, its merged tap index."""
    pairs = []
    offs = sorted({i * d - (k - 1) * d // 2 for d in dilations for i in range(k)})
    index = {o: j for j, o in enumerate(offs)}
    for b, d in enumerate(dilations):
        for i in range(k):
            pairs.append((b, i, index[i * d - (k - 1) * d // 2]))
    return ConvGeom(tuple(offs), tuple(0 for _ in offs), 1, 1, len(pairs) / len(offs)), pairs


def conv_out_len(l_in: int, k: int, stride: int, dilation: int = 1) -> int:
    pad = (k - 1) * dilation // 2
    return (l_in + 2 * pad - dilation * (k - 1) - 1) // stride + 1


class MergeTapsFn(torch.autograd.Function):
    """w_eff[T, Co, Ci] = sum over (branch, k) landing on tap t of sw[branch] * W_branch[:, :, k]
    (the weight algebra of MultiScaleDilatedConv1d, models.py:752-759) in one kernel each way."""

    @staticmethod
    def forward(ctx, sw, tap_index: Tuple[int, ...], n_taps: int, *branch_w):
        lib = _lib.load()
        nb = len(branch_w)
        co, ci, k = branch_w[0].shape
        ws = [w.detach().float().contiguous() for w in branch_w]
        swf = _f32c(sw)
        w_eff = torch.empty((n_taps, co, ci), dtype=torch.float32, device=swf.device)
        ptrs = (C.c_void_p * nb)(*[w.data_ptr() for w in ws])
        taps = (C.c_int32 * (nb * k))(*tap_index)
        check(lib.unet_merge_taps_fwd(ptrs, _p(swf), nb, k, co, ci, taps, n_taps, _p(w_eff), _stream()),
              "unet_merge_taps_fwd")
        ctx.save_for_backward(swf, *ws)
        ctx.meta = (tuple(tap_index), n_taps, sw.dtype, [w.dtype for w in branch_w])
        return w_eff

    @staticmethod
    def backward(ctx, dw_eff):
        lib = _lib.load()
        swf, *ws = ctx.saved_tensors
        tap_index, n_taps, sw_dtype, w_dtypes = ctx.meta
        nb = len(ws)
        co, ci, k = ws[0].shape
        dw_eff = dw_eff.float().contiguous()
        dws = [torch.empty_like(w) for w in ws]
        parts_n = int(lib.unet_merge_taps_parts(co, ci))
        dsw_part = torch.empty((parts_n, nb), dtype=torch.float32, device=dw_eff.device)
        ptrs = (C.c_void_p * nb)(*[w.data_ptr() for w in ws])
        dptrs = (C.c_void_p * nb)(*[w.data_ptr() for w in dws])
        taps = (C.c_int32 * (nb * k))(*tap_index)
        check(lib.unet_merge_taps_bwd(ptrs, _p(swf), nb, k, co, ci, taps, n_taps, _p(dw_eff), dptrs, _p(dsw_part),
                                      _stream()), "unet_merge_taps_bwd")
        dsw = torch.empty((1, nb), dtype=torch.float32, device=dw_eff.device)
        check(lib.unet_reduce_rows(_p(dsw_part), parts_n, nb, _p(dsw), _stream()), "unet_reduce_rows")
        return (dsw[0].to(sw_dtype), None, None) + tuple(d.to(t) for d, t in zip(dws, w_dtypes))


class TapOrderFn(torch.autograd.Function):
    """w[order] along the tap axis for a PERMUTATION ``order`` (the transposed conv's taps in phase order): one copy each
    way.  torch.stack of slices has the same forward, but its backward is a zero-fill, a copy and an add per slice
    (11 launches per ConvTranspose1d and step)."""

    @staticmethod
    def forward(ctx, w, order: Tuple[int, ...]):
        if sorted(order) != list(range(w.shape[0])):
            raise ValueError(f"{order} is not a permutation of the {w.shape[0]} taps")
        ctx.order = order
        return torch.stack([w[k] for k in order])

    @staticmethod
    def backward(ctx, g):
        inv = [0] * len(ctx.order)
        for j, k in enumerate(ctx.order):
            inv[k] = j
        return torch.stack([g[j] for j in inv]), None


class MergeBranchesFn(torch.autograd.Function):
    """(w_eff, b_eff) of MultiScaleDilatedConv1d (models.py:752-759) from the RAW scale_weights, the branch weights
    and the branch biases: softmax over the branches, merged taps and the effective bias in one launch
    (unet_merge_branches_fwd); backward in two (branch gradients + partials, then the sum of the partials taken
    through the softmax).  As separate torch ops around MergeTapsFn this was 5 launches forward and 8 backward per
    conv."""

    @staticmethod
    def forward(ctx, scale_weights, tap_index: Tuple[int, ...], n_taps: int, nb: int, *wb):
        lib = _lib.load()
        branch_w, branch_b = wb[:nb], wb[nb:]
        co, ci, k = branch_w[0].shape
        ws = [_f32c(w.detach()) for w in branch_w]
        bs = [_f32c(b.detach()) for b in branch_b]
        raw = _f32c(scale_weights.detach())
        dev = raw.device
        w_eff = torch.empty((n_taps, co, ci), dtype=torch.float32, device=dev)
        b_eff = torch.empty(co, dtype=torch.float32, device=dev)
        sw = torch.empty(nb, dtype=torch.float32, device=dev)
        wp = (C.c_void_p * nb)(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * nb)(*[b.data_ptr() for b in bs])
        taps = (C.c_int32 * (nb * k))(*tap_index)
        check(lib.unet_merge_branches_fwd(wp, bp, _p(raw), 1, nb, k, co, ci, taps, n_taps, _p(w_eff), _p(b_eff), _p(sw),
                                          _stream()), "unet_merge_branches_fwd")
        ctx.save_for_backward(sw, *ws, *bs)
        ctx.meta = (tuple(tap_index), n_taps, nb, scale_weights.dtype, [t.dtype for t in wb])
        return w_eff, b_eff

    @staticmethod
    def backward(ctx, dw_eff, db_eff):
        lib = _lib.load()
        tap_index, n_taps, nb, sw_dtype, dtypes = ctx.meta
        sw, *rest = ctx.saved_tensors
        ws, bs = rest[:nb], rest[nb:]
        co, ci, k = ws[0].shape
        dev = sw.device
        dw_eff = _f32c(dw_eff) if dw_eff is not None else torch.zeros((n_taps, co, ci), dtype=torch.float32, device=dev)
        db_eff = _f32c(db_eff) if db_eff is not None else torch.zeros(co, dtype=torch.float32, device=dev)
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty_like(b) for b in bs]
        parts_n = int(lib.unet_merge_taps_parts(co, ci))
        dsw_part = torch.empty((parts_n, nb), dtype=torch.float32, device=dev)
        draw = torch.empty(nb, dtype=torch.float32, device=dev)
        wp = (C.c_void_p * nb)(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * nb)(*[b.data_ptr() for b in bs])
        dwp = (C.c_void_p * nb)(*[w.data_ptr() for w in dws])
        dbp = (C.c_void_p * nb)(*[b.data_ptr() for b in dbs])
        taps = (C.c_int32 * (nb * k))(*tap_index)
        check(lib.unet_merge_branches_bwd(wp, bp, _p(sw), 1, nb, k, co, ci, taps, n_taps, _p(dw_eff), _p(db_eff), dwp, dbp,
                                          _p(dsw_part), _p(draw), _stream()), "unet_merge_branches_bwd")
        grads = [d.to(t) for d, t in zip(dws + dbs, dtypes)]
        return (draw.to(sw_dtype), None, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
# conv launchers
# ---------------------------------------------------------------------------------------------
def _desc(geom: ConvGeom, x1: Tensor, x2: Optional[Tensor], y: Tensor, res: Optional[Tensor], act: int = 0) -> ConvDesc:
    if geom.n_taps > _lib.MAX_TAPS:
        raise ValueError(f"{geom.n_taps} taps exceed UNET_MAX_TAPS={_lib.MAX_TAPS}")
    d = ConvDesc()
    d.dtype = _dt(x1)
    d.n, d.l_in, d.c1 = x1.shape
    d.x1_ld, d.x1_ns = x1.stride(1), x1.stride(0)
    if x2 is not None:
        if x2.shape[:2] != x1.shape[:2] or x2.dtype != x1.dtype:
            raise ValueError("concat sources must agree in batch, length and dtype")
        d.c2, d.x2_ld, d.x2_ns = x2.shape[2], x2.stride(1), x2.stride(0)
    else:
        d.c2, d.x2_ld, d.x2_ns = 0, 0, 0
    d.l_out, d.c_out = y.shape[1], y.shape[2]
    d.y_ld, d.y_ns = y.stride(1), y.stride(0)
    if res is not None:
        if res.shape != y.shape or res.dtype != y.dtype:
            raise ValueError(f"residual {tuple(res.shape)}/{res.dtype} does not match output {tuple(y.shape)}/{y.dtype}")
        d.res_ld, d.res_ns = res.stride(1), res.stride(0)
    d.stride, d.up, d.n_taps = geom.stride, geom.up, geom.n_taps
    for j, (o, r) in enumerate(zip(geom.offs, geom.phases)):
        d.tap_off[j] = o
        d.tap_phase[j] = r
    d.act = act
    return d


def pack_weights(w: Tensor, dtype: int, *, transpose: bool, tap_src: Optional[Sequence[int]], n_taps: int,
                 row_lo: int, rows: int, c1: int, c2: int) -> Tensor:
    """w: fp32 [J, Cout, Ctot] (any strides).  transpose=False packs rows=co, K=ci (forward);
    transpose=True packs rows=ci[row_lo:row_lo+rows], K=co (data gradient)."""
    lib = _lib.load()
    nbytes = lib.unet_conv1d_packed_bytes(dtype, n_taps, rows, c1, c2)
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    ts, s1, s2 = w.stride()
    rs, cs = (s2, s1) if transpose else (s1, s2)
    src = None
    if tap_src is not None:
        src = (C.c_int32 * n_taps)(*tap_src)
    check(lib.unet_conv1d_pack(dtype, _p(w), ts, rs, cs, n_taps, src, row_lo, rows, 0, c1, c2, _p(out), _stream()),
          "unet_conv1d_pack")
    return out


def pack_weights_multi(w: Tensor, dtype: int, jobs: Sequence[dict]) -> List[Tensor]:
    """Several ``pack_weights`` images in ONE launch (``jobs``: the keyword arguments of each; a job may name its own
    weight tensor under ``"w"``, otherwise it packs ``w``)."""
    lib = _lib.load()
    if not 1 <= len(jobs) <= _lib.PACK_MAX_JOBS:
        raise ValueError(f"1..{_lib.PACK_MAX_JOBS} images per launch, got {len(jobs)}")
    arr = (_lib.PackJob * len(jobs))()
    outs = []
    for a, j in zip(arr, jobs):
        wj = j.get("w", w)
        ts, s1, s2 = wj.stride()
        out = torch.empty(lib.unet_conv1d_packed_bytes(dtype, j["n_taps"], j["rows"], j["c1"], j["c2"]),
                          dtype=torch.uint8, device=wj.device)
        outs.append(out)
        a.w, a.tap_stride = wj.data_ptr(), ts
        a.row_stride, a.col_stride = (s2, s1) if j["transpose"] else (s1, s2)
        a.n_taps, a.has_tap_src = j["n_taps"], int(j["tap_src"] is not None)
        if j["tap_src"] is not None:
            for t, v in enumerate(j["tap_src"]):
                a.tap_src[t] = v
        a.row_lo, a.rows, a.col_lo, a.c1, a.c2 = j["row_lo"], j["rows"], 0, j["c1"], j["c2"]
        a.packed = out.data_ptr()
    check(lib.unet_conv1d_pack_multi(dtype, arr, len(jobs), _stream()), "unet_conv1d_pack_multi")
    return outs


def conv_forward_raw(geom: ConvGeom, x1: Tensor, x2: Optional[Tensor], w_packed: Tensor, bias: Optional[Tensor],
                     res: Optional[Tensor], l_out: int, c_out: int, act: int = 0, out: Optional[Tensor] = None,
                     want_stats: bool = False) -> Tensor:
    """``out``: write into this [N, l_out, c_out] tensor (any batch / row pitch with unit channel stride) instead of a
    new one -- how a group of convs fills one buffer without a gather afterwards.
    ``want_stats``: a BatchNorm follows; where the kernel can, it emits the per-tile sum / sum of squares of y from its
    epilogue and they travel beside y (``_take_stats``) so that the norm skips its statistics pass."""
    lib = _lib.load()
    if out is not None:
        if tuple(out.shape) != (x1.shape[0], l_out, c_out) or out.dtype != x1.dtype or out.stride(2) != 1:
            raise ValueError(f"out {tuple(out.shape)}/{out.dtype} does not fit [{x1.shape[0]}, {l_out}, {c_out}]/{x1.dtype}")
        y = out
    else:
        y = torch.empty((x1.shape[0], l_out, c_out), dtype=x1.dtype, device=x1.device)
    d = _desc(geom, x1, x2, y, res, act)
    max_parts = int(lib.unet_conv1d_stats_max_parts(C.byref(d))) if (want_stats and EPILOGUE_STATS) else 0
    with _Timed("conv_fwd", geom.alg_flops(d.n, l_out, c_out, d.c1 + d.c2), geom.alg_scale):
        if max_parts > 0:
            sbuf = torch.empty((2, d.n * max_parts * c_out), dtype=torch.float32, device=x1.device)
            got = C.c_int32(0)
            check(lib.unet_conv1d_fwd_stats(C.byref(d), _p(x1), _p(x2), _p(w_packed), _p(bias), _p(res), _p(y),
                                            _p(sbuf[0]), _p(sbuf[1]), C.byref(got), _stream()), "unet_conv1d_fwd_stats")
            if got.value > 0:
                k = d.n * got.value * c_out
                _offer_stats(y, sbuf[0, :k].view(d.n, got.value, c_out), sbuf[1, :k].view(d.n, got.value, c_out))
        else:
            check(lib.unet_conv1d_fwd(C.byref(d), _p(x1), _p(x2), _p(w_packed), _p(bias), _p(res), _p(y), _stream()),
                  "unet_conv1d_fwd")
    return y


FILM_GATED_EPILOGUE = os.environ.get("UNET_FILM_GATED", "1") != "0"   # 0: out_proj, then unet_gated_add_fwd (A/B)


def conv_forward_gated_raw(geom: ConvGeom, x1: Tensor, w_packed: Tensor, bias: Optional[Tensor], res: Tensor,
                           gate: Tensor, out: Tensor, keep_o: bool):
    """``out = res + gate[n, c] * o`` with ``o = conv(x1) + bias`` in ONE kernel (unet_conv1d_fwd_gated).  Returns
    ``(o or None, launched)``: ``launched`` False means the streaming kernel does not take this shape, nothing ran, and
    the caller falls back to the conv followed by unet_gated_add_fwd.  ``o`` is stored only when ``keep_o``."""
    lib = _lib.load()
    n, l_out, c_out = out.shape
    o = torch.empty_like(out) if keep_o else None
    d = _desc(geom, x1, None, out, res, 0)
    with _Timed("conv_fwd", geom.alg_flops(d.n, l_out, c_out, d.c1), geom.alg_scale):
        rc = lib.unet_conv1d_fwd_gated(C.byref(d), _p(x1), _p(w_packed), _p(bias), _p(res), _p(gate), _p(o), _p(out),
                                       _stream())
    if rc == 1:
        return None, False
    check(rc, "unet_conv1d_fwd_gated")
    return o, True


def conv_wgrad_raw(geom: ConvGeom, x1: Tensor, x2: Optional[Tensor], dy: Tensor, c_out: int) -> Tensor:
    """dW[j][co][ci] (fp32) of the conv described by geom, from its input(s) and the output gradient."""
    lib = _lib.load()
    n = x1.shape[0]
    ctot = x1.shape[2] + (x2.shape[2] if x2 is not None else 0)
    d = _desc(geom, x1, x2, dy, None)
    ws_bytes = lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dy.device)
    dw = torch.empty((geom.n_taps, c_out, ctot), dtype=torch.float32, device=dy.device)
    with _Timed("conv_wgrad", geom.alg_flops(n, dy.shape[1], c_out, ctot), geom.alg_scale):
        check(lib.unet_conv1d_wgrad(C.byref(d), _p(x1), _p(x2), _p(dy), _p(dw), _p(ws), ws_bytes, _stream()),
              "unet_conv1d_wgrad")
    return dw


def _dgrad_pack_job(geom: ConvGeom, wf: Tensor, row_lo: int, rows: int) -> dict:
    j, c_out, _ = wf.shape
    return dict(transpose=True, tap_src=geom.dgrad()[1], n_taps=j, row_lo=row_lo, rows=rows, c1=c_out, c2=0)


def conv_dgrad_raw(geom: ConvGeom, wf: Tensor, dy: Tensor, l_in: int, row_lo: int, rows: int,
                   res: Optional[Tensor] = None, out: Optional[Tensor] = None, wp: Optional[Tensor] = None) -> Tensor:
    """dx[:, :, row_lo:row_lo+rows] of the conv (wf: fp32 [J, Cout, Ctot]); ``res`` is added in the epilogue, which
    is how a second gradient path into the same tensor is merged without an extra pass.  ``wp``: the transposed
    packed image when the caller already has it (ConvFn packs it beside the forward image)."""
    gd, _ = geom.dgrad()
    if wp is None:
        wp = pack_weights(wf, _dt(dy), **_dgrad_pack_job(geom, wf, row_lo, rows))
    return conv_forward_raw(gd, dy, None, wp, None, res, l_in, rows, out=out)


# Bias-gradient hand-off.  The backward pass of norm + activation writes the gradient `dy` of a conv's output and can
# add its per-part column sums for free (unet_affine_act_bwd_apply_sum); the conv's backward would otherwise spend a
# pass of its own over dy (unet_colsum) on the bias gradient.  The partials travel beside the tensor: keyed by the
# Python object's id and guarded by a weak reference, so an entry can only ever be used with the very tensor it was
# computed from (a dead or recycled object never matches; the fallback is the separate pass).
import weakref

_BIAS_HINT = {}
BIAS_HINT_HITS = 0          # diagnostic: how many conv bias gradients came from a hand-off


def _ver(t: Tensor) -> int:
    try:
        return t._version
    except RuntimeError:       # inference tensors keep no version counter (and cannot be written in place either)
        return -1


EPILOGUE_STATS = os.environ.get("UNET_EPILOGUE_STATS", "1") != "0"   # 0: separate statistics pass (A/B measurements)
_STATS_HINT = {}


def _offer_stats(y: Tensor, s1p: Tensor, s2p: Tensor):
    """Same hand-off for BatchNorm statistics: the conv that wrote ``y`` also wrote its per-tile sum / sum of squares."""
    if len(_STATS_HINT) > 64:
        _STATS_HINT.clear()
    _STATS_HINT[id(y)] = (weakref.ref(y), s1p, s2p, _ver(y))


STATS_HINT_HITS = 0          # diagnostic: how many BatchNorm statistics came from a conv epilogue


def _take_stats(y: Tensor):
    global STATS_HINT_HITS
    h = _STATS_HINT.pop(id(y), None)
    # the same tensor object, not written to since the conv (an in-place op in between would make the sums stale)
    if (h is not None and h[0]() is y and h[3] == _ver(y) and h[1].shape[0] == y.shape[0]
            and h[1].shape[2] == y.shape[2]):
        STATS_HINT_HITS += 1
        return h[1], h[2]
    return None


def _offer_colsum(dy: Tensor, parts: Tensor):
    if len(_BIAS_HINT) > 64:
        _BIAS_HINT.clear()
    _BIAS_HINT[id(dy)] = (weakref.ref(dy), parts, _ver(dy))


def _take_colsum(dy: Tensor) -> Optional[Tensor]:
    h = _BIAS_HINT.pop(id(dy), None)
    if (h is not None and h[0]() is dy and h[2] == _ver(dy) and h[1].shape[0] == dy.shape[0]
            and h[1].shape[2] == dy.shape[2]):
        return h[1]
    return None


DUAL_PACK = os.environ.get("UNET_DUAL_PACK", "1") != "0"     # 0: one pack launch per image (A/B measurements)


class GradStash:
    """Hand-off of ONE gradient between the two consumers of a skip tensor (CondUNet1D: an encoder output feeds the
    next encoder block's conv AND a decoder block).  Autograd would add the two gradients with a full-tensor pass of
    its own (3 touches); instead the decoder-side consumer parks its gradient here (``StashGradFn``) and the
    encoder-side conv adds it in the epilogue of its data-gradient kernel (``ConvFn.backward``: ``res``).
    Either order of the two backward calls is correct: if the conv's backward ran first (``consumed``), the parked
    path hands its gradient to autograd as usual."""
    __slots__ = ("g", "armed", "consumed")

    def __init__(self):
        self.g, self.armed, self.consumed = None, False, False


SKIP_GRAD_FUSE = os.environ.get("UNET_SKIP_GRAD_FUSE", "1") != "0"     # 0: autograd's own accumulation (A/B runs)
SKIP_GRAD_HITS = 0          # diagnostic: gradients merged in a data-gradient epilogue


class StashGradFn(torch.autograd.Function):
    """Identity whose backward parks the incoming gradient in ``stash`` (and reports none to autograd)."""

    @staticmethod
    def forward(ctx, x, stash: GradStash):
        ctx.stash = stash
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        st = ctx.stash
        if st.consumed:          # the conv's backward has already run (or ran in an earlier backward over this graph)
            return g, None
        st.g = g if st.g is None else st.g + g
        return None, None


def stash_for(x: Tensor) -> Optional[GradStash]:
    """Mark ``x`` as a tensor whose next conv consumer (``ops.conv(x, ...)``) may take a second gradient through its
    data-gradient epilogue; returns the stash (``armed`` once such a consumer has accepted it)."""
    if not (SKIP_GRAD_FUSE and torch.is_grad_enabled() and x.requires_grad):
        return None
    st = GradStash()
    x._unet_grad_stash = st
    return st


class ConvFn(torch.autograd.Function):
    """y = conv(geom; [x1 | x2], w[J, Cout, C1+C2], bias) (+ res).  See include/unet_hip.h."""

    @staticmethod
    def forward(ctx, x1, x2, w, bias, res, geom: ConvGeom, l_out: int, want_stats: bool = False,
                x1_stash: Optional[GradStash] = None):
        _need_cuda(x1)
        ctx.x1_stash = x1_stash if (x1_stash is not None and ctx.needs_input_grad[0]) else None
        if ctx.x1_stash is not None:
            ctx.x1_stash.armed = True
        x1 = _nlc_ok(x1)
        x2 = _nlc_ok(x2) if x2 is not None else None
        res = _nlc_ok(res) if res is not None else None
        dt = _dt(x1)
        c1 = x1.shape[2]
        c2 = x2.shape[2] if x2 is not None else 0
        j, c_out, ctot = w.shape
        if j != geom.n_taps or ctot != c1 + c2:
            raise ValueError(f"weight {tuple(w.shape)} does not fit taps={geom.n_taps}, channels={c1}+{c2}")
        wf = w.detach()
        if wf.dtype != torch.float32:
            wf = wf.float()
        # the forward image and the transposed images the data gradients will read, in one launch: the weights do
        # not change between this call and backward (save_for_backward checks that), and a pack is launch-bound
        jobs = [dict(transpose=False, tap_src=None, n_taps=j, row_lo=0, rows=c_out, c1=c1, c2=c2)]
        ctx.wp_d1 = ctx.wp_d2 = None
        want_d1 = DUAL_PACK and ctx.needs_input_grad[0]
        want_d2 = DUAL_PACK and x2 is not None and ctx.needs_input_grad[1]
        if want_d1:
            jobs.append(_dgrad_pack_job(geom, wf, 0, c1))
        if want_d2:
            jobs.append(_dgrad_pack_job(geom, wf, c1, c2))
        if len(jobs) > 1:
            imgs = pack_weights_multi(wf, dt, jobs)
            wp = imgs[0]
            if want_d1:
                ctx.wp_d1 = imgs[1]
            if want_d2:
                ctx.wp_d2 = imgs[-1]
        else:
            wp = pack_weights(wf, dt, **jobs[0])
        b = _f32c(bias) if bias is not None else None
        y = conv_forward_raw(geom, x1, x2, wp, b, res, l_out, c_out, want_stats=want_stats)
        ctx.geom, ctx.l_out = geom, l_out
        ctx.has_x2, ctx.has_bias, ctx.has_res = x2 is not None, bias is not None, res is not None
        ctx.w_dtype = w.dtype
        ctx.save_for_backward(x1, x2, wf)
        return y

    @staticmethod
    def backward(ctx, dy):
        global BIAS_HINT_HITS
        x1, x2, wf = ctx.saved_tensors
        geom: ConvGeom = ctx.geom
        hint = _take_colsum(dy)
        dy = dy.contiguous()
        n, l_in, c1 = x1.shape
        c2 = x2.shape[2] if x2 is not None else 0
        j, c_out, ctot = wf.shape
        dx1 = dx2 = dw = db = dres = None
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = dy
        if ctx.has_bias and ctx.needs_input_grad[3]:
            if hint is not None:
                db = _reduce_all_parts(hint)
                BIAS_HINT_HITS += 1
            else:
                db = colsum(dy)
        if ctx.needs_input_grad[2]:
            dw = conv_wgrad_raw(geom, x1, x2, dy, c_out)
            if ctx.w_dtype != torch.float32:
                dw = dw.to(ctx.w_dtype)
        if ctx.needs_input_grad[0]:
            extra = None
            st = ctx.x1_stash
            if st is not None:       # the gradient the tensor's OTHER consumer parked: merged in the kernel's epilogue
                global SKIP_GRAD_HITS
                extra, st.g, st.consumed = st.g, None, True
            fused = extra is not None and extra.shape == x1.shape and extra.dtype == dy.dtype and extra.is_contiguous()
            dx1 = conv_dgrad_raw(geom, wf, dy, l_in, 0, c1, wp=ctx.wp_d1, res=extra if fused else None)
            if fused:
                SKIP_GRAD_HITS += 1
            elif extra is not None:
                dx1 = dx1 + extra.to(dx1.dtype)
        if ctx.has_x2 and ctx.needs_input_grad[1]:
            dx2 = conv_dgrad_raw(geom, wf, dy, l_in, c1, c2, wp=ctx.wp_d2)
        return dx1, dx2, dw, db, dres, None, None, None, None


# Packed-weight cache for INFERENCE.  A training step must repack every weight (the optimizer has just changed it; the
# packed bf16 image is a per-call temporary), but under torch.no_grad() the same weights serve call after call
# (sliding-window inference, evaluate()): the packed image is then kept per conv, keyed on the identity of the
# parameters it derives from and revalidated against their ``_version`` counters ("packed-weight caches
# are torch tensors owned by the Python module, rebuilt when weight._version changes").  Version counters are not
# enough on their own -- torch's FUSED optimizers update parameters without bumping them -- so CondUNet1D.train()
# also drops the cache whenever the model (re-)enters training mode.  Never used while a HIP graph is being
# captured: a cache hit would leave the pack kernels out of the graph, and replays would run on stale images.
_PACK_CACHE = {}


_PACK_GENERATION = 0     # bumped by every clear: what a captured inference graph checks before it replays


def clear_pack_cache():
    global _PACK_GENERATION
    _PACK_GENERATION += 1
    _PACK_CACHE.clear()


def pack_cache_generation() -> int:
    """Counts ``clear_pack_cache()`` calls (``CondUNet1D.train()``, ``load_state_dict``, ``.to()``): a graph captured
    over cached weight images (``graphs.GraphedInference``) is stale once this has moved."""
    return _PACK_GENERATION


_CACHE_IN_CAPTURE = False


class pack_cache_in_capture:
    """Context manager for the capture of an INFERENCE graph (graphs.GraphedInference): cached weight images may be
    used while capturing, so the graph holds the convs alone and not the ~40 pack launches of a forward pass.  The
    graph then reads the cached tensors on every replay: whoever captures under this must keep them alive
    (``pack_cache_tensors``) and re-capture when a parameter changes."""

    def __enter__(self):
        global _CACHE_IN_CAPTURE
        self.prev, _CACHE_IN_CAPTURE = _CACHE_IN_CAPTURE, True
        return self

    def __exit__(self, *exc):
        global _CACHE_IN_CAPTURE
        _CACHE_IN_CAPTURE = self.prev
        return False


def pack_cache_tensors() -> list:
    """Every tensor the pack cache holds right now (a capture under ``pack_cache_in_capture`` keeps this list so that
    clearing the cache later cannot free memory its graph still reads)."""
    return [t for entry in _PACK_CACHE.values() for t in entry[1] if torch.is_tensor(t)]


def _cacheable(params) -> bool:
    return (params is not None and not torch.is_grad_enabled() and os.environ.get("UNET_PACK_CACHE", "1") != "0"
            and (_CACHE_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()))


def packed_for_inference(tag, params: Sequence[Tensor], dtype: int, build):
    """``build()`` -> tuple of tensors (packed weights, fp32 bias, ...) derived from ``params``; cached per
    (tag, dtype, parameter identities) while every parameter's ``_version`` is unchanged.  Identity means the
    Parameter OBJECT: the entry holds weak references and a hit requires ``ref() is p`` for every parameter, and the
    entry is dropped when any of them dies -- a process that evaluates checkpoint after checkpoint gets freed blocks
    back at the same addresses with the same version counts, so neither ``data_ptr()`` nor ``id()`` alone names a
    tensor."""
    key = (tag, dtype) + tuple(id(p) for p in params)
    versions = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params)
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == versions and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    val = build()

    def evict(_ref, key=key):
        _PACK_CACHE.pop(key, None)

    _PACK_CACHE[key] = (versions, val, tuple(weakref.ref(p, evict) for p in params))
    return val


def conv(x1: Tensor, w, bias: Optional[Tensor], geom: ConvGeom, l_out: int, x2: Optional[Tensor] = None,
         res: Optional[Tensor] = None, params: Optional[Sequence[Tensor]] = None, want_stats: bool = False) -> Tensor:
    """``w``: [J, Cout, C1+C2] weights, or a callable returning ``(w, bias)`` (weights that take work to derive: the
    merged dilated taps, the transposed conv's tap order).  ``params``: the parameters ``w`` and ``bias`` derive
    from; given, inference calls reuse the packed image instead of rebuilding it (see above)."""
    if _cacheable(params):
        _need_cuda(x1)
        x1 = _nlc_ok(x1)
        x2 = _nlc_ok(x2) if x2 is not None else None
        res = _nlc_ok(res) if res is not None else None
        dt = _dt(x1)
        c1, c2 = x1.shape[2], (x2.shape[2] if x2 is not None else 0)

        def build():
            wv, bv = w() if callable(w) else (w, bias)
            wf = wv.detach().float()
            if wf.shape[0] != geom.n_taps or wf.shape[2] != c1 + c2:
                raise ValueError(f"weight {tuple(wf.shape)} does not fit taps={geom.n_taps}, channels={c1}+{c2}")
            wp = pack_weights(wf, dt, transpose=False, tap_src=None, n_taps=wf.shape[0], row_lo=0, rows=wf.shape[1], c1=c1, c2=c2)
            return wp, (_f32c(bv) if bv is not None else None), wf.shape[1]

        wp, b32, c_out = packed_for_inference(("conv", geom, c1, c2), params, dt, build)
        return conv_forward_raw(geom, x1, x2, wp, b32, res, l_out, c_out)
    wv, bv = w() if callable(w) else (w, bias)
    return ConvFn.apply(x1, x2, wv, bv, res, geom, l_out, want_stats, getattr(x1, "_unet_grad_stash", None))


# ---------------------------------------------------------------------------------------------
# streaming kernels
# ---------------------------------------------------------------------------------------------
def _dims(x: Tensor):
    n, rows, c = x.shape
    return n, rows, c


def _part_buf(x: Tensor, k: int = 1):
    n, rows, c = x.shape
    return [torch.empty((n, parts(rows), c), dtype=torch.float32, device=x.device) for _ in range(k)]


def sum_parts(buf: Tensor) -> Tensor:
    """[N, parts, C] fp32 partials -> [N, C] (one coalesced HIP pass instead of a strided torch reduction)"""
    n, p, c = buf.shape
    out = torch.empty((n, c), dtype=torch.float32, device=buf.device)
    check(_lib.load().unet_reduce_parts(_p(buf), n, p, c, _p(out), _stream()), "unet_reduce_parts")
    return out


def colsum(x: Tensor) -> Tensor:
    """sum over windows and rows -> [C] fp32 (two HIP stages, fp64 accumulation in the second)"""
    lib = _lib.load()
    x = x.contiguous()
    (buf,) = _part_buf(x)
    n, rows, c = x.shape
    check(lib.unet_colsum(_dt(x), _p(x), _p(buf), n, rows, c, _stream()), "unet_colsum")
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib.unet_reduce_rows(_p(buf), buf.shape[0] * buf.shape[1], c, _p(out), _stream()), "unet_reduce_rows")
    return out


REDUCE_MULTI = os.environ.get("UNET_REDUCE_MULTI", "1") != "0"        # 0: one launch per sum (A/B measurements)


def reduce_multi(bufs: Sequence[Tensor], per_window: Sequence[bool]) -> List[Tensor]:
    """``sum_parts`` (per_window True: [N, P, C] -> [N, C]) and ``_reduce_all_parts`` (False: -> [C]) of several partial
    buffers in ONE launch (unet_reduce_multi): each of these sums is a few microseconds of launch-bound work."""
    lib = _lib.load()
    if not REDUCE_MULTI:
        return [sum_parts(b) if pw else _reduce_all_parts(b) for b, pw in zip(bufs, per_window)]
    outs: List[Tensor] = []
    for lo in range(0, len(bufs), _lib.REDUCE_MAX_JOBS):
        chunk = list(zip(bufs, per_window))[lo:lo + _lib.REDUCE_MAX_JOBS]
        arr = (_lib.ReduceJob * len(chunk))()
        for i, (buf, pw) in enumerate(chunk):
            n, p, c = buf.shape
            if buf.dtype != torch.float32 or not buf.is_contiguous():
                raise ValueError("reduce_multi takes contiguous fp32 [N, parts, C] buffers")
            out = torch.empty((n, c) if pw else (c,), dtype=torch.float32, device=buf.device)
            arr[i].src, arr[i].out = buf.data_ptr(), out.data_ptr()
            arr[i].units, arr[i].rows, arr[i].c = (n, p, c) if pw else (1, n * p, c)
            outs.append(out)
        check(lib.unet_reduce_multi(arr, len(chunk), _stream()), "unet_reduce_multi")
    return outs


def _reduce_all_parts(buf: Tensor) -> Tensor:
    """[N, parts, C] partial column sums -> [C] (second stage of colsum)"""
    c = buf.shape[2]
    out = torch.empty(c, dtype=torch.float32, device=buf.device)
    check(_lib.load().unet_reduce_rows(_p(buf), buf.shape[0] * buf.shape[1], c, _p(out), _stream()), "unet_reduce_rows")
    return out


_CONST_CACHE = {}


def _const_nc(n: int, c: int, device, value: float) -> Tensor:
    key = (n, c, str(device), value)
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.full((n, c), value, dtype=torch.float32, device=device)
        _CONST_CACHE[key] = t
    return t


def rowstats(y: Tensor) -> Tuple[Tensor, Tensor]:
    """per-(window, channel) sum and sum of squares over rows -> two [N, C] fp64 tensors"""
    lib = _lib.load()
    s1, s2 = _part_buf(y, 2)
    check(lib.unet_rowstats(_dt(y), _p(y), _p(s1), _p(s2), *_dims(y), _stream()), "unet_rowstats")
    return s1.double().sum(dim=1), s2.double().sum(dim=1)


class AffineActFn(torch.autograd.Function):
    """z = act(a*y + b) with per-(window, channel) a, b; optional pooled sum over rows.

    Backward uses a "normalisation set" description so one pair of kernels serves BatchNorm
    (train/eval), InstanceNorm, GroupNorm and plain activations:
        dy = ca*g + cb*y + cc,   g = (dz + dpool) * act'(a*y + b)
    with ca/cb/cc computed from the two reductions sum_t g and sum_t g*y.
    """

    @staticmethod
    def forward(ctx, y, gamma, beta, stats, act_id: int, want_pool: bool):
        # stats: None (identity affine) or dict(mean[N,C], rstd[N,C], mode, groups, count, coupled)
        lib = _lib.load()
        _need_cuda(y)
        y = y.contiguous()
        n, rows, c = y.shape
        dev = y.device
        if stats is None:
            a = _const_nc(n, c, dev, 1.0)
            b = _const_nc(n, c, dev, 0.0)
        else:
            g = gamma.detach().double() if gamma is not None else torch.ones(c, dtype=torch.float64, device=dev)
            bt = beta.detach().double() if beta is not None else torch.zeros(c, dtype=torch.float64, device=dev)
            a64 = stats["rstd"] * g[None, :]
            a = a64.float().contiguous()
            b = (bt[None, :] - stats["mean"] * a64).float().contiguous()
        z = torch.empty_like(y)
        pool = None
        pbuf = None
        if want_pool:
            (pbuf,) = _part_buf(y)
        check(lib.unet_affine_act_fwd(_dt(y), _p(y), _p(a), _p(b), act_id, _p(z), _p(pbuf), n, rows, c, _stream()),
              "unet_affine_act_fwd")
        if want_pool:
            pool = sum_parts(pbuf)
        ctx.stats, ctx.act_id, ctx.want_pool = stats, act_id, want_pool
        ctx.has_affine = gamma is not None
        ctx.save_for_backward(y, a, b, gamma)
        if want_pool:
            return z, pool
        return z

    @staticmethod
    def backward(ctx, dz, dpool=None):
        lib = _lib.load()
        y, a, b, gamma = ctx.saved_tensors
        stats = ctx.stats
        n, rows, c = y.shape
        dz = dz.contiguous()
        dp = _f32c(dpool) if (ctx.want_pool and dpool is not None) else None
        coupled = stats is not None and stats["coupled"]
        need_param = ctx.has_affine and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dgamma = dbeta = None
        if coupled or need_param:
            g1p, g2p = _part_buf(y, 2)
            check(lib.unet_affine_act_bwd_reduce(_dt(y), _p(y), _p(dz), _p(dp), _p(a), _p(b), ctx.act_id, _p(g1p),
                                                 _p(g2p), n, rows, c, _stream()), "unet_affine_act_bwd_reduce")
            g1 = g1p.double().sum(dim=1)
            g2 = g2p.double().sum(dim=1)
        if stats is None:
            ca, cb, cc = a, b, b  # a = 1, b = 0 constants
        else:
            mean, rstd = stats["mean"], stats["rstd"]
            gm = gamma.detach().double() if gamma is not None else torch.ones(c, dtype=torch.float64, device=y.device)
            if coupled or need_param:
                gy = rstd * (g2 - mean * g1)  # sum_t g * yhat
            if need_param:
                dgamma = gy.sum(dim=0).to(gamma.dtype)
                dbeta = g1.sum(dim=0).to(gamma.dtype)
            if coupled:
                red = stats["reduce"]
                cnt = stats["count"]
                m1 = red(gm[None, :] * g1) / cnt
                m2 = red(gm[None, :] * gy) / cnt
                ca = a
                cb = (-rstd * rstd * m2).float().contiguous()
                cc = (-rstd * m1 + rstd * rstd * mean * m2).float().contiguous()
            else:
                ca, cb, cc = a, torch.zeros_like(a), torch.zeros_like(a)
        dy = None
        if ctx.needs_input_grad[0]:
            dy = torch.empty_like(y)
            (bsum,) = _part_buf(y)
            check(lib.unet_affine_act_bwd_apply_sum(_dt(y), _p(y), _p(dz), _p(dp), _p(a), _p(b), ctx.act_id, _p(ca),
                                                    _p(cb), _p(cc), _p(dy), _p(bsum), n, rows, c, _stream()),
                  "unet_affine_act_bwd_apply_sum")
            _offer_colsum(dy, bsum)
        return dy, dgamma, dbeta, None, None, None


def _counter_ok(t: Optional[Tensor]) -> Optional[Tensor]:
    """BatchNorm's ``num_batches_tracked`` (int64 scalar on the GPU), incremented by the finalisation kernel itself
    instead of a launch of its own per norm layer and step."""
    if t is None:
        return None
    if t.dtype != torch.int64 or t.numel() != 1 or not t.is_cuda:
        raise TypeError("num_batches_tracked must be an int64 scalar on the GPU")
    return t


class BatchNormActFn(torch.autograd.Function):
    """z = act(BatchNorm1d_train(y)) (+ pooled sum of z): statistics, running-buffer update and the
    affine coefficients come from ONE finalisation kernel (unet_bn_finalize_fwd); backward likewise."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, momentum: float, eps: float, act_id: int,
                want_pool: bool, batches_tracked: Optional[Tensor] = None):
        lib = _lib.load()
        _need_cuda(y)
        y = y.contiguous()
        n, rows, c = y.shape
        dt, st = _dt(y), _stream()
        hint = _take_stats(y)
        if hint is not None:
            s1p, s2p = hint
        else:
            s1p, s2p = _part_buf(y, 2)
            check(lib.unet_rowstats(dt, _p(y), _p(s1p), _p(s2p), n, rows, c, st), "unet_rowstats")
        coef = torch.empty((4, n, c), dtype=torch.float32, device=y.device)
        mean, rstd, a, b = coef[0], coef[1], coef[2], coef[3]
        gf = _f32c(gamma) if gamma is not None else None
        bf = _f32c(beta) if beta is not None else None
        check(lib.unet_bn_finalize_fwd(_p(s1p), _p(s2p), n, s1p.shape[1], c, rows, eps, _p(gf), _p(bf),
                                       _p(running_mean), _p(running_var), momentum, _p(mean), _p(rstd), _p(a), _p(b),
                                       _p(_counter_ok(batches_tracked)), st), "unet_bn_finalize_fwd")
        z = torch.empty_like(y)
        pbuf = (s1p if hint is None else _part_buf(y)[0]) if want_pool else None  # own partial buffer no longer needed: reuse
        check(lib.unet_affine_act_fwd(dt, _p(y), _p(a), _p(b), act_id, _p(z), _p(pbuf), n, rows, c, st),
              "unet_affine_act_fwd")
        ctx.act_id, ctx.want_pool = act_id, want_pool
        ctx.has_affine = gamma is not None
        ctx.gdtype = gamma.dtype if gamma is not None else torch.float32
        ctx.save_for_backward(y, coef, gf)
        if want_pool:
            return z, sum_parts(pbuf)
        return z

    @staticmethod
    def backward(ctx, dz, dpool=None):
        lib = _lib.load()
        y, coef, gf = ctx.saved_tensors
        mean, rstd, a, b = coef[0], coef[1], coef[2], coef[3]
        n, rows, c = y.shape
        dt, st = _dt(y), _stream()
        dz = dz.contiguous()
        dp = _f32c(dpool) if (ctx.want_pool and dpool is not None) else None
        g1p, g2p = _part_buf(y, 2)
        check(lib.unet_affine_act_bwd_reduce(dt, _p(y), _p(dz), _p(dp), _p(a), _p(b), ctx.act_id, _p(g1p), _p(g2p),
                                             n, rows, c, st), "unet_affine_act_bwd_reduce")
        cf = torch.empty((3, n, c), dtype=torch.float32, device=y.device)
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device)
        check(lib.unet_bn_finalize_bwd(_p(g1p), _p(g2p), n, g1p.shape[1], c, rows, _p(mean), _p(rstd), _p(gf),
                                       _p(cf[0]), _p(cf[1]), _p(cf[2]), _p(dgb[0]), _p(dgb[1]), st),
              "unet_bn_finalize_bwd")
        dy = torch.empty_like(y)
        check(lib.unet_affine_act_bwd_apply_sum(dt, _p(y), _p(dz), _p(dp), _p(a), _p(b), ctx.act_id, _p(cf[0]), _p(cf[1]),
                                                _p(cf[2]), _p(dy), _p(g1p), n, rows, c, st), "unet_affine_act_bwd_apply_sum")
        _offer_colsum(dy, g1p)       # g1p is free again: bn_finalize_bwd has consumed it
        dgamma = dgb[0].to(ctx.gdtype) if ctx.has_affine else None
        dbeta = dgb[1].to(ctx.gdtype) if ctx.has_affine else None
        return dy, dgamma, dbeta, None, None, None, None, None, None, None


def se_gate(pool_sum: Tensor, w0: Tensor, w2: Tensor, rows: int) -> Tensor:
    """sigmoid(W2 gelu(W0 (pool_sum / rows))): the SE excitation MLP (models.py:660-661), two launches; the mean's
    1/rows is the first map's scale."""
    (h,) = small_linear_group(pool_sum, ((w0, None, "gelu", 1.0 / rows),))
    (s,) = small_linear_group(h, ((w2, None, "sigmoid", 1.0),))
    return s


class BatchNormActSEFn(torch.autograd.Function):
    """z = act(BatchNorm1d_train(y)) * s,  s = SE-MLP(mean over rows of act(BatchNorm(y)))  -- the norm, GELU and
    SqueezeExcitation1D of a ModernConvBlock (models.py:801-839, 637-662) as one autograd node.

    Forward: statistics pass, pool-only pass, MLP, fused apply pass; the unscaled activation is never stored.
    Backward: one five-sum pass over (y, dz) (csrc/stream.hip, AffineActSEBwdReduceOp), the MLP backward on [N, C],
    a combine on [N, C], BatchNorm's finalisation, one apply pass: 2 + 3 activation touches instead of the 3 + 2 + 3
    of SE-excite backward + norm reduction + norm apply."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, momentum: float, eps: float, act_id: int, w0, w2,
                batches_tracked: Optional[Tensor] = None):
        lib = _lib.load()
        _need_cuda(y)
        y = y.contiguous()
        n, rows, c = y.shape
        dt, st = _dt(y), _stream()
        hint = _take_stats(y)
        if hint is not None:
            s1p, s2p = hint
        else:
            s1p, s2p = _part_buf(y, 2)
            check(lib.unet_rowstats(dt, _p(y), _p(s1p), _p(s2p), n, rows, c, st), "unet_rowstats")
        coef = torch.empty((4, n, c), dtype=torch.float32, device=y.device)
        mean, rstd, a, b = coef[0], coef[1], coef[2], coef[3]
        gf = _f32c(gamma) if gamma is not None else None
        bf = _f32c(beta) if beta is not None else None
        check(lib.unet_bn_finalize_fwd(_p(s1p), _p(s2p), n, s1p.shape[1], c, rows, eps, _p(gf), _p(bf),
                                       _p(running_mean), _p(running_var), momentum, _p(mean), _p(rstd), _p(a), _p(b),
                                       _p(_counter_ok(batches_tracked)), st), "unet_bn_finalize_fwd")
        pbuf = s1p if hint is None else _part_buf(y)[0]
        check(lib.unet_affine_pool(dt, _p(y), _p(a), _p(b), act_id, _p(pbuf), n, rows, c, st), "unet_affine_pool")
        pool = sum_parts(pbuf)
        # the SE MLP (models.py:660-661) by raw launches; its backward runs in the middle of this node's backward
        w0f, w2f = _f32c(w0), _f32c(w2)
        gelu, sig = _lib.SMALL_ACT["gelu"], _lib.SMALL_ACT["sigmoid"]
        (h,), (hpre,) = _small_fwd(pool, (gelu,), (1.0 / rows,), [w0f], [None])
        (sd,), _ = _small_fwd(h, (sig,), (1.0,), [w2f], [None])
        z = torch.empty_like(y)
        check(lib.unet_affine_act_scale_fwd(dt, _p(y), _p(a), _p(b), act_id, _p(sd), _p(z), n, rows, c, st),
              "unet_affine_act_scale_fwd")
        ctx.act_id = act_id
        ctx.has_affine = gamma is not None
        ctx.gdtype = gamma.dtype if gamma is not None else torch.float32
        ctx.wdtypes = (w0.dtype, w2.dtype)
        ctx.save_for_backward(y, coef, gf, sd, pool, h, hpre, w0f, w2f)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        y, coef, gf, sd, pool, h, hpre, w0f, w2f = ctx.saved_tensors
        mean, rstd, a, b = coef[0], coef[1], coef[2], coef[3]
        n, rows, c = y.shape
        dt, st = _dt(y), _stream()
        dz = dz.contiguous()
        np_ = parts(rows)
        pp = torch.empty((5, n, np_, c), dtype=torch.float32, device=y.device)
        check(lib.unet_affine_act_se_bwd_reduce(dt, _p(y), _p(dz), _p(a), _p(b), ctx.act_id, _p(pp), n, rows, c, st),
              "unet_affine_act_se_bwd_reduce")
        ps = sum_parts(pp.view(5 * n, np_, c))                       # [5*N, C] = P0..P4 of [N, C]
        gelu, sig = _lib.SMALL_ACT["gelu"], _lib.SMALL_ACT["sigmoid"]
        dh, (dw2,), _ = _small_bwd(h, (sig,), (1.0,), [w2f], [sd], [None], [ps[:n]], True, [True], [False])
        dpool, (dw0,), _ = _small_bwd(pool, (gelu,), (1.0 / rows,), [w0f], [h], [hpre], [dh], True, [True], [False])
        g12 = torch.empty((2, n, 1, c), dtype=torch.float32, device=y.device)
        check(lib.unet_se_combine(_p(ps), _p(sd), _p(dpool), _p(g12[0]), _p(g12[1]), n, c, st), "unet_se_combine")
        cf = torch.empty((3, n, c), dtype=torch.float32, device=y.device)
        dgb = torch.empty((2, c), dtype=torch.float32, device=y.device)
        check(lib.unet_bn_finalize_bwd(_p(g12[0]), _p(g12[1]), n, 1, c, rows, _p(mean), _p(rstd), _p(gf),
                                       _p(cf[0]), _p(cf[1]), _p(cf[2]), _p(dgb[0]), _p(dgb[1]), st),
              "unet_bn_finalize_bwd")
        dy = torch.empty_like(y)
        bsum = pp[0]                                                 # consumed by sum_parts: free for the column sums
        check(lib.unet_affine_act_se_bwd_apply(dt, _p(y), _p(dz), _p(dpool), _p(sd), _p(a), _p(b), ctx.act_id, _p(cf[0]),
                                               _p(cf[1]), _p(cf[2]), _p(dy), _p(bsum), n, rows, c, st),
              "unet_affine_act_se_bwd_apply")
        _offer_colsum(dy, bsum)
        dgamma = dgb[0].to(ctx.gdtype) if ctx.has_affine else None
        dbeta = dgb[1].to(ctx.gdtype) if ctx.has_affine else None
        return dy, dgamma, dbeta, None, None, None, None, None, dw0.to(ctx.wdtypes[0]), dw2.to(ctx.wdtypes[1]), None


def norm_act_se_inference(y: Tensor, a: Tensor, b: Tensor, act_id: int, w0: Tensor, w2: Tensor) -> Tensor:
    """act(a*y+b) * SE-gate for inference (no autograd): pool-only pass, the SE MLP, one fused apply pass -- 3 activation
    touches instead of the 4 of norm+act(+pool) followed by the SE scale.  a, b: [N, C] fp32."""
    lib = _lib.load()
    _need_cuda(y)
    y = y.contiguous()
    n, rows, c = y.shape
    dt, st = _dt(y), _stream()
    (pbuf,) = _part_buf(y)
    check(lib.unet_affine_pool(dt, _p(y), _p(a), _p(b), act_id, _p(pbuf), n, rows, c, st), "unet_affine_pool")
    pool = sum_parts(pbuf)
    gelu, sig = _lib.SMALL_ACT["gelu"], _lib.SMALL_ACT["sigmoid"]
    (h,), _ = _small_fwd(pool, (gelu,), (1.0 / rows,), [_f32c(w0)], [None])
    (sd,), _ = _small_fwd(h, (sig,), (1.0,), [_f32c(w2)], [None])
    z = torch.empty_like(y)
    check(lib.unet_affine_act_scale_fwd(dt, _p(y), _p(a), _p(b), act_id, _p(sd), _p(z), n, rows, c, st),
          "unet_affine_act_scale_fwd")
    return z


def activation(x: Tensor, act_id: int) -> Tensor:
    return AffineActFn.apply(x, None, None, None, act_id, False)


class ScaleFn(torch.autograd.Function):
    """out = x * s[n, c]  (SE excite, models.py:662)"""

    @staticmethod
    def forward(ctx, x, s):
        lib = _lib.load()
        _need_cuda(x)
        x = x.contiguous()
        sf = _f32c(s)
        out = torch.empty_like(x)
        check(lib.unet_scale_fwd(_dt(x), _p(x), _p(sf), _p(out), *_dims(x), _stream()), "unet_scale_fwd")
        ctx.save_for_backward(x, sf)
        ctx.s_dtype = s.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, sf = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        (dsp,) = _part_buf(x)
        check(lib.unet_scale_bwd(_dt(x), _p(x), _p(dout), _p(sf), _p(dx), _p(dsp), *_dims(x), _stream()),
              "unet_scale_bwd")
        return dx, sum_parts(dsp).to(ctx.s_dtype)


class FilmSoftmaxFn(torch.autograd.Function):
    """h = softmax_c(q * kk[n, :]) * v[n, :]  (models.py:514-515; kk already holds C^-1/2)"""

    @staticmethod
    def forward(ctx, q, kk, v):
        lib = _lib.load()
        _need_cuda(q)
        q = q.contiguous()
        kf, vf = _f32c(kk), _f32c(v)
        h = torch.empty_like(q)
        check(lib.unet_film_softmax_fwd(_dt(q), _p(q), _p(kf), _p(vf), _p(h), *_dims(q), _stream()),
              "unet_film_softmax_fwd")
        ctx.save_for_backward(q, kf, vf)
        ctx.kdt, ctx.vdt = kk.dtype, v.dtype
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        q, kf, vf = ctx.saved_tensors
        dh = dh.contiguous()
        dq = torch.empty_like(q)
        dkp, dvp = _part_buf(q, 2)
        check(lib.unet_film_softmax_bwd(_dt(q), _p(q), _p(dh), _p(kf), _p(vf), _p(dq), _p(dkp), _p(dvp), _p(None), *_dims(q),
                                        _stream()), "unet_film_softmax_bwd")
        return dq, sum_parts(dkp).to(ctx.kdt), sum_parts(dvp).to(ctx.vdt)


class GatedAddFn(torch.autograd.Function):
    """out = res + g[n, c] * o  (models.py:517)"""

    @staticmethod
    def forward(ctx, res, o, g):
        lib = _lib.load()
        _need_cuda(o)
        res, o = res.contiguous(), o.contiguous()
        gf = _f32c(g)
        out = torch.empty_like(o)
        check(lib.unet_gated_add_fwd(_dt(o), _p(res), _p(o), _p(gf), _p(out), *_dims(o), _stream()),
              "unet_gated_add_fwd")
        ctx.save_for_backward(o, gf)
        ctx.gdt = g.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        o, gf = ctx.saved_tensors
        dout = dout.contiguous()
        d_o = torch.empty_like(o)
        (dgp,) = _part_buf(o)
        check(lib.unet_gated_add_bwd(_dt(o), _p(o), _p(dout), _p(gf), _p(d_o), _p(dgp), _p(None), *_dims(o), _stream()),
              "unet_gated_add_bwd")
        return dout, d_o, sum_parts(dgp).to(ctx.gdt)


class FilmFn(torch.autograd.Function):
    """FiLM.forward (models.py:504-517) as ONE autograd node:
        q = Wq z + bq;  h = softmax_c(q * kk) * v;  o = Wo h + bo;  out = z + g * o
    Same kernels as the separate ops; what the fusion buys is the backward: the gradient of the residual
    path (dout) enters dz through the epilogue of q_proj's data-gradient conv, so autograd never runs a
    separate full-tensor add, and q / h / o are kept exactly once."""

    @staticmethod
    def forward(ctx, z, wq, bq, wo, bo, kk, v, g, differentiated=None, cache_params=None):
        """``differentiated``: will this call be back-propagated (``film`` below passes it)?  It cannot be read in
        here: grad mode is always off inside an autograd Function's forward, and ``ctx.needs_input_grad`` only repeats
        the inputs' ``requires_grad`` flags -- True for the parameters under ``torch.no_grad()`` as well.
        ``cache_params``: the Parameter objects the weights derive from (``wq`` / ``wo`` are usually fresh permuted
        views, which cannot key a cache); with them an inference call reuses the packed images."""
        lib = _lib.load()
        _need_cuda(z)
        z = _nlc_ok(z)
        dt, st = _dt(z), _stream()
        n, rows, c = z.shape
        g1 = geom_conv(1)
        wqf, wof = wq.detach().float(), wo.detach().float()          # [1, C, C] views of the 1x1 conv weights
        kf, vf, gf = _f32c(kk), _f32c(v), _f32c(g)
        pk = lambda wf: pack_weights(wf, dt, transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=c, c1=c, c2=0)
        ctx.wp_dq = ctx.wp_do = None
        # a differentiated (training) call must never take the inference cache: the optimizer kernels update the
        # weights without bumping their version counters
        training = any(ctx.needs_input_grad) if differentiated is None else bool(differentiated)
        if not training and _cacheable(cache_params):
            wqp, bq32, wop, bo32 = packed_for_inference(("film", c), tuple(cache_params), dt,
                                                        lambda: (pk(wqf), _f32c(bq), pk(wof), _f32c(bo)))
        elif DUAL_PACK and training and _lib.PACK_MAX_JOBS >= 4:
            # the two forward images and the two transposed images backward will read, in ONE launch (a pack is
            # launch-bound: four 5 us launches per block and step otherwise)
            fj = dict(transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=c, c1=c, c2=0)
            wqp, wop, ctx.wp_dq, ctx.wp_do = pack_weights_multi(wqf, dt, [
                dict(fj, w=wqf), dict(fj, w=wof), dict(_dgrad_pack_job(g1, wqf, 0, c), w=wqf),
                dict(_dgrad_pack_job(g1, wof, 0, c), w=wof)])
            bq32, bo32 = _f32c(bq), _f32c(bo)
        else:
            wqp, bq32, wop, bo32 = pk(wqf), _f32c(bq), pk(wof), _f32c(bo)
        q = conv_forward_raw(g1, z, None, wqp, bq32, None, rows, c)
        h = torch.empty_like(q)
        check(lib.unet_film_softmax_fwd(dt, _p(q), _p(kf), _p(vf), _p(h), n, rows, c, st), "unet_film_softmax_fwd")
        out = torch.empty_like(z)
        o, done = conv_forward_gated_raw(g1, h, wop, bo32, z, gf, out, keep_o=training) if FILM_GATED_EPILOGUE \
            else (None, False)
        if not done:
            o = conv_forward_raw(g1, h, None, wop, bo32, None, rows, c)
            check(lib.unet_gated_add_fwd(dt, _p(z), _p(o), _p(gf), _p(out), n, rows, c, st), "unet_gated_add_fwd")
        ctx.save_for_backward(z, q, h, o, wqf, wof, kf, vf, gf)
        ctx.dtypes = (wq.dtype, bq.dtype, wo.dtype, bo.dtype, kk.dtype, v.dtype, g.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        z, q, h, o, wqf, wof, kf, vf, gf = ctx.saved_tensors
        dout = dout.contiguous()
        dt, st = _dt(z), _stream()
        n, rows, c = z.shape
        g1 = geom_conv(1)
        d_o = torch.empty_like(o)
        dgp, dop = _part_buf(o, 2)
        check(lib.unet_gated_add_bwd(dt, _p(o), _p(dout), _p(gf), _p(d_o), _p(dgp), _p(dop), n, rows, c, st),
              "unet_gated_add_bwd")
        dwo = conv_wgrad_raw(g1, h, None, d_o, c)
        dh = conv_dgrad_raw(g1, wof, d_o, rows, 0, c, wp=ctx.wp_do)
        dq = torch.empty_like(q)
        dkp, dvp, dqp = _part_buf(q, 3)
        check(lib.unet_film_softmax_bwd(dt, _p(q), _p(dh), _p(kf), _p(vf), _p(dq), _p(dkp), _p(dvp), _p(dqp), n, rows, c,
                                        st), "unet_film_softmax_bwd")
        dwq = conv_wgrad_raw(g1, z, None, dq, c)
        dz = conv_dgrad_raw(g1, wqf, dq, rows, 0, c, res=dout, wp=ctx.wp_dq) if ctx.needs_input_grad[0] else None
        # the five sums over partials this backward ends in, in one launch: the bias gradients of out_proj and q_proj
        # (column sums of d_o and dq, taken in the passes above) and the gradients of the embedding maps kk, v, g
        dbo, dbq, dk, dv, dg = reduce_multi((dop, dqp, dkp, dvp, dgp), (False, False, True, True, True))
        t = ctx.dtypes
        return (dz, dwq.to(t[0]), dbq.to(t[1]), dwo.to(t[2]), dbo.to(t[3]), dk.to(t[4]), dv.to(t[5]), dg.to(t[6]), None, None)


def film(z: Tensor, wq: Tensor, bq: Tensor, wo: Tensor, bo: Tensor, kk: Tensor, v: Tensor, g: Tensor,
         params: Optional[Sequence[Tensor]] = None) -> Tensor:
    """``FilmFn`` with the one fact its forward cannot find out itself: whether autograd is recording this call.
    ``params``: the parameters behind ``wq, bq, wo, bo`` (see ``FilmFn.forward``)."""
    diff = torch.is_grad_enabled() and any(t.requires_grad for t in (z, wq, bq, wo, bo, kk, v, g))
    return FilmFn.apply(z, wq, bq, wo, bo, kk, v, g, diff, tuple(params) if params is not None else None)


# ---------------------------------------------------------------------------------------------
# per-window vector algebra: groups of small Linear layers (csrc/small.hip)
# ---------------------------------------------------------------------------------------------
def _small_fwd(xf: Tensor, acts, scales, ws, bs):
    """Launch of unet_small_linear_fwd: xf [N, I] fp32, ws / bs fp32 (bias may be None) -> (ys, pres)."""
    lib = _lib.load()
    np_ = len(acts)
    n, n_in = xf.shape
    ys, pres = [], []
    splits = int(lib.unet_small_linear_splits(n_in))
    parts_ = []          # split-K partial sums of wide maps, consumed by the finishing launch
    probs = (_lib.SmallLinear * np_)()
    for i in range(np_):
        n_out = ws[i].shape[0]
        if ws[i].shape[1] != n_in:
            raise ValueError(f"weight {tuple(ws[i].shape)} does not take {n_in} inputs")
        y = torch.empty((n, n_out), dtype=torch.float32, device=xf.device)
        pre = torch.empty_like(y) if acts[i] == _lib.SMALL_ACT["gelu"] else None
        q = probs[i]
        q.x, q.w, q.b, q.y, q.pre = xf.data_ptr(), ws[i].data_ptr(), (bs[i].data_ptr() if bs[i] is not None else None), \
            y.data_ptr(), (pre.data_ptr() if pre is not None else None)
        q.n_in, q.n_out, q.act, q.scale = n_in, n_out, acts[i], float(scales[i])
        if splits > 1:
            parts_.append(torch.empty((splits, n, n_out), dtype=torch.float32, device=xf.device))
            q.part = parts_[-1].data_ptr()
        ys.append(y)
        pres.append(pre)
    check(lib.unet_small_linear_fwd(probs, np_, n, _stream()), "unet_small_linear_fwd")
    return ys, pres


def _small_bwd(xf: Tensor, acts, scales, ws, ys, pres, dys, need_x: bool, need_w, need_b):
    """Launch of unet_small_linear_bwd -> (dx | None, dws, dbs), all fp32."""
    lib = _lib.load()
    np_ = len(acts)
    n, n_in = xf.shape
    tiles = [int(lib.unet_small_linear_tiles(w.shape[0])) for w in ws]
    dx_part = torch.empty((sum(tiles), n * n_in), dtype=torch.float32, device=xf.device) if need_x else None
    probs = (_lib.SmallLinear * np_)()
    keep, dws, dbs, slot = [], [], [], 0
    for i in range(np_):
        n_out = ws[i].shape[0]
        dy = dys[i]
        dy = torch.zeros((n, n_out), dtype=torch.float32, device=xf.device) if dy is None else _f32c(dy)
        pre = pres[i]
        dw = torch.empty_like(ws[i]) if need_w[i] else None
        db = torch.empty(n_out, dtype=torch.float32, device=xf.device) if need_b[i] else None
        q = probs[i]
        q.x, q.w, q.b, q.y, q.pre = xf.data_ptr(), ws[i].data_ptr(), None, ys[i].data_ptr(), (pre.data_ptr() if pre is not None else None)
        q.n_in, q.n_out, q.act, q.scale = n_in, n_out, acts[i], float(scales[i])
        q.dy = dy.data_ptr()
        q.dw = dw.data_ptr() if dw is not None else None
        q.db = db.data_ptr() if db is not None else None
        q.dx_part = dx_part.data_ptr() if need_x else None
        q.dx_slot0 = slot
        slot += tiles[i]
        keep.append(dy)
        dws.append(dw)
        dbs.append(db)
    check(lib.unet_small_linear_bwd(probs, np_, n, _stream()), "unet_small_linear_bwd")
    dx = None
    if need_x:
        dx = torch.empty((n, n_in), dtype=torch.float32, device=xf.device)
        check(lib.unet_reduce_rows(_p(dx_part), dx_part.shape[0], n * n_in, _p(dx), _stream()), "unet_reduce_rows")
    return dx, dws, dbs


class SmallLinearGroupFn(torch.autograd.Function):
    """y_p = act_p(scale_p * (x W_p^T + b_p)) for up to 4 Linear maps of ONE input x [N, I]: FiLM's k_proj / v_proj /
    gate on the conditioning embedding (models.py:506-509) and the SE excitation MLP (models.py:660-661).  One launch
    forward, one backward (+ the sum of the input-gradient partials), fp32."""

    @staticmethod
    def forward(ctx, x, acts: Tuple[int, ...], scales: Tuple[float, ...], *wb):
        _need_cuda(x)
        np_ = len(acts)
        if np_ < 1 or np_ > _lib.SMALL_MAX or len(wb) != 2 * np_:
            raise ValueError(f"1..{_lib.SMALL_MAX} (weight, bias) pairs")
        xf = _f32c(x)
        ws = [_f32c(wb[2 * i]) for i in range(np_)]
        bs = [(_f32c(wb[2 * i + 1]) if wb[2 * i + 1] is not None else None) for i in range(np_)]
        ys, pres = _small_fwd(xf, acts, scales, ws, bs)
        ctx.acts, ctx.scales, ctx.np = acts, scales, np_
        ctx.has_b = [b is not None for b in bs]
        ctx.has_pre = [p is not None for p in pres]
        ctx.dtypes = (x.dtype,) + tuple(t.dtype if t is not None else None for t in wb)
        ctx.save_for_backward(xf, *ws, *ys, *[p for p in pres if p is not None])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        np_ = ctx.np
        saved = ctx.saved_tensors
        xf, ws, ys = saved[0], saved[1:1 + np_], saved[1 + np_:1 + 2 * np_]
        pres_it = iter(saved[1 + 2 * np_:])
        pres = [next(pres_it) if ctx.has_pre[i] else None for i in range(np_)]
        need_w = [ctx.needs_input_grad[3 + 2 * i] for i in range(np_)]
        need_b = [ctx.has_b[i] and ctx.needs_input_grad[4 + 2 * i] for i in range(np_)]
        dx, dws, dbs = _small_bwd(xf, ctx.acts, ctx.scales, ws, ys, pres, dys, ctx.needs_input_grad[0], need_w, need_b)
        out = [dx.to(ctx.dtypes[0]) if dx is not None else None, None, None]
        for i in range(np_):
            wd, bd = ctx.dtypes[1 + 2 * i], ctx.dtypes[2 + 2 * i]
            out.append(dws[i].to(wd) if dws[i] is not None else None)
            out.append(dbs[i].to(bd) if dbs[i] is not None else None)
        return tuple(out)


def small_linear_group(x: Tensor, maps) -> Tuple[Tensor, ...]:
    """maps: sequence of (weight [O, I], bias [O] | None, activation "none" | "gelu" | "sigmoid", scale)."""
    out = ()
    for i in range(0, len(maps), _lib.SMALL_MAX):
        grp = maps[i:i + _lib.SMALL_MAX]
        acts = tuple(_lib.SMALL_ACT[m[2]] for m in grp)
        scales = tuple(float(m[3]) for m in grp)
        wb = []
        for m in grp:
            wb += [m[0], m[1]]
        out += SmallLinearGroupFn.apply(x, acts, scales, *wb)
    return out


# ---------------------------------------------------------------------------------------------
# model boundary: NCL <-> NLC with the z-score of models.py:3235 fused into the way in
# ---------------------------------------------------------------------------------------------
def ncl_moments(x: Tensor, eps: float) -> Tuple[Tensor, Tensor]:
    lib = _lib.load()
    n, c, t = x.shape
    mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(lib.unet_ncl_moments(_dt(x), _p(x), _p(mean), _p(rstd), n, c, t, eps, _stream()), "unet_ncl_moments")
    return mean, rstd


class ZScoreInFn(torch.autograd.Function):
    """x_nlc[n, t, c] = (ob[n, c, t] - mean) / (std_unbiased + 1e-8)   in the compute dtype"""

    @staticmethod
    def forward(ctx, ob, out_dtype):
        lib = _lib.load()
        _need_cuda(ob)
        ob = ob.contiguous()
        n, c, t = ob.shape
        mean, rstd = ncl_moments(ob, 1e-8)
        x = torch.empty((n, t, c), dtype=out_dtype, device=ob.device)
        check(lib.unet_ncl_to_nlc(_dt(ob), _dt(x), _p(ob), _p(mean), _p(rstd), _p(x), n, c, t, _stream()),
              "unet_ncl_to_nlc")
        ctx.save_for_backward(x, rstd)
        ctx.in_dtype = ob.dtype
        return x

    @staticmethod
    def backward(ctx, dx):
        lib = _lib.load()
        x, rstd = ctx.saved_tensors
        dx = dx.contiguous()
        n, t, c = x.shape
        ones = torch.ones((n, c), dtype=torch.float32, device=x.device)
        zeros = torch.zeros_like(ones)
        g1p, g2p = _part_buf(x, 2)
        check(lib.unet_affine_act_bwd_reduce(_dt(x), _p(x), _p(dx), _p(None), _p(ones), _p(zeros), 0, _p(g1p), _p(g2p),
                                             n, t, c, _stream()), "unet_affine_act_bwd_reduce")
        g1 = g1p.double().sum(dim=1)
        g2 = g2p.double().sum(dim=1)          # sum_t g * xhat
        r = rstd.double()
        sigma = 1.0 / r - 1e-8
        ga = rstd
        gb = (-g2 / (max(t - 1, 1) * sigma)).float().contiguous()
        be = (-r * g1 / t).float().contiguous()
        dob = torch.empty((n, c, t), dtype=ctx.in_dtype, device=x.device)
        check(lib.unet_nlc_to_ncl(_dt(x), _dt(dob), _p(dx), _p(x), _p(ga), _p(gb), _p(be), _p(dob), n, c, t, _stream()),
              "unet_nlc_to_ncl")
        return dob, None


class ToNCLFn(torch.autograd.Function):
    """plain layout change NLC -> NCL (the model's output side)"""

    @staticmethod
    def forward(ctx, x, out_dtype):
        lib = _lib.load()
        _need_cuda(x)
        x = x.contiguous()
        n, t, c = x.shape
        y = torch.empty((n, c, t), dtype=out_dtype, device=x.device)
        check(lib.unet_nlc_to_ncl(_dt(x), _dt(y), _p(x), _p(None), _p(None), _p(None), _p(None), _p(y), n, c, t,
                                  _stream()), "unet_nlc_to_ncl")
        ctx.x_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        n, c, t = dy.shape
        dx = torch.empty((n, t, c), dtype=ctx.x_dtype, device=dy.device)
        check(lib.unet_ncl_to_nlc(_dt(dy), _dt(dx), _p(dy), _p(None), _p(None), _p(dx), n, c, t, _stream()),
              "unet_ncl_to_nlc")
        return dx, None


class ToNCLAffineFn(torch.autograd.Function):
    """y[n, c, t] = gamma[n, c] * x[n, t, c] + beta[n, c]: the model's output-side layout change with
    SessionAdaptiveScaling's per-(window, channel) affine (models.py:2068) applied on the way -- one pass instead of a
    transpose followed by torch's broadcast multiply-add over [N, C, T]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, out_dtype):
        lib = _lib.load()
        _need_cuda(x)
        x = x.contiguous()
        n, t, c = x.shape
        ga, be = _f32c(gamma), _f32c(beta)
        if ga.shape != (n, c) or be.shape != (n, c):
            raise ValueError(f"gamma / beta must be [{n}, {c}], got {tuple(ga.shape)} / {tuple(be.shape)}")
        y = torch.empty((n, c, t), dtype=out_dtype, device=x.device)
        check(lib.unet_nlc_to_ncl(_dt(x), _dt(y), _p(x), _p(None), _p(ga), _p(None), _p(be), _p(y), n, c, t,
                                  _stream()), "unet_nlc_to_ncl")
        ctx.save_for_backward(x, ga)
        ctx.io = (gamma.dtype, beta.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, ga = ctx.saved_tensors
        dy = dy.contiguous()
        n, t, c = x.shape
        g = torch.empty((n, t, c), dtype=x.dtype, device=dy.device)         # dy in the kernels' layout
        check(lib.unet_ncl_to_nlc(_dt(dy), _dt(g), _p(dy), _p(None), _p(None), _p(g), n, c, t, _stream()),
              "unet_ncl_to_nlc")
        dgamma = dbeta = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # sum_t g and sum_t g * x per (window, channel): the two-sum reduction of the norm backward with the
            # identity affine (a = 1, b = 0, no activation)
            ones = torch.ones((n, c), dtype=torch.float32, device=x.device)
            zeros = torch.zeros_like(ones)
            g1p, g2p = _part_buf(x, 2)
            check(lib.unet_affine_act_bwd_reduce(_dt(x), _p(x), _p(g), _p(None), _p(ones), _p(zeros), 0, _p(g1p),
                                                 _p(g2p), n, t, c, _stream()), "unet_affine_act_bwd_reduce")
            dbeta = sum_parts(g1p).to(ctx.io[1])
            dgamma = sum_parts(g2p).to(ctx.io[0])
        if ctx.needs_input_grad[0]:
            dx = g * ga.unsqueeze(1).to(g.dtype)
        return dx, dgamma, dbeta, None


class ToNLCFn(torch.autograd.Function):
    """plain layout change NCL -> NLC (used when a block is called directly with an NCL tensor)"""

    @staticmethod
    def forward(ctx, x, out_dtype):
        lib = _lib.load()
        _need_cuda(x)
        x = x.contiguous()
        n, c, t = x.shape
        y = torch.empty((n, t, c), dtype=out_dtype, device=x.device)
        check(lib.unet_ncl_to_nlc(_dt(x), _dt(y), _p(x), _p(None), _p(None), _p(y), n, c, t, _stream()),
              "unet_ncl_to_nlc")
        ctx.x_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        n, t, c = dy.shape
        dx = torch.empty((n, c, t), dtype=ctx.x_dtype, device=dy.device)
        check(lib.unet_nlc_to_ncl(_dt(dy), _dt(dx), _p(dy), _p(None), _p(None), _p(None), _p(None), _p(dx), n, c, t,
                                  _stream()), "unet_nlc_to_ncl")
        return dx, None


# ---------------------------------------------------------------------------------------------
# band-pair attention of CrossFrequencyCouplingAttentionV2 (csrc/attn.hip)
# ---------------------------------------------------------------------------------------------
def cfc_attn_forward_raw(q: Tensor, k: Tensor, v: Tensor, coupling: Tensor, n_heads: int, scale: float, keep: bool):
    """q, k, v: [nb, N, L, C] bf16; coupling: [nb, nb] fp32 (softmaxed).  -> out [N, L, nb*C] (+ o_pairs, lse when
    ``keep``: what backward needs)."""
    for t in (q, k, v, coupling):
        _need_cuda(t)
    nb, n, l, c = q.shape
    if q.dtype not in (torch.bfloat16, torch.float32) or k.dtype != q.dtype or v.dtype != q.dtype or \
            not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
        raise TypeError("cfc attention takes contiguous [nb, N, L, C] stacks, all bfloat16 (MFMA flash kernels) or all "
                        "float32 (exact-fp32 kernels)")
    coupling = coupling.float().contiguous()
    out = torch.empty((n, l, nb * c), dtype=q.dtype, device=q.device)
    # the pair outputs are stored also when nothing is kept for backward: the kernel variant that sums them in
    # registers instead is slower (see include/unet_hip.h) and rounds differently from the training path
    o_pairs = torch.empty((nb, nb, n, l, c), dtype=q.dtype, device=q.device)
    lse = torch.empty((nb, nb, n, n_heads, l), dtype=torch.float32, device=q.device) if keep else None
    check(_lib.load().unet_cfc_attn_fwd(_dt(q), _p(q), _p(k), _p(v), _p(coupling), n, l, c, n_heads, nb, float(scale), _p(out),
                                  _p(o_pairs), _p(lse), _stream()), "unet_cfc_attn_fwd")
    return out, o_pairs, lse


class CfcAttnFn(torch.autograd.Function):
    """out[n, l, i*C + h*hd + d] = sum_j coupling[i, j] * softmax(q_i k_j^T * scale) v_j  (models.py:431-462)."""

    @staticmethod
    def forward(ctx, q, k, v, coupling, n_heads, scale):
        keep = any(ctx.needs_input_grad[:4])
        coupling = coupling.float().contiguous()
        out, o_pairs, lse = cfc_attn_forward_raw(q, k, v, coupling, n_heads, scale, keep)
        if keep:
            ctx.save_for_backward(q, k, v, coupling, o_pairs, lse)
        ctx.n_heads, ctx.scale = n_heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, coupling, o_pairs, lse = ctx.saved_tensors
        nb, n, l, c = q.shape
        lib = _lib.load()
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        parts = torch.empty(lib.unet_cfc_attn_bwd_partials(n, l, nb), dtype=torch.float32, device=q.device).view(nb * nb, -1)
        delta = torch.empty_like(lse)
        if dout.dtype != q.dtype:
            dout = dout.to(q.dtype)
        check(lib.unet_cfc_attn_bwd(_dt(q), _p(q), _p(k), _p(v), _p(coupling), _p(o_pairs), _p(lse), _p(dout), n, l, c,
                                    ctx.n_heads, nb, float(ctx.scale), _p(dq), _p(dk), _p(dv), _p(parts), _p(delta),
                                    _stream()), "unet_cfc_attn_bwd")
        return dq, dk, dv, parts.sum(dim=1).view(nb, nb), None, None


def cfc_attention(q: Tensor, k: Tensor, v: Tensor, coupling: Tensor, n_heads: int, scale: float) -> Tensor:
    """q, k, v: [nb, N, L, C] bf16 or fp32 stacks of the per-band projections; coupling: [nb, nb] softmaxed weights."""
    return CfcAttnFn.apply(q, k, v, coupling, n_heads, scale)


class BandProjectFn(torch.autograd.Function):
    """The per-band 1x1 q / k / v projections of the cross-frequency attention (models.py:420-426 grouped Conv1d,
    models.py:193-199 per-band Conv1d) as one autograd node: bands [N, nb, L, C] -> three [nb, N, L, C] stacks
    written in place by the conv kernel (no gather), and backward sums the three data-gradients of a band in the
    conv epilogue (``res``) straight into its slice of d(bands) (no zero-fill, no adds).
    w*: [nb, C, C] (out, in), b*: [nb, C]."""

    @staticmethod
    def forward(ctx, bands, wq, bq, wk, bk, wv, bv):
        _need_cuda(bands)
        n, nb, l, c = bands.shape
        if bands.stride(3) != 1:
            bands = bands.contiguous()
        g1 = geom_conv(1)
        dt = _dt(bands)
        outs, wfs = [], []
        for w, b in ((wq, bq), (wk, bk), (wv, bv)):
            wf = w.detach().float()
            out = torch.empty((nb, n, l, c), dtype=bands.dtype, device=bands.device)
            for i in range(nb):
                wp = pack_weights(wf[i][None], dt, transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=c, c1=c, c2=0)
                conv_forward_raw(g1, bands[:, i], None, wp, _f32c(b[i]), None, l, c, out=out[i])
            outs.append(out)
            wfs.append(wf)
        ctx.save_for_backward(bands, *wfs)
        ctx.w_dtype = wq.dtype
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        bands, wfq, wfk, wfv = ctx.saved_tensors
        n, nb, l, c = bands.shape
        g1 = geom_conv(1)
        grads = [g.contiguous() for g in (dq, dk, dv)]
        need_x = ctx.needs_input_grad[0]
        dbands = torch.empty_like(bands) if need_x else None
        dws, dbs = [], []
        for p, (g, wf) in enumerate(zip(grads, (wfq, wfk, wfv))):
            dws.append(torch.stack([conv_wgrad_raw(g1, bands[:, i], None, g[i], c)[0] for i in range(nb)]).to(ctx.w_dtype))
            dbs.append(torch.stack([colsum(g[i]) for i in range(nb)]))
        if need_x:
            for i in range(nb):
                acc = conv_dgrad_raw(g1, wfq[i][None], grads[0][i], l, 0, c)
                acc = conv_dgrad_raw(g1, wfk[i][None], grads[1][i], l, 0, c, res=acc)
                conv_dgrad_raw(g1, wfv[i][None], grads[2][i], l, 0, c, res=acc, out=dbands[:, i])
        return dbands, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2]


def band_project(bands: Tensor, wq, bq, wk, bk, wv, bv):
    return BandProjectFn.apply(bands, wq, bq, wk, bk, wv, bv)
