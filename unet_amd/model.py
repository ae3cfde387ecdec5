"""CondUNet1D on hand-written CDNA4 kernels -- drop-in for the reference's model class.

Same constructor keywords, ``forward`` signature, attribute names and ``state_dict()``
key/shape layout as ``/root/reference/models.py:2865-3375`` (so checkpoints written by the
reference's ``train.py:653-657`` load with ``strict=True`` and vice versa), and the same
parameter creation order, so ``torch.manual_seed(s)`` before the constructor gives the same
initial weights.

What differs is everything underneath: parameters live in ordinary ``nn`` containers
(``nn.Conv1d`` etc. are used as *parameter holders only* and are never called); the forward
pass runs on channels-last bf16/fp32 activations through ``unet_amd.ops`` (HIP kernels via
the C ABI).  Blocks hand each other ``[N, C, L]``-shaped views of channels-last storage, so
forward hooks / ``extract_all_activations`` keep seeing the reference's tensor shapes.

Hot-path scope: conv_type standard|modern, skip add|concat, every
norm_type, cond none|cross_attn_gated, attention none|basic, scalar adaptive scaling.
Out-of-scope constructor options raise ``NotImplementedError`` (never a silent fallback).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import ACT_IDS

Tensor = torch.Tensor

BASE_CHANNELS = 128  # models.py BASE_CHANNELS
NUM_ODORS = 7        # models.py NUM_ODORS


def get_activation_id(name: str) -> int:
    """models.py:50-68 -- same names, same error."""
    if name not in ("relu", "leaky_relu", "gelu", "silu", "mish"):
        raise ValueError(f"Unknown activation: {name}. Available: ['relu', 'leaky_relu', 'gelu', 'silu', 'mish']")
    return ACT_IDS[name]


def _to_nlc(x: Tensor) -> Tensor:
    """[N, C, L]-shaped tensor -> contiguous channels-last [N, L, C] (free when x is already a
    view of channels-last storage)."""
    y = x.permute(0, 2, 1)
    if y.is_contiguous():
        return y
    if x.is_contiguous() and x.is_cuda:
        return ops.ToNLCFn.apply(x, x.dtype)
    return y.contiguous()


def _as_ncl(x: Tensor) -> Tensor:
    return x.permute(0, 2, 1)


# ---------------------------------------------------------------------------------------------
# normalisation (models.py:537-555, 801-817) + activation (+ SE squeeze) on the HIP kernels
# ---------------------------------------------------------------------------------------------
def _uses_batch_stats(norm: nn.Module) -> bool:
    """BatchNorm that normalises with the statistics of its own input: the conv in front of it can deliver them."""
    return isinstance(norm, nn.BatchNorm1d) and (norm.training or norm.running_mean is None) and torch.is_grad_enabled()


def _make_norm(norm_type: str, c: int) -> nn.Module:
    if norm_type == "instance":
        return nn.InstanceNorm1d(c, affine=True)
    if norm_type == "batch":
        return nn.BatchNorm1d(c)
    if norm_type == "group":
        g = min(8, c)
        while c % g != 0:
            g -= 1
        return nn.GroupNorm(g, c)
    if norm_type in ("layer", "rms"):
        return nn.GroupNorm(1, c)
    if norm_type == "none":
        return nn.Identity()
    raise ValueError(f"Unknown norm_type: {norm_type}. Available: batch, instance, group, layer, rms, none")


HOIST_FILM = os.environ.get("UNET_HOIST_FILM", "1") != "0"   # 0: per-block FiLM projections (A/B measurements)
FUSE_SE = os.environ.get("UNET_FUSE_SE", "1") != "0"    # 0: the unfused kernels (A/B measurements)


def _bn_eval_affine(norm: nn.BatchNorm1d, n: int):
    """Eval-mode BatchNorm as ``a * y + b``: the [N, C] fp32 coefficient arrays the row kernels take, from the running
    statistics (fp64 arithmetic).  Under no_grad they are cached like the packed weight images (``ops._PACK_CACHE``,
    revalidated against the buffers' / parameters' version counters): the dozen tiny torch launches that derive them
    are most of what a batch-1 forward used to launch per block."""
    def build():
        rstd = torch.rsqrt(norm.running_var.double() + norm.eps)
        g = norm.weight.detach().double() if norm.weight is not None else torch.ones_like(rstd)
        bt = norm.bias.detach().double() if norm.bias is not None else torch.zeros_like(rstd)
        a = (rstd * g).float()
        b = (bt - norm.running_mean.double() * rstd * g).float()
        c = a.shape[0]
        return a[None, :].expand(n, c).contiguous(), b[None, :].expand(n, c).contiguous()

    params = tuple(t for t in (norm.running_mean, norm.running_var, norm.weight, norm.bias) if t is not None)
    if ops._cacheable(params):
        return ops.packed_for_inference(("bn_eval_affine", n, float(norm.eps)), params, 0, build)
    return build()


def _norm_act_se(y: Tensor, norm: nn.Module, act_id: int, se: "SqueezeExcitation1D") -> Tensor:
    """SE(act(norm(y))) for y in NLC (models.py:801-839).  Training-mode BatchNorm -- the production path -- is one
    fused autograd node; every other norm takes the general kernels followed by the SE scale."""
    if FUSE_SE and isinstance(norm, nn.BatchNorm1d) and (norm.training or norm.running_mean is None):
        return _norm_act(y, norm, act_id, False, se=se)
    if FUSE_SE and isinstance(norm, nn.BatchNorm1d) and not torch.is_grad_enabled() and y.is_cuda:
        # eval-mode BatchNorm under no_grad (sliding-window inference, evaluate()): running statistics as the affine
        a, b = _bn_eval_affine(norm, y.shape[0])
        return ops.norm_act_se_inference(y, a, b, act_id, se.fc[0].weight, se.fc[2].weight)
    z, pool = _norm_act(y, norm, act_id, True)
    return ops.ScaleFn.apply(z, se.gate_from_pool(pool, z.shape[1]))


def _norm_act(y: Tensor, norm: nn.Module, act_id: int, want_pool: bool, se=None):
    """z = act(norm(y)) (+ per-(window, channel) sum of z over rows) for y in NLC."""
    n, rows, c = y.shape
    if isinstance(norm, nn.Identity):
        return ops.AffineActFn.apply(y, None, None, None, act_id, want_pool)
    y = y.contiguous()
    if isinstance(norm, nn.BatchNorm1d):
        use_batch = norm.training or norm.running_mean is None
        if use_batch:
            track = norm.training and norm.track_running_stats and norm.running_mean is not None
            mom = norm.momentum
            nbt = None
            if track:
                nbt = norm.num_batches_tracked
                if mom is None or nbt is None or not nbt.is_cuda:
                    with torch.no_grad():
                        norm.num_batches_tracked += 1
                    nbt = None           # counted here; otherwise the finalisation kernel increments it (no launch)
                if mom is None:  # cumulative moving average (not used by the reference; needs a host read)
                    mom = 1.0 / float(norm.num_batches_tracked)
            rm = norm.running_mean if track else None
            rv = norm.running_var if track else None
            if rm is not None and (rm.dtype != torch.float32 or not rm.is_contiguous()):
                raise TypeError("BatchNorm running buffers must be contiguous fp32")
            if se is not None:
                return ops.BatchNormActSEFn.apply(y, norm.weight, norm.bias, rm, rv, float(mom or 0.0), float(norm.eps),
                                                  act_id, se.fc[0].weight, se.fc[2].weight, nbt)
            return ops.BatchNormActFn.apply(y, norm.weight, norm.bias, rm, rv, float(mom or 0.0), float(norm.eps),
                                            act_id, want_pool, nbt)
        else:
            mean_c = norm.running_mean.double()
            rstd_c = torch.rsqrt(norm.running_var.double() + norm.eps)
            stats = dict(mean=mean_c[None, :].expand(n, c), rstd=rstd_c[None, :].expand(n, c), coupled=False)
        return ops.AffineActFn.apply(y, norm.weight, norm.bias, stats, act_id, want_pool)
    if isinstance(norm, nn.InstanceNorm1d):
        s1, s2 = ops.rowstats(y)
        mean = s1 / rows
        var = (s2 / rows - mean * mean).clamp_min(0.0)
        stats = dict(mean=mean, rstd=torch.rsqrt(var + norm.eps), coupled=True, count=float(rows),
                     reduce=lambda v: v)
        return ops.AffineActFn.apply(y, norm.weight, norm.bias, stats, act_id, want_pool)
    if isinstance(norm, nn.GroupNorm):
        g = norm.num_groups
        s1, s2 = ops.rowstats(y)
        cnt = rows * (c // g)

        def red(v, g=g, n=n, c=c):
            return v.reshape(n, g, c // g).sum(-1, keepdim=True).expand(n, g, c // g).reshape(n, c)

        mean = red(s1) / cnt
        var = (red(s2) / cnt - mean * mean).clamp_min(0.0)
        stats = dict(mean=mean, rstd=torch.rsqrt(var + norm.eps), coupled=True, count=float(cnt), reduce=red)
        return ops.AffineActFn.apply(y, norm.weight, norm.bias, stats, act_id, want_pool)
    raise TypeError(f"unsupported norm module {type(norm)}")


# ---------------------------------------------------------------------------------------------
# blocks (parameter layout = the reference's; compute = HIP ops)
# ---------------------------------------------------------------------------------------------
class SelfAttention1D(nn.Module):
    """models.py:75-97.  1x1 projections on the HIP conv kernel; the attention itself runs on the flash kernels of
    csrc/attn.hip (bf16, query/key width a multiple of 8, i.e. channels >= 64) or csrc/attn_f32.hip (fp32, a multiple
    of 4, i.e. channels >= 32); only narrower layers evaluate softmax(q k^T) v with torch."""

    def __init__(self, channels: int, reduction: int = 8):
        super().__init__()
        self.channels = channels
        self.query = nn.Conv1d(channels, channels // reduction, kernel_size=1)
        self.key = nn.Conv1d(channels, channels // reduction, kernel_size=1)
        self.value = nn.Conv1d(channels, channels, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.use_flash = True          # tests switch it off to compare with the torch evaluation in the same dtype

    def forward_nlc(self, x: Tensor) -> Tensor:
        g1 = ops.geom_conv(1)
        l = x.shape[1]
        q = ops.conv(x, self.query.weight.permute(2, 0, 1), self.query.bias, g1, l)
        k = ops.conv(x, self.key.weight.permute(2, 0, 1), self.key.bias, g1, l)
        v = ops.conv(x, self.value.weight.permute(2, 0, 1), self.value.bias, g1, l)
        dq, rep = q.shape[2], self.channels // q.shape[2]
        if self.use_flash and dq % (8 if x.dtype == torch.bfloat16 else 4) == 0 and dq <= 256 and rep * dq == self.channels:
            # flash kernel of csrc/attn.hip (one band, no coupling, scale 1): the value channels are split into
            # `rep` heads of dq channels that all see the same scores, so q and k are tiled across the heads;
            # no [N, L, L] score tensor, one launch forward, three backward
            one = torch.ones((1, 1), dtype=torch.float32, device=x.device)
            out = ops.cfc_attention(q.repeat(1, 1, rep)[None], k.repeat(1, 1, rep)[None], v[None], one, rep, 1.0)
        else:
            att = torch.softmax(torch.bmm(q, k.transpose(1, 2)).float(), dim=-1).to(x.dtype)
            out = torch.bmm(att, v)
        return (x.float() + self.gamma.float() * out.float()).to(x.dtype)  # fp32 so d(gamma) reduces in fp32

    def forward(self, x: Tensor) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x)))


def _band_pair_attention(q: Tensor, k: Tensor, v: Tensor, coupling: Tensor, n_heads: int, scale: float) -> Tensor:
    """q, k, v: [nb, M, l, C] stacks of the per-band projections -> [M, l, nb*C]:
    out[:, :, i*C + h*hd + d] = sum_j coupling[i, j] * softmax(q_i k_j^T * scale) v_j   (models.py:431-462 / 191-216).
    ONE launch of the band-pair flash kernel (bf16: csrc/attn.hip, MFMA 32x32x16; fp32: csrc/attn_f32.hip, exact fp32
    MFMA 16x16x4; three launches backward).  Head sizes the kernels do not take: the same products with
    ``scaled_dot_product_attention``."""
    nb, m, l, c = q.shape
    hd = c // n_heads
    if hd % (8 if q.dtype == torch.bfloat16 else 4) == 0 and hd <= 256:
        return ops.cfc_attention(q.contiguous(), k.contiguous(), v.contiguous(), coupling, n_heads, scale)
    sp = lambda z: z.reshape(nb, m, l, n_heads, hd).transpose(2, 3)                    # [nb, M, heads, l, hd]
    qs, ks, vs = sp(q), sp(k), sp(v)
    outs = []
    for i in range(nb):
        acc = None
        for j in range(nb):
            a = F.scaled_dot_product_attention(qs[i], ks[j], vs[j], dropout_p=0.0, is_causal=False, scale=scale)
            a = a.float() * coupling[i, j]
            acc = a if acc is None else acc + a
        outs.append(acc.transpose(1, 2).reshape(m, l, c))                               # channel = head * hd + d
    return torch.cat(outs, dim=2).to(q.dtype)


class CrossFrequencyCouplingAttentionV2(nn.Module):
    """models.py:227-474 (``attention_type="cross_freq_v2"``, the phase-four / phase-five default): FFT band split,
    grouped 1x1 q/k/v projections per band, every query band attends to every key/value band, outputs mixed by
    softmax(coupling_weights), 1x1 out_proj, gated residual.  Same parameters / state-dict keys as the reference.

    Channels-last throughout: the band split runs along the row axis of the ``[N, L, C]`` tensor (fp32: rocFFT;
    bf16: the same linear operator as one library GEMM, see ``_band_operator``; no transposes), the per-band projections and out_proj run on the HIP conv kernels (the grouped Conv1d is five
    independent C x C convs on strided views of one buffer), and in bf16 the 25 attention products, their
    coupling-weighted sum and the concatenation are ONE launch of the band-pair flash kernel (``csrc/attn.hip``,
    head_dim = C/4 = 256 at config B, 107 GFLOP per window; backward = 3 launches).  fp32 runs take the same route
    through the exact-fp32 flash kernels of ``csrc/attn_f32.hip``; ``scaled_dot_product_attention`` is left only for
    head sizes neither kernel family takes (not a multiple of 8 / 4, or above 256).
    """

    def __init__(self, channels: int, n_heads: int = 4, sample_rate: float = 1000.0, bands=None, chunk_size: int = 256,
                 enable_flash: bool = True):
        super().__init__()
        self.channels = channels
        self.n_heads = n_heads
        self.head_dim = channels // n_heads
        self.sample_rate = sample_rate
        self.enable_flash = enable_flash
        self.chunk_size = chunk_size
        if bands is None:
            bands = [(1, 4), (4, 8), (8, 12), (12, 30), (30, 80)]
        self.bands = bands
        self.n_bands = len(bands)
        nb = self.n_bands
        self.proj_q = nn.Conv1d(channels * nb, channels * nb, 1, groups=nb)
        self.proj_k = nn.Conv1d(channels * nb, channels * nb, 1, groups=nb)
        self.proj_v = nn.Conv1d(channels * nb, channels * nb, 1, groups=nb)
        self.out_proj = nn.Conv1d(channels * nb, channels, 1)
        self.coupling_weights = nn.Parameter(torch.ones(nb, nb) / nb)
        self.gate = nn.Parameter(torch.zeros(1))
        self._filters_cache: Optional[Tensor] = None
        self._cache_len: int = 0
        self._filters_on = {}
        self.scale = 1.0 / math.sqrt(self.head_dim)

    def _build_bandpass_filters_batched(self, T: int, device) -> Tensor:
        """models.py:293-312; the device copy is cached (no host transfer per step)."""
        freq_len = T // 2 + 1
        if self._filters_cache is None or self._cache_len != freq_len:
            freqs = torch.fft.rfftfreq(T, d=1.0 / self.sample_rate)
            self._filters_cache = torch.stack([torch.sigmoid((freqs - lo) / 2) * torch.sigmoid((hi - freqs) / 2)
                                               for lo, hi in self.bands], dim=0)
            self._cache_len = freq_len
            self._filters_on = {}
        key = str(torch.device(device))
        if key not in self._filters_on:
            self._filters_on[key] = self._filters_cache.to(device)
        return self._filters_on[key]

    GEMM_SPLIT_MAX_LEN = 4096

    def _band_operator(self, T: int, device) -> Tensor:
        """The band split irfft(rfft(x) * filt_b) is linear in x: row (b, l) of this [nb*T, 2T] bf16 matrix is the
        response of band b at position l to an impulse at each position, as a bf16 high part followed by the bf16
        remainder (K = 2T against [x; x]: bf16 inputs are exact, products accumulate in fp32, so the operator is
        applied to ~2^-17 and the only rounding left is the bf16 store the FFT path has as well)."""
        key = (T, str(torch.device(device)))
        cache = self.__dict__.setdefault("_band_ops", {})
        if key not in cache:
            filt = self._build_bandpass_filters_batched(T, device).float()                        # [nb, F]
            eye_f = torch.fft.rfft(torch.eye(T, device=device, dtype=torch.float32), dim=0)       # [F, T]
            h = torch.fft.irfft(filt[:, :, None] * eye_f[None], n=T, dim=1)                       # [nb, T(l), T(l')]
            h = h.reshape(self.n_bands * T, T)
            hi = h.bfloat16()
            lo = (h - hi.float()).bfloat16()
            cache.clear()
            cache[key] = torch.cat([hi, lo], dim=1).contiguous()
        return cache[key]

    def _split_bands(self, x: Tensor) -> Tensor:
        """[N, L, C] -> [N, nb, L, C] band signals (models.py:314-340)."""
        n, t, c = x.shape
        if x.dtype == torch.bfloat16 and t <= self.GEMM_SPLIT_MAX_LEN:
            # one plain library GEMM per window batch ([nb*L, 2L] @ [2L, C]) instead of rFFT, a complex product
            # over [N, nb, F, C], nb inverse FFTs and the casts around them; backward is the transposed GEMM
            op = self._band_operator(t, x.device)
            return torch.matmul(op, torch.cat([x, x], dim=1)).view(n, self.n_bands, t, c)
        filt = self._build_bandpass_filters_batched(t, x.device)                       # [nb, F]
        xf = torch.fft.rfft(x.float(), dim=1)                                          # [N, F, C]
        bands = torch.fft.irfft(xf.unsqueeze(1) * filt[None, :, :, None], n=t, dim=2)  # [N, nb, L, C]
        return bands.to(x.dtype)

    def forward_nlc(self, x: Tensor) -> Tensor:
        n, t, c = x.shape
        nb, nh, hd = self.n_bands, self.n_heads, self.head_dim
        g1 = ops.geom_conv(1)
        # the grouped Conv1d is nb independent C x C convs; one autograd node runs all 15 on strided views of the
        # band tensor and writes the [nb, N, L, C] stacks the attention kernels read (channel = head * hd + d)
        wb = lambda conv: (conv.weight.view(nb, c, c), conv.bias.view(nb, c))
        q, k, v = ops.band_project(self._split_bands(x), *wb(self.proj_q), *wb(self.proj_k), *wb(self.proj_v))
        coupling = F.softmax(self.coupling_weights.float(), dim=-1)
        cat = _band_pair_attention(q, k, v, coupling, nh, self.scale)
        out = ops.conv(cat, self.out_proj.weight.permute(2, 0, 1), self.out_proj.bias, g1, t)
        return (x.float() + self.gate.float() * out.float()).to(x.dtype)

    def forward(self, x: Tensor) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x)))

    def extra_repr(self) -> str:
        band_str = ", ".join(f"{l}-{h}Hz" for l, h in self.bands)
        return f"channels={self.channels}, n_heads={self.n_heads}, flash={self.enable_flash}, bands=[{band_str}]"


class CrossFrequencyCouplingAttention(CrossFrequencyCouplingAttentionV2):
    """models.py:100-224 (``attention_type="cross_freq"``), the chunked V1 module: one Conv1d 1x1 per band and
    projection (``band_q / band_k / band_v`` ModuleLists: the reference's state-dict keys), and a query attends to
    the keys of its own chunk of ``chunk_size`` positions only (the last chunk may be shorter).  A chunk is a batch
    item of the same band-pair kernels: [N, L, C] -> [N * L/chunk, chunk, C] is a view when chunk divides L."""

    def __init__(self, channels: int, n_heads: int = 4, sample_rate: float = 1000.0, bands=None, chunk_size: int = 256):
        nn.Module.__init__(self)
        self.channels = channels
        self.n_heads = n_heads
        self.head_dim = channels // n_heads
        self.sample_rate = sample_rate
        self.chunk_size = chunk_size
        if bands is None:
            bands = [(1, 4), (4, 8), (8, 12), (12, 30), (30, 80)]
        self.bands = bands
        self.n_bands = len(bands)
        self.band_q = nn.ModuleList([nn.Conv1d(channels, channels, 1) for _ in bands])
        self.band_k = nn.ModuleList([nn.Conv1d(channels, channels, 1) for _ in bands])
        self.band_v = nn.ModuleList([nn.Conv1d(channels, channels, 1) for _ in bands])
        self.out_proj = nn.Conv1d(channels * self.n_bands, channels, 1)
        self.coupling_weights = nn.Parameter(torch.ones(self.n_bands, self.n_bands) / self.n_bands)
        self.gate = nn.Parameter(torch.zeros(1))
        self._filters_cache: Optional[Tensor] = None
        self._cache_len: int = 0
        self._filters_on = {}
        self.scale = 1.0 / math.sqrt(self.head_dim)

    def forward_nlc(self, x: Tensor) -> Tensor:
        n, t, c = x.shape
        nb, nh, cs = self.n_bands, self.n_heads, self.chunk_size
        g1 = ops.geom_conv(1)
        wb = lambda convs: (torch.stack([m.weight[:, :, 0] for m in convs]), torch.stack([m.bias for m in convs]))
        q, k, v = ops.band_project(self._split_bands(x), *wb(self.band_q), *wb(self.band_k), *wb(self.band_v))
        coupling = F.softmax(self.coupling_weights.float(), dim=-1)
        full = (t // cs) * cs
        parts = []
        if full:
            chunked = lambda z: z[:, :, :full].reshape(nb, n * (full // cs), cs, c)
            a = _band_pair_attention(chunked(q), chunked(k), chunked(v), coupling, nh, self.scale)
            parts.append(a.view(n, full, nb * c))
        if full < t:
            tail = lambda z: z[:, :, full:]
            parts.append(_band_pair_attention(tail(q), tail(k), tail(v), coupling, nh, self.scale))
        cat = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        out = ops.conv(cat, self.out_proj.weight.permute(2, 0, 1), self.out_proj.bias, g1, t)
        return (x.float() + self.gate.float() * out.float()).to(x.dtype)

    def extra_repr(self) -> str:
        band_str = ", ".join(f"{l}-{h}Hz" for l, h in self.bands)
        return f"channels={self.channels}, chunk_size={self.chunk_size}, bands=[{band_str}]"


class FiLM(nn.Module):
    """models.py:477-517 (cross-attention with output gating; 'none' bypasses)."""

    def __init__(self, channels: int, emb_dim: int, cond_mode: str = "cross_attn_gated"):
        super().__init__()
        self.cond_mode = cond_mode
        self.channels = channels
        self.emb_dim = emb_dim
        if cond_mode == "cross_attn_gated":
            self.q_proj = nn.Conv1d(channels, channels, 1)
            self.k_proj = nn.Linear(emb_dim, channels)
            self.v_proj = nn.Linear(emb_dim, channels)
            self.out_proj = nn.Conv1d(channels, channels, 1)
            self.scale_factor = channels ** -0.5
            self.gate = nn.Sequential(nn.Linear(emb_dim, channels), nn.Sigmoid())
        elif cond_mode != "none":
            raise ValueError(f"Unknown cond_mode: {cond_mode}. Supported: 'none', 'cross_attn_gated'")

    def emb_maps(self):
        """k_proj (with the C^-1/2 of models.py:514 folded in), v_proj and the sigmoid gate as small-linear maps."""
        gate = self.gate[0]
        return ((self.k_proj.weight, self.k_proj.bias, "none", self.scale_factor),
                (self.v_proj.weight, self.v_proj.bias, "none", 1.0),
                (gate.weight, gate.bias, "sigmoid", 1.0))

    _pre = None     # (emb, kk, v, g) set by CondUNet1D.forward: the projections of EVERY block's FiLM in one launch

    def forward_nlc(self, x: Tensor, emb: Optional[Tensor]) -> Tensor:
        if self.cond_mode == "none" or emb is None:
            return x
        pre = self._pre
        if pre is not None and pre[0] is emb:
            kk, v, g = pre[1:]
        else:                                           # stand-alone use: one launch for this module's three maps
            kk, v, g = ops.small_linear_group(emb, self.emb_maps())
        return ops.film(x, self.q_proj.weight.permute(2, 0, 1), self.q_proj.bias,
                        self.out_proj.weight.permute(2, 0, 1), self.out_proj.bias, kk, v, g,
                        params=(self.q_proj.weight, self.q_proj.bias, self.out_proj.weight, self.out_proj.bias))

    def forward(self, x: Tensor, emb: Optional[Tensor] = None) -> Tensor:
        if self.cond_mode == "none" or emb is None:
            return x
        return _as_ncl(self.forward_nlc(_to_nlc(x), emb))


class ConvBlock(nn.Module):
    """models.py:520-569: film(Sequential[Conv1d k3, norm, GELU, (Dropout1d)](x), emb)."""

    def __init__(self, in_c: int, out_c: int, emb_dim: int, downsample: bool = False, dropout: float = 0.0,
                 norm_type: str = "batch", cond_mode: str = "cross_attn_gated"):
        super().__init__()
        self.stride = 2 if downsample else 1
        self.cond_mode = cond_mode
        norm = _make_norm(norm_type, out_c)
        layers: List[nn.Module] = [nn.Conv1d(in_c, out_c, kernel_size=3, stride=self.stride, padding=1), norm, nn.GELU()]
        if dropout > 0:
            layers.append(nn.Dropout1d(p=dropout))
        self.dropout_p = dropout
        self.conv = nn.Sequential(*layers)
        self.film = FiLM(out_c, emb_dim, cond_mode=cond_mode)

    def _main_conv(self, x1: Tensor, x2: Optional[Tensor]) -> Tensor:
        c = self.conv[0]
        l_out = ops.conv_out_len(x1.shape[1], 3, self.stride)
        return ops.conv(x1, c.weight.permute(2, 0, 1), c.bias, ops.geom_conv(3, self.stride), l_out, x2=x2,
                        params=(c.weight, c.bias), want_stats=_uses_batch_stats(self.conv[1]))

    def forward_nlc(self, x1: Tensor, emb: Optional[Tensor], x2: Optional[Tensor] = None) -> Tensor:
        y = self._main_conv(x1, x2)
        z = _norm_act(y, self.conv[1], ACT_IDS["gelu"], False)
        z = _dropout1d_nlc(z, self.dropout_p, self.training)
        return self.film.forward_nlc(z, emb)

    def forward(self, x: Tensor, emb: Optional[Tensor] = None) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x), emb))


def _dropout1d_nlc(z: Tensor, p: float, training: bool) -> Tensor:
    """nn.Dropout1d zeroes whole (window, channel) rows; on NLC that is a [N, 1, C] mask.
    p = 0 in every production config; the mask uses torch's generator."""
    if p <= 0 or not training:
        return z
    n, _, c = z.shape
    keep = (torch.rand((n, c), device=z.device) >= p).to(torch.float32) / (1.0 - p)
    return ops.ScaleFn.apply(z, keep)


class SqueezeExcitation1D(nn.Module):
    """models.py:637-662.  The squeeze (mean over L) arrives as a by-product of the fused
    norm+GELU pass; only the [N, C] MLP runs here."""

    def __init__(self, channels: int, reduction: int = 4):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool1d(1)
        reduced = max(channels // reduction, 8)
        self.fc = nn.Sequential(nn.Linear(channels, reduced, bias=False), nn.GELU(),
                                nn.Linear(reduced, channels, bias=False), nn.Sigmoid())

    def gate_from_pool(self, pool_sum: Tensor, rows: int) -> Tensor:
        """sigmoid(W2 gelu(W0 (pool_sum / rows))): two launches (models.py:660-661; the mean's 1/rows is the first
        map's scale)."""
        return ops.se_gate(pool_sum, self.fc[0].weight, self.fc[2].weight, rows)

    def forward(self, x: Tensor) -> Tensor:
        xn = _to_nlc(x)
        z, pool = ops.AffineActFn.apply(xn, None, None, None, 0, True)
        return _as_ncl(ops.ScaleFn.apply(z, self.gate_from_pool(pool, xn.shape[1])))


class DilatedConv1d(nn.Module):
    """models.py:665-702 (parameter holder for one branch)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 7, stride: int = 1, dilation: int = 1):
        super().__init__()
        padding = (kernel_size - 1) * dilation // 2
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=True)


class MultiScaleDilatedConv1d(nn.Module):
    """models.py:705-759.  The weighted sum of the parallel dilated branches is evaluated as ONE
    tap-list convolution: W_eff[tap] = sum over (branch, k) landing on that tap of
    softmax(scale_weights)[branch] * W_branch[:, :, k]; gradients reach every branch weight,
    bias and scale_weights through autograd on this (tiny) weight algebra."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 7, stride: int = 1,
                 dilations: Tuple[int, ...] = (1, 4, 16, 32)):
        super().__init__()
        self.n_scales = len(dilations)
        self.kernel_size, self.stride, self.dilations = kernel_size, stride, tuple(dilations)
        self.branches = nn.ModuleList([
            DilatedConv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, dilation=d)
            for d in dilations])
        self.scale_weights = nn.Parameter(torch.ones(self.n_scales) / self.n_scales)
        g, pairs = ops.merged_dilated_taps(kernel_size, self.dilations)
        self._geom = ops.ConvGeom(g.offs, g.phases, stride, 1, g.alg_scale)
        self._tap_list = tuple(p[2] for p in pairs)

    def effective_weights(self) -> Tuple[Tensor, Tensor]:
        ws = [b.conv.weight for b in self.branches]
        if ws[0].is_cuda:      # softmax over the branches, merged taps and effective bias: one launch
            return ops.MergeBranchesFn.apply(self.scale_weights, self._tap_list, self._geom.n_taps, len(ws), *ws,
                                             *[b.conv.bias for b in self.branches])
        # parameter-only algebra, used by CPU-side layout checks; the conv itself has no CPU path
        w = F.softmax(self.scale_weights.float(), dim=0)
        st = (torch.stack(ws, dim=0).float() * w[:, None, None, None]).permute(0, 3, 1, 2)
        st = st.reshape(-1, st.shape[2], st.shape[3])
        w_eff = torch.zeros((self._geom.n_taps,) + tuple(st.shape[1:]), dtype=st.dtype).index_add(
            0, torch.tensor(self._tap_list, dtype=torch.long), st)
        b_eff = (torch.stack([b.conv.bias for b in self.branches], dim=0).float() * w[:, None]).sum(0)
        return w_eff, b_eff

    def forward_nlc(self, x1: Tensor, x2: Optional[Tensor] = None, want_stats: bool = False) -> Tensor:
        k, s = self.kernel_size, self.stride
        l_out = ops.conv_out_len(x1.shape[1], k, s, 1)
        return ops.conv(x1, self.effective_weights, None, self._geom, l_out, x2=x2, params=tuple(self.parameters()),
                        want_stats=want_stats)

    def forward(self, x: Tensor) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x)))


class ModernConvBlock(nn.Module):
    """models.py:762-840: film(Sequential[multi-dilated conv, norm, GELU, (SE), (Dropout1d)](x), emb)."""

    def __init__(self, in_c: int, out_c: int, emb_dim: int, downsample: bool = False, dropout: float = 0.0,
                 norm_type: str = "batch", cond_mode: str = "cross_attn_gated", use_se: bool = True,
                 kernel_size: int = 7, dilations: Tuple[int, ...] = (1, 4, 16, 32)):
        super().__init__()
        stride = 2 if downsample else 1
        self.use_se = use_se
        norm = _make_norm(norm_type, out_c)
        conv = MultiScaleDilatedConv1d(in_c, out_c, kernel_size=kernel_size, stride=stride, dilations=dilations)
        layers: List[nn.Module] = [conv, norm, nn.GELU()]
        if self.use_se:
            layers.append(SqueezeExcitation1D(out_c))
        if dropout > 0:
            layers.append(nn.Dropout1d(p=dropout))
        self.dropout_p = dropout
        self.conv = nn.Sequential(*layers)
        self.film = FiLM(out_c, emb_dim, cond_mode=cond_mode)

    def forward_nlc(self, x1: Tensor, emb: Optional[Tensor], x2: Optional[Tensor] = None) -> Tensor:
        y = self.conv[0].forward_nlc(x1, x2, want_stats=_uses_batch_stats(self.conv[1]))
        if self.use_se:
            z = _norm_act_se(y, self.conv[1], ACT_IDS["gelu"], self.conv[3])
        else:
            z = _norm_act(y, self.conv[1], ACT_IDS["gelu"], False)
        z = _dropout1d_nlc(z, self.dropout_p, self.training)
        return self.film.forward_nlc(z, emb)

    def forward(self, x: Tensor, emb: Optional[Tensor] = None) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x), emb))


class _UpMixin:
    """models.py:611-630 / 916-936: ConvTranspose1d(k4,s2,p1) -> centre-crop -> concat | add -> block.
    The transposed conv runs as a two-phase tap-list conv; `add` is fused into its epilogue and
    `concat` is never materialised (the block's first conv reads two sources)."""

    def _up_forward(self, x: Tensor, skip: Tensor, emb: Optional[Tensor]) -> Tensor:
        geom, kidx = ops.geom_convt_k4s2p1()

        def w():                                                   # tap order (no index tensor: graph-capture safe)
            return ops.TapOrderFn.apply(self.up.weight.permute(2, 1, 0), tuple(kidx)), self.up.bias   # [k, Cout, Cin] view

        up_params = (self.up.weight, self.up.bias)
        l_up = 2 * x.shape[1]
        l_skip = skip.shape[1]
        if self.skip_type == "add" and self.skip_proj is not None:
            skip = ops.conv(skip, self.skip_proj.weight.permute(2, 0, 1), self.skip_proj.bias, ops.geom_conv(1), l_skip,
                            params=(self.skip_proj.weight, self.skip_proj.bias))
        if self.skip_type == "add" and l_up == l_skip:
            u = ops.conv(x, w, None, geom, l_up, res=skip, params=up_params)
            return self.block.forward_nlc(u, emb)
        u = ops.conv(x, w, None, geom, l_up, params=up_params)
        if l_up < l_skip:
            s = (l_skip - l_up) // 2
            skip = skip[:, s:s + l_up, :]
        elif l_up > l_skip:
            s = (l_up - l_skip) // 2
            u = u[:, s:s + l_skip, :]
        if self.skip_type == "concat":
            return self.block.forward_nlc(u, emb, x2=skip)
        return self.block.forward_nlc(u + skip, emb)


class UpBlock(nn.Module, _UpMixin):
    def __init__(self, in_c: int, skip_c: int, out_c: int, emb_dim: int, dropout: float = 0.0,
                 norm_type: str = "batch", cond_mode: str = "cross_attn_gated", skip_type: str = "concat"):
        super().__init__()
        self.skip_type = skip_type
        self.up = nn.ConvTranspose1d(in_c, out_c, kernel_size=4, stride=2, padding=1)
        if skip_type == "concat":
            self.block = ConvBlock(out_c + skip_c, out_c, emb_dim, downsample=False, dropout=dropout,
                                   norm_type=norm_type, cond_mode=cond_mode)
            self.skip_proj = None
        elif skip_type == "add":
            self.block = ConvBlock(out_c, out_c, emb_dim, downsample=False, dropout=dropout, norm_type=norm_type,
                                   cond_mode=cond_mode)
            self.skip_proj = nn.Conv1d(skip_c, out_c, kernel_size=1) if skip_c != out_c else None
        else:
            raise ValueError(f"Unknown skip_type: {skip_type}. Available: concat, add")

    def forward_nlc(self, x, skip, emb):
        return self._up_forward(x, skip, emb)

    def forward(self, x: Tensor, skip: Tensor, emb: Optional[Tensor] = None) -> Tensor:
        return _as_ncl(self._up_forward(_to_nlc(x), _to_nlc(skip), emb))


class ModernUpBlock(nn.Module, _UpMixin):
    def __init__(self, in_c: int, skip_c: int, out_c: int, emb_dim: int, dropout: float = 0.0,
                 norm_type: str = "batch", cond_mode: str = "cross_attn_gated", use_se: bool = True,
                 kernel_size: int = 7, dilations: Tuple[int, ...] = (1, 4, 16, 32), skip_type: str = "concat"):
        super().__init__()
        self.skip_type = skip_type
        self.up = nn.ConvTranspose1d(in_c, out_c, kernel_size=4, stride=2, padding=1)
        kw = dict(downsample=False, dropout=dropout, norm_type=norm_type, cond_mode=cond_mode, use_se=use_se,
                  kernel_size=kernel_size, dilations=dilations)
        if skip_type == "concat":
            self.block = ModernConvBlock(out_c + skip_c, out_c, emb_dim, **kw)
            self.skip_proj = None
        elif skip_type == "add":
            self.block = ModernConvBlock(out_c, out_c, emb_dim, **kw)
            self.skip_proj = nn.Conv1d(skip_c, out_c, kernel_size=1) if skip_c != out_c else None
        else:
            raise ValueError(f"Unknown skip_type: {skip_type}. Available: concat, add")

    def forward_nlc(self, x, skip, emb):
        return self._up_forward(x, skip, emb)

    def forward(self, x: Tensor, skip: Tensor, emb: Optional[Tensor] = None) -> Tensor:
        return _as_ncl(self._up_forward(_to_nlc(x), _to_nlc(skip), emb))


class _Act(nn.Module):
    """Parameter-free stand-in for the activation modules of ``self.mid`` (keeps the
    reference's Sequential indices: mid.0 conv, mid.1 act, [mid.2 attention], conv, act)."""

    def __init__(self, name: str):
        super().__init__()
        self.name = name
        self.act_id = get_activation_id(name)

    def forward_nlc(self, x: Tensor) -> Tensor:
        return ops.activation(x, self.act_id)

    def forward(self, x: Tensor) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x)))

    def extra_repr(self) -> str:
        return self.name


class _Bottleneck(nn.Sequential):
    """``self.mid`` (models.py:3029-3041) -- an nn.Sequential so indices/keys match; forward is
    overridden to stay channels-last between its layers."""

    def forward_nlc(self, x: Tensor) -> Tensor:
        for m in self:
            if isinstance(m, nn.Conv1d):
                x = ops.conv(x, m.weight.permute(2, 0, 1), m.bias, ops.geom_conv(3), x.shape[1], params=(m.weight, m.bias))
            else:
                x = m.forward_nlc(x)
        return x

    def forward(self, x: Tensor) -> Tensor:
        return _as_ncl(self.forward_nlc(_to_nlc(x)))


class SessionAdaptiveScaling(nn.Module):
    """models.py:1998-2068 (scalar mode).  Statistics come from the HIP moments kernel (``unet_ncl_moments``) and the
    final affine rides the model's output-side layout change (``ops.ToNCLAffineFn`` in ``CondUNet1D``); the 64-wide
    predictor MLP on [N, 64] stays in torch."""

    def __init__(self, n_channels: int, n_input_channels: int = 32, hidden_dim: int = 64):
        super().__init__()
        self.n_channels = n_channels
        self.n_input_channels = n_input_channels
        input_dim = n_input_channels * 2
        self.predictor = nn.Sequential(
            nn.Linear(input_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, n_channels * 2))
        nn.init.zeros_(self.predictor[-1].weight)
        with torch.no_grad():
            self.predictor[-1].bias[:n_channels] = 1.0
            self.predictor[-1].bias[n_channels:] = 0.0

    def scale_params(self, raw_input: Tensor) -> Tuple[Tensor, Tensor]:
        """gamma, beta [N, C_out] fp32 from the raw input's per-channel mean / unbiased std (models.py:2058-2066)."""
        if raw_input.is_cuda and not (raw_input.requires_grad and torch.is_grad_enabled()):
            xin = raw_input if raw_input.dtype in (torch.float32, torch.bfloat16) else raw_input.float()
            mean, rstd = ops.ncl_moments(xin.contiguous(), 0.0)          # one HIP pass: mean and 1 / std
            std = rstd.reciprocal()
        else:     # an input that takes a gradient: torch's differentiable reductions (the reference's graph)
            mean = raw_input.float().mean(dim=-1)
            std = raw_input.float().std(dim=-1)
        params = self.predictor(torch.cat([mean, std], dim=-1))
        return params[:, :self.n_channels], params[:, self.n_channels:]

    def forward(self, x: Tensor, raw_input: Tensor) -> Tensor:
        gamma, beta = self.scale_params(raw_input)
        return x * gamma.unsqueeze(-1).to(x.dtype) + beta.unsqueeze(-1).to(x.dtype)


class CondUNet1D(nn.Module):
    """Drop-in for models.py:2865-3375 on the MI355X kernels (see module docstring)."""

    def __init__(
        self,
        in_channels: int = 32,
        out_channels: int = 32,
        base: int = BASE_CHANNELS,
        n_odors: int = NUM_ODORS,
        emb_dim: int = 128,
        dropout: float = 0.0,
        use_attention: bool = True,
        attention_type: str = "basic",
        norm_type: str = "batch",
        cond_mode: str = "cross_attn_gated",
        n_downsample: int = 2,
        conv_type: str = "standard",
        use_se: bool = True,
        conv_kernel_size: int = 7,
        dilations: Tuple[int, ...] = (1, 4, 16, 32),
        use_adaptive_scaling: bool = False,
        adaptive_scaling_mode: str = "scalar",
        adaptive_scaling_cross_channel: bool = False,
        use_session_stats: bool = False,
        session_emb_dim: int = 32,
        session_use_spectral: bool = False,
        use_session_embedding: bool = False,
        n_sessions: int = 0,
        activation: str = "relu",
        n_heads: int = 4,
        skip_type: str = "add",
        sampling_rate: float = 1000.0,
    ):
        super().__init__()
        self.cond_mode = cond_mode
        self.emb_dim = emb_dim
        self.attention_type = attention_type
        self.conv_type = conv_type
        self.n_downsample = n_downsample
        self.use_session_stats = use_session_stats
        self.use_session_embedding = use_session_embedding
        self.session_emb_dim = session_emb_dim
        self.n_sessions = n_sessions
        self.activation_name = activation
        self.n_heads = n_heads
        self.skip_type = skip_type

        if use_session_stats or use_session_embedding:
            raise NotImplementedError(
                "session statistics / session embedding conditioning is outside the hot-path scope "
                "(SURVEY.md 2.1: ablation-only); use the reference implementation for it")
        if use_session_embedding and n_sessions <= 0:
            raise ValueError(f"use_session_embedding=True requires n_sessions > 0, got {n_sessions}")

        self.embed = nn.Embedding(n_odors, emb_dim) if cond_mode != "none" else None
        self.session_stats_encoder = None
        self.session_proj = None
        self.session_embed = None
        self.session_embed_proj = None

        channels = [base]
        for i in range(n_downsample):
            channels.append(min(base * (2 ** (i + 1)), base * 8))

        if conv_type == "modern":
            kw = dict(use_se=use_se, kernel_size=conv_kernel_size, dilations=dilations)
            self.inc = ModernConvBlock(in_channels, channels[0], emb_dim, dropout=dropout, norm_type=norm_type,
                                       cond_mode=cond_mode, **kw)
            self.encoders = nn.ModuleList([
                ModernConvBlock(channels[i], channels[i + 1], emb_dim, downsample=True, dropout=dropout,
                                norm_type=norm_type, cond_mode=cond_mode, **kw) for i in range(n_downsample)])
            self.decoders = nn.ModuleList([
                ModernUpBlock(channels[n_downsample - i], channels[n_downsample - i - 1], channels[n_downsample - i - 1],
                              emb_dim, dropout=dropout, norm_type=norm_type, cond_mode=cond_mode, skip_type=skip_type,
                              **kw) for i in range(n_downsample)])
        else:
            self.inc = ConvBlock(in_channels, channels[0], emb_dim, dropout=dropout, norm_type=norm_type,
                                 cond_mode=cond_mode)
            self.encoders = nn.ModuleList([
                ConvBlock(channels[i], channels[i + 1], emb_dim, downsample=True, dropout=dropout, norm_type=norm_type,
                          cond_mode=cond_mode) for i in range(n_downsample)])
            self.decoders = nn.ModuleList([
                UpBlock(channels[n_downsample - i], channels[n_downsample - i - 1], channels[n_downsample - i - 1],
                        emb_dim, dropout=dropout, norm_type=norm_type, cond_mode=cond_mode, skip_type=skip_type)
                for i in range(n_downsample)])

        bottleneck_ch = channels[n_downsample]
        mid_layers: List[nn.Module] = [nn.Conv1d(bottleneck_ch, bottleneck_ch, kernel_size=3, padding=1), _Act(activation)]
        if use_attention and attention_type != "none":
            mid_layers.append(self._build_attention(bottleneck_ch, attention_type))
        mid_layers.extend([nn.Conv1d(bottleneck_ch, bottleneck_ch, kernel_size=3, padding=1), _Act(activation)])
        self.mid = _Bottleneck(*mid_layers)

        self.outc = nn.Conv1d(channels[0], out_channels, kernel_size=3, padding=1)
        self.use_residual = (in_channels == out_channels)
        if not self.use_residual:
            self.residual_conv = nn.Conv1d(in_channels, out_channels, kernel_size=1)
        nn.init.zeros_(self.outc.weight)
        nn.init.zeros_(self.outc.bias)

        self.use_adaptive_scaling = use_adaptive_scaling
        self.adaptive_scaling_mode = adaptive_scaling_mode
        if use_adaptive_scaling:
            if adaptive_scaling_mode == "scalar" and not adaptive_scaling_cross_channel:
                self.adaptive_scaler = SessionAdaptiveScaling(n_channels=out_channels, n_input_channels=in_channels)
            else:
                raise NotImplementedError("SessionAdaptiveScalingV3 (spectral/harmonic/cross-channel) is outside the "
                                          "hot-path scope (SURVEY.md 2.1)")

        self.in_channels = in_channels
        self.out_channels = out_channels
        self.base = base
        self.use_attention = use_attention
        self.gradient_checkpointing = False
        self.compute_dtype: Optional[torch.dtype] = None  # None: follow the input's dtype

    # -- reference API surface ---------------------------------------------------------------
    def _build_attention(self, channels: int, attention_type: str) -> nn.Module:
        if attention_type == "basic":
            return SelfAttention1D(channels)
        if attention_type == "cross_freq_v2":
            return CrossFrequencyCouplingAttentionV2(channels, n_heads=self.n_heads)
        if attention_type == "cross_freq":
            return CrossFrequencyCouplingAttention(channels, n_heads=self.n_heads)
        raise ValueError(f"Unknown attention_type: {attention_type}. Available: none, basic, cross_freq, cross_freq_v2")

    def set_gradient_checkpointing(self, enable: bool = True):
        self.gradient_checkpointing = enable

    def late_grad_parameters(self) -> List[nn.Parameter]:
        """Parameters whose gradients are produced at the END of backward although they are registered all over the
        model: the embedding-side FiLM maps (k_proj, v_proj, gate) of every block, which ``forward`` evaluates for
        all blocks in one node in front of the first block (≈1 M parameters at config B).  A bucketed gradient
        exchange sends them with the last (tail) bucket (``dp.BucketedGradAllReduce(late_params=...)``)."""
        if not HOIST_FILM:
            return []
        blocks = [self.inc, *self.encoders, *[getattr(d, "block", None) for d in self.decoders]]
        films = [b.film for b in blocks if isinstance(getattr(b, "film", None), FiLM) and b.film.cond_mode != "none"]
        if len(films) <= 1:
            return []
        out = []
        for f in films:
            for w, b, _act, _scale in f.emb_maps():
                out.extend(t for t in (w, b) if t is not None and t.requires_grad)
        return out

    def train(self, mode: bool = True):
        """Entering training mode drops the inference cache of packed weights (``ops.conv``): fused optimizers update
        parameters WITHOUT bumping their version counters (torch's fused AdamW), so version checks alone would let
        an ``eval()`` after further training steps run on stale images.  Weights changed by such an optimizer while
        the model stays in eval mode need an explicit ``ops.clear_pack_cache()``."""
        if mode:
            ops.clear_pack_cache()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        ops.clear_pack_cache()              # copy_ bumps the versions anyway; loading is rare, so be unconditional
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        ops.clear_pack_cache()              # .to() / .cuda() / .half(): new storages behind the same Parameter objects
        return super()._apply(fn, *args, **kwargs)

    def set_compute_dtype(self, dtype: Optional[torch.dtype]):
        """bf16 or fp32 kernels regardless of the input dtype (None = follow the input)."""
        if dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32, torch.bfloat16 or None")
        self.compute_dtype = dtype

    def enable_adabn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.training = True
                m.reset_running_stats()

    def disable_adabn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.training = False

    def calibrate_bn_stats(self, dataloader, device: torch.device, max_batches: int = 50):
        self.train()
        with torch.no_grad():
            for i, batch in enumerate(dataloader):
                if i >= max_batches:
                    break
                x = batch[0] if isinstance(batch, (list, tuple)) else batch
                _ = self(x.to(device))
        self.eval()

    # -- introspection / analysis surface (models.py:3381-3703) -------------------------------------
    def get_layer_names(self) -> List[str]:
        """models.py:3381-3392 (names used by the recorder, not module paths)."""
        return (["inc"] + [f"encoder_{i}" for i in range(self.n_downsample)] + ["mid"] +
                [f"decoder_{i}" for i in range(self.n_downsample)] + ["outc"])

    def _named_blocks(self, mid_name: str = "mid"):
        """(reference layer name, module) in network order."""
        return ([("inc", self.inc)] + [(f"encoder_{i}", m) for i, m in enumerate(self.encoders)] +
                [(mid_name, self.mid)] + [(f"decoder_{i}", m) for i, m in enumerate(self.decoders)] +
                [("outc", self.outc)])

    def register_activation_hooks(self, callback) -> List:
        """models.py:3394-3437: callback(layer_name, activation) on every forward; activations are the
        reference's ``[B, C, L]`` tensors (views of the channels-last storage), ``outc`` sees ``delta`` before the
        residual.  Returns the hook handles."""
        return [mod.register_forward_hook(lambda m, i, o, n=name: callback(n, o)) for name, mod in self._named_blocks()]

    def get_encoder_layers(self) -> List[nn.Module]:
        return [self.inc] + list(self.encoders)

    def get_decoder_layers(self) -> List[nn.Module]:
        return list(self.decoders) + [self.outc]

    def get_attention_module(self) -> Optional[nn.Module]:
        """The cross-frequency attention module of the bottleneck, if any (models.py:3455-3464)."""
        for module in self.mid.modules():
            if hasattr(module, "coupling_weights"):
                return module
        return None

    def _get_first_conv_weight(self, module) -> Optional[Tensor]:
        """The tensor the reference's recorder calls a layer's "first conv weight" (models.py:3470-3495).  The
        search order is part of the contract: own ``weight``; then attributes conv / conv1 / conv_in; then the
        items of a Sequential / ModuleList; then direct Conv1d / Conv2d / Linear children.  (For a modern block
        that is the BatchNorm weight: the multi-scale conv has no ``weight`` of its own; for an up block with
        equal skip / output channels it is None.)"""
        if module is None:
            return None
        w = getattr(module, "weight", None)
        if w is not None:
            return w
        for attr in ("conv", "conv1", "conv_in"):
            if hasattr(module, attr):
                w = self._get_first_conv_weight(getattr(module, attr))
                if w is not None:
                    return w
        if isinstance(module, (nn.Sequential, nn.ModuleList)):
            for child in module:
                w = self._get_first_conv_weight(child)
                if w is not None:
                    return w
        for child in module.children():
            if getattr(child, "weight", None) is not None and isinstance(child, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                return child.weight
        return None

    def _recorder_weights(self):
        """(key, tensor) pairs behind get_gradient_norms / get_weight_norms: inc, encoder_i, mid_j (one per item of
        the bottleneck Sequential that has a weight), decoder_i, outc."""
        out = [("inc", self._get_first_conv_weight(self.inc))]
        out += [(f"encoder_{i}", self._get_first_conv_weight(m)) for i, m in enumerate(self.encoders)]
        out += [(f"mid_{j}", self._get_first_conv_weight(m)) for j, m in enumerate(self.mid)]
        out += [(f"decoder_{i}", self._get_first_conv_weight(m)) for i, m in enumerate(self.decoders)]
        out += [("outc", self._get_first_conv_weight(self.outc))]
        return [(k, w) for k, w in out if w is not None]

    def get_gradient_norms(self) -> dict:
        """models.py:3497-3535 (read by train.py:1549-1551 after backward)."""
        return {k: w.grad.norm().item() for k, w in self._recorder_weights() if w.grad is not None}

    def get_weight_norms(self) -> dict:
        """models.py:3537-3573."""
        return {k: w.norm().item() for k, w in self._recorder_weights()}

    def get_layer_parameter_counts(self) -> dict:
        """models.py:3575-3605."""
        counts = {name: sum(p.numel() for p in mod.parameters()) for name, mod in self._named_blocks()}
        if self.embed is not None:
            counts["embed"] = sum(p.numel() for p in self.embed.parameters())
        return counts

    def get_architecture_summary(self) -> dict:
        """models.py:3607-3627."""
        return {
            "n_downsample": self.n_downsample, "in_channels": self.in_channels, "out_channels": self.out_channels,
            "base_channels": self.base, "cond_mode": self.cond_mode, "skip_type": self.skip_type,
            "use_attention": self.use_attention, "attention_type": self.attention_type,
            "use_residual": self.use_residual,
            "total_params": sum(p.numel() for p in self.parameters()),
            "trainable_params": sum(p.numel() for p in self.parameters() if p.requires_grad),
            "layer_param_counts": self.get_layer_parameter_counts(),
            "layer_names": self.get_layer_names(),
        }

    @torch.no_grad()
    def extract_all_activations(self, x: Tensor, odor_ids: Optional[Tensor] = None, cond_emb: Optional[Tensor] = None,
                                session_ids: Optional[Tensor] = None) -> dict:
        """models.py:3629-3672: one forward pass, every block's output (the bottleneck under the key 'bottleneck')."""
        activations = {}
        hooks = [mod.register_forward_hook(lambda m, i, o, n=name: activations.__setitem__(n, o.detach().clone()))
                 for name, mod in self._named_blocks(mid_name="bottleneck")]
        try:
            self.forward(x, odor_ids=odor_ids, cond_emb=cond_emb, session_ids=session_ids)
        finally:
            for h in hooks:
                h.remove()
        return activations

    @torch.no_grad()
    def get_bottleneck_features(self, x: Tensor, odor_ids: Optional[Tensor] = None, cond_emb: Optional[Tensor] = None,
                                session_ids: Optional[Tensor] = None, pool: bool = True) -> Tensor:
        """models.py:3674-3703: pooled [B, C] or temporal [B, C, T'] bottleneck features."""
        _, bottleneck = self.forward(x, odor_ids=odor_ids, cond_emb=cond_emb, session_ids=session_ids,
                                     return_bottleneck=pool, return_bottleneck_temporal=not pool)
        return bottleneck

    # -- forward --------------------------------------------------------------------------------
    def forward(
        self,
        ob: Tensor,
        odor_ids: Optional[Tensor] = None,
        cond_emb: Optional[Tensor] = None,
        session_ids: Optional[Tensor] = None,
        return_bottleneck: bool = False,
        return_bottleneck_temporal: bool = False,
    ) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        if ob.dim() != 3 or ob.shape[1] != self.in_channels:
            raise ValueError(f"expected input [B, {self.in_channels}, T], got {tuple(ob.shape)}")
        if not ob.is_cuda:
            raise RuntimeError("unet_amd.CondUNet1D runs on the GPU only (HIP kernels); move the input to cuda")
        raw_input = ob
        io_dtype = ob.dtype if ob.dtype in (torch.float32, torch.bfloat16) else torch.float32
        if ob.dtype != io_dtype:
            ob = ob.to(io_dtype)
        cdt = self.compute_dtype or io_dtype

        x0 = ops.ZScoreInFn.apply(ob, cdt)                       # models.py:3235, now channels-last

        if cond_emb is not None:
            emb = cond_emb
        elif self.embed is not None and odor_ids is not None:
            emb = self.embed(odor_ids)
        else:
            emb = None

        # The FiLM projections depend on the embedding alone: all blocks' k / v / gate maps run as ONE launch here
        # (and one backward launch after the last block's backward) instead of one per block on the critical path.
        # Not under gradient checkpointing, whose recomputation must replay each block's own launches.
        # In a data-parallel job the hoisted node's backward runs after the LAST block's, so every block's k / v / gate
        # gradients arrive at the very end: ``late_grad_parameters()`` names them, and dp.BucketedGradAllReduce puts
        # them into its small tail bucket instead of letting them hold back the buckets of their blocks.
        films = []
        if HOIST_FILM and emb is not None and not (self.gradient_checkpointing and self.training):
            blocks = [self.inc, *self.encoders, *[getattr(d, "block", None) for d in self.decoders]]
            films = [b.film for b in blocks if isinstance(getattr(b, "film", None), FiLM) and b.film.cond_mode != "none"]
            if len(films) > 1:
                proj = ops.small_linear_group(emb, [m for f in films for m in f.emb_maps()])
                for i, f in enumerate(films):
                    f._pre = (emb, *proj[3 * i:3 * i + 3])
        try:
            return self._forward_blocks(ob, raw_input, x0, emb, io_dtype, return_bottleneck, return_bottleneck_temporal)
        finally:
            for f in films:
                f._pre = None

    def _forward_blocks(self, ob, raw_input, x0, emb, io_dtype, return_bottleneck, return_bottleneck_temporal):
        def run(block, *a):
            if self.gradient_checkpointing and self.training:
                from torch.utils.checkpoint import checkpoint
                return checkpoint(block.forward_nlc, *a, use_reentrant=False)
            return block.forward_nlc(*a)

        # A skip tensor has two consumers (the next encoder block's conv and a decoder block); instead of autograd's
        # full-tensor add of their two gradients, the decoder side parks its gradient and the encoder conv's
        # data-gradient kernel adds it in its epilogue (ops.GradStash).  Gradient checkpointing keeps working: the
        # stash is created and parked out here, outside the checkpointed block functions, and a recomputation only
        # re-arms the same stash.
        fuse = True
        skips = [_HookedCall.run(self.inc, run, x0, emb)]
        stashes = []
        for enc in self.encoders:
            stashes.append(ops.stash_for(skips[-1]) if fuse else None)
            skips.append(_HookedCall.run(enc, run, skips[-1], emb))

        bottleneck = _HookedCall.run(self.mid, lambda blk, x: blk.forward_nlc(x), skips[-1])
        if return_bottleneck_temporal:
            bott = _as_ncl(bottleneck).to(io_dtype)
        elif return_bottleneck:
            bott = bottleneck.float().mean(dim=1).to(io_dtype)
        else:
            bott = None

        x = bottleneck
        for i, dec in enumerate(self.decoders):
            j = self.n_downsample - i - 1
            skip = skips[j]
            if stashes[j] is not None and stashes[j].armed:
                skip = ops.StashGradFn.apply(skip, stashes[j])
            x = _HookedCall.run(dec, run, x, skip, emb)

        l = x.shape[1]
        if self.use_residual:
            res = x0
        else:
            res = ops.conv(x0, self.residual_conv.weight.permute(2, 0, 1), self.residual_conv.bias, ops.geom_conv(1), l,
                           params=(self.residual_conv.weight, self.residual_conv.bias))
        if self.outc._forward_hooks or self.outc._forward_pre_hooks:
            # a hook on outc must see delta before the residual (models.py:3359-3366): unfused
            def outc_delta(_blk, xin):
                return ops.conv(xin, self.outc.weight.permute(2, 0, 1), self.outc.bias, ops.geom_conv(3), l,
                                params=(self.outc.weight, self.outc.bias))
            out_nlc = _HookedCall.run(self.outc, outc_delta, x) + res
        else:
            out_nlc = ops.conv(x, self.outc.weight.permute(2, 0, 1), self.outc.bias, ops.geom_conv(3), l, res=res,
                               params=(self.outc.weight, self.outc.bias))
        if self.use_adaptive_scaling and isinstance(self.adaptive_scaler, SessionAdaptiveScaling) \
                and not (self.adaptive_scaler._forward_hooks or self.adaptive_scaler._forward_pre_hooks):
            # models.py:3369-3370 folded into the output-side layout change: out = gamma * out + beta in the same pass
            gamma, beta = self.adaptive_scaler.scale_params(raw_input)
            out = ops.ToNCLAffineFn.apply(out_nlc, gamma, beta, io_dtype)
        else:
            out = ops.ToNCLFn.apply(out_nlc, io_dtype)
            if self.use_adaptive_scaling:
                out = self.adaptive_scaler(out, raw_input)
        if return_bottleneck or return_bottleneck_temporal:
            return out, bott
        return out


class _HookedCall:
    """Runs a block's channels-last path while still honouring forward (pre-)hooks registered on
    the block (register_activation_hooks / extract_all_activations in the reference): hooks see
    and may replace ``[N, C, L]``-shaped views, exactly the tensors the reference's hooks get."""

    @staticmethod
    def run(block: nn.Module, runner, *args):
        if not (block._forward_hooks or block._forward_pre_hooks):
            return runner(block, *args)
        ncl_args = tuple(_as_ncl(a) if (torch.is_tensor(a) and a.dim() == 3) else a for a in args)
        for hook in block._forward_pre_hooks.values():
            r = hook(block, ncl_args)
            if r is not None:
                ncl_args = r if isinstance(r, tuple) else (r,)
        nlc_args = tuple(_to_nlc(a) if (torch.is_tensor(a) and a.dim() == 3) else a for a in ncl_args)
        out = runner(block, *nlc_args)
        out_ncl = _as_ncl(out)
        for hook in block._forward_hooks.values():
            r = hook(block, ncl_args, out_ncl)
            if r is not None:
                out_ncl = r
        return _to_nlc(out_ncl)
