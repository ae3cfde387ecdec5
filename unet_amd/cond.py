"""SpectroTemporalEncoder (models.py:1135-1395) on the HIP path -- the reference's default conditioning
source (``train.py --conditioning spectro_temporal``, train.py:2026-2030): it runs on every batch before the
U-Net and its output is the ``cond_emb`` the FiLM layers consume.

Same constructor, sub-module names, parameter creation order (so the same seed gives the same initial
weights) and ``state_dict`` keys as the reference.  What runs where:

* spectral branch: one-sided spectrum by ``torch.fft.rfft`` (rocFFT; any T), then ONE kernel
  (``unet_band_log_power``) for |X|^2, the 10 soft band masks and the log -- the reference materialises the
  power spectrum and an einsum.  The input is data (no gradient), so this branch has no backward through
  the FFT; gradients start at the MLP.
* temporal branch: the three dilated k=7 convs (32 -> 64 each) read the same input and their outputs are
  concatenated, so they are ONE tap-list conv with 19 distinct offsets and 192 output channels (weights
  scattered to their taps, zero elsewhere -- 0.3 % of the U-Net's FLOPs) -> GELU -> 1x1 conv 192 -> 64
  (streaming kernel) -> BatchNorm + GELU + mean over time in one pass (``unet_affine_act_fwd`` with the pooled
  sum) -- the reference writes three feature maps, a concat, a conv, a BN, a GELU and a pool.
* the [B, 128]-sized MLP / LayerNorm / gate algebra stays in torch (a few kFLOP per window).

The reference runs the encoder in ``eval()`` during training (train.py:1410-1411: BatchNorm uses its running
statistics, dropout is off) while still training its weights; ``train()`` mode (batch statistics, dropout)
is supported as well.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ACT_IDS, check
from .model import _norm_act

Tensor = torch.Tensor
SAMPLING_RATE_HZ = 1000   # models.py:23
MAX_FREQ_HZ = 100         # models.py:24


def band_log_power(x: Tensor, masks: Tensor, eps: float = 1e-10) -> Tensor:
    """x [B, C, T] -> log band power [B, C, K] (models.py:1294-1301); masks [K, T//2+1] fp32."""
    if not x.is_cuda:
        raise RuntimeError("unet_amd.cond runs on the GPU only; there is no CPU path")
    b, c, t = x.shape
    spec = torch.fft.rfft(x.float(), dim=-1).contiguous()            # [B, C, F] complex64
    f = spec.shape[-1]
    masks = masks.to(device=x.device, dtype=torch.float32).contiguous()
    if masks.shape[1] != f:
        raise ValueError(f"masks have {masks.shape[1]} bins, the spectrum {f}")
    out = torch.empty((b, c, masks.shape[0]), dtype=torch.float32, device=x.device)
    check(_lib.load().unet_band_log_power(ops._p(torch.view_as_real(spec)), ops._p(masks), b * c, f, masks.shape[0], eps,
                                          ops._p(out), ops._stream()), "unet_band_log_power")
    return out


class SpectroTemporalEncoder(nn.Module):
    def __init__(self, in_channels: int = 32, emb_dim: int = 128, n_freq_bands: int = 10, temporal_hidden: int = 64,
                 temporal_dilations: Tuple[int, ...] = (1, 4, 16), temporal_kernel: int = 7,
                 sample_rate: float = SAMPLING_RATE_HZ, max_freq: float = MAX_FREQ_HZ, dropout: float = 0.1):
        super().__init__()
        self.in_channels = in_channels
        self.emb_dim = emb_dim
        self.n_freq_bands = n_freq_bands
        self.sample_rate = sample_rate
        self.max_freq = max_freq
        self.temporal_dilations = tuple(temporal_dilations)
        self.temporal_kernel = temporal_kernel
        self.temporal_hidden = temporal_hidden
        if n_freq_bands > 16:
            raise NotImplementedError("unet_band_log_power supports up to 16 bands")
        self.spectral_encoder = nn.Sequential(
            nn.Linear(in_channels * n_freq_bands, emb_dim), nn.LayerNorm(emb_dim), nn.GELU(), nn.Dropout(dropout),
            nn.Linear(emb_dim, emb_dim // 2), nn.LayerNorm(emb_dim // 2))
        self.temporal_convs = nn.ModuleList([
            nn.Conv1d(in_channels, temporal_hidden, kernel_size=temporal_kernel, dilation=d,
                      padding=(temporal_kernel - 1) * d // 2) for d in temporal_dilations])
        self.temporal_fusion = nn.Sequential(
            nn.Conv1d(temporal_hidden * len(temporal_dilations), temporal_hidden, kernel_size=1),
            nn.BatchNorm1d(temporal_hidden), nn.GELU())
        self.temporal_pool = nn.AdaptiveAvgPool1d(1)
        self.temporal_proj = nn.Sequential(nn.Linear(temporal_hidden, emb_dim // 2), nn.LayerNorm(emb_dim // 2))
        self.output_gate = nn.Sequential(nn.Linear(emb_dim, emb_dim), nn.Sigmoid())
        self.output_proj = nn.Linear(emb_dim, emb_dim)
        self._freq_masks: Optional[Tensor] = None
        self._cached_length: int = 0
        self._masks_on = {}
        self.compute_dtype: Optional[torch.dtype] = None
        # merged tap list of the dilated branches: tap index of (branch, k)
        k = temporal_kernel
        offs = sorted({i * d - (k - 1) * d // 2 for d in self.temporal_dilations for i in range(k)})
        index = {o: j for j, o in enumerate(offs)}
        self._geom = ops.ConvGeom(tuple(offs), tuple(0 for _ in offs), 1, 1, len(self.temporal_dilations) * k / len(offs))
        self._tap_of = [[index[i * d - (k - 1) * d // 2] for i in range(k)] for d in self.temporal_dilations]
        self._idx_cache = {}

    def set_compute_dtype(self, dtype: Optional[torch.dtype]):
        """bf16: the temporal convs run on the bf16 kernels (fp32 master weights); None / fp32: exact fp32 kernels."""
        if dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype

    def _build_freq_masks(self, T: int, device) -> Tensor:
        """models.py:1245-1275 (built on the host in fp32, cached per length; the device copy is cached too, so a
        step has no host-to-device transfer and can be captured into a HIP graph)."""
        n_fft = T // 2 + 1
        if self._freq_masks is None or self._cached_length != n_fft:
            freqs = torch.fft.rfftfreq(T, d=1.0 / self.sample_rate)
            band_edges = torch.logspace(math.log10(1.0), math.log10(self.max_freq), self.n_freq_bands + 1)
            masks = []
            for i in range(self.n_freq_bands):
                low, high = band_edges[i], band_edges[i + 1]
                mask = torch.sigmoid((freqs - low) * 2) * torch.sigmoid((high - freqs) * 2)
                masks.append(mask / (mask.sum() + 1e-8))
            self._freq_masks = torch.stack(masks, dim=0)
            self._cached_length = n_fft
            self._masks_on = {}
        key = str(torch.device(device))
        if key not in self._masks_on:
            self._masks_on[key] = self._freq_masks.to(device)
        return self._masks_on[key]

    @property
    def _param_dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    def _spectral_forward(self, x: Tensor) -> Tensor:
        b = x.shape[0]
        masks = self._build_freq_masks(x.shape[-1], x.device)
        band_flat = band_log_power(x, masks).reshape(b, -1)                      # [B, C * n_bands]
        return self.spectral_encoder(band_flat.to(self._param_dtype))

    def _merged_weight(self) -> Tuple[Tensor, Tensor]:
        """[taps, n_scales * hidden, Cin] weight of the single tap-list conv (zeros where a branch has no tap) and
        the concatenated bias; built with one scatter so autograd routes dW back to the branch weights."""
        ns, h, cin, k = len(self.temporal_convs), self.temporal_hidden, self.in_channels, self.temporal_kernel
        stacked = torch.stack([c.weight for c in self.temporal_convs]).float()        # [ns, h, cin, k]
        src = stacked.permute(0, 3, 1, 2).reshape(ns * k, h, cin)
        dev = stacked.device
        idx = self._idx_cache.get(dev)
        if idx is None:   # index tensors are uploaded once per device (no host copy inside a captured step)
            idx = (torch.tensor([t for row in self._tap_of for t in row], device=dev),
                   torch.arange(ns, device=dev).repeat_interleave(k))
            self._idx_cache[dev] = idx
        tap, br = idx
        w = torch.zeros((self._geom.n_taps, ns, h, cin), dtype=torch.float32, device=dev)
        w = w.index_put((tap, br), src)
        bias = torch.cat([c.bias for c in self.temporal_convs]).float()
        return w.reshape(self._geom.n_taps, ns * h, cin), bias

    def _temporal_forward(self, x: Tensor) -> Tensor:
        cdt = self.compute_dtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        x_nlc = ops.ToNLCFn.apply(x.contiguous(), cdt)
        t = x.shape[-1]
        w, bias = self._merged_weight()
        gelu = ACT_IDS["gelu"]
        multi = ops.activation(ops.conv(x_nlc, w, bias, self._geom, t), gelu)           # [B, T, ns*hidden]
        fconv, bn = self.temporal_fusion[0], self.temporal_fusion[1]
        fused = ops.conv(multi, fconv.weight.permute(2, 0, 1), fconv.bias, ops.geom_conv(1), t)
        _, pool = _norm_act(fused, bn, gelu, True)                                       # sum over T of GELU(BN(.))
        pooled = (pool / float(t)).to(self._param_dtype)
        return self.temporal_proj(pooled)

    def forward(self, x: Tensor, odor_emb: Optional[Tensor] = None) -> Tensor:
        original_dtype = x.dtype
        param_dtype = self._param_dtype
        spectral_emb = self._spectral_forward(x).to(param_dtype)
        temporal_emb = self._temporal_forward(x).to(param_dtype)
        style_emb = torch.cat([spectral_emb, temporal_emb], dim=-1)
        gate = self.output_gate(style_emb)
        style_emb = self.output_proj(style_emb) * gate
        if odor_emb is not None:
            style_emb = style_emb + odor_emb.to(param_dtype)
        return style_emb.to(original_dtype)

    def extra_repr(self) -> str:
        return (f"in_channels={self.in_channels}, emb_dim={self.emb_dim}, n_freq_bands={self.n_freq_bands}, "
                f"dilations={self.temporal_dilations}")
