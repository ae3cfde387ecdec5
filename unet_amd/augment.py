"""NoiseAugmentor (models.py:4835-4946) on the HIP path.

Same constructor and behaviour as the reference: in training mode, with probability ``prob`` (a host-side
``torch.rand(1)`` exactly as the reference draws it), add Gaussian noise and 1/f noise scaled by the batch's
standard deviation, drop random channels and time points, add line hum.  The random numbers are drawn with
torch's generator in the reference's order (``randn_like(x)``, complex ``randn`` for the pink spectrum,
``rand`` for the two masks, ``rand`` twice for the hum), so on the same device and seed the stream is the one
the reference consumes.  Everything after the draws -- which the reference spreads over about twelve
elementwise passes -- is ONE kernel (``unet_noise_augment``): x, the Gaussian draw and the pink series are read
once, y is written once; the two standard deviations stay on the device (no host read).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .ops import _dt, _p, _stream

Tensor = torch.Tensor


def noise_augment_apply(x: Tensor, gauss: Optional[Tensor], pink: Optional[Tensor], scales: Optional[Tensor],
                        ch_u: Optional[Tensor] = None, channel_dropout: float = 0.0, t_u: Optional[Tensor] = None,
                        temporal_dropout: float = 0.0, line_amp: Optional[Tensor] = None, line_phase: Optional[Tensor] = None,
                        line_noise_hz: float = 0.0, sample_rate: float = 1000.0) -> Tensor:
    """The arithmetic of models.py:4916-4944 after the draws (see include/unet_hip.h unet_noise_augment)."""
    if not x.is_cuda:
        raise RuntimeError("unet_amd.augment runs on the GPU only; there is no CPU path")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("NoiseAugmentor supports float32 and bfloat16 inputs")
    x = x.contiguous()
    n, c, t = x.shape
    f32 = lambda v: None if v is None else v.reshape(-1).to(torch.float32).contiguous()
    same = lambda v: None if v is None else v.to(x.dtype).contiguous()
    gauss, pink = same(gauss), same(pink)
    ch_u, t_u, line_amp, line_phase, scales = f32(ch_u), f32(t_u), f32(line_amp), f32(line_phase), f32(scales)
    y = torch.empty_like(x)
    check(_lib.load().unet_noise_augment(_dt(x), _p(x), _p(gauss), _p(pink), _p(scales), _p(ch_u), float(channel_dropout),
                                         _p(t_u), float(temporal_dropout), _p(line_amp), _p(line_phase), float(line_noise_hz),
                                         float(sample_rate), n, c, t, _p(y), _stream()), "unet_noise_augment")
    return y


class NoiseAugmentor(nn.Module):
    def __init__(self, gaussian_std: float = 0.1, pink_noise: bool = False, pink_noise_std: float = 0.05,
                 channel_dropout: float = 0.0, temporal_dropout: float = 0.0, line_noise_hz: float = 0.0,
                 sample_rate: float = 1000.0, prob: float = 0.5):
        super().__init__()
        self.gaussian_std = gaussian_std
        self.pink_noise = pink_noise
        self.pink_noise_std = pink_noise_std
        self.channel_dropout = channel_dropout
        self.temporal_dropout = temporal_dropout
        self.line_noise_hz = line_noise_hz
        self.sample_rate = sample_rate
        self.prob = prob
        self._filters = {}

    def _pink_filter(self, T: int, device) -> Tensor:
        """models.py:4881-4884, cached per (length, device)."""
        key = (T, str(device))
        f = self._filters.get(key)
        if f is None:
            freqs = torch.fft.rfftfreq(T, d=1.0 / self.sample_rate, device=device)
            freqs[0] = 1e-6
            f = 1.0 / torch.sqrt(freqs)
            f = f / f.max()
            self._filters[key] = f
        return f

    def _pink_raw(self, shape, device) -> Tensor:
        """models.py:4875-4890 up to the irfft (the division by its std is folded into the kernel's scale)."""
        B, C, T = shape
        white = torch.randn(B, C, T // 2 + 1, device=device, dtype=torch.complex64)
        return torch.fft.irfft(white * self._pink_filter(T, device), n=T)

    def forward(self, x: Tensor) -> Tensor:
        if not self.training or torch.rand(1).item() > self.prob:
            return x
        B, C, T = x.shape
        device, dtype = x.device, x.dtype
        data_std = x.float().std().clamp(min=1e-6)
        gauss = torch.randn_like(x) if self.gaussian_std > 0 else None
        pink = None
        sp = data_std.new_zeros(())
        if self.pink_noise and self.pink_noise_std > 0:
            pink = self._pink_raw((B, C, T), device)
            sp = self.pink_noise_std * data_std / (pink.std() + 1e-6)
        scales = torch.stack([self.gaussian_std * data_std, sp]) if (gauss is not None or pink is not None) else None
        ch_u = torch.rand(B, C, 1, device=device) if self.channel_dropout > 0 else None
        t_u = torch.rand(B, 1, T, device=device) if self.temporal_dropout > 0 else None
        amp = phase = None
        if self.line_noise_hz > 0:
            amp = torch.rand(B, 1, 1, device=device, dtype=dtype).float() * 0.02 * data_std
            phase = torch.rand(B, 1, 1, device=device, dtype=dtype).float() * (2 * torch.pi)
        return noise_augment_apply(x, gauss, pink, scales, ch_u, self.channel_dropout, t_u, self.temporal_dropout, amp, phase,
                                   self.line_noise_hz, self.sample_rate)
