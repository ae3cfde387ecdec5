"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

CPU restatement of the data path in front of the model: NoiseAugmentor (models.py:4835-4946) and
ECoGDataset.__getitem__ (data.py:4586-4612).  Pinned by ``tests/golden/data_path.npz``, frozen from the
reference's own classes by ``oracle/make_golden_data.py`` (the augmentor is run on the CPU under a fixed seed;
``forward`` below draws from torch's generator in the reference's order, so it reproduces those outputs).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def pink_from_white(white: torch.Tensor, T: int, sample_rate: float) -> torch.Tensor:
    """models.py:4880-4895 after the draw of ``white`` (complex64 [B, C, T//2+1])."""
    freqs = torch.fft.rfftfreq(T, d=1.0 / sample_rate)
    freqs[0] = 1e-6
    pink_filter = 1.0 / torch.sqrt(freqs)
    pink_filter = pink_filter / pink_filter.max()
    pink = torch.fft.irfft(white * pink_filter, n=T)
    return pink / (pink.std() + 1e-6)


def apply(x, gauss=None, pink=None, ch_u=None, t_u=None, amp_u=None, phase_u=None, *, gaussian_std=0.1, pink_noise_std=0.05,
          channel_dropout=0.0, temporal_dropout=0.0, line_noise_hz=0.0, sample_rate=1000.0):
    """models.py:4913-4946 given the random draws (``pink`` already normalised to unit std)."""
    B, C, T = x.shape
    dtype = x.dtype
    data_std = x.std().clamp(min=1e-6)
    if gauss is not None:
        x = x + gauss * gaussian_std * data_std
    if pink is not None:
        x = x + pink.to(dtype) * pink_noise_std * data_std
    if ch_u is not None:
        x = x * (ch_u > channel_dropout).to(dtype)
    if t_u is not None:
        x = x * (t_u > temporal_dropout).to(dtype)
    if amp_u is not None:
        t = torch.linspace(0, T / sample_rate, T, dtype=dtype)
        amp = amp_u * 0.02 * data_std
        phase = phase_u * 2 * np.pi
        x = x + amp * torch.sin(2 * np.pi * line_noise_hz * t + phase)
    return x


def forward(x, *, gaussian_std=0.1, pink_noise=False, pink_noise_std=0.05, channel_dropout=0.0, temporal_dropout=0.0,
            line_noise_hz=0.0, sample_rate=1000.0, prob=0.5, training=True):
    """models.py:4897-4946 with the draws made from torch's global generator in the reference's order."""
    if not training or torch.rand(1).item() > prob:
        return x
    B, C, T = x.shape
    gauss = torch.randn_like(x) if gaussian_std > 0 else None
    pink = None
    if pink_noise and pink_noise_std > 0:
        pink = pink_from_white(torch.randn(B, C, T // 2 + 1, dtype=torch.complex64), T, sample_rate)
    ch_u = torch.rand(B, C, 1) if channel_dropout > 0 else None
    t_u = torch.rand(B, 1, T) if temporal_dropout > 0 else None
    amp_u = phase_u = None
    if line_noise_hz > 0:
        amp_u = torch.rand(B, 1, 1, dtype=x.dtype)
        phase_u = torch.rand(B, 1, 1, dtype=x.dtype)
    return apply(x, gauss, pink, ch_u, t_u, amp_u, phase_u, gaussian_std=gaussian_std, pink_noise_std=pink_noise_std,
                 channel_dropout=channel_dropout, temporal_dropout=temporal_dropout, line_noise_hz=line_noise_hz,
                 sample_rate=sample_rate)


def ecog_item(subjects, windows, idx, window_size, zscore_per_window):
    """data.py:4586-4612: (source, target) float32 arrays of window ``idx``."""
    subj_idx, start = windows[idx]
    src = subjects[subj_idx]["source"][:, start:start + window_size].copy()
    tgt = subjects[subj_idx]["target"][:, start:start + window_size].copy()
    if zscore_per_window:
        src = (src - src.mean(axis=1, keepdims=True)) / (src.std(axis=1, keepdims=True) + 1e-8)
        tgt = (tgt - tgt.mean(axis=1, keepdims=True)) / (tgt.std(axis=1, keepdims=True) + 1e-8)
    return src.astype(np.float32), tgt.astype(np.float32)
