"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

CPU restatement (torch fp32, functional over a state_dict) of the reference's SpectroTemporalEncoder
(models.py:1135-1395).  Pinned by ``tests/golden/cond_spectro_temporal.npz``, frozen from the reference's own
module by ``oracle/make_golden_cond.py``; ``tests/test_oracle_cond.py`` checks this file against it.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def freq_masks(T: int, sample_rate: float = 1000.0, max_freq: float = 100.0, n_bands: int = 10) -> torch.Tensor:
    """models.py:1245-1275."""
    freqs = torch.fft.rfftfreq(T, d=1.0 / sample_rate)
    edges = torch.logspace(math.log10(1.0), math.log10(max_freq), n_bands + 1)
    masks = []
    for i in range(n_bands):
        m = torch.sigmoid((freqs - edges[i]) * 2) * torch.sigmoid((edges[i + 1] - freqs) * 2)
        masks.append(m / (m.sum() + 1e-8))
    return torch.stack(masks, 0)


def band_log_power(x: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """models.py:1294-1301."""
    power = torch.fft.rfft(x.float(), dim=-1).abs().square()
    return torch.log(torch.einsum("bcf,kf->bck", power, masks) + 1e-10)


def forward(sd, x: torch.Tensor, odor_emb=None, dilations=(1, 4, 16), kernel=7, training: bool = False) -> torch.Tensor:
    """models.py:1277-1389 in eval mode (how train.py runs it, :1410-1411), or with batch statistics when
    ``training`` (dropout left out: it is random)."""
    B = x.shape[0]
    ln = lambda v, p: F.layer_norm(v, (v.shape[-1],), sd[p + ".weight"], sd[p + ".bias"])
    bl = band_log_power(x, freq_masks(x.shape[-1])).reshape(B, -1)
    s = F.linear(bl, sd["spectral_encoder.0.weight"], sd["spectral_encoder.0.bias"])
    s = F.gelu(ln(s, "spectral_encoder.1"))
    s = ln(F.linear(s, sd["spectral_encoder.4.weight"], sd["spectral_encoder.4.bias"]), "spectral_encoder.5")
    feats = [F.gelu(F.conv1d(x, sd[f"temporal_convs.{i}.weight"], sd[f"temporal_convs.{i}.bias"], dilation=d,
                             padding=(kernel - 1) * d // 2)) for i, d in enumerate(dilations)]
    f = F.conv1d(torch.cat(feats, 1), sd["temporal_fusion.0.weight"], sd["temporal_fusion.0.bias"])
    f = F.batch_norm(f, sd["temporal_fusion.1.running_mean"].clone(), sd["temporal_fusion.1.running_var"].clone(),
                     sd["temporal_fusion.1.weight"], sd["temporal_fusion.1.bias"], training, 0.1, 1e-5)
    t = ln(F.linear(F.gelu(f).mean(-1), sd["temporal_proj.0.weight"], sd["temporal_proj.0.bias"]), "temporal_proj.1")
    style = torch.cat([s, t], -1)
    gate = torch.sigmoid(F.linear(style, sd["output_gate.0.weight"], sd["output_gate.0.bias"]))
    out = F.linear(style, sd["output_proj.weight"], sd["output_proj.bias"]) * gate
    return out + odor_emb if odor_emb is not None else out
