"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

CPU restatement (numpy fp64 / torch fp32 on the host) of the reference's loss and evaluation
reductions.  Pinned by ``tests/golden/metrics.npz``, which ``oracle/make_golden_host.py`` freezes from
the reference's own functions (``models.py``, ``train.py``, ``utils/dsp_metrics.py``);
``tests/test_oracle_golden.py`` checks this file against it.  Each function cites what it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

CROP = (500, 4500)  # data.py:102-115

BANDS = {"delta": (0.5, 4.0), "theta": (4.0, 8.0), "alpha": (8.0, 13.0), "beta": (13.0, 30.0),
         "gamma": (30.0, 100.0)}                                    # models.py:4799-4805
DSP_BANDS = {"delta": (0.5, 4.0), "theta": (4.0, 8.0), "alpha": (8.0, 13.0), "beta": (13.0, 30.0),
             "low_gamma": (30.0, 50.0), "high_gamma": (50.0, 100.0), "gamma": (30.0, 100.0)}  # dsp_metrics.py:36-44


def crop(x: np.ndarray) -> np.ndarray:
    """data.py:2392-2394."""
    return x[..., CROP[0]:CROP[1]]


def per_channel_normalize(x: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """train.py:476-495: unbiased std, clamped from below."""
    x = x.astype(np.float64)
    m = x.mean(axis=-1, keepdims=True)
    s = np.maximum(x.std(axis=-1, ddof=1, keepdims=True), eps)
    return (x - m) / s


def window_zscore(x: np.ndarray) -> np.ndarray:
    """data.py:3120-3124: numpy std (biased) + 1e-8."""
    x = x.astype(np.float64)
    return (x - x.mean(axis=-1, keepdims=True)) / (x.std(axis=-1, keepdims=True) + 1e-8)


def l1_crop(pred: np.ndarray, target: np.ndarray, weight: float = 1.0, lo: int = CROP[0], hi: int = CROP[1]):
    """train.py:1481-1486 and its gradient w.r.t. pred (sign / count inside the crop, 0 outside)."""
    p, t = pred[..., lo:hi].astype(np.float64), target[..., lo:hi].astype(np.float64)
    d = p - t
    loss = weight * np.abs(d).mean()
    g = np.zeros(pred.shape, dtype=np.float64)
    g[..., lo:hi] = weight * np.sign(d) / d.size
    return loss, g


def pooled_sums(pred: np.ndarray, target: np.ndarray) -> np.ndarray:
    """train.py:941-952: n, St, Stt, Sp, Spp, Spt, S(p-t)^2, S|p-t| (fp64 here)."""
    p, t = pred.astype(np.float64).ravel(), target.astype(np.float64).ravel()
    d = (pred.astype(np.float32).ravel() - target.astype(np.float32).ravel()).astype(np.float64)  # fp32 difference, as the reference's tensors
    return np.array([p.size, t.sum(), (t * t).sum(), p.sum(), (p * p).sum(), (p * t).sum(), (d * d).sum(), np.abs(d).sum()])


def pooled_metrics(s) -> Dict[str, float]:
    """train.py:1098-1135."""
    n, st, stt, sp, spp, spt, sse, sae = [float(v) for v in s]
    if n <= 0:
        return dict(mse=0.0, mae=0.0, corr=0.0, r2=0.0)
    tmean, pmean = st / n, sp / n
    ss_tot = stt - st * st / n
    r2 = 1.0 - sse / ss_tot if ss_tot > 1e-12 else 0.0
    cov = spt / n - pmean * tmean
    vp, vt = spp / n - pmean ** 2, stt / n - tmean ** 2
    corr = max(-1.0, min(1.0, cov / (math.sqrt(vp) * math.sqrt(vt)))) if (vp > 1e-12 and vt > 1e-12) else 0.0
    return dict(mse=sse / n, mae=sae / n, corr=corr, r2=r2)


def pearson_rows(pred: np.ndarray, target: np.ndarray) -> np.ndarray:
    """models.py:4264-4268 per (batch, channel) row."""
    p, t = pred.astype(np.float64), target.astype(np.float64)
    pc, tc = p - p.mean(-1, keepdims=True), t - t.mean(-1, keepdims=True)
    return (pc * tc).sum(-1) / np.sqrt((pc ** 2).sum(-1) * (tc ** 2).sum(-1) + 1e-8)


def pearson_batch(pred, target) -> float:
    return float(pearson_rows(pred, target).mean())


def pearson_per_channel(pred, target) -> np.ndarray:
    return pearson_rows(pred, target).mean(axis=0)


def explained_variance(pred, target) -> float:
    """models.py:4331-4362."""
    p, t = pred.astype(np.float64), target.astype(np.float64)
    return float(1.0 - ((p - t) ** 2).mean() / max(t.var(ddof=1), 1e-12))


def nrmse(pred, target) -> float:
    """models.py:4365-4397."""
    p, t = pred.astype(np.float64), target.astype(np.float64)
    return float(math.sqrt(((p - t) ** 2).mean()) / max(t.std(ddof=1), 1e-12))


def hann(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic) == scipy get_window('hann', n)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def welch_psd(x: np.ndarray, fs: float, nperseg: int, scipy_convention: bool = False) -> np.ndarray:
    """models.py:4512-4550 (unfold(nperseg, hop) * hann -> rfft -> mean |X|^2 * 2/(fs*S2)); with
    scipy_convention the defaults of scipy.signal.welch (detrend='constant', DC/Nyquist not doubled)."""
    x = x.astype(np.float64).reshape(-1, x.shape[-1])
    hop = nperseg // 2
    n_seg = (x.shape[-1] - nperseg) // hop + 1
    w = hann(nperseg)
    seg = np.stack([x[:, s * hop:s * hop + nperseg] for s in range(n_seg)], axis=1)
    if scipy_convention:
        seg = seg - seg.mean(axis=-1, keepdims=True)
    X = np.fft.rfft(seg * w, axis=-1)
    psd = (np.abs(X) ** 2).mean(axis=1) * (2.0 / (fs * (w ** 2).sum()))
    if scipy_convention:
        psd[:, 0] *= 0.5
        psd[:, -1] *= 0.5
    return psd


def _mask(nperseg, fs, lo, hi):
    f = np.fft.rfftfreq(nperseg, d=1.0 / fs).astype(np.float32)   # torch.fft.rfftfreq is fp32
    return (f >= lo) & (f <= hi)


def psd_error_db(pred, target, fs=1000.0, nperseg=512, freq_range=(0.0, 100.0), normalize=False, signed=False) -> float:
    """models.py:4553-4629 (and the signed variant :4632-4697)."""
    pp, pt = welch_psd(pred, fs, nperseg), welch_psd(target, fs, nperseg)
    m = _mask(nperseg, fs, *freq_range)
    pp, pt = pp[:, m], pt[:, m]
    if normalize:
        pp = pp / np.maximum(pp.sum(-1, keepdims=True), 1e-12)
        pt = pt / np.maximum(pt.sum(-1, keepdims=True), 1e-12)
    d = 10 * np.log10(pp + 1e-12) - 10 * np.log10(pt + 1e-12)
    return float(d.mean() if signed else np.abs(d).mean())


def psd_diagnostics(pred, target, fs=1000.0, nperseg=512, freq_range=(0.0, 100.0)) -> dict:
    """models.py:4700-4827."""
    eps = 1e-12
    pp, pt = welch_psd(pred, fs, nperseg).mean(0), welch_psd(target, fs, nperseg).mean(0)
    f = np.fft.rfftfreq(nperseg, d=1.0 / fs).astype(np.float32)
    m = (f >= freq_range[0]) & (f <= freq_range[1])
    pf, tf, ff = pp[m], pt[m], f[m]
    ptot, ttot = pf.sum(), tf.sum()
    pdb, tdb = 10 * np.log10(pf + eps), 10 * np.log10(tf + eps)
    pn, tn = pf / (ptot + eps), tf / (ttot + eps)
    pc, tc = pn - pn.mean(), tn - tn.mean()
    bands = {}
    for k, (lo, hi) in BANDS.items():
        bm = (ff >= lo) & (ff <= hi)
        bands[k] = float(10 * np.log10((pf[bm].sum() + eps) / (tf[bm].sum() + eps))) if bm.any() else 0.0
    ps, ts = float(pred.astype(np.float64).std(ddof=1)), float(target.astype(np.float64).std(ddof=1))
    return dict(total_power_ratio_db=float(10 * np.log10((ptot + eps) / (ttot + eps))),
                psd_error_db=float(np.abs(pdb - tdb).mean()), psd_bias_db=float((pdb - tdb).mean()),
                spectral_correlation=float((pc * tc).sum() / ((math.sqrt((pc ** 2).sum()) + eps) * (math.sqrt((tc ** 2).sum()) + eps))),
                band_power_ratios_db=bands, pred_std=ps, target_std=ts, std_ratio=ps / (ts + eps))


def compute_psd(sig: np.ndarray, fs: float = 1000.0, nperseg: int = 256) -> dict:
    """utils/dsp_metrics.py:128-194 for one 1-D signal (scipy.signal.welch defaults restated in welch_psd)."""
    psd = welch_psd(sig[None, :], fs, nperseg, scipy_convention=True)[0]
    f = np.fft.rfftfreq(nperseg, d=1.0 / fs)
    bands = {}
    for k, (lo, hi) in DSP_BANDS.items():
        bm = (f >= lo) & (f <= hi)
        bands[k] = float(np.trapz(psd[bm], f[bm])) if bm.any() else 0.0
    pn = psd / (psd.sum() + 1e-10)
    return dict(frequencies=f, psd=psd, band_powers=bands, total_power=float(np.trapz(psd, f)),
                dominant_frequency=float(f[np.argmax(psd)]), spectral_entropy=float(-(pn * np.log2(pn + 1e-10)).sum()))


def overlap_average(preds: np.ndarray, n_samples: int, stride: int) -> np.ndarray:
    """phase_five/sliding_window.py:230-251 ('average' mode): preds[w] covers [w*stride, w*stride+W)."""
    nw, c, w = preds.shape
    out = np.zeros((c, n_samples), dtype=np.float32)
    cnt = np.zeros(n_samples, dtype=np.float32)
    for i in range(nw):
        out[:, i * stride:i * stride + w] += preds[i]
        cnt[i * stride:i * stride + w] += 1
    return out / np.maximum(cnt, 1)
