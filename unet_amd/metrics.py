"""Loss / evaluation path of the reference on the HIP reduction kernels (csrc/metrics.hip).

Mirrors, name for name, the functions the reference's ``train.py`` calls around the model:

    crop_to_target_torch        data.py:2392-2394   (absolute [500:4500] crop)
    per_channel_normalize       train.py:476-495    (target z-score, unbiased std clamped at eps)
    l1_crop_loss                train.py:1481-1486  (w * F.l1_loss(crop(pred), crop(target)), with backward)
    pearson_batch / pearson_per_channel / explained_variance_torch / normalized_rmse_torch
                                models.py:4262-4287, 4331-4397
    PooledStats                 train.py:939-952, 1074-1135 (7 pooled sums -> R^2, Pearson, MSE, MAE)
    psd_error_db_torch / psd_diff_db_torch / psd_diagnostics_torch
                                models.py:4553-4827 (Welch PSD, dB error / bias, band-power ratios)
    compute_psd                 utils/dsp_metrics.py:128-194 (scipy.signal.welch + np.trapz band powers)

Every pass over a ``[B, C, T]`` tensor is ONE HIP kernel that reads it once; what remains in torch is
arithmetic on ``[B*C, 8]`` / ``[B*C, 257]`` results.  GPU only -- there is no CPU path (tests compare
against oracle/metrics_oracle.py).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .ops import _dt, _need_cuda, _p, _stream

Tensor = torch.Tensor

SAMPLING_RATE_HZ = 1000      # models.py:23
MAX_FREQ_HZ = 100            # models.py:24
CROP_START, CROP_END = 500, 4500   # data.py:102-115

BAND_DEFINITIONS = {         # models.py:4799-4805
    "delta": (0.5, 4.0), "theta": (4.0, 8.0), "alpha": (8.0, 13.0), "beta": (13.0, 30.0), "gamma": (30.0, 100.0),
}
FREQUENCY_BANDS = {          # utils/dsp_metrics.py:36-44
    "delta": (0.5, 4.0), "theta": (4.0, 8.0), "alpha": (8.0, 13.0), "beta": (13.0, 30.0),
    "low_gamma": (30.0, 50.0), "high_gamma": (50.0, 100.0), "gamma": (30.0, 100.0),
}


def crop_to_target_torch(x: Tensor) -> Tensor:
    return x[..., CROP_START:CROP_END]


def _rows(x: Tensor) -> Tuple[Tensor, int, int, int]:
    """View a [..., T] tensor as rows with a common pitch; returns (tensor, rows, pitch, T).
    Accepts time-slices of contiguous tensors (the crop) without copying."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    t = x.shape[-1]
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if x.stride(-1) != 1:
        x = x.contiguous()
    lead = x.shape[:-1]
    rows = 1
    for s in lead:
        rows *= s
    # rows are equally spaced iff the leading dims collapse with one stride
    pitch = x.stride(-2) if x.dim() >= 2 else t
    ok = True
    expect = pitch
    for d in range(x.dim() - 2, -1, -1):
        if x.shape[d] != 1 and x.stride(d) != expect:
            ok = False
            break
        expect *= x.shape[d]
    if not ok:
        x = x.contiguous()
        pitch = t
    return x, rows, pitch, t


def pair_moments(pred: Tensor, target: Tensor) -> Tensor:
    """[rows, 8] fp64: n, St, Stt, Sp, Spp, Spt, S(p-t)^2, S|p-t| per (window, channel) row."""
    _need_cuda(pred)
    if pred.shape != target.shape:
        raise ValueError(f"shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
    if pred.dtype != target.dtype:
        pred, target = pred.float(), target.float()
    p, rows, ld_p, t = _rows(pred)
    q, _, ld_t, _ = _rows(target)
    out = torch.empty((rows, 8), dtype=torch.float64, device=pred.device)
    check(_lib.load().unet_pair_moments(_dt(p), _p(p), _p(q), rows, ld_p, ld_t, 0, t, _p(out), _stream()),
          "unet_pair_moments")
    return out


def _total(rows8: Tensor) -> Tensor:
    out = torch.empty(rows8.shape[1], dtype=torch.float64, device=rows8.device)
    check(_lib.load().unet_sum_rows_f64(_p(rows8), rows8.shape[0], rows8.shape[1], _p(out), _stream()),
          "unet_sum_rows_f64")
    return out


def _pearson_rows(m: Tensor) -> Tensor:
    n, st, stt, sp, spp, spt = (m[:, i] for i in range(6))
    num = spt - sp * st / n
    den = torch.sqrt((spp - sp * sp / n).clamp_min(0) * (stt - st * st / n).clamp_min(0) + 1e-8)
    return num / den


def pearson_batch(pred: Tensor, target: Tensor) -> Tensor:
    """models.py:4262-4269."""
    return _pearson_rows(pair_moments(pred, target)).mean().float()


def pearson_per_channel(pred: Tensor, target: Tensor) -> Tensor:
    """models.py:4272-4287 -> [C]."""
    b, c, _ = pred.shape
    return _pearson_rows(pair_moments(pred, target)).reshape(b, c).mean(dim=0).float()


def explained_variance_torch(pred: Tensor, target: Tensor) -> Tensor:
    """models.py:4331-4362: 1 - mse / var_unbiased(target), over ALL elements."""
    s = _total(pair_moments(pred, target))
    n = s[0]
    var = ((s[2] - s[1] * s[1] / n) / (n - 1)).clamp_min(1e-12)
    return (1.0 - (s[6] / n) / var).float()


def normalized_rmse_torch(pred: Tensor, target: Tensor) -> Tensor:
    """models.py:4365-4397."""
    s = _total(pair_moments(pred, target))
    n = s[0]
    std = torch.sqrt(((s[2] - s[1] * s[1] / n) / (n - 1)).clamp_min(0)).clamp_min(1e-12)
    return (torch.sqrt(s[6] / n) / std).float()


class PooledStats:
    """train.py:939-952 + 1074-1135: accumulate the 7 pooled sums over batches ON THE DEVICE (the
    reference does 8 ``.item()`` syncs per batch), all-reduce once, then R^2 / Pearson / MSE / MAE."""

    def __init__(self, device):
        self.sums = torch.zeros(8, dtype=torch.float64, device=device)
        self.corr_list, self.r2_list = [], []

    def update(self, pred_c: Tensor, target_c: Tensor):
        m = pair_moments(pred_c, target_c)
        tot = _total(m)
        self.sums += tot
        # legacy per-batch metrics (train.py:931-934), kept on device
        self.corr_list.append(_pearson_rows(m).mean())
        n = tot[0]
        var = ((tot[2] - tot[1] * tot[1] / n) / (n - 1)).clamp_min(1e-12)
        self.r2_list.append(1.0 - (tot[6] / n) / var)

    def all_reduce(self, group=None):
        from .dp import group_active
        if group_active(group):                                             # dp.NO_GROUP: a fold job, nothing to pool
            dist.all_reduce(self.sums, op=dist.ReduceOp.SUM, group=group)   # train.py:1087

    def finalize(self) -> Dict[str, float]:
        return pooled_metrics_from_sums(self.sums.tolist(),
                                        [float(c) for c in self.corr_list], [float(r) for r in self.r2_list])


def pooled_metrics_from_sums(s, corr_list=(), r2_list=()) -> Dict[str, float]:
    """train.py:1098-1146 (host arithmetic on 8 numbers)."""
    n, st, stt, sp, spp, spt, sse, sae = s
    if n > 0:
        mse, mae = sse / n, sae / n
        tmean = st / n
        ss_tot = stt - st * st / n
        r2 = 1.0 - sse / ss_tot if ss_tot > 1e-12 else 0.0
        pmean = sp / n
        cov = spt / n - pmean * tmean
        vp, vt = spp / n - pmean * pmean, stt / n - tmean * tmean
        if vp > 1e-12 and vt > 1e-12:
            corr = max(-1.0, min(1.0, cov / (math.sqrt(vp) * math.sqrt(vt))))
        else:
            corr = 0.0
    else:
        mse = mae = r2 = corr = 0.0
    out = {"mse": float(mse), "mae": float(mae), "corr": float(corr), "r2": float(r2)}
    if len(corr_list):
        out["corr_batch_avg"] = float(sum(corr_list) / len(corr_list))
        out["r2_batch_avg"] = float(sum(r2_list) / len(r2_list))
    return out


# ---------------------------------------------------------------------------------------------
# target normalisation and the training loss
# ---------------------------------------------------------------------------------------------
def row_zscore(x: Tensor, eps: float, mode: int, unbiased: bool, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    _need_cuda(x)
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = x.contiguous()
    y = torch.empty_like(x, dtype=out_dtype or x.dtype)
    t = x.shape[-1]
    check(_lib.load().unet_row_zscore(_dt(x), _dt(y), _p(x), _p(y), x.numel() // t, t, eps, mode, int(unbiased),
                                      _stream()), "unet_row_zscore")
    return y


def per_channel_normalize(x: Tensor, eps: float = 1e-6) -> Tensor:
    """train.py:476-495 (no gradient: it is applied to the target batch)."""
    return row_zscore(x, eps, 1, True)


def window_zscore(x: Tensor) -> Tensor:
    """data.py:3120-3124 / 4648-4654 per-window z-score of the datasets (numpy std: biased, +1e-8)."""
    return row_zscore(x, 1e-8, 0, False)


class _L1CropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lo, hi, weight):
        _need_cuda(pred)
        if pred.shape != target.shape:
            raise ValueError(f"shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        pred_c = pred if pred.is_contiguous() else pred.contiguous()
        tgt = target if (target.is_contiguous() and target.dtype == pred_c.dtype) else target.to(pred_c.dtype).contiguous()
        t = pred_c.shape[-1]
        hi = min(hi, t)
        rows = pred_c.numel() // t
        m = torch.empty((rows, 8), dtype=torch.float64, device=pred.device)
        lib = _lib.load()
        check(lib.unet_pair_moments(_dt(pred_c), _p(pred_c), _p(tgt), rows, t, t, lo, hi, _p(m), _stream()),
              "unet_pair_moments")
        tot = _total(m)
        ctx.save_for_backward(pred_c, tgt)
        ctx.geom = (rows, t, lo, hi, float(weight))
        return (tot[7] / tot[0] * weight).float()

    @staticmethod
    def backward(ctx, g):
        pred_c, tgt = ctx.saved_tensors
        rows, t, lo, hi, weight = ctx.geom
        gs = g.reshape(1).float().contiguous()
        dpred = torch.empty_like(pred_c)
        check(_lib.load().unet_l1_crop_bwd(_dt(pred_c), _dt(dpred), _p(pred_c), _p(tgt), rows, t, t, t, lo, hi, _p(gs),
                                           weight / (rows * (hi - lo)), _p(dpred), _stream()), "unet_l1_crop_bwd")
        return dpred, None, None, None, None


def l1_crop_loss(pred: Tensor, target: Tensor, weight: float = 1.0, lo: int = CROP_START, hi: int = CROP_END) -> Tensor:
    """``weight * F.l1_loss(crop_to_target_torch(pred), crop_to_target_torch(target))`` (train.py:1481-1486)
    as one reduction kernel forward and one sign kernel backward; the crop is never materialised."""
    return _L1CropFn.apply(pred, target, lo, hi, weight)


# ---------------------------------------------------------------------------------------------
# Welch PSD family
# ---------------------------------------------------------------------------------------------
def _hann_s2(nperseg: int) -> float:
    w = torch.hann_window(nperseg, dtype=torch.float32)
    return float((w ** 2).sum())


def welch_psd(x: Tensor, fs: float, nperseg: int, scipy_convention: bool = False) -> Tensor:
    """[..., T] -> [rows, nperseg//2+1] fp32.  Default = models.py:4512-4550 (no detrend, every bin x2);
    scipy_convention = scipy.signal.welch defaults (constant detrend, DC/Nyquist not doubled)."""
    _need_cuda(x)
    xr, rows, ld, t = _rows(x)
    s2 = _hann_s2(nperseg)
    scale = 2.0 / (fs * s2)
    edge = 1.0 / (fs * s2) if scipy_convention else scale
    psd = torch.empty((rows, nperseg // 2 + 1), dtype=torch.float32, device=x.device)
    check(_lib.load().unet_welch_psd(_dt(xr), _p(xr), rows, ld, 0, t, nperseg, int(scipy_convention), scale, edge,
                                     _p(psd), _stream()), "unet_welch_psd")
    return psd


def _freq_bins(nperseg: int, fs: float, lo: float, hi: float) -> Tuple[int, int]:
    """first/last rfft bin with lo <= f <= hi, f = torch.fft.rfftfreq(nperseg, 1/fs) (fp32, as the reference)."""
    f = torch.fft.rfftfreq(nperseg, d=1.0 / fs)
    idx = torch.nonzero((f >= lo) & (f <= hi)).flatten()
    if idx.numel() == 0:
        return 0, -1
    return int(idx[0]), int(idx[-1])


def _psd_db_rows(pred: Tensor, target: Tensor, fs, nperseg, freq_range, normalize) -> Tuple[Tensor, int]:
    pp = welch_psd(pred, fs, nperseg)
    pt = welch_psd(target, fs, nperseg)
    k_lo, k_hi = _freq_bins(nperseg, fs, *freq_range)
    out = torch.empty((pp.shape[0], 2), dtype=torch.float64, device=pp.device)
    check(_lib.load().unet_psd_db_stats(_p(pp), _p(pt), pp.shape[0], pp.shape[1], k_lo, k_hi, int(normalize), _p(out),
                                        _stream()), "unet_psd_db_stats")
    return out, (k_hi - k_lo + 1)


def psd_error_db_torch(pred: Tensor, target: Tensor, fs: float = SAMPLING_RATE_HZ, nperseg: int = 512,
                       freq_range: tuple = (0.0, MAX_FREQ_HZ), normalize_psd: bool = False) -> Tensor:
    """models.py:4553-4629."""
    rows, nk = _psd_db_rows(pred, target, fs, nperseg, freq_range, normalize_psd)
    return (rows[:, 0].sum() / (rows.shape[0] * nk)).float()


def psd_diff_db_torch(pred: Tensor, target: Tensor, fs: float = SAMPLING_RATE_HZ, nperseg: int = 512,
                      freq_range: tuple = (0.0, MAX_FREQ_HZ)) -> Tensor:
    """models.py:4632-4697."""
    rows, nk = _psd_db_rows(pred, target, fs, nperseg, freq_range, False)
    return (rows[:, 1].sum() / (rows.shape[0] * nk)).float()


def band_power(psd: Tensor, nperseg: int, fs: float, bands: Dict[str, Tuple[float, float]], mode: str = "sum",
               within: Optional[Tuple[float, float]] = None) -> Tensor:
    """[rows, nfreq] -> [rows, len(bands)]: inclusive-mask band sums (mode 'sum', models.py:4808-4812) or
    np.trapz over the masked bins (mode 'trapz', dsp_metrics.py:168-177)."""
    lo_l, hi_l = [], []
    for f_lo, f_hi in bands.values():
        a, b = _freq_bins(nperseg, fs, f_lo, f_hi)
        if within is not None:
            wa, wb = _freq_bins(nperseg, fs, *within)
            a, b = max(a, wa), min(b, wb)
        lo_l.append(a)
        hi_l.append(b)
    nb = len(lo_l)
    psd = psd.contiguous()
    out = torch.empty((psd.shape[0], nb), dtype=torch.float32, device=psd.device)
    lo_a, hi_a = (C.c_int32 * nb)(*lo_l), (C.c_int32 * nb)(*hi_l)
    check(_lib.load().unet_band_power(_p(psd), psd.shape[0], psd.shape[1], nb, lo_a, hi_a, 1 if mode == "trapz" else 0,
                                      fs / nperseg, _p(out), _stream()), "unet_band_power")
    return out


def psd_diagnostics_torch(pred: Tensor, target: Tensor, fs: float = SAMPLING_RATE_HZ, nperseg: int = 512,
                          freq_range: tuple = (0.0, MAX_FREQ_HZ)) -> dict:
    """models.py:4700-4827 (same keys)."""
    eps = 1e-12
    mp = _total(pair_moments(pred, pred))      # n, S x, S x^2 of pred (target slots unused)
    mt = _total(pair_moments(target, target))

    def std_all(s):
        n = s[0]
        return float(torch.sqrt(((s[2] - s[1] * s[1] / n) / (n - 1)).clamp_min(0)))

    pred_std, target_std = std_all(mp), std_all(mt)
    pp = welch_psd(pred, fs, nperseg).mean(dim=0, keepdim=True)       # [1, nfreq] average across signals
    pt = welch_psd(target, fs, nperseg).mean(dim=0, keepdim=True)
    k_lo, k_hi = _freq_bins(nperseg, fs, *freq_range)
    db = torch.empty((1, 2), dtype=torch.float64, device=pp.device)
    check(_lib.load().unet_psd_db_stats(_p(pp), _p(pt), 1, pp.shape[1], k_lo, k_hi, 0, _p(db), _stream()),
          "unet_psd_db_stats")
    nk = k_hi - k_lo + 1
    allb = {"_total": (freq_range[0], freq_range[1])}
    allb.update(BAND_DEFINITIONS)
    bp = band_power(torch.cat([pp, pt], 0), nperseg, fs, allb, "sum", within=freq_range).double().cpu()
    ptot, ttot = bp[0, 0], bp[1, 0]
    pf, tf = pp[0, k_lo:k_hi + 1].double(), pt[0, k_lo:k_hi + 1].double()
    pn, tn = pf / (ptot.item() + eps), tf / (ttot.item() + eps)
    pc, tc = pn - pn.mean(), tn - tn.mean()
    spectral_corr = (pc * tc).sum() / ((pc.pow(2).sum().sqrt() + eps) * (tc.pow(2).sum().sqrt() + eps))
    ratios = {}
    for i, name in enumerate(BAND_DEFINITIONS):
        a, b = _freq_bins(nperseg, fs, *BAND_DEFINITIONS[name])
        a, b = max(a, k_lo), min(b, k_hi)
        ratios[name] = float(10 * torch.log10((bp[0, i + 1] + eps) / (bp[1, i + 1] + eps))) if b >= a else 0.0
    return {
        "total_power_ratio_db": float(10 * torch.log10((ptot + eps) / (ttot + eps))),
        "psd_error_db": float(db[0, 0]) / nk,
        "psd_bias_db": float(db[0, 1]) / nk,
        "spectral_correlation": float(spectral_corr),
        "band_power_ratios_db": ratios,
        "pred_std": pred_std,
        "target_std": target_std,
        "std_ratio": pred_std / (target_std + eps),
    }


def compute_psd(signal: Tensor, fs: float = 1000.0, nperseg: int = 256) -> Dict[str, Tensor]:
    """utils/dsp_metrics.py:128-194 for every row of ``signal`` at once (the reference takes one 1-D
    signal): scipy.signal.welch defaults + np.trapz band powers, total power, dominant frequency,
    spectral entropy.  Returns tensors with a leading row axis."""
    psd = welch_psd(signal, fs, nperseg, scipy_convention=True)
    freqs = torch.arange(nperseg // 2 + 1, dtype=torch.float64) * (fs / nperseg)
    bands = dict(FREQUENCY_BANDS)
    bands["_total"] = (0.0, float(freqs[-1]))
    bp = band_power(psd, nperseg, fs, bands, "trapz")
    pn = psd.double() / (psd.double().sum(dim=1, keepdim=True) + 1e-10)
    return {
        "frequencies": freqs,
        "psd": psd,
        "band_powers": {k: bp[:, i] for i, k in enumerate(FREQUENCY_BANDS)},
        "total_power": bp[:, len(FREQUENCY_BANDS)],
        "dominant_frequency": freqs.to(psd.device)[psd.argmax(dim=1)],
        "spectral_entropy": -(pn * torch.log2(pn + 1e-10)).sum(dim=1),
    }
