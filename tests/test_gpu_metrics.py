"""GPU parity of the loss / evaluation kernels (csrc/metrics.hip through the C ABI) against the golden
vectors frozen from the reference (tests/golden/metrics.npz) and against oracle/metrics_oracle.py at
other sizes.  Tolerances: fp64-accumulated sums 1e-6 rel; fp32 FFT-based PSD 2e-4 rel / 1e-3 dB."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

pytestmark = pytest.mark.gpu

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Z = np.load(os.path.join(GD, "metrics.npz"), allow_pickle=False)


def _m():
    from unet_amd import metrics
    return metrics


def _pt():
    return torch.from_numpy(Z["pred"]).cuda(), torch.from_numpy(Z["target"]).cuda()


def test_pointwise_metrics_match_reference():
    M = _m()
    pred, target = _pt()
    pc, tc = M.crop_to_target_torch(pred), M.crop_to_target_torch(target)
    assert abs(float(M.pearson_batch(pc, tc)) - float(Z["pearson_batch"])) < 1e-5
    assert np.allclose(M.pearson_per_channel(pc, tc).cpu().numpy(), Z["pearson_per_channel"], atol=1e-5)
    assert abs(float(M.explained_variance_torch(pc, tc)) - float(Z["explained_variance"])) < 1e-5
    assert abs(float(M.normalized_rmse_torch(pc, tc)) - float(Z["nrmse"])) < 1e-5
    assert np.allclose(M.per_channel_normalize(target).cpu().numpy(), Z["per_channel_normalize"], rtol=1e-5, atol=1e-6)


def test_pooled_stats_match_reference_and_oracle():
    M = _m()
    pred, target = _pt()
    pc, tc = M.crop_to_target_torch(pred), M.crop_to_target_torch(target)
    st = M.PooledStats(pred.device)
    st.update(pc[:2], tc[:2])
    st.update(pc[2:], tc[2:])
    sums = st.sums.cpu().numpy()
    assert sums[0] == Z["pooled_sums"][0]
    assert np.allclose(sums, Z["pooled_sums"], rtol=2e-6)                   # reference sums in fp32
    assert np.allclose(sums, MO.pooled_sums(Z["pred"][..., 500:4500], Z["target"][..., 500:4500]), rtol=1e-12, atol=1e-9)
    res, ref = st.finalize(), MO.pooled_metrics(Z["pooled_sums"])
    for k in ("mse", "mae", "corr", "r2"):
        assert abs(res[k] - ref[k]) < 1e-5, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_l1_crop_loss_and_gradient(dtype):
    M = _m()
    pred, target = _pt()
    tn = M.per_channel_normalize(target)
    if dtype == torch.float32:
        p = pred.clone().requires_grad_(True)
        loss = M.l1_crop_loss(p, tn)
        loss.backward()
        assert abs(float(loss) - float(Z["l1_crop_loss"])) < 1e-6
        assert torch.equal(torch.sign(p.grad).cpu(), torch.from_numpy(np.sign(Z["l1_crop_grad"])))
        assert np.allclose(p.grad.cpu().numpy(), Z["l1_crop_grad"], rtol=1e-6, atol=1e-12)
    else:
        p = pred.to(dtype).requires_grad_(True)
        loss = M.l1_crop_loss(p, tn.to(dtype), weight=0.5, lo=100, hi=3000)
        (loss * 3).backward()
        rl, rg = MO.l1_crop(p.detach().float().cpu().numpy(), tn.to(dtype).float().cpu().numpy(), 0.5, 100, 3000)
        assert abs(float(loss) - rl) < 1e-5
        assert np.allclose(p.grad.float().cpu().numpy(), 3 * rg, rtol=1e-2, atol=1e-12)   # bf16 rounding of the constant


def test_welch_family_matches_reference():
    M = _m()
    pred, target = _pt()
    pc, tc = M.crop_to_target_torch(pred), M.crop_to_target_torch(target)
    psd = M.welch_psd(pc, 1000.0, 512).cpu().numpy()
    assert np.allclose(psd, Z["welch_psd_pred"], rtol=2e-4, atol=1e-9)
    assert abs(float(M.psd_error_db_torch(pc, tc)) - float(Z["psd_error_db"])) < 1e-3
    assert abs(float(M.psd_error_db_torch(pc, tc, normalize_psd=True)) - float(Z["psd_error_db_norm"])) < 1e-3
    assert abs(float(M.psd_diff_db_torch(pc, tc)) - float(Z["psd_diff_db"])) < 1e-3
    assert abs(float(M.psd_error_db_torch(pc, tc, fs=500.0, nperseg=256, freq_range=(1.0, 80.0))) - float(Z["psd_error_db_256_fs500"])) < 1e-3
    d = M.psd_diagnostics_torch(pc, tc)
    for k in ("total_power_ratio_db", "psd_error_db", "psd_bias_db", "spectral_correlation", "pred_std", "target_std", "std_ratio"):
        assert abs(d[k] - float(Z["diag/" + k])) < 1e-3, k
    for k, v in d["band_power_ratios_db"].items():
        assert abs(v - float(Z["diag/band/" + k])) < 1e-3, k


def test_dsp_metrics_compute_psd_matches_reference():
    M = _m()
    pred, _ = _pt()
    pc = M.crop_to_target_torch(pred)
    res = M.compute_psd(pc[0, :2], 1000.0, 256)
    for r in range(2):
        assert np.allclose(res["psd"][r].cpu().numpy(), Z[f"dsp/{r}/psd"], rtol=2e-4, atol=1e-10)
        assert abs(float(res["total_power"][r]) / float(Z[f"dsp/{r}/total_power"]) - 1) < 2e-4
        assert float(res["dominant_frequency"][r]) == float(Z[f"dsp/{r}/dominant_frequency"])
        assert abs(float(res["spectral_entropy"][r]) - float(Z[f"dsp/{r}/spectral_entropy"])) < 1e-3
        for k in M.FREQUENCY_BANDS:
            ref = float(Z[f"dsp/{r}/band/{k}"])      # a one-bin band has zero trapezoid area
            assert abs(float(res["band_powers"][k][r]) - ref) <= 2e-4 * abs(ref) + 1e-12, k


@pytest.mark.parametrize("nperseg,T", [(128, 1000), (1024, 5000), (2048, 4100)])
def test_welch_other_sizes_vs_oracle(nperseg, T):
    M = _m()
    x = torch.randn(5, 3, T, generator=torch.Generator().manual_seed(T)) * 3 + 1
    for conv in (False, True):
        got = M.welch_psd(x.cuda(), 250.0, nperseg, scipy_convention=conv).cpu().numpy()
        ref = MO.welch_psd(x.numpy(), 250.0, nperseg, scipy_convention=conv)
        assert np.allclose(got, ref, rtol=3e-4, atol=1e-7 * ref.max()), (nperseg, conv)


def test_edge_cases_and_errors():
    M = _m()
    x = torch.randn(2, 3, 700).cuda()
    # odd length, unaligned crop start, non-contiguous rows
    a, b = x[..., 3:698], (x * 0.5 + 0.1)[..., 3:698]
    s = M._total(M.pair_moments(a, b)).cpu().numpy()
    assert np.allclose(s, MO.pooled_sums(a.cpu().numpy(), b.cpu().numpy()), rtol=1e-10, atol=1e-9)
    with pytest.raises(RuntimeError, match="shorter than nperseg"):
        M.welch_psd(x, 1000.0, 1024)
    with pytest.raises(RuntimeError, match="power of two"):
        M.welch_psd(x, 1000.0, 300)
    with pytest.raises(ValueError, match="shape mismatch"):
        M.pair_moments(x, x[:1])
    with pytest.raises(RuntimeError, match="GPU only"):
        M.pair_moments(x.cpu(), x.cpu())
    z = M.window_zscore(x).cpu().numpy()
    assert np.allclose(z, MO.window_zscore(x.cpu().numpy()), rtol=1e-5, atol=1e-6)
