"""The metric oracle (oracle/metrics_oracle.py) against tests/golden/metrics.npz, frozen from the
reference's own functions by oracle/make_golden_host.py.  CPU only."""
import os

import numpy as np

from oracle import metrics_oracle as MO

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz"), allow_pickle=False)
PRED, TARGET = Z["pred"], Z["target"]
PC, TC = MO.crop(PRED), MO.crop(TARGET)


def close(a, b, rtol):
    return np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=rtol * 1e-2)


def test_pointwise_metrics():
    assert close(MO.pearson_batch(PC, TC), Z["pearson_batch"], 1e-5)
    assert close(MO.pearson_per_channel(PC, TC), Z["pearson_per_channel"], 1e-5)
    assert close(MO.explained_variance(PC, TC), Z["explained_variance"], 1e-5)
    assert close(MO.nrmse(PC, TC), Z["nrmse"], 1e-5)
    assert close(MO.per_channel_normalize(TARGET), Z["per_channel_normalize"], 1e-5)


def test_pooled_sums_and_loss():
    s = MO.pooled_sums(PC, TC)
    assert np.allclose(s, Z["pooled_sums"], rtol=2e-6)   # the reference sums in fp32
    m = MO.pooled_metrics(s)
    assert close(m["r2"], MO.pooled_metrics(Z["pooled_sums"])["r2"], 1e-5)
    tn = MO.per_channel_normalize(TARGET)
    loss, g = MO.l1_crop(PRED, tn)
    assert close(loss, Z["l1_crop_loss"], 1e-6)
    assert np.array_equal(np.sign(g), np.sign(Z["l1_crop_grad"]))
    assert np.allclose(g, Z["l1_crop_grad"], rtol=1e-6, atol=1e-12)


def test_welch_family():
    assert close(MO.welch_psd(PC, 1000.0, 512), Z["welch_psd_pred"], 2e-4)
    assert abs(MO.psd_error_db(PC, TC) - float(Z["psd_error_db"])) < 1e-4
    assert abs(MO.psd_error_db(PC, TC, normalize=True) - float(Z["psd_error_db_norm"])) < 1e-4
    assert abs(MO.psd_error_db(PC, TC, signed=True) - float(Z["psd_diff_db"])) < 1e-4
    assert abs(MO.psd_error_db(PC, TC, fs=500.0, nperseg=256, freq_range=(1.0, 80.0)) - float(Z["psd_error_db_256_fs500"])) < 1e-4
    d = MO.psd_diagnostics(PC, TC)
    for k in ("total_power_ratio_db", "psd_error_db", "psd_bias_db", "spectral_correlation", "pred_std", "target_std", "std_ratio"):
        assert abs(d[k] - float(Z["diag/" + k])) < 1e-4, k
    for k, v in d["band_power_ratios_db"].items():
        assert abs(v - float(Z["diag/band/" + k])) < 1e-4, k


def test_dsp_metrics_compute_psd():
    for r in range(2):
        res = MO.compute_psd(PC[0, r], 1000.0, 256)
        assert close(res["psd"], Z[f"dsp/{r}/psd"], 5e-6)  # scipy keeps the float32 input precision
        assert close(res["total_power"], Z[f"dsp/{r}/total_power"], 5e-6)
        assert res["dominant_frequency"] == float(Z[f"dsp/{r}/dominant_frequency"])
        assert close(res["spectral_entropy"], Z[f"dsp/{r}/spectral_entropy"], 5e-6)
        for k, v in res["band_powers"].items():
            assert close(v, Z[f"dsp/{r}/band/{k}"], 5e-6), k


def test_overlap_average_restates_reference():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "overlap_add.npz"))
    sig, rec = z["signal"], z["recon"]
    w, s = 4096, 1024
    nw = (sig.shape[1] - w) // s + 1
    ramp = np.linspace(-1, 1, w, dtype=np.float32)
    preds = np.stack([sig[:, i * s:i * s + w] * 0.5 + ramp * np.roll(sig[:, i * s:i * s + w], 1, axis=0) for i in range(nw)])
    assert np.allclose(MO.overlap_average(preds.astype(np.float32), sig.shape[1], s), rec, rtol=1e-6, atol=1e-6)
