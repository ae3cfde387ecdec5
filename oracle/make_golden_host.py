"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/metrics.npz, host_logic.json, overlap_add.npz, checkpoint_layout.json.

Runs ONLY in the build container (reference mounted at /root/reference): imports the reference's own
``models.py`` / ``utils/dsp_metrics.py`` / ``phase_five/sliding_window.py`` / ``phase_four/data_splitter.py``
/ ``data.py`` / ``train.py`` and freezes their OUTPUTS on seeded inputs.  Fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_host.py
"""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def metrics(ref):
    g = torch.Generator().manual_seed(77)
    B, C, T = 3, 4, 4096
    tt = torch.arange(T) / 1000.0
    base = torch.sin(2 * np.pi * 6.0 * tt)[None, None] * torch.rand(B, C, 1, generator=g) \
        + 0.5 * torch.sin(2 * np.pi * 40.0 * tt + 1.0)[None, None]
    target = base + 0.7 * torch.randn(B, C, T, generator=g) + 0.3
    pred = 0.8 * base + 0.5 * torch.randn(B, C, T, generator=g)
    out = {"pred": pred.numpy(), "target": target.numpy()}
    pc, tc = pred[..., 500:4500], target[..., 500:4500]
    out["pearson_batch"] = ref.pearson_batch(pc, tc).numpy()
    out["pearson_per_channel"] = ref.pearson_per_channel(pc, tc).numpy()
    out["explained_variance"] = ref.explained_variance_torch(pc, tc).numpy()
    out["nrmse"] = ref.normalized_rmse_torch(pc, tc).numpy()
    out["psd_error_db"] = ref.psd_error_db_torch(pc, tc).numpy()
    out["psd_error_db_norm"] = ref.psd_error_db_torch(pc, tc, normalize_psd=True).numpy()
    out["psd_diff_db"] = ref.psd_diff_db_torch(pc, tc).numpy()
    out["psd_error_db_256_fs500"] = ref.psd_error_db_torch(pc, tc, fs=500.0, nperseg=256, freq_range=(1.0, 80.0)).numpy()
    w = torch.hann_window(512)
    out["welch_psd_pred"] = ref._compute_welch_psd_torch(pc.reshape(B * C, -1), 1000.0, 512, w,
                                                        2.0 / (1000.0 * (w ** 2).sum())).numpy()
    d = ref.psd_diagnostics_torch(pc, tc)
    bands = d.pop("band_power_ratios_db")
    for k, v in d.items():
        out["diag/" + k] = np.float64(v)
    for k, v in bands.items():
        out["diag/band/" + k] = np.float64(v)
    # train.py helpers
    import train as ref_train
    out["per_channel_normalize"] = ref_train.per_channel_normalize(target).numpy()
    import torch.nn.functional as F
    p2 = pred.clone().requires_grad_(True)
    tn = ref_train.per_channel_normalize(target)
    import data as ref_data
    loss = 1.0 * F.l1_loss(ref_data.crop_to_target_torch(p2), ref_data.crop_to_target_torch(tn))
    loss.backward()
    out["l1_crop_loss"] = loss.detach().numpy()
    out["l1_crop_grad"] = p2.grad.numpy()
    # pooled sums exactly as train.py:941-952 computes them (fp32 tensor sums -> python floats)
    pf, tf = pc.flatten(), tc.flatten()
    out["pooled_sums"] = np.array([pf.numel(), tf.sum().item(), (tf ** 2).sum().item(), pf.sum().item(),
                                   (pf ** 2).sum().item(), (pf * tf).sum().item(), ((pf - tf) ** 2).sum().item(),
                                   (pf - tf).abs().sum().item()], dtype=np.float64)
    # utils/dsp_metrics.compute_psd on two rows
    from utils import dsp_metrics as dm
    for r in range(2):
        res = dm.compute_psd(pc[0, r].numpy(), fs=1000.0, nperseg=256)
        out[f"dsp/{r}/psd"] = res.psd.astype(np.float64)
        out[f"dsp/{r}/total_power"] = np.float64(res.total_power)
        out[f"dsp/{r}/dominant_frequency"] = np.float64(res.dominant_frequency)
        out[f"dsp/{r}/spectral_entropy"] = np.float64(res.spectral_entropy)
        for k, v in res.band_powers.items():
            out[f"dsp/{r}/band/{k}"] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics.npz:", len(out), "arrays")


def host_logic():
    from phase_five import sliding_window as sw
    from phase_four import data_splitter as ds
    import data as ref_data
    out = {}
    # window counts / starts
    cases = [(300000, 5000, 2500), (5000, 5000, 2500), (4999, 5000, 2500), (12345, 4096, 1024), (10001, 5000, 5000),
             (7500, 5000, 2500), (7499, 5000, 2500), (100, 5000, 2500)]
    wc = []
    for n, w, s in cases:
        src = np.zeros((n, 2), dtype=np.float32)
        d5 = sw.SlidingWindowDataset(src, src, w, s)
        cont = ref_data.ContinuousLFPDataset(np.zeros((2, n), np.float32), np.zeros((2, n), np.float32), w, s)
        # process_continuous start list
        starts = []
        st = 0
        inf = sw.ContinuousInference.__new__(sw.ContinuousInference)
        inf.window_size, inf.stride = w, s
        inf.process_window = lambda window: window
        for start, _ in inf.process_continuous(np.zeros((2, n), np.float32)):
            starts.append(int(start))
        wc.append(dict(n=n, w=w, s=s, phase5_len=len(d5), data_len=len(cont), coverage=float(d5.coverage),
                       infer_starts=starts))
    out["window_cases"] = wc
    # multi-session mapping + ECoG windows
    sess = [dict(ob=np.zeros((2, n), np.float32), pcx=np.zeros((2, n), np.float32), session=f"s{i}")
            for i, n in enumerate([12000, 4000, 20001])]
    ms = ref_data.MultiSessionContinuousDataset(sess, 5000, 2500)
    out["multisession_mapping"] = [[int(a), int(b)] for a, b in ms.window_mapping]
    subj = [dict(source=np.zeros((3, n), np.float32), target=np.zeros((2, n), np.float32), n_samples=n,
                 n_source_channels=3, n_target_channels=2, subject_id=f"p{i}") for i, n in enumerate([11000, 30000])]
    ec = ref_data.ECoGDataset(subj, 5000, 5000, verbose=False)
    out["ecog_windows"] = [[int(a), int(b)] for a, b in ec.windows]
    # LOSO temporal split point (data.py:3275) for a few lengths
    out["loso_split_points"] = [[n, int(n * (1.0 - 0.3))] for n in (300000, 123457, 10, 99999)]
    # overlap-average reconstruction with a deterministic stand-in model
    class Toy(torch.nn.Module):
        def forward(self, x):
            ramp = torch.linspace(-1, 1, x.shape[-1])
            return x * 0.5 + ramp * x.roll(1, dims=1)
    np.random.seed(3)
    sig = np.random.randn(3, 14000).astype(np.float32)
    rec = sw.continuous_inference(Toy(), sig, 4096, 1024, device="cpu")
    np.savez_compressed(os.path.join(OUT, "overlap_add.npz"), signal=sig, recon=rec)
    # synthetic signal generator (np.random.seed(cv_seed + fold), phase_five/runner.py:240-242)
    np.random.seed(42 + 1)
    s_src, s_tgt = sw.generate_continuous_signal(3.0, 1000.0, 4)
    out["synthetic_signal"] = dict(src_sum=float(s_src.astype(np.float64).sum()), tgt_sum=float(s_tgt.astype(np.float64).sum()),
                                   src_abs=float(np.abs(s_src.astype(np.float64)).sum()),
                                   tgt_abs=float(np.abs(s_tgt.astype(np.float64)).sum()), shape=list(s_src.shape))
    # session splitters
    rs = np.random.RandomState(0)
    sid = rs.randint(0, 7, size=200)
    sp = ds.SessionSplitter(sid, val_ratio=0.1, test_ratio=0.2, random_state=42)
    a = sp.split(ds.SplitMode.INTRA_SESSION)
    b = sp.split(ds.SplitMode.INTER_SESSION)
    c = sp.leave_one_session_out(3)
    conv = lambda v: [int(i) for i in v]
    out["session_splitter"] = dict(
        session_ids=conv(sid),
        intra=dict(train=conv(a.train_indices), val=conv(a.val_indices), test=conv(a.test_indices)),
        inter=dict(train=conv(b.train_indices), val=conv(b.val_indices), test=conv(b.test_indices),
                   train_sessions=conv(b.train_sessions), test_sessions=conv(b.test_sessions)),
        loso3=dict(train=conv(c.train_indices), val=conv(c.val_indices), test=conv(c.test_indices),
                   train_sessions=conv(c.train_sessions)))
    # data.load_or_create_session_splits: random and explicit (LOSO: --test-sessions <held-out>, force recreate)
    odors = rs.randint(0, 7, size=200)
    idx_to_session = {i: f"sess{i:02d}" for i in range(7)}
    tr, va, te, info = ref_data.load_or_create_session_splits(sid, odors, n_test_sessions=1, n_val_sessions=2, seed=42,
                                                              force_recreate=True, idx_to_session=idx_to_session)
    out["session_split_random"] = dict(train=conv(tr), val=conv(va), test=conv(te))
    tr, va, te, info = ref_data.load_or_create_session_splits(sid, odors, seed=45, force_recreate=True,
                                                              idx_to_session=idx_to_session, test_sessions=["sess04"],
                                                              separate_val_sessions=False)
    out["session_split_explicit"] = dict(train=conv(tr), val=conv(va), test=conv(te))
    tr, va, te, info = ref_data.load_or_create_session_splits(sid, odors, n_val_sessions=2, seed=7, force_recreate=True,
                                                              idx_to_session=idx_to_session, no_test_set=True,
                                                              separate_val_sessions=True)
    out["session_split_noval"] = dict(train=conv(tr), test=conv(te), val=conv(va),
                                      val_per_session={k: conv(v) for k, v in info["val_idx_per_session"].items()})
    json.dump(out, open(os.path.join(OUT, "host_logic.json"), "w"))
    print("host_logic.json written")


def checkpoint_layout():
    """Layout of the file the reference's save_checkpoint writes (train.py:652-701) and load_checkpoint's
    behaviour on a cond_mode mismatch (train.py:523-534), on a tiny module."""
    import tempfile
    from pathlib import Path
    import train as ref_train
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    model(torch.randn(5, 4)).sum().backward()
    opt.step()
    config = {"cond_mode": "cross_attn_gated", "lr": 1e-3, "checkpoint_prefix": "fold3", "base_channels": 256}
    data = {"split_info": {"test_sessions": ["s1"], "n_val": 2}, "val_idx": np.arange(3), "test_idx": np.array([7, 8]),
            "train_idx": [0, 1, 2]}
    d = Path(tempfile.mkdtemp(prefix="unet_ckpt_"))
    ref_train.save_checkpoint(model, opt, 7, d / "a.pt", config=config, data=data)
    ck = torch.load(d / "a.pt", map_location="cpu", weights_only=True)        # our own freshly written file
    out = {"keys": sorted(ck.keys()), "epoch": ck["epoch"], "cond_mode": ck["cond_mode"], "config": ck["config"],
           "split_info": ck["split_info"], "model_keys": list(ck["model"].keys()),
           "optimizer_keys": sorted(ck["optimizer"].keys()),
           "optimizer_state_keys": sorted(ck["optimizer"]["state"][0].keys())}
    ref_train.save_checkpoint(model, opt, 1, d / "b.pt")
    ck = torch.load(d / "b.pt", map_location="cpu", weights_only=True)
    out["keys_minimal"] = sorted(ck.keys())
    out["cond_mode_minimal"] = ck["cond_mode"]
    m2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    out["load_returns"] = ref_train.load_checkpoint(d / "a.pt", m2, expected_cond_mode="cross_attn_gated")
    out["loaded_equal"] = bool(all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values())))
    try:
        ref_train.load_checkpoint(d / "a.pt", m2, expected_cond_mode="film")
        out["mismatch_error"] = None
    except ValueError as e:
        out["mismatch_error"] = str(e)
    try:
        ref_train.load_checkpoint(d / "missing.pt", m2)
    except FileNotFoundError as e:
        out["missing_error"] = str(e).replace(str(d), "<dir>")
    json.dump(out, open(os.path.join(OUT, "checkpoint_layout.json"), "w"), indent=1)
    print("checkpoint_layout.json written")


def main():
    import tempfile
    # the reference's split helper saves .npy files under UNET_DATA_DIR (data.py:88, 803-808): point it at a scratch dir
    os.environ["UNET_DATA_DIR"] = tempfile.mkdtemp(prefix="unet_golden_")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    import models as ref
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if "--only-checkpoint" in sys.argv:
        checkpoint_layout()
        return
    metrics(ref)
    host_logic()
    checkpoint_layout()


if __name__ == "__main__":
    main()
