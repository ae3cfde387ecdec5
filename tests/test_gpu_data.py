"""Data path on the HIP kernels (unet_gather_windows, unet_noise_augment) against the reference's outputs
(tests/golden/data_path.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_path.npz"))


def _draws(x, kw, gen_device):
    """The reference's draw order (models.py:4906-4942) on ``gen_device``; returns the oracle-side arguments."""
    from oracle import data_oracle as DO
    B, C, T = x.shape
    coin = torch.rand(1).item()
    assert coin <= kw.get("prob", 0.5)
    d = {}
    d["gauss"] = torch.randn_like(x.to(gen_device)) if kw.get("gaussian_std", 0.1) > 0 else None
    d["pink"] = None
    if kw.get("pink_noise", False) and kw.get("pink_noise_std", 0.05) > 0:
        white = torch.randn(B, C, T // 2 + 1, dtype=torch.complex64, device=gen_device)
        d["pink"] = DO.pink_from_white(white.cpu(), T, kw.get("sample_rate", 1000.0))
    d["ch_u"] = torch.rand(B, C, 1, device=gen_device) if kw.get("channel_dropout", 0) > 0 else None
    d["t_u"] = torch.rand(B, 1, T, device=gen_device) if kw.get("temporal_dropout", 0) > 0 else None
    d["amp_u"] = d["phase_u"] = None
    if kw.get("line_noise_hz", 0) > 0:
        d["amp_u"] = torch.rand(B, 1, 1, device=gen_device)
        d["phase_u"] = torch.rand(B, 1, 1, device=gen_device)
    return {k: (v.cpu() if v is not None else None) for k, v in d.items()}


@pytest.mark.parametrize("name", ["loso", "line", "all"])
def test_noise_kernel_reproduces_reference_given_the_draws(name):
    """Same CPU draws as the fixture (seed 123) fed to the HIP kernel -> the reference's output."""
    from oracle.make_golden_data import AUG_CASES
    from unet_amd.augment import noise_augment_apply
    kw = AUG_CASES[name]
    x = torch.from_numpy(Z["aug/x"])
    torch.manual_seed(123)
    d = _draws(x, kw, "cpu")
    std = x.std().clamp(min=1e-6)
    scales = torch.stack([kw.get("gaussian_std", 0.1) * std, kw.get("pink_noise_std", 0.05) * std])
    cu = lambda v: None if v is None else v.cuda()
    amp = d["amp_u"] * 0.02 * std if d["amp_u"] is not None else None
    ph = d["phase_u"] * 2 * np.pi if d["phase_u"] is not None else None
    y = noise_augment_apply(x.cuda(), cu(d["gauss"]), cu(d["pink"]), scales.cuda(), cu(d["ch_u"]), kw.get("channel_dropout", 0.0),
                            cu(d["t_u"]), kw.get("temporal_dropout", 0.0), cu(amp), cu(ph), kw.get("line_noise_hz", 0.0),
                            kw.get("sample_rate", 1000.0))
    # the hum's phase w*t reaches ~750 rad at fp32 (ATen's vectorised linspace rounds t differently again), so its
    # sine agrees to ~5e-5 rad: 3e-6 absolute at amplitude 0.05; everything else agrees to fp32 rounding
    atol = 5e-6 if kw.get("line_noise_hz", 0) > 0 else 2e-6
    assert np.allclose(y.cpu().numpy(), Z[f"aug/{name}/y"][0], rtol=1e-5, atol=atol)


@pytest.mark.parametrize("name", ["loso", "all"])
def test_module_draws_in_the_reference_order(name):
    """Module on the GPU == oracle arithmetic on draws made on the GPU in the reference's order under the same seeds."""
    from oracle import data_oracle as DO
    from oracle.make_golden_data import AUG_CASES
    from unet_amd.augment import NoiseAugmentor
    kw = AUG_CASES[name]
    x = torch.from_numpy(Z["aug/x"])
    aug = NoiseAugmentor(**kw).train()
    torch.manual_seed(7)
    y = aug(x.cuda())
    torch.manual_seed(7)
    d = _draws(x, kw, "cuda")
    want = DO.apply(x, d["gauss"], d["pink"], d["ch_u"], d["t_u"], d["amp_u"], d["phase_u"],
                    **{k: v for k, v in kw.items() if k not in ("prob", "pink_noise")})
    assert np.allclose(y.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    torch.manual_seed(7)
    assert torch.equal(aug(x.cuda()), y)                       # deterministic under a seed
    assert aug.eval()(x.cuda()).data_ptr() != 0 and aug(x) is x   # eval: identity, no copy (even for a CPU tensor)


def test_module_statistics_and_bf16():
    from unet_amd.augment import NoiseAugmentor
    torch.manual_seed(0)
    x = (torch.randn(8, 16, 4096) * 3.0).cuda()
    aug = NoiseAugmentor(gaussian_std=0.1, pink_noise=True, pink_noise_std=0.05, prob=1.0).train()
    y = aug(x)
    noise = (y - x).float()
    std = float(x.std())
    assert abs(float(noise.std()) - std * (0.1 ** 2 + 0.05 ** 2) ** 0.5) < 0.01 * std
    drop = NoiseAugmentor(gaussian_std=0.0, channel_dropout=0.25, temporal_dropout=0.1, prob=1.0).train()
    z = drop(x)
    dead_rows = (z.abs().sum(-1) == 0).float().mean().item()
    dead_cols = (z.abs().sum(1) == 0).float().mean().item()
    assert 0.1 < dead_rows < 0.45 and 0.07 < dead_cols < 0.13
    kept = z != 0
    assert torch.equal(z[kept], x[kept])
    yb = aug(x.bfloat16())
    assert yb.dtype == torch.bfloat16 and abs(float((yb.float() - x).std()) - float(noise.std())) < 0.02 * std
    never = NoiseAugmentor(prob=0.0).train()
    assert never(x) is x


def test_device_recordings_match_reference_items():
    from unet_amd.windows import DeviceRecordings
    subjects = [dict(source=Z[f"ecog/subj{i}/source"], target=Z[f"ecog/subj{i}/target"], n_samples=Z[f"ecog/subj{i}/source"].shape[1],
                     n_source_channels=5, n_target_channels=4) for i in range(3)]
    idx = [int(i) for i in Z["ecog/idx"]]
    for z in (0, 1):
        rec = DeviceRecordings(subjects, 5000, 2500, zscore_per_window=bool(z), n_source_channels=4, n_target_channels=3)
        assert np.array_equal(np.array(rec.windows), Z["ecog/windows"]) and len(rec) == len(Z["ecog/windows"])
        b = rec.batch(idx)
        assert b["source"].shape == (len(idx), 4, 5000) and b["target"].shape == (len(idx), 3, 5000)
        tol = dict(rtol=0, atol=0) if not z else dict(rtol=1e-5, atol=2e-6)
        assert np.allclose(b["source"].cpu().numpy(), Z[f"ecog/z{z}/source"], **tol)
        assert np.allclose(b["target"].cpu().numpy(), Z[f"ecog/z{z}/target"], **tol)
        assert b["subject_idx"].tolist() == [int(Z["ecog/windows"][i][0]) for i in idx]
        assert b["start_sample"].tolist() == [int(Z["ecog/windows"][i][1]) for i in idx]
        bb = rec.batch(idx[:2], out_dtype=torch.bfloat16)["source"]
        assert bb.dtype == torch.bfloat16 and np.allclose(bb.float().cpu().numpy(), Z[f"ecog/z{z}/source"][:2], rtol=1e-2, atol=2e-2)
        assert rec.batch([])["source"].shape == (0, 4, 5000)
    with pytest.raises(IndexError):
        rec.batch([len(rec)])


def test_trainer_applies_augmentation_to_the_input_only():
    import unet_amd
    from unet_amd.trainer import Trainer
    cfg = dict(in_channels=8, out_channels=8, base=16, emb_dim=32, conv_type="modern", attention_type="none", activation="gelu")
    g = torch.Generator().manual_seed(0)
    ob, tgt = torch.randn(2, 8, 4608, generator=g).cuda(), torch.randn(2, 8, 4608, generator=g).cuda()
    ids = torch.tensor([0, 1]).cuda()
    losses = {}
    for name, aug in (("none", None), ("never", unet_amd.NoiseAugmentor(prob=0.0)),
                      ("always", unet_amd.NoiseAugmentor(gaussian_std=0.5, pink_noise=True, prob=1.0))):
        torch.manual_seed(1)
        model = unet_amd.CondUNet1D(**cfg).cuda()
        with torch.no_grad():
            model.outc.weight.normal_(0, 0.05)
        tr = Trainer(model, lr=1e-3, compute_dtype=torch.float32, noise_augmentor=aug)
        losses[name] = float(tr.step(ob, tgt, ids).detach())
        res = tr.evaluate([(ob, tgt, ids)])          # evaluation never augments (train.py:776-1291)
        assert np.isfinite(res["loss"])
    assert losses["none"] == losses["never"]
    assert abs(losses["always"] - losses["none"]) > 1e-6
