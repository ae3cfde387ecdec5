"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/data_path.npz.

Runs ONLY in the build container (reference mounted at /root/reference): runs the reference's NoiseAugmentor
(models.py) on the CPU under fixed seeds and its ECoGDataset (data.py) on synthetic recordings, and freezes the
OUTPUTS.  Fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_data.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

AUG_CASES = {
    "loso": dict(gaussian_std=0.1, pink_noise=True, pink_noise_std=0.05, channel_dropout=0.05, temporal_dropout=0.02, prob=1.0),
    "line": dict(gaussian_std=0.0, pink_noise=False, line_noise_hz=50.0, prob=1.0),
    "all": dict(gaussian_std=0.2, pink_noise=True, pink_noise_std=0.1, channel_dropout=0.3, temporal_dropout=0.1,
                line_noise_hz=60.0, sample_rate=500.0, prob=1.0),
    "coin": dict(gaussian_std=0.1, prob=0.5),
}


def main():
    os.environ["UNET_DATA_DIR"] = tempfile.mkdtemp(prefix="unet_golden_")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    import models as ref
    import data as ref_data
    torch.set_num_threads(2)
    out = {}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 6, 1001, generator=g) * 2.5 + 0.4
    out["aug/x"] = x.numpy()
    for name, kw in AUG_CASES.items():
        aug = ref.NoiseAugmentor(**kw)
        aug.train()
        torch.manual_seed(123)
        ys = [aug(x).numpy().copy() for _ in range(4 if name == "coin" else 1)]
        out[f"aug/{name}/y"] = np.stack(ys)
        aug.eval()
        assert aug(x) is x
    rng = np.random.default_rng(4)
    subjects = []
    for i, n in enumerate([11000, 7300, 30000]):
        subjects.append(dict(source=(rng.standard_normal((5, n)) * (i + 1) + i).astype(np.float32),
                             target=(rng.standard_normal((4, n)) * 0.5 - i).astype(np.float32), n_samples=n,
                             n_source_channels=5, n_target_channels=4, subject_id=f"p{i}"))
    for i, s in enumerate(subjects):
        out[f"ecog/subj{i}/source"], out[f"ecog/subj{i}/target"] = s["source"], s["target"]
    for z in (False, True):
        ds = ref_data.ECoGDataset(subjects, 5000, 2500, zscore_per_window=z, verbose=False, n_source_channels=4,
                                  n_target_channels=3)
        out["ecog/windows"] = np.array(ds.windows, dtype=np.int64)
        idx = [0, 1, 2, 3, len(ds) - 1]
        out["ecog/idx"] = np.array(idx)
        out[f"ecog/z{int(z)}/source"] = np.stack([ds[i]["source"].numpy() for i in idx])
        out[f"ecog/z{int(z)}/target"] = np.stack([ds[i]["target"].numpy() for i in idx])
    np.savez_compressed(os.path.join(OUT, "data_path.npz"), **out)
    print("data_path.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
