"""Data-path oracle (oracle/data_oracle.py) against fixtures frozen from the reference's NoiseAugmentor and
ECoGDataset (tests/golden/data_path.npz) -- CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import data_oracle as DO
from oracle.make_golden_data import AUG_CASES
from unet_amd.windows import ecog_windows

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_path.npz"))


@pytest.mark.parametrize("name", list(AUG_CASES))
def test_noise_augmentor_oracle_reproduces_reference(name):
    x = torch.from_numpy(Z["aug/x"])
    want = Z[f"aug/{name}/y"]
    torch.manual_seed(123)
    for k in range(want.shape[0]):
        y = DO.forward(x, **AUG_CASES[name])
        assert np.allclose(y.numpy(), want[k], rtol=1e-6, atol=1e-6), (name, k)
    if name == "coin":       # the coin came up both ways in the fixture: some calls return x untouched
        same = [np.array_equal(want[k], Z["aug/x"]) for k in range(want.shape[0])]
        assert any(same) and not all(same)
    assert DO.forward(x, training=False, **AUG_CASES[name]) is x


def test_ecog_items_and_window_table():
    subjects = [dict(source=Z[f"ecog/subj{i}/source"], target=Z[f"ecog/subj{i}/target"]) for i in range(3)]
    windows = ecog_windows([s["source"].shape[1] for s in subjects], 5000, 2500)
    assert np.array_equal(np.array(windows), Z["ecog/windows"])
    for z in (0, 1):
        for j, idx in enumerate(Z["ecog/idx"]):
            sub = [dict(source=s["source"][:4], target=s["target"][:3]) for s in subjects]
            src, tgt = DO.ecog_item(sub, windows, int(idx), 5000, bool(z))
            assert np.allclose(src, Z[f"ecog/z{z}/source"][j], rtol=1e-6, atol=1e-6)
            assert np.allclose(tgt, Z[f"ecog/z{z}/target"][j], rtol=1e-6, atol=1e-6)
