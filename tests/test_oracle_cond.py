"""The conditioning-encoder oracle (oracle/cond_oracle.py) against fixtures frozen from the reference's own
SpectroTemporalEncoder (tests/golden/cond_spectro_temporal.npz), and the product module's constructor parity
(state-dict keys, same seed -> same initial weights) -- CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cond_oracle as CO

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cond_spectro_temporal.npz"))


def _sd(requires_grad=False):
    sd = {k[3:]: torch.from_numpy(Z[k]).clone() for k in Z.files if k.startswith("sd/")}
    if requires_grad:
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
    return sd


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("T", [1024, 1250])
def test_oracle_matches_reference(T):
    sd = _sd(True)
    x, odor, gy = (torch.from_numpy(Z[f"T{T}/{k}"]) for k in ("x", "odor", "gy"))
    assert torch.allclose(CO.freq_masks(T), torch.from_numpy(Z[f"T{T}/masks"]), rtol=1e-6, atol=1e-12)
    assert rel(CO.band_log_power(x, CO.freq_masks(T)), Z[f"T{T}/band_log"]) < 1e-6
    y = CO.forward(sd, x, odor)
    assert rel(y.detach(), Z[f"T{T}/y"]) < 1e-6
    assert rel(CO.forward(_sd(), x), Z[f"T{T}/y_no_odor"]) < 1e-6
    (y * gy).sum().backward()
    for k, v in sd.items():
        if v.requires_grad:
            assert rel(v.grad, Z[f"T{T}/grad/{k}"]) < 1e-5, k


def test_module_constructor_parity():
    """Same seed -> same initial parameters as the reference (creation order), same state-dict keys and shapes."""
    from unet_amd.cond import SpectroTemporalEncoder
    torch.manual_seed(11)
    enc = SpectroTemporalEncoder(in_channels=32, emb_dim=128)
    sd = enc.state_dict()
    want = [k[3:] for k in Z.files if k.startswith("sd/")]
    assert list(sd.keys()) == want
    for k in want:
        assert tuple(sd[k].shape) == Z["sd/" + k].shape, k
    sums = np.array([float(p.double().sum()) for p in enc.parameters()])
    abss = np.array([float(p.double().abs().sum()) for p in enc.parameters()])
    bn = [i for i, (k, _) in enumerate(enc.named_parameters()) if k.startswith("temporal_fusion.1")]
    keep = [i for i in range(len(sums)) if i not in bn]      # the fixture re-draws the BN affine after construction
    assert np.allclose(sums[keep], Z["init_checksum"][keep], rtol=1e-12, atol=1e-12)
    assert np.allclose(abss[keep], Z["init_abs_checksum"][keep], rtol=1e-12, atol=1e-12)
    assert torch.allclose(enc._build_freq_masks(1250, "cpu"), torch.from_numpy(Z["T1250/masks"]), rtol=1e-6, atol=1e-12)
    with pytest.raises(RuntimeError, match="GPU only"):
        enc(torch.zeros(1, 32, 256))
