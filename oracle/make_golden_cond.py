"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/cond_spectro_temporal.npz.

Runs ONLY in the build container (reference mounted at /root/reference): imports the reference's ``models.py``
and freezes OUTPUTS of its SpectroTemporalEncoder on seeded inputs (eval mode, as train.py:1410-1411 runs it):
state dict, inputs, band log power, output, gradients of every parameter under a fixed linear loss, for two
window lengths (power of two and not).  Fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cond.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    sys.path.insert(0, REF)
    import models as ref
    torch.set_num_threads(4)
    torch.manual_seed(11)
    enc = ref.SpectroTemporalEncoder(in_channels=32, emb_dim=128)
    with torch.no_grad():   # non-trivial running statistics and affine terms, as after some training
        enc.temporal_fusion[1].running_mean.normal_(0, 0.2)
        enc.temporal_fusion[1].running_var.uniform_(0.5, 1.5)
        enc.temporal_fusion[1].weight.uniform_(0.5, 1.5)
        enc.temporal_fusion[1].bias.normal_(0, 0.1)
    enc.eval()
    out = {"init_checksum": np.array([float(p.double().sum()) for p in enc.parameters()]),
           "init_abs_checksum": np.array([float(p.double().abs().sum()) for p in enc.parameters()])}
    for k, v in enc.state_dict().items():
        out["sd/" + k] = v.numpy()
    g = torch.Generator().manual_seed(5)
    for T in (1024, 1250):
        tt = torch.arange(T) / 1000.0
        x = torch.randn(3, 32, T, generator=g) + 2.0 * torch.sin(2 * np.pi * 12.0 * tt)[None, None] \
            * torch.rand(3, 32, 1, generator=g)
        odor = torch.randn(3, 128, generator=g) * 0.3
        gy = torch.randn(3, 128, generator=g)
        enc.zero_grad()
        y = enc(x, odor)
        (y * gy).sum().backward()
        masks = enc._build_freq_masks(T, x.device)
        power = torch.fft.rfft(x.float(), dim=-1).abs().square()
        out[f"T{T}/x"], out[f"T{T}/odor"], out[f"T{T}/gy"], out[f"T{T}/y"] = x.numpy(), odor.numpy(), gy.numpy(), y.detach().numpy()
        out[f"T{T}/masks"] = masks.numpy()
        out[f"T{T}/band_log"] = torch.log(torch.einsum("bcf,kf->bck", power, masks) + 1e-10).numpy()
        out[f"T{T}/y_no_odor"] = enc(x).detach().numpy()
        for k, p in enc.named_parameters():
            out[f"T{T}/grad/{k}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "cond_spectro_temporal.npz"), **out)
    print("cond_spectro_temporal.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
