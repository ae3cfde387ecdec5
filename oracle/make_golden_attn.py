"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/attn_cross_freq_v2.npz and attn_cross_freq_v1.npz.

Runs ONLY in the build container (reference mounted at /root/reference): imports the reference's ``models.py``
and freezes OUTPUTS of its CrossFrequencyCouplingAttentionV2 (models.py:227-474) on seeded inputs at head sizes
the model fixtures do not reach (head_dim 32 and 64, lengths that are not multiples of the kernel's tiles):
state dict, input, output, gradients of the input and of every parameter under a fixed linear loss.
Fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_attn.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CASES = {"c64_h2_l160": (64, 2, 160, 2), "c128_h2_l200": (128, 2, 200, 1)}   # channels, heads, length, batch


def main():
    sys.path.insert(0, REF)
    import models as ref
    torch.set_num_threads(4)
    out = {}
    for name, (c, nh, l, n) in CASES.items():
        torch.manual_seed(23)
        m = ref.CrossFrequencyCouplingAttentionV2(c, n_heads=nh)
        with torch.no_grad():   # as after some training: an open gate and uneven coupling
            m.gate.fill_(0.7)
            m.coupling_weights.normal_(0, 1.0)
            for conv in (m.proj_q, m.proj_k):
                conv.weight.mul_(3.0)        # peaked softmaxes (fresh-init scores are nearly uniform)
        m.train()
        g = torch.Generator().manual_seed(3)
        tt = torch.arange(l) / 1000.0
        x = torch.randn(n, c, l, generator=g) + 1.5 * torch.sin(2 * np.pi * 10.0 * tt)[None, None] * torch.rand(n, c, 1, generator=g)
        x.requires_grad_(True)
        gy = torch.randn(n, c, l, generator=g)
        y = m(x)
        (y * gy).sum().backward()
        out[f"{name}/cfg"] = np.array([c, nh, l, n])
        out[f"{name}/x"], out[f"{name}/gy"], out[f"{name}/y"] = x.detach().numpy(), gy.numpy(), y.detach().numpy()
        out[f"{name}/dx"] = x.grad.numpy().copy()
        for k, v in m.state_dict().items():
            out[f"{name}/sd/{k}"] = v.numpy().copy()
        for k, p in m.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "attn_cross_freq_v2.npz"), **out)
    print("attn_cross_freq_v2.npz:", len(out), "arrays")

    # the chunked V1 module (models.py:100-224): a length that chunk_size divides and one with a ragged last chunk
    out = {}
    for name, (c, nh, l, n, cs) in {"c64_h2_l512_cs256": (64, 2, 512, 2, 256), "c64_h2_l300_cs128": (64, 2, 300, 1, 128)}.items():
        torch.manual_seed(29)
        m = ref.CrossFrequencyCouplingAttention(c, n_heads=nh, chunk_size=cs)
        with torch.no_grad():
            m.gate.fill_(0.6)
            m.coupling_weights.normal_(0, 1.0)
            for conv in list(m.band_q) + list(m.band_k):
                conv.weight.mul_(3.0)
        m.train()
        g = torch.Generator().manual_seed(7)
        tt = torch.arange(l) / 1000.0
        x = torch.randn(n, c, l, generator=g) + 1.5 * torch.sin(2 * np.pi * 10.0 * tt)[None, None] * torch.rand(n, c, 1, generator=g)
        x.requires_grad_(True)
        gy = torch.randn(n, c, l, generator=g)
        y = m(x)
        (y * gy).sum().backward()
        out[f"{name}/cfg"] = np.array([c, nh, l, n, cs])
        out[f"{name}/x"], out[f"{name}/gy"], out[f"{name}/y"] = x.detach().numpy(), gy.numpy(), y.detach().numpy()
        out[f"{name}/dx"] = x.grad.numpy().copy()
        for k, v in m.state_dict().items():
            out[f"{name}/sd/{k}"] = v.numpy().copy()
        for k, p in m.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "attn_cross_freq_v1.npz"), **out)
    print("attn_cross_freq_v1.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
