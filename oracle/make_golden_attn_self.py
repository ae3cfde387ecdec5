"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/attn_self.npz: OUTPUTS of the reference's SelfAttention1D
(models.py:75-97) at the config-A bottleneck width (C = 512, query/key width 64) for L = 300 and L = 1024, with an
open gate and peaked scores.  Module weights and inputs are regenerated from seeds on both sides (``build``); the
record holds checksums of the weights (so a different init order would be caught) plus checksums and strided slices
of the output, the input gradient and every parameter gradient under a fixed linear loss.

Runs ONLY in the build container (imports /root/reference/models.py; nothing is copied):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_attn_self.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_fullwidth import strided, sums  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "..", "tests", "golden")
CASES = {"c512_l300": (512, 300, 2), "c512_l1024": (512, 1024, 1)}     # channels, length, batch


def build(cls, c, l, n, seed=31):
    torch.manual_seed(seed)
    m = cls(c)
    with torch.no_grad():
        m.gamma.fill_(0.8)
        m.query.weight.mul_(2.0)       # no 1/sqrt(d) in this layer: scores are peaked
        m.key.weight.mul_(2.0)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(n, c, l, generator=g)
    gy = torch.randn(n, c, l, generator=g)
    return m, x, gy


def main():
    sys.path.insert(0, REF)
    import models as ref
    torch.set_num_threads(8)
    out = {}
    for name, (c, l, n) in CASES.items():
        m, x, gy = build(ref.SelfAttention1D, c, l, n)
        x.requires_grad_(True)
        y = m(x)
        (y * gy).sum().backward()
        for k, v in m.state_dict().items():
            out[f"{name}/sd/{k}"] = sums(v)
        for key, t in (("y", y), ("dx", x.grad)):
            out[f"{name}/{key}/sums"], out[f"{name}/{key}/slice"] = sums(t), strided(t).numpy()
        for k, p in m.named_parameters():
            out[f"{name}/grad/{k}/sums"], out[f"{name}/grad/{k}/slice"] = sums(p.grad), strided(p.grad).numpy()
    np.savez_compressed(os.path.join(OUT, "attn_self.npz"), **out)
    print("attn_self.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
