"""Achieved HBM GB/s of the row-streaming kernels at config-B shapes (64 windows, bf16).

    python scripts/bench_rows.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from unet_amd import ops
from unet_amd._lib import ACT_IDS


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    n = 64
    for rows, c in ((4096, 256), (2048, 512), (1024, 1024)):
        y = torch.randn(n, rows, c, device="cuda").bfloat16()
        dz = torch.randn_like(y)
        nb = y.numel() * 2
        a = torch.rand(n, c, device="cuda") + 0.5
        gelu = ACT_IDS["gelu"]
        stats = dict(mean=torch.zeros(n, c, device="cuda", dtype=torch.float64), rstd=torch.ones(n, c, device="cuda", dtype=torch.float64),
                     coupled=False)
        gamma = torch.ones(c, device="cuda")
        cases = [
            ("act none", 2, lambda: ops.activation(y, 0)),
            ("act relu", 2, lambda: ops.activation(y, ACT_IDS["relu"])),
            ("act gelu", 2, lambda: ops.activation(y, gelu)),
            ("act silu", 2, lambda: ops.activation(y, ACT_IDS["silu"])),
            ("rowstats", 1, lambda: ops.rowstats(y)),
            ("colsum", 1, lambda: ops.colsum(y)),
            ("affine_act_fwd", 2, lambda: ops.AffineActFn.apply(y, gamma, gamma, stats, gelu, False)),
            ("affine_act_fwd+pool", 2, lambda: ops.AffineActFn.apply(y, gamma, gamma, stats, gelu, True)),
            ("scale_fwd", 2, lambda: ops.ScaleFn.apply(y, a)),
            ("gated_add_fwd", 3, lambda: ops.GatedAddFn.apply(y, dz, a)),
        ]
        for name, touches, fn in cases:
            us = timed(fn)
            print(f"[{rows}x{c}] {name:22s} {us:7.1f} us  {touches * nb / us / 1e3:7.1f} GB/s ({touches} touches, incl. small follow-up kernels)", flush=True)


if __name__ == "__main__":
    main()
