"""Timing of the data-path kernels at config-B batch shape (64 windows x 32 ch x 4096): achieved HBM GB/s.

    python scripts/bench_data.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from unet_amd.augment import NoiseAugmentor, noise_augment_apply
from unet_amd.windows import DeviceRecordings


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
    B, C, T = 64, 32, 4096
    x = torch.randn(B, C, T, device="cuda")
    g, p = torch.randn_like(x), torch.randn_like(x)
    scales = torch.tensor([0.1, 0.05], device="cuda")
    chu, tu = torch.rand(B, C, device="cuda"), torch.rand(B, T, device="cuda")
    us = timed(lambda: noise_augment_apply(x, g, p, scales, chu, 0.05, tu, 0.02))
    nbytes = 4 * x.numel() * 4
    print(f"noise_augment kernel   {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s (3 reads + 1 write of {x.numel() * 4 / 1e6:.1f} MB)")
    aug = NoiseAugmentor(gaussian_std=0.1, pink_noise=True, pink_noise_std=0.05, channel_dropout=0.05, temporal_dropout=0.02,
                         prob=1.0).train()
    us = timed(lambda: aug(x))
    print(f"NoiseAugmentor module  {us:8.1f} us  (draws by torch's generator + rocFFT irfft + std + the kernel)")
    rng = np.random.default_rng(0)
    subj = [dict(source=rng.standard_normal((C, 300000)).astype(np.float32), target=rng.standard_normal((C, 300000)).astype(np.float32),
                 n_samples=300000, n_source_channels=C, n_target_channels=C) for _ in range(4)]
    for z in (False, True):
        rec = DeviceRecordings(subj, 5000, 2500, zscore_per_window=z)
        idx = torch.randperm(len(rec))[:B]
        us = timed(lambda: rec.batch(idx))
        nbytes = 2 * 2 * B * C * 5000 * 4
        print(f"gather_windows z={int(z)}     {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s (source + target, {B} windows of 5000, incl. host index handling)")


if __name__ == "__main__":
    main()
