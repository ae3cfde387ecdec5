"""Sliding-window inference throughput (BASELINE configs[4] shape: 32 channels, window 5000, stride 2500) on one GPU.

    python scripts/bench_infer.py [--seconds 300] [--attention none|cross_freq_v2] [--batch 64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import unet_amd
from unet_amd.infer import continuous_inference


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=300)
    ap.add_argument("--attention", default="none")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    cfg = dict(in_channels=32, out_channels=32, base=256, conv_type="modern", attention_type=args.attention,
               activation="gelu", skip_type="add", norm_type="batch", cond_mode="cross_attn_gated", n_downsample=2)
    torch.manual_seed(0)
    model = unet_amd.CondUNet1D(**cfg).cuda()
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    rng = np.random.default_rng(0)
    signal = rng.standard_normal((32, args.seconds * 1000)).astype(np.float32)
    w, s = 5000, 2500
    n_windows = (signal.shape[1] - w) // s + 1
    kw = dict(model_kwargs={"odor_ids": None})
    for _ in range(2):
        continuous_inference(model, signal, w, s, batch_windows=args.batch, return_tensor=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 3
    for _ in range(reps):
        out = continuous_inference(model, signal, w, s, batch_windows=args.batch, return_tensor=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"sliding-window inference, config B, attention={args.attention}: {args.seconds} s x 32 ch, {n_windows} windows "
          f"of {w}/{s}, batch {args.batch}: {ms:.1f} ms per recording = {n_windows / ms * 1e3:.0f} windows/s "
          f"(host signal -> device included)", flush=True)


if __name__ == "__main__":
    main()
