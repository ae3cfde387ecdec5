"""Inference numbers in the reference's own terms (phase_five/benchmark.py:189-278, targets phase_five/config.py:31-35:
latency < 100 ms at batch 1, > 1000 samples/s) plus sliding-window throughput over a long recording
(BASELINE configs[4] shape: 32 channels, window 5000, stride 2500) on one GPU.

    python scripts/bench_infer.py [--model B|phase5] [--seconds 300] [--batch 64] [--dtype bf16|fp32] [--graph]

--graph         the forward replayed as one HIP graph per batch shape (unet_amd.graphs.GraphedInference) instead of
                ~150 eager launches; the sliding-window run reuses one graph for its full batches

--model B       config B (train.py / LOSO default), FiLM ON: odor ids are passed (without them the reference's FiLM is an
                identity, models.py:505, and the workload is lighter than what training runs)
--model phase5  the phase-five model (phase_five/runner.py:205-213): base=64, cross_freq_v2 bottleneck attention,
                cond_mode="none", called as model(x)

Reports, as LatencyBenchmark does (10 warm-ups, 100 iterations, device-resident input, synchronise around every call):
latency mean / p50 / p95 / p99 at batch 1 (and 8, 32), windows/s and samples/s at each batch size, and the whole
sliding-window reconstruction of a recording (host signal -> device upload and overlap-average included).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import unet_amd
from unet_amd.graphs import GraphedInference
from unet_amd.infer import continuous_inference

N_WARMUP, N_ITER = 10, 100            # phase_five/config.py:93-94


def build(name, dtype):
    if name == "B":
        cfg = dict(in_channels=32, out_channels=32, base=256, conv_type="modern", attention_type="none",
                   activation="gelu", skip_type="add", norm_type="batch", cond_mode="cross_attn_gated", n_downsample=2)
    else:
        cfg = dict(in_channels=32, out_channels=32, base=64, n_downsample=2, attention_type="cross_freq_v2",
                   cond_mode="none")
    torch.manual_seed(0)
    model = unet_amd.CondUNet1D(**cfg).cuda()
    with torch.no_grad():
        model.outc.weight.normal_(0, 0.02)
    model.set_compute_dtype(dtype)
    model.eval()
    return model, cfg


def latency(model, x, kw):
    with torch.no_grad():
        for _ in range(N_WARMUP):
            model(x, **kw)
        torch.cuda.synchronize()
        lat = []
        for _ in range(N_ITER):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(x, **kw)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
    return np.array(lat)


def throughput(model, x, kw):
    with torch.no_grad():
        for _ in range(N_WARMUP):
            model(x, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N_ITER):
            model(x, **kw)
        torch.cuda.synchronize()
    return x.shape[0] * N_ITER / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="B", choices=["B", "phase5"])
    ap.add_argument("--seconds", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64, help="windows per forward of the sliding-window reconstruction")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--window", type=int, default=5000)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model, cfg = build(args.model, dtype)
    film = cfg["cond_mode"] != "none"
    tag = f"model={args.model} ({'FiLM on' if film else 'cond_mode=none'}), {args.dtype}, window {args.window}" + \
        (", HIP graph" if args.graph else "")
    w = args.window
    graphed = GraphedInference(model) if args.graph else None
    call = graphed if args.graph else model
    for bs in (1, 8, 32):
        x = torch.randn(bs, 32, w, device="cuda")
        kw = {"odor_ids": torch.zeros(bs, dtype=torch.long, device="cuda")} if film else {}
        lat = latency(call, x, kw)
        wps = throughput(call, x, kw)
        print(f"{tag}: batch {bs:2d}  latency mean {lat.mean():.2f} ms  p50 {np.percentile(lat, 50):.2f}  "
              f"p95 {np.percentile(lat, 95):.2f}  p99 {np.percentile(lat, 99):.2f}  |  {wps:.0f} windows/s = "
              f"{wps * w:.3e} samples/s", flush=True)
    rng = np.random.default_rng(0)
    signal = rng.standard_normal((32, args.seconds * 1000)).astype(np.float32)
    s = w // 2
    n_windows = (signal.shape[1] - w) // s + 1
    kw = {"odor_ids": torch.zeros(n_windows, dtype=torch.long, device="cuda")} if film else None
    for _ in range(2):
        continuous_inference(model, signal, w, s, batch_windows=args.batch, return_tensor=True, model_kwargs=kw,
                             graphs=graphed or False)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        continuous_inference(model, signal, w, s, batch_windows=args.batch, return_tensor=True, model_kwargs=kw,
                             graphs=graphed or False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{tag}: sliding-window reconstruction of {args.seconds} s x 32 ch ({n_windows} windows of {w}/{s}, batch "
          f"{args.batch}): {ms:.1f} ms per recording = {n_windows / ms * 1e3:.0f} windows/s (host signal -> device "
          f"upload and overlap-average included)", flush=True)


if __name__ == "__main__":
    main()
