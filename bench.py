#!/usr/bin/env python
"""Headline benchmark: train windows/sec of the CondUNet1D hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload (BASELINE.json configs[1]): the train.py/LOSO default model
(in=out=32, base=256, conv_type=modern, attention none, gelu, skip add, batch norm,
cross_attn_gated FiLM; 41,966,772 parameters), synthetic N(0,1) windows of 32 ch x 4096 samples,
64 windows per GPU, bf16 kernels with fp32 master weights.  One step = forward + L1 on the
[500:4500] crop + backward + bucketed gradient all-reduce (N>1) + grad-clip 5.0 + AdamW step,
i.e. everything the reference's train_epoch does per batch (train.py:1395-1560).
``--workload B2`` = the same with skip_type="concat" (optimal_config.json); ``--workload C`` = BASELINE.json
configs[2], the wide ECoG stress (128 channels in/out, n_downsample=4, ladder 256-512-1024-2048-2048,
446,645,540 parameters).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel (conv_fwd2_kernel: implicit-GEMM conv forward/data-grad on bf16
                  MFMA), timed live with HIP events around every launch (include/unet_hip.h
                  unet_conv1d_last_variant attributes each launch to its kernel); "traffic" = HBM bytes
                  per launch from the committed rocprofv3 --pmc passes (scripts/pmc_traffic.sh)
  cpu_baseline -- the CPU oracle (oracle/condunet_oracle.py, a port of the reference's algorithm
                  in torch fp32) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.nn.functional as F

CONFIG_B = dict(in_channels=32, out_channels=32, base=256, conv_type="modern", attention_type="none",
                activation="gelu", skip_type="add", norm_type="batch", cond_mode="cross_attn_gated",
                n_downsample=2)
WORKLOADS = {
    # name: (constructor overrides, algorithmic forward GFLOP per window, description)
    "B": ({}, 122.339, "config B: CondUNet1D train.py/LOSO default (base=256, modern 4x dilated k7 + SE, FiLM, skip add, BN)"),
    "B2": (dict(skip_type="concat"), 167.437, "config B2: config B with skip_type=concat (optimal_config.json)"),
    "C": (dict(in_channels=128, out_channels=128, n_downsample=4), 472.178,
          "config C: wide ECoG stress (128 ch in/out, n_downsample=4, ladder 256-512-1024-2048-2048, 446.6 M parameters)"),
}
T = 4096
CROP = (500, 4500)
BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16
TRAFFIC_FILE = "r03_hbm_traffic.json"
KERNEL_DESCR = {
    "conv_fwd2_kernel": "conv_fwd2_kernel_128/_192/_256 (bf16 implicit-GEMM tap-list conv, forward and data-grad launches)",
    "conv_stream_kernel": "conv_stream_kernel_t<0..3> (bf16 single-tap conv, persistent 256x128 tiles; plain / gated / residual epilogue)",
    "conv_wgrad2_kernel": "conv_wgrad2_kernel (bf16 weight-grad, two taps per block, split-K slabs)",
    "conv_wgrad3_kernel": "conv_wgrad3_kernel (bf16 weight-grad of single-tap convs at >= 512 channels, 256x256 tiles)",
    "conv_fwd_kernel": "conv_fwd_kernel (generic / fp32 implicit-GEMM conv)",
    "conv_wgrad_kernel": "conv_wgrad_kernel (generic / fp32 weight-grad)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="B", choices=sorted(WORKLOADS), help="B (default, the headline config), B2, C")
    ap.add_argument("--batch", type=int, default=64, help="windows per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=8, help="windows per CPU-baseline step (1 warm-up + 3 timed steps "
                    "at each thread count)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--optimizer", default="native", choices=["native", "torch"],
                    help="native: unet_amd.optim.ClipAdamW (clip + AdamW in three HIP launches); torch: "
                         "clip_grad_norm_ + torch.optim.AdamW(fused=True)")
    ap.add_argument("--conditioning", default="odor_onehot", choices=["odor_onehot", "spectro_temporal"],
                    help="spectro_temporal: the SpectroTemporalEncoder (train.py --conditioning default) runs inside the "
                         "step and trains with the model; the headline config is odor_onehot")
    ap.add_argument("--attention", default="none", choices=["none", "cross_freq_v2"],
                    help="cross_freq_v2: the phase-four / phase-five bottleneck attention (band-pair flash kernels); the "
                         "headline config B has none")
    ap.add_argument("--graph", action="store_true", help="N > 1: two captured graphs with the all-reduce between them "
                    "(no overlap of communication with backward)")
    ap.add_argument("--graph-overlap", action="store_true", help="N > 1 (opt-in; the default there is --eager): ONE captured graph with the "
                    "bucket all-reduces inside it, forked off the backward stream (graphs.GraphedTrainStep(overlap=True)); "
                    "if any rank's capture is refused, every rank falls back to the eager hook-driven path")
    ap.add_argument("--dp-selftest", action="store_true", help="N = 1 only: run the N > 1 code path (eager launches, bucket "
                    "hooks, one RCCL all-reduce per bucket) through a one-rank process group with the collectives forced on; "
                    "measures what the data-parallel machinery costs beside the step, not the xGMI transport")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one (hook-driven overlapped "
                    "all-reduce) instead of replaying the captured HIP graph")
    return ap.parse_args()


def _cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(n_windows: int, cfg, gflop_fwd: float):
    """Oracle (CPU port of the reference path) on a bounded sample of the same workload, as BASELINE.md section 3
    plans it: 1 warm-up step + 3 timed steps, median, at k = 8 threads (comparable with the 8-thread container
    anchors of BASELINE.md) and at k = this process's CPU share (torch's default thread count: OMP_NUM_THREADS,
    16 on a one-GPU box).  ``value`` is the larger thread count's."""
    from oracle import condunet_oracle as O
    share = torch.get_num_threads()
    sd = O.init_state_dict(cfg, seed=0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    run_sd = dict(sd)
    run_sd.update(leaves)
    g = torch.Generator().manual_seed(0)
    ob = torch.randn(n_windows, cfg["in_channels"], T, generator=g)
    tgt = torch.randn(n_windows, cfg["out_channels"], T, generator=g)
    ids = torch.zeros(n_windows, dtype=torch.long)
    rates, total = {}, 0.0
    for k in sorted({min(8, share), share}):
        torch.set_num_threads(k)
        times = []
        for rep in range(4):                     # the first one is the warm-up
            for v in leaves.values():
                v.grad = None
            t0 = time.perf_counter()
            O.train_step_loss(run_sd, cfg, ob, tgt, odor_ids=ids).backward()
            times.append(time.perf_counter() - t0)
        total += sum(times)
        rates[k] = n_windows / sorted(times[1:])[1]
    torch.set_num_threads(share)
    gf = 3 * gflop_fwd
    return {"value": rates[share], "unit": "windows/s", "cores": share, "kind": "port",
            "by_threads": {str(k): v for k, v in rates.items()},
            "sample": f"fwd+bwd steps of the same config on {n_windows} windows each (fp32 torch CPU ops), 1 warm-up + 3 timed, "
                      f"median, at " + " and ".join(f"{k} threads ({v:.2f} windows/s = {v * gf:.0f} GFLOP/s)" for k, v in rates.items())
                      + f"; {total:.1f} s of CPU work on '{_cpu_name()}'"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # N > 1 default: kernels launched one by one with the hook-driven bucket all-reduces overlapping backward -- the
    # protocol every multi-rank RCCL job runs.  On one rank it measures the same as the captured step (23.21 against
    # 23.29 ms, --dp-selftest).  --graph-overlap captures the whole step INCLUDING the collectives as one graph; that
    # path has only ever met a one-rank group (no multi-GPU box in this build), so an unattended scaling run does not
    # depend on it: a capture that hung instead of raising would cost every N > 1 point.
    if world > 1 and not (args.graph or args.graph_overlap):
        args.eager = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif args.dp_selftest:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29733", world_size=1, rank=0,
                                device_id=torch.device("cuda", 0))
        args.eager = not args.graph_overlap      # --dp-selftest --graph-overlap: the captured N > 1 step on one GPU
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import unet_amd
    from unet_amd import ops
    from unet_amd.dp import BucketedGradAllReduce, broadcast_parameters

    torch.manual_seed(0)
    over, gflop_fwd, wl_descr = WORKLOADS[args.workload]
    cfg = dict(CONFIG_B, attention_type=args.attention, **over)
    c_in, c_out = cfg["in_channels"], cfg["out_channels"]
    model = unet_amd.CondUNet1D(**cfg)
    with torch.no_grad():  # the reference zero-initialises outc; give it values so backward is not trivially zero
        model.outc.weight.normal_(0, 0.02)
    model = model.to(dev)
    from unet_amd.dp import coalesced_grad_mean
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.set_compute_dtype(cdt)
    model.train()
    broadcast_parameters(model)
    n_params = sum(p.numel() for p in model.parameters())

    enc = None
    groups = [{"params": list(model.parameters())}]
    if args.conditioning == "spectro_temporal":
        from unet_amd.cond import SpectroTemporalEncoder
        enc = SpectroTemporalEncoder(in_channels=c_in, emb_dim=128).to(dev)
        enc.set_compute_dtype(cdt)
        enc.eval()                                                         # train.py:1410-1411
        broadcast_parameters(enc)
        groups.append({"params": list(enc.parameters()), "name": "cond_encoder"})  # train.py:2225-2227
    BETAS = (0.7595905764360957, 0.920298282605139)                         # train.py:183-193

    def make_optimizer(capturable):
        if args.optimizer == "native":     # K17: clip 5.0 + AdamW as three HIP launches (the encoder keeps its own clip)
            from unet_amd.optim import ClipAdamW
            groups[0]["max_grad_norm"] = 5.0
            for grp in groups[1:]:
                grp["max_grad_norm"] = None
            return ClipAdamW(groups, lr=2e-4, betas=BETAS, weight_decay=0.0)
        return torch.optim.AdamW(groups, lr=2e-4, betas=BETAS, weight_decay=0.0, fused=True, capturable=capturable)

    opt = make_optimizer(not args.eager)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    B = args.batch
    ob = torch.randn(B, c_in, T, device=dev, generator=g).to(cdt)
    tgt = torch.randn(B, c_out, T, device=dev, generator=g) * 1.7 + 0.3     # raw target batch
    ids = torch.zeros(B, dtype=torch.long, device=dev)

    from unet_amd import metrics

    def loss_fn(pred, target):
        # train.py:1405 (target z-score, per batch) + train.py:1481-1486 (L1 on the [500:4500] crop)
        return metrics.l1_crop_loss(pred, metrics.per_channel_normalize(target), 1.0, CROP[0], CROP[1])

    mode = "eager"
    graphed = None
    if not args.eager:
        try:
            from unet_amd.graphs import GraphedTrainStep
            fwd = (lambda o, i: model(o, i, cond_emb=enc(o))) if enc is not None else None
            graphed = GraphedTrainStep(model, loss_fn, opt, [ob, ids], [tgt], clip_norm=5.0, forward=fwd,
                                       extra_modules=[enc] if enc is not None else [], overlap=args.graph_overlap,
                                       force_collectives=args.dp_selftest)
            mode = "hipgraph+rccl-in-graph" if graphed.overlap else "hipgraph"
        except Exception as e:  # capture unsupported in this environment: run the same step eagerly
            import traceback
            traceback.print_exc()
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}); running eagerly", file=sys.stderr)
            graphed = None
        if world > 1:           # one rank on the eager path, the others replaying: keep them on the same protocol
            ok = torch.tensor([1.0 if graphed is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                graphed = None
        if graphed is None:
            mode = "eager"
            opt = make_optimizer(False)
    if graphed is None:
        reducer = BucketedGradAllReduce(model, force=args.dp_selftest)

    def step():
        if graphed is not None:
            return graphed()
        reducer.zero_grad()
        if enc is not None:
            for p in enc.parameters():
                p.grad = None
        loss = loss_fn(model(ob, ids, cond_emb=enc(ob)) if enc is not None else model(ob, ids), tgt)
        loss.backward()
        reducer.finish()
        if not getattr(opt, "clips_gradients", False):
            torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)  # train.py:1553 GRAD_CLIP
        if enc is not None:                                      # train.py:1556-1558
            torch.nn.utils.clip_grad_norm_(enc.parameters(), 5.0)
            coalesced_grad_mean(list(enc.parameters()))
        opt.step()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # per-launch HIP events inside the timed region cost ~0.25 ms per step; a data-parallel run (whose per-N values the
    # driver turns into scaling efficiency against the graph-replayed N = 1 run) takes them after the region instead
    in_region = not args.no_kernel_timing and graphed is None and world == 1 and not args.dp_selftest
    if in_region:
        ops.PROFILE = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        loss = step()
        evs[i + 1].record()
    host_dt = time.perf_counter() - t0   # time to ENQUEUE the steps (host side); < dt when the GPU is the bottleneck
    sync()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    timing_mode = "HIP events around every conv launch inside the timed region (eager launches)"
    if not in_region and not args.no_kernel_timing:
        # A replayed HIP graph has no host-visible launch boundaries, so the per-kernel durations are taken
        # from eager launches of the SAME step (same tensors, same kernels, same stream) right after the timed
        # region; profiles/ holds the rocprofv3 summary of the graph replay itself for cross-checking.
        ops.PROFILE = []
        for _ in range(min(3, args.steps)):
            if graphed is None:
                step()            # every rank runs the same full step (its collectives included)
                continue
            model.zero_grad(set_to_none=True)
            if enc is not None:
                enc.zero_grad(set_to_none=True)
            loss_fn(model(ob, ids, cond_emb=enc(ob)) if enc is not None else model(ob, ids), tgt).backward()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        prof_steps = min(3, args.steps)
        timing_mode = ("HIP events around every conv launch, eager launches of the same step right after the "
                       + ("graph-replayed" if graphed is not None else "data-parallel") + " timed region")
    else:
        prof_steps = args.steps
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    windows = B * world * args.steps
    value = windows / dt
    out = {
        "metric": "train windows/sec (fwd+bwd)", "value": value, "unit": "windows/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic N(0,1) windows, random-init weights (reference init, seed 0)",
        "config": {"workload": wl_descr + f", {c_in}ch x {T} windows, {B} windows/GPU, fwd + L1[500:4500] + bwd + "
                               "grad all-reduce + clip 5.0 + AdamW", "windows_per_gpu": B, "params": n_params,
                   "parallelism": f"dp{world}" + (" (one-rank RCCL group, collectives forced)" if args.dp_selftest else ""), "launch": mode, "conditioning": args.conditioning,
                   "attention": args.attention, "optimizer": args.optimizer},
        "final_loss": float(loss.item()), "host_enqueue_ms_per_step": 1e3 * host_dt / args.steps,
        "ms_each_step": [round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(args.steps)],
        "model_tflops_algorithmic": value * 3 * gflop_fwd / 1e3,
        "frac_of_bf16_mfma_peak_end_to_end": value * 3 * gflop_fwd / 1e3 / (BF16_DENSE_PEAK_TFLOPS * world),
    }
    if prof:
        agg = {}
        for kind, flops, execf, e0, e1 in prof:
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += execf
        kern = {k: {"launches": v[2], "avg_us": 1e6 * v[1] / v[2], "tflops_algorithmic": v[0] / v[1] / 1e12,
                    "tflops_executed": v[3] / v[1] / 1e12, "ms_per_step": 1e3 * v[1] / prof_steps} for k, v in agg.items()}
        # dominant kernel = the conv kernel with the largest share of the step (conv_fwd2_kernel: tiled
        # implicit-GEMM forward / data-grad); its HBM traffic per launch comes from the committed PMC passes
        dom_name = max(agg, key=lambda k: agg[k][1])
        dom = agg[dom_name]
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(tpath) and B == 64 and args.dtype == "bf16" and args.workload == "B":
            tk = json.load(open(tpath))["kernels"]
            hit = [v for k, v in tk.items() if k.startswith(dom_name)]
            if hit:
                traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit)
                traffic_src = f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this step, FETCH_SIZE doubled for gfx950)"
        # achieved / frac price the kernel on the FLOPs it EXECUTES (23 merged taps); the reference algorithm's count
        # (28 branch taps) is reported beside it and is what the windows/s-derived figures use
        out["roofline"] = {"bound": "mfma", "kernel": KERNEL_DESCR.get(dom_name, dom_name),
                           "achieved": dom[3] / dom[1] / 1e12, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": dom[3] / dom[1] / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                           "achieved_algorithmic": dom[0] / dom[1] / 1e12,
                           "frac_algorithmic": dom[0] / dom[1] / 1e12 / BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                           "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                           "flops_per_launch_algorithmic": dom[0] / dom[2], "flops_per_launch_executed": dom[3] / dom[2],
                           "avg_launch_us": 1e6 * dom[1] / dom[2], "launches": dom[2], "timing": timing_mode}
        out["kernels"] = kern
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_windows if args.workload != "C" else max(1, args.cpu_windows // 4), cfg, gflop_fwd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or args.dp_selftest:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
