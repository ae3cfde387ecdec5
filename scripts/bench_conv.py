"""Micro-benchmark of the conv kernels on config-B layer shapes (N=64 windows, bf16).

    python scripts/bench_conv.py [--only NAME] [--iters 20] [--kind fwd|dgrad|wgrad|all]

Prints one line per (layer, kind): average microseconds over `iters` back-to-back launches (HIP events on
the launch stream) and algorithmic / executed TFLOP/s.  Used under rocprofv3 --pmc for counter passes.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from unet_amd import _lib, ops

# name: (kind of geometry, cin, cout, L_in, stride)
LAYERS = {
    "inc_m": ("merged", 32, 256, 4096, 1),
    "enc0_m": ("merged", 256, 512, 4096, 2),
    "enc1_m": ("merged", 512, 1024, 2048, 2),
    "dec0_m": ("merged", 512, 512, 2048, 1),
    "dec1_m": ("merged", 256, 256, 4096, 1),
    "film256": ("k1", 256, 256, 4096, 1),
    "film512": ("k1", 512, 512, 2048, 1),
    "film1024": ("k1", 1024, 1024, 1024, 1),
    "mid": ("k3", 1024, 1024, 1024, 1),
    "up0": ("convt", 1024, 512, 1024, 1),
    "up1": ("convt", 512, 256, 2048, 1),
    "outc": ("k3", 256, 32, 4096, 1),
}


def geom_of(kind, stride):
    if kind == "merged":
        g, _ = ops.merged_dilated_taps(7, (1, 4, 16, 32))
        return ops.ConvGeom(g.offs, g.phases, stride, 1, g.alg_scale)
    if kind == "k1":
        return ops.geom_conv(1)
    if kind == "k3":
        return ops.geom_conv(3, stride)
    if kind == "convt":
        return ops.geom_convt_k4s2p1()[0]
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kind", default="all")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lib = _lib.load()
    names = [args.only] if args.only else list(LAYERS)
    for name in names:
        kind, cin, cout, l_in, stride = LAYERS[name]
        geom = geom_of(kind, stride)
        if kind == "convt":
            l_out = 2 * l_in
        else:
            l_out = (l_in - 1) // stride + 1
        n = args.n
        x = torch.randn(n, l_in, cin, device=dev).to(dt)
        w = torch.randn(geom.n_taps, cout, cin, device=dev) / (cin * geom.n_taps) ** 0.5
        dy = torch.randn(n, l_out, cout, device=dev).to(dt)
        flops = geom.alg_flops(n, l_out, cout, cin)
        execf = flops / geom.alg_scale
        kinds = ["fwd", "dgrad", "wgrad"] if args.kind == "all" else [args.kind]
        if args.kind == "pair":      # data-grad + weight-grad of one layer: back to back vs on two streams
            gd, src = geom.dgrad()
            wpt = ops.pack_weights(w, ops._dt(x), transpose=True, tap_src=src, n_taps=geom.n_taps, row_lo=0, rows=cin,
                                   c1=cout, c2=0)
            d = ops._desc(geom, x, None, dy, None)
            ws_bytes = lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d))
            ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
            dw = torch.empty((geom.n_taps, cout, cin), dtype=torch.float32, device=dev)
            dgrad = lambda: ops.conv_forward_raw(gd, dy, None, wpt, None, None, l_in, cin)
            wgrad = lambda: _lib.check(lib.unet_conv1d_wgrad(C.byref(d), ops._p(x), None, ops._p(dy), ops._p(dw), ops._p(ws),
                                                             ws_bytes, ops._stream()), "wgrad")
            side = torch.cuda.Stream()
            main_s = torch.cuda.current_stream()

            def both_seq():
                dgrad()
                wgrad()

            def both_conc():
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    wgrad()
                dgrad()
                main_s.wait_stream(side)

            res = {}
            for label, fn in (("seq", both_seq), ("conc", both_conc)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[label] = e0.elapsed_time(e1) * 1e3 / args.iters
            print(f"{name:9s} dgrad+wgrad  sequential {res['seq']:8.1f} us   two streams {res['conc']:8.1f} us "
                  f"({100 * (res['conc'] / res['seq'] - 1):+.1f} %)", flush=True)
            continue
        if args.kind == "glue":      # weight-grad (MFMA-bound) beside the HBM-bound row kernels of a block's backward
            d = ops._desc(geom, x, None, dy, None)
            ws_bytes = lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d))
            ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
            dw = torch.empty((geom.n_taps, cout, cin), dtype=torch.float32, device=dev)
            wgrad = lambda: _lib.check(lib.unet_conv1d_wgrad(C.byref(d), ops._p(x), None, ops._p(dy), ops._p(dw), ops._p(ws),
                                                             ws_bytes, ops._stream()), "wgrad")
            y = torch.randn(n, l_out, cout, device=dev).to(dt)
            dz = torch.randn(n, l_out, cout, device=dev).to(dt)
            o1, o2 = torch.empty_like(y), torch.empty_like(y)
            a = torch.rand(n, cout, device=dev) + 0.5
            b = torch.randn(n, cout, device=dev) * 0.1
            p1, p2 = ops._part_buf(y, 2)
            st = ops._stream

            def glue():      # ~ SE-excite bwd, BN/GELU bwd reduce + apply, gated-add bwd: 11 tensor touches
                _lib.check(lib.unet_scale_bwd(ops._dt(y), ops._p(y), ops._p(dz), ops._p(a), ops._p(o1), ops._p(p1), n, l_out, cout, st()), "s")
                _lib.check(lib.unet_affine_act_bwd_reduce(ops._dt(y), ops._p(y), ops._p(o1), None, ops._p(a), ops._p(b), 3, ops._p(p1), ops._p(p2), n, l_out, cout, st()), "r")
                _lib.check(lib.unet_affine_act_bwd_apply(ops._dt(y), ops._p(y), ops._p(o1), None, ops._p(a), ops._p(b), 3, ops._p(a), ops._p(b), ops._p(b), ops._p(o2), n, l_out, cout, st()), "a")
                _lib.check(lib.unet_gated_add_bwd(ops._dt(y), ops._p(y), ops._p(dz), ops._p(a), ops._p(o1), ops._p(p1), ops._p(p2), n, l_out, cout, st()), "g")

            side = torch.cuda.Stream()
            main_s = torch.cuda.current_stream()

            def seq():
                wgrad()
                glue()

            def conc():
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    wgrad()
                glue()
                main_s.wait_stream(side)

            def conc2():     # row kernels dispatched first
                side.wait_stream(main_s)
                glue()
                with torch.cuda.stream(side):
                    wgrad()
                main_s.wait_stream(side)

            res = {}
            for label, fn in (("wgrad", wgrad), ("glue", glue), ("seq", seq), ("conc", conc), ("conc2", conc2)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[label] = e0.elapsed_time(e1) * 1e3 / args.iters
            print(f"{name:9s} wgrad {res['wgrad']:7.1f}  glue {res['glue']:7.1f}  sequential {res['seq']:7.1f}  two streams "
                  f"{res['conc']:7.1f} us ({100 * (res['conc'] / res['seq'] - 1):+.1f} %)  glue-first {res['conc2']:7.1f} us "
                  f"({100 * (res['conc2'] / res['seq'] - 1):+.1f} %)", flush=True)
            continue
        for k in kinds:
            if k == "fwd":
                wp = ops.pack_weights(w, ops._dt(x), transpose=False, tap_src=None, n_taps=geom.n_taps, row_lo=0,
                                      rows=cout, c1=cin, c2=0)
                fn = lambda: ops.conv_forward_raw(geom, x, None, wp, None, None, l_out, cout)
            elif k == "fwdstats":   # forward with the BatchNorm statistics from the epilogue (UNET_EPILOGUE_STATS=0: without)
                wp = ops.pack_weights(w, ops._dt(x), transpose=False, tap_src=None, n_taps=geom.n_taps, row_lo=0,
                                      rows=cout, c1=cin, c2=0)
                fn = lambda: ops.conv_forward_raw(geom, x, None, wp, None, None, l_out, cout, want_stats=True)
            elif k == "dgrad":
                gd, src = geom.dgrad()
                wp = ops.pack_weights(w, ops._dt(x), transpose=True, tap_src=src, n_taps=geom.n_taps, row_lo=0,
                                      rows=cin, c1=cout, c2=0)
                fn = lambda: ops.conv_forward_raw(gd, dy, None, wp, None, None, l_in, cin)
            else:
                d = ops._desc(geom, x, None, dy, None)
                ws_bytes = lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d))
                ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
                dw = torch.empty((geom.n_taps, cout, cin), dtype=torch.float32, device=dev)
                fn = lambda: _lib.check(lib.unet_conv1d_wgrad(C.byref(d), ops._p(x), None, ops._p(dy), ops._p(dw), ops._p(ws),
                                                              ws_bytes, ops._stream()), "wgrad")
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            print(f"{name:9s} {k:5s} {us:8.1f} us  alg {flops / us / 1e6:7.1f} TF  exec {execf / us / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
