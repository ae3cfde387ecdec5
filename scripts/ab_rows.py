"""Same-process A/B of the row-streaming kernels of two builds of libunet_hip.so (raw C-ABI launches, interleaved
rounds, HIP events; config-B tensor shapes, 64 windows, bf16).

    python scripts/ab_rows.py unet_amd/libunet_hip_A.so unet_amd/libunet_hip.so [substring of the case names to run]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from unet_amd import _lib


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    libs = [bind(p) for p in sys.argv[1:3]]
    n = 64
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    for rows, c in ((4096, 256), (2048, 512), (1024, 1024)):
        y = torch.randn(n, rows, c, device="cuda").bfloat16()
        dz, o1 = torch.randn_like(y), torch.empty_like(y)
        a = torch.rand(n, c, device="cuda") + 0.5
        b = torch.randn(n, c, device="cuda") * 0.1
        parts = int(libs[0].unet_parts(rows))
        p1 = torch.empty(n, parts, c, device="cuda")
        p2 = torch.empty_like(p1)
        p5 = torch.empty(5, n, parts, c, device="cuda")
        nb = y.numel() * 2
        cases = [
            ("rowstats", 1, lambda L: L.unet_rowstats(0, P(y), P(p1), P(p2), n, rows, c, st)),
            ("colsum", 1, lambda L: L.unet_colsum(0, P(y), P(p1), n, rows, c, st)),
            ("affine_act_fwd+pool", 2, lambda L: L.unet_affine_act_fwd(0, P(y), P(a), P(b), 3, P(o1), P(p1), n, rows, c, st)),
            ("affine_act_bwd_reduce", 2, lambda L: L.unet_affine_act_bwd_reduce(0, P(y), P(dz), None, P(a), P(b), 3, P(p1), P(p2), n, rows, c, st)),
            ("affine_act_bwd_apply", 3, lambda L: L.unet_affine_act_bwd_apply(0, P(y), P(dz), None, P(a), P(b), 3, P(a), P(b), P(b), P(o1), n, rows, c, st)),
            ("affine_pool", 1, lambda L: L.unet_affine_pool(0, P(y), P(a), P(b), 3, P(p1), n, rows, c, st)),
            ("affine_act_scale_fwd", 2, lambda L: L.unet_affine_act_scale_fwd(0, P(y), P(a), P(b), 3, P(a), P(o1), n, rows, c, st)),
            ("se_bwd_reduce", 2, lambda L: L.unet_affine_act_se_bwd_reduce(0, P(y), P(dz), P(a), P(b), 3, P(p5), n, rows, c, st)),
            ("se_bwd_apply", 3, lambda L: L.unet_affine_act_se_bwd_apply(0, P(y), P(dz), P(b), P(a), P(a), P(b), 3, P(a), P(b), P(b), P(o1), P(p1), n, rows, c, st)),
            ("film_softmax_fwd", 2, lambda L: L.unet_film_softmax_fwd(0, P(y), P(a), P(b), P(o1), n, rows, c, st)),
            ("film_softmax_bwd+sum", 3, lambda L: L.unet_film_softmax_bwd(0, P(y), P(dz), P(a), P(b), P(o1), P(p1), P(p2), P(p5[0]), n, rows, c, st)),
            ("scale_fwd", 2, lambda L: L.unet_scale_fwd(0, P(y), P(a), P(o1), n, rows, c, st)),
            ("scale_bwd", 3, lambda L: L.unet_scale_bwd(0, P(y), P(dz), P(a), P(o1), P(p1), n, rows, c, st)),
            ("gated_add_fwd", 3, lambda L: L.unet_gated_add_fwd(0, P(y), P(dz), P(a), P(o1), n, rows, c, st)),
            ("gated_add_bwd+sum", 3, lambda L: L.unet_gated_add_bwd(0, P(y), P(dz), P(a), P(o1), P(p1), P(p2), n, rows, c, st)),
        ]
        only = sys.argv[3] if len(sys.argv) > 3 else ""
        for name, touches, fn in cases:
            if only not in name:
                continue
            best = [1e9, 1e9]
            for rnd in range(4):
                for i in (0, 1, 1, 0):
                    for _ in range(2):
                        assert fn(libs[i]) == 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        fn(libs[i])
                    e1.record()
                    torch.cuda.synchronize()
                    best[i] = min(best[i], e0.elapsed_time(e1) * 100)
            print(f"[{rows}x{c}] {name:22s} A {best[0]:6.1f} us {touches * nb / best[0] / 1e6:5.2f} TB/s   "
                  f"B {best[1]:6.1f} us {touches * nb / best[1] / 1e6:5.2f} TB/s   ({100 * (best[1] / best[0] - 1):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
