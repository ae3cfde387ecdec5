"""GPU check + timing of a weight-gradient dispatch switch: runs unet_conv1d_wgrad on config-B layers with the
environment variable ENV set to 0 and to 1 alternately (A B A B on one box), prints microseconds per call (slab sum
included), the kernel variant each picked and the difference of the two results.

    python scripts/ab_wgrad_env.py UNET_WGRAD_QUAD [layer ...]
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import _lib, ops
from scripts.bench_conv import LAYERS, geom_of

env = sys.argv[1]
lib = _lib.load()
dev = torch.device("cuda")
for name in sys.argv[2:] or ["dec0_m", "enc1_m", "enc0_m", "dec1_m", "mid", "up0", "outc"]:
    kind, cin, cout, l_in, stride = LAYERS[name]
    geom = geom_of(kind, stride)
    l_out = 2 * l_in if kind == "convt" else (l_in - 1) // stride + 1
    n = 64
    x = torch.randn(n, l_in, cin, device=dev).bfloat16()
    dy = torch.randn(n, l_out, cout, device=dev).bfloat16()
    d = ops._desc(geom, x, None, dy, None)
    res = {}
    for flag in ("0", "1", "0", "1"):
        os.environ[env] = flag
        ws_bytes = lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d))
        ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
        dw = torch.zeros((geom.n_taps, cout, cin), dtype=torch.float32, device=dev)
        call = lambda: _lib.check(lib.unet_conv1d_wgrad(C.byref(d), ops._p(x), None, ops._p(dy), ops._p(dw), ops._p(ws),
                                                        ws_bytes, ops._stream()), "wgrad")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(flag, []).append((e0.elapsed_time(e1) * 1e3 / 20, dw.clone(), ws_bytes))
    a, b = res["0"][0][1], res["1"][0][1]
    rel = ((a - b).norm() / a.norm()).item()
    print(f"{name:8s} {env}=0 {res['0'][0][0]:8.1f} {res['0'][1][0]:8.1f} us (ws {res['0'][0][2] >> 20} MiB)   =1 "
          f"{res['1'][0][0]:8.1f} {res['1'][1][0]:8.1f} us (ws {res['1'][0][2] >> 20} MiB)   rel diff {rel:.2e}", flush=True)
os.environ.pop(env, None)
