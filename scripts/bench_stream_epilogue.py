"""The streaming 1x1 conv with its epilogue variants at config-B FiLM shapes (64 windows): microseconds per launch of
  plain (conv + bias), gated keeping o (training), gated without o (inference), residual add,
against what they replace: plain + unet_gated_add_fwd, and the tap-loop kernel with a residual (UNET_STREAM_NORES=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from unet_amd import _lib, ops
from scripts.bench_rows import timed


def main():
    lib = _lib.load()
    n = 64
    g1 = ops.geom_conv(1)
    for rows, c in ((4096, 256), (2048, 512), (1024, 1024)):
        h = torch.randn(n, rows, c, device="cuda").bfloat16()
        res = torch.randn(n, rows, c, device="cuda").bfloat16()
        gate = torch.randn(n, c, device="cuda")
        w = torch.randn(1, c, c, device="cuda") / c ** 0.5
        b = torch.randn(c, device="cuda")
        wp = ops.pack_weights(w, _lib.BF16, transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=c, c1=c, c2=0)
        out = torch.empty_like(res)
        o = torch.empty_like(res)
        st = torch.cuda.current_stream().cuda_stream

        def plain():
            ops.conv_forward_raw(g1, h, None, wp, b, None, rows, c, out=o)

        def gated_add():
            lib.unet_gated_add_fwd(_lib.BF16, res.data_ptr(), o.data_ptr(), gate.data_ptr(), out.data_ptr(), n, rows, c, st)

        def with_res():
            ops.conv_forward_raw(g1, h, None, wp, b, res, rows, c, out=o)

        t = {}
        t["plain"] = timed(plain)
        t["gated_add"] = timed(gated_add)
        t["gated keep o"] = timed(lambda: ops.conv_forward_gated_raw(g1, h, wp, b, res, gate, out, keep_o=True))
        t["gated out only"] = timed(lambda: ops.conv_forward_gated_raw(g1, h, wp, b, res, gate, out, keep_o=False))
        t["res (stream)"] = timed(with_res)
        os.environ["UNET_STREAM_NORES"] = "1"
        t["res (tap loop)"] = timed(with_res)
        del os.environ["UNET_STREAM_NORES"]
        t["plain again"] = timed(plain)
        print(f"[{rows}x{c}] " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()), flush=True)


if __name__ == "__main__":
    main()
