"""FiLM channel-softmax kernels (forward / backward) at config-B shapes: microseconds and achieved GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from unet_amd import ops
from scripts.bench_rows import timed


def main():
    n = 64
    for rows, c in ((4096, 256), (2048, 512), (1024, 1024)):
        q = torch.randn(n, rows, c, device="cuda").bfloat16().requires_grad_(True)
        kk = torch.randn(n, c, device="cuda").requires_grad_(True)
        vv = torch.randn(n, c, device="cuda").requires_grad_(True)
        nb = q.numel() * 2
        us = timed(lambda: ops.FilmSoftmaxFn.apply(q.detach(), kk.detach(), vv.detach()))
        print(f"[{rows}x{c}] film_softmax fwd {us:7.1f} us {2 * nb / us / 1e3:7.1f} GB/s", flush=True)
        h = ops.FilmSoftmaxFn.apply(q, kk, vv)
        g = torch.randn_like(h)
        us = timed(lambda: torch.autograd.grad(h, (q, kk, vv), g, retain_graph=True))
        print(f"[{rows}x{c}] film_softmax bwd {us:7.1f} us {3 * nb / us / 1e3:7.1f} GB/s (incl. partial-sum reduction)", flush=True)


if __name__ == "__main__":
    main()
