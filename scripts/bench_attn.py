"""Forward+backward time of the cross-frequency attention at the config-B bottleneck shape ([N, 1024, 1024]).

    python scripts/bench_attn.py [--n 16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import unet_amd
from unet_amd.model import CrossFrequencyCouplingAttentionV2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--c", type=int, default=1024)
    ap.add_argument("--l", type=int, default=1024)
    args = ap.parse_args()
    m = CrossFrequencyCouplingAttentionV2(args.c).cuda()
    with torch.no_grad():
        m.gate.fill_(0.5)
    x = torch.randn(args.n, args.l, args.c, device="cuda").bfloat16().requires_grad_(True)
    for _ in range(2):
        m.forward_nlc(x).float().square().mean().backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    y = m.forward_nlc(x)
    e[1].record()
    y.float().square().mean().backward()
    e[2].record()
    torch.cuda.synchronize()
    hd = args.c // 4
    flops = 25 * args.n * 4 * (4.0 * args.l * args.l * hd) + 2.0 * args.n * args.l * args.c * args.c * 20
    print(f"cross_freq_v2 N={args.n} C={args.c} L={args.l}: fwd {e[0].elapsed_time(e[1]):.2f} ms "
          f"({flops / e[0].elapsed_time(e[1]) / 1e9:.0f} TFLOP/s), bwd {e[1].elapsed_time(e[2]):.2f} ms")


if __name__ == "__main__":
    main()
