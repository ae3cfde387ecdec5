"""Band-pair attention kernel against an fp32 torch evaluation of the same expression (bf16-rounded inputs).

    python scripts/check_attn.py            # a ladder of shapes, forward (+ backward once it exists), then timing
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from unet_amd import ops


def reference(q, k, v, coupling, nh, scale):
    nb, n, l, c = q.shape
    hd = c // nh
    sp = lambda t: t.float().view(nb, n, l, nh, hd).permute(0, 1, 3, 2, 4)      # [nb, N, nh, L, hd]
    qf, kf, vf = sp(q), sp(k), sp(v)
    outs = []
    for i in range(nb):
        acc = 0
        for j in range(nb):
            p = torch.softmax(qf[i] @ kf[j].transpose(-1, -2) * scale, dim=-1)
            acc = acc + coupling[i, j] * (p @ vf[j])
        outs.append(acc.permute(0, 2, 1, 3).reshape(n, l, c))
    return torch.cat(outs, dim=2)


def main():
    torch.manual_seed(0)
    dev = "cuda"
    shapes = [(2, 64, 32, 4, 5), (1, 200, 64, 2, 3), (2, 130, 256, 4, 5), (1, 257, 512, 4, 2), (1, 300, 1024, 4, 5),
              (2, 100, 96, 4, 5)]
    if "--time-only" in sys.argv:
        shapes = []
    for n, l, c, nh, nb in shapes:
        q, k, v = (torch.randn(nb, n, l, c, device=dev).mul_(1.5).bfloat16() for _ in range(3))
        coupling = torch.softmax(torch.randn(nb, nb, device=dev), dim=-1)
        scale = 1.0 / math.sqrt(c // nh)
        out, o_pairs, lse = ops.cfc_attn_forward_raw(q, k, v, coupling, nh, scale, keep=True)
        torch.cuda.synchronize()
        ref = reference(q, k, v, coupling, nh, scale)
        err = (out.float() - ref).abs().max().item()
        print(f"fwd N={n} L={l} C={c} heads={nh} bands={nb}: max abs err {err:.3e} (ref max {ref.abs().max().item():.2f})",
              flush=True)
        assert err < 3e-2, err
        # backward against autograd through the fp32 expression
        qr, kr, vr, cr = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
        g = torch.randn_like(ref).bfloat16()
        reference(qr, kr, vr, cr, nh, scale).backward(g.float())
        qh, kh, vh, ch = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
        ops.cfc_attention(qh, kh, vh, ch, nh, scale).backward(g)
        torch.cuda.synchronize()
        for name, a, b in (("dq", qh.grad, qr.grad), ("dk", kh.grad, kr.grad), ("dv", vh.grad, vr.grad),
                           ("dcoupling", ch.grad, cr.grad)):
            rel = ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
            mx = (a.float() - b.float()).abs().max().item()
            print(f"    {name}: rel l2 err {rel:.3e}, max abs {mx:.3e} (ref max {b.abs().max().item():.3f})", flush=True)
            assert rel < 2e-2, (name, rel)
    # timing at the config-B bottleneck shape
    n, l, c, nh, nb = (4 if "--time-only" in sys.argv else 16), 1024, 1024, 4, 5
    q, k, v = (torch.randn(nb, n, l, c, device=dev).bfloat16() for _ in range(3))
    coupling = torch.softmax(torch.randn(nb, nb, device=dev), dim=-1)
    for keep in (False, True):
        for _ in range(2):
            ops.cfc_attn_forward_raw(q, k, v, coupling, nh, 1 / 16.0, keep=keep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.cfc_attn_forward_raw(q, k, v, coupling, nh, 1 / 16.0, keep=keep)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flops = nb * nb * n * nh * 4.0 * l * l * (c // nh)
        print(f"fwd keep={keep} N={n}: {ms:.2f} ms, {flops / ms / 1e9:.0f} TFLOP/s", flush=True)
    qh, kh, vh, ch = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
    g = torch.randn(n, l, nb * c, device=dev).bfloat16()
    for rep in range(3):
        out = ops.cfc_attention(qh, kh, vh, ch, nh, 1 / 16.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out.backward(g)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"bwd N={n}: {ms:.2f} ms, {3.5 * flops / ms / 1e9:.0f} TFLOP/s (7 products)", flush=True)


if __name__ == "__main__":
    main()
