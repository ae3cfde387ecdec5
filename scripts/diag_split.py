"""Diagnostic (GPU): which block's eval-mode output depends on the batch size?  Runs config B in bf16 on a 64-window
batch and on sub-batches of it and prints, per hooked block, whether the shared windows agree bit for bit."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import unet_amd
from oracle.make_golden_fullwidth import CASES, block_names, build_case


def main():
    kw, batch, T, _ = CASES["B"]
    model, ob, tgt, ids = build_case(unet_amd.CondUNet1D, kw, batch, T)
    model = model.cuda()
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 32, 4096, generator=g).cuda()
    id64 = torch.randint(0, 7, (64,), generator=g).cuda()

    def run(xs, idx):
        acts, hooks = {}, []
        for lname in block_names(model) + ["outc"]:
            hooks.append(model.get_submodule(lname).register_forward_hook(
                lambda _m, _i, o, lname=lname: acts.__setitem__(lname, o.detach().float().clone())))
        with torch.no_grad():
            out = model(xs, idx)
        for h in hooks:
            h.remove()
        acts["out"] = out.float()
        return acts

    a64 = run(x, id64)
    for lo, hi in ((0, 2), (8, 16), (0, 1), (32, 64)):
        a = run(x[lo:hi].contiguous(), id64[lo:hi].contiguous())
        for k in a:
            ref = a64[k][lo:hi]
            d = (a[k] - ref).norm() / ref.norm()
            print(f"[{lo}:{hi}] {k:12s} equal={torch.equal(a[k], ref)} rel={d.item():.3e}")


if __name__ == "__main__":
    main()
