"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/fullwidth_*.npz: the reference's CondUNet1D at the FULL widths
of BASELINE.json's configurations (B: base=256 modern/add, B2: concat, C: in=out=128, n_downsample=4, 446 M
parameters), frozen as SMALL records.

A 42 M- or 446 M-parameter state dict cannot be committed, so nothing large is stored: weights and inputs are
regenerated on both sides from seeds by ``build_case`` below (constructor under ``torch.manual_seed`` -- that
unet_amd consumes the init RNG stream like the reference is pinned by tests/golden/init_checksums.npz -- plus
seeded perturbations of the zero-initialised tensors), and the record holds what the reference computed from them:
loss, output / eval-output / bottleneck / block-activation checksums and strided slices, input-gradient slice,
per-parameter gradient norms and a strided slice of every gradient, BatchNorm running-stat checksums.

Runs ONLY in the build container (imports /root/reference/models.py; nothing is copied):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_fullwidth.py [--only=B]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

B_KW = dict(in_channels=32, out_channels=32, base=256, conv_type="modern", attention_type="none", activation="gelu",
            norm_type="batch", cond_mode="cross_attn_gated", n_downsample=2)        # train.py:1915-1946 defaults
CASES = {
    # name: (ctor kwargs, batch, T, loss window)
    "B": (dict(B_KW), 2, 4096, (500, 4500)),
    "B2": (dict(B_KW, skip_type="concat"), 2, 4096, (500, 4500)),
    "C": (dict(B_KW, in_channels=128, out_channels=128, n_downsample=4), 1, 4096, (500, 4500)),
    # BASELINE.json configs[1] exactly: config B at the bench batch of 64 windows, TRAINING-mode BatchNorm over the
    # whole batch (about half an hour of reference CPU time and ~25 GB of autograd state to generate)
    "B64": (dict(B_KW), 64, 4096, (500, 4500)),
}
SLICE = 4096            # elements kept per strided slice


def _prime_at_most(n: int) -> int:
    """Largest prime <= n (n >= 2), by trial division (n is at most a few 10^5 here)."""
    while n > 2:
        if n % 2 and all(n % d for d in range(3, int(n ** 0.5) + 1, 2)):
            return n
        n -= 1
    return 2


def strided(t: torch.Tensor) -> torch.Tensor:
    """<= SLICE elements of t at a fixed stride (same rule on both sides).  The stride is the largest PRIME not
    above numel / SLICE: coprime to every tensor extent, tile size and tap count, so the samples run through all
    taps, all residues of the channel / position tiles and the tile edges instead of one lattice of them (a
    power-of-two or 3 * 2^k stride visits only tap 0 of a [1024, 1024, 3] gradient and one position residue of a
    [2, C, 4096] activation)."""
    f = t.detach().reshape(-1)
    q = f.numel() // SLICE
    step = _prime_at_most(q) if q >= 2 else 1
    return f[::step][:SLICE].clone()


def scattered(t: torch.Tensor) -> torch.Tensor:
    """SLICE elements of t at seeded random flat indices (same draw on both sides: the seed is the element count):
    a second, lattice-free sample that also reaches the tail of the tensor the strided one may stop short of."""
    f = t.detach().reshape(-1)
    if f.numel() <= SLICE:
        return f.clone()
    gen = torch.Generator().manual_seed(1000003 + f.numel() % 2147483629)
    idx = torch.randint(0, f.numel(), (SLICE,), generator=gen)
    return f[idx.to(f.device)].clone()


def put(blob: dict, key: str, t: torch.Tensor):
    blob[key + "/sums"] = sums(t)
    blob[key + "/slice"] = strided(t).cpu().numpy()
    blob[key + "/rslice"] = scattered(t).cpu().numpy()


def sums(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item(), d.pow(2).sum().sqrt().item()])


def build_case(model_cls, kw, batch, T, seed=4321):
    """Model + inputs of a full-width case.  Used with the reference's class here and with unet_amd's class in
    tests/test_gpu_fullwidth.py: every random draw comes from torch's seeded CPU generator in a fixed order."""
    torch.manual_seed(seed)
    model = model_cls(**kw)
    with torch.no_grad():
        # the reference zero-initialises outc (and BN starts at identity): perturb so that every gradient is live
        model.outc.weight.normal_(0, 0.02)
        model.outc.bias.normal_(0, 0.05)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.8, 1.2)
                m.weight.uniform_(0.8, 1.2)
                m.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(seed + 1)
    ob = torch.randn(batch, kw["in_channels"], T, generator=g) * 1.7 + 0.3
    tgt = torch.randn(batch, kw["out_channels"], T, generator=g)
    ids = torch.randint(0, 7, (batch,), generator=g)
    return model, ob, tgt, ids


def block_names(model):
    return ["inc"] + [f"encoders.{i}" for i in range(len(model.encoders))] + ["mid"] + \
           [f"decoders.{i}" for i in range(len(model.decoders))]


def main():
    sys.path.insert(0, REF)
    import models as ref  # the reference's own file

    torch.set_num_threads(8)
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for name, (kw, batch, T, (lo, hi)) in CASES.items():
        if only and name not in only:
            continue
        model, ob, tgt, ids = build_case(ref.CondUNet1D, kw, batch, T)
        ob.requires_grad_(True)
        acts, hooks = {}, []
        for lname in block_names(model):
            hooks.append(model.get_submodule(lname).register_forward_hook(
                lambda _m, _i, o, lname=lname: acts.__setitem__(lname, o.detach())))
        model.train()
        pred, bott = model(ob, ids, return_bottleneck=True)
        loss = F.l1_loss(pred[..., lo:hi], tgt[..., lo:hi])
        loss.backward()
        for h in hooks:
            h.remove()
        blob = {"loss": np.float64(loss.item()), "loss_window": np.array([lo, hi])}
        put(blob, "pred", pred)
        put(blob, "bott", bott)
        put(blob, "grad_ob", ob.grad)
        for k, v in acts.items():
            put(blob, f"act/{k}", v)
        for k, p in model.named_parameters():
            if p.grad is not None:
                put(blob, f"grad/{k}", p.grad)
        for k, v in model.state_dict().items():
            if "running_" in k:
                put(blob, f"bn/{k}", v)
        # a second, LINEAR loss <pred, G> (the bf16 gradient bar is stated under it, see tests/test_gpu_model.py):
        # G is regenerated from its seed on the other side
        model.zero_grad(set_to_none=True)
        gG = torch.Generator().manual_seed(99)
        G = torch.randn(pred.shape, generator=gG) / pred.numel()
        for m in model.modules():              # keep the BN buffers of the first step (no second update)
            if isinstance(m, torch.nn.BatchNorm1d):
                m.momentum = 0.0
        pred2 = model(ob.detach(), ids)
        (pred2 * G).sum().backward()
        lin = {}
        for k, p in model.named_parameters():
            if p.grad is not None:
                put(blob, f"lingrad/{k}", p.grad)
                if p.numel() <= 4:
                    lin[k] = p.grad.detach().clone()
        # bf16 sensitivity of the <= 4-element tensors (scale_weights): the REFERENCE itself, fp32 arithmetic, with
        # only its inputs rounded to bf16 (every weight matrix and the input window).  Whatever evaluates the model
        # in bf16 inherits at least this deviation, so the bf16 bar of such a tensor is derived from it
        # (tests/test_gpu_fullwidth.py) instead of being fitted to what the kernels happen to produce.
        saved = {k: p.detach().clone() for k, p in model.named_parameters() if p.dim() >= 2}
        with torch.no_grad():
            for k, p in model.named_parameters():
                if p.dim() >= 2:
                    p.copy_(p.bfloat16().float())
        model.zero_grad(set_to_none=True)
        pred3 = model(ob.detach().bfloat16().float(), ids)
        (pred3 * G).sum().backward()
        for k, p in model.named_parameters():
            if k in lin:
                d = (p.grad.double() - lin[k].double()).norm() / lin[k].double().norm().clamp_min(1e-300)
                blob[f"bf16in/{k}"] = np.float64(d.item())
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k in saved:
                    p.copy_(saved[k])
        del saved, pred3
        model.eval()
        with torch.no_grad():
            pe = model(ob.detach(), ids)
        put(blob, "pred_eval", pe)
        meta = dict(kw)
        meta.update(batch=batch, T=T, n_params=sum(p.numel() for p in model.parameters()))
        blob["meta"] = np.array(repr(meta))
        path = os.path.join(OUT, f"fullwidth_{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: params={meta['n_params']} loss={loss.item():.6f} -> {os.path.getsize(path)/1e3:.0f} kB", flush=True)
        del model, pred, pred2, pe, acts


if __name__ == "__main__":
    main()
