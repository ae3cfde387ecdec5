"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/model_*.npz.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference: imports the reference's own ``models.py`` (nothing is copied),
runs ``CondUNet1D`` on seeded inputs and freezes inputs, the full state_dict,
block activations, loss, parameter/input gradients and BatchNorm side effects.
The fixtures are data only; the reference never travels to the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CASES = {
    # name: (ctor kwargs, batch, T, cond: 'ids' | 'emb' | 'none', loss window)
    "A_std_attn_add_bn": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16), 2, 256, "ids", (16, 240)),
    "B_modern_add_bn_gelu": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, conv_type="modern",
                                  attention_type="none", activation="gelu"), 2, 256, "emb", (16, 240)),
    "B2_modern_concat": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, conv_type="modern",
                              attention_type="none", activation="gelu", skip_type="concat"), 2, 256, "ids", (16, 240)),
    "std_concat_instance_leaky": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, skip_type="concat",
                                       norm_type="instance", activation="leaky_relu", use_attention=False),
                                  2, 256, "ids", (16, 240)),
    "modern_group_silu_resconv_oddT": (dict(in_channels=6, out_channels=4, base=8, emb_dim=16, conv_type="modern",
                                            norm_type="group", activation="silu", attention_type="none"),
                                       2, 251, "ids", (10, 230)),
    "std_nonorm_nocond_mish_d3_oddT": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, norm_type="none",
                                            cond_mode="none", activation="mish", n_downsample=3,
                                            attention_type="none"), 2, 203, "none", (5, 200)),
    "modern_adaptive_nose": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, conv_type="modern",
                                  use_se=False, use_adaptive_scaling=True, attention_type="none",
                                  activation="gelu"), 2, 256, "emb", (16, 240)),
    "std_layernorm_noids": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, norm_type="layer",
                                 attention_type="none"), 2, 256, "none", (16, 240)),
    "B_modern_crop4096": (dict(in_channels=4, out_channels=4, base=8, emb_dim=16, conv_type="modern",
                               attention_type="none", activation="gelu"), 2, 4096, "ids", (500, 4500)),
    "std_crossfreq_v2": (dict(in_channels=8, out_channels=8, base=8, emb_dim=16, attention_type="cross_freq_v2"),
                         2, 256, "ids", (16, 240)),
    "modern_d4_cap": (dict(in_channels=8, out_channels=8, base=4, emb_dim=16, conv_type="modern",
                           attention_type="none", activation="gelu", n_downsample=4), 2, 512, "ids", (16, 496)),
}


def main():
    sys.path.insert(0, REF)
    import models as ref  # the reference's own file

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for name, (kw, B, T, cond, (lo, hi)) in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(1234)
        model = ref.CondUNet1D(**kw)
        with torch.no_grad():
            # the reference zero-initialises outc and the attention gate, which would
            # make every upstream gradient vanish -- perturb them (values are frozen below)
            model.outc.weight.normal_(0, 0.08)
            model.outc.bias.normal_(0, 0.05)
            for m in model.modules():
                if isinstance(m, ref.SelfAttention1D):
                    m.gamma.fill_(0.5)
                if isinstance(m, ref.CrossFrequencyCouplingAttentionV2):
                    m.gate.fill_(0.5)
                    m.coupling_weights.normal_(0, 0.5)
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.8, 1.2)
                    m.weight.uniform_(0.8, 1.2)
                    m.bias.normal_(0, 0.1)
            if kw.get("use_adaptive_scaling"):
                model.adaptive_scaler.predictor[-1].weight.normal_(0, 0.05)
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        cin, cout = kw["in_channels"], kw["out_channels"]
        ob = (torch.randn(B, cin, T) * 1.7 + 0.3).requires_grad_(True)
        tgt = torch.randn(B, cout, T)
        ids = torch.randint(0, 7, (B,))
        emb = torch.randn(B, kw["emb_dim"])
        fkw = {}
        if cond == "ids":
            fkw["odor_ids"] = ids
        elif cond == "emb":
            fkw["cond_emb"] = emb

        acts = {}
        hooks = []
        for lname, mod in [("inc", model.inc), ("mid", model.mid), ("outc", model.outc)] + \
                [(f"encoders.{i}", m) for i, m in enumerate(model.encoders)] + \
                [(f"decoders.{i}", m) for i, m in enumerate(model.decoders)]:
            hooks.append(mod.register_forward_hook(
                lambda _m, _i, o, lname=lname: acts.__setitem__(lname, o.detach().clone())))
        model.train()
        pred, bott = model(ob, return_bottleneck=True, **fkw)
        loss = F.l1_loss(pred[..., lo:hi], tgt[..., lo:hi])
        loss.backward()
        for h in hooks:
            h.remove()
        sd1 = model.state_dict()
        model.eval()
        with torch.no_grad():
            pred_eval = model(ob.detach(), **fkw)

        blob = {
            "ob": ob.detach().numpy(), "target": tgt.numpy(), "odor_ids": ids.numpy(),
            "cond_emb": emb.numpy(), "pred": pred.detach().numpy(), "bottleneck": bott.detach().numpy(),
            "pred_eval": pred_eval.numpy(), "loss": np.float64(loss.item()),
            "loss_window": np.array([lo, hi]), "grad_ob": ob.grad.numpy(),
        }
        for k, v in sd0.items():
            blob["sd/" + k] = v.numpy()
        for k, v in sd1.items():
            if "running_" in k or "num_batches" in k:
                blob["sd_after/" + k] = v.detach().numpy()
        for k, p in model.named_parameters():
            blob["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
            blob["hasgrad/" + k] = np.bool_(p.grad is not None)
        for k, v in acts.items():
            blob["act/" + k] = v.numpy()
        meta = dict(kw)
        meta.update(batch=B, T=T, cond=cond)
        blob["meta"] = np.array(repr(meta))
        path = os.path.join(OUT, f"model_{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: loss={loss.item():.6f} keys={len(sd0)} -> {os.path.getsize(path)/1e3:.0f} kB")


if __name__ == "__main__" and "--init-only" not in sys.argv:
    main()


def init_checksums():
    """Per-tensor (sum, abs-sum) of the reference's freshly constructed models at seed 7 -- pins that
    unet_amd consumes the init RNG stream in the same order (same seed -> same initial weights)."""
    sys.path.insert(0, REF)
    import models as ref
    blob = {}
    for tag, kw in (("A", {}), ("B", dict(base=32, conv_type="modern", attention_type="none", activation="gelu")),
                    ("B2", dict(base=32, conv_type="modern", skip_type="concat", use_adaptive_scaling=True,
                                in_channels=16, out_channels=8))):
        torch.manual_seed(7)
        m = ref.CondUNet1D(**kw)
        for k, v in m.state_dict().items():
            if v.is_floating_point():
                blob[f"{tag}/{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    np.savez_compressed(os.path.join(OUT, "init_checksums.npz"), **blob)
    print("init_checksums:", len(blob), "tensors")


if __name__ == "__main__":
    init_checksums()
