"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/introspect_*.npz (+ .json): outputs of the reference's own
introspection / analysis methods of CondUNet1D (models.py:3086-3092, 3322-3326, 3349-3353, 3381-3703) and of its
gradient-checkpointed training step, frozen for three small configurations.

Runs ONLY in the build container (imports /root/reference/models.py; nothing is copied):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_introspect.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CASES, OUT, REF  # noqa: E402  (constructor keywords of the model fixtures)

NAMES = ["A_std_attn_add_bn", "B_modern_add_bn_gelu", "std_crossfreq_v2"]


def perturb(model, ref):
    with torch.no_grad():
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


def main():
    sys.path.insert(0, REF)
    import models as ref  # the reference's own file

    torch.set_num_threads(4)
    for name in NAMES:
        kw, B, T, cond, (lo, hi) = CASES[name]
        torch.manual_seed(4242)
        model = ref.CondUNet1D(**kw)
        perturb(model, ref)
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ob = torch.randn(B, kw["in_channels"], T) * 1.7 + 0.3
        tgt = torch.randn(B, kw["out_channels"], T)
        ids = torch.randint(0, 7, (B,))
        blob = {"ob": ob.numpy(), "target": tgt.numpy(), "odor_ids": ids.numpy(), "loss_window": np.array([lo, hi])}
        for k, v in sd0.items():
            blob["sd/" + k] = v.numpy()
        info = {"cfg": kw, "batch": B, "T": T}

        # ---- plain training step: hooks through register_activation_hooks, then the norm dictionaries ----------
        model.train()
        seen = []
        acts = {}

        def cb(lname, act):
            seen.append(lname)
            acts[lname] = act.detach().clone()

        handles = model.register_activation_hooks(cb)
        pred = model(ob, ids)
        for h in handles:
            h.remove()
        loss = F.l1_loss(pred[..., lo:hi], tgt[..., lo:hi])
        loss.backward()
        info["hook_order"] = seen
        for k, v in acts.items():
            blob["hook/" + k] = v.numpy()
        info["layer_names"] = model.get_layer_names()
        info["gradient_norms"] = model.get_gradient_norms()
        info["weight_norms"] = model.get_weight_norms()
        info["param_counts"] = model.get_layer_parameter_counts()
        summ = model.get_architecture_summary()
        info["architecture_summary"] = {k: v for k, v in summ.items()}
        info["encoder_layers"] = [type(m).__name__ for m in model.get_encoder_layers()]
        info["decoder_layers"] = [type(m).__name__ for m in model.get_decoder_layers()]
        att = model.get_attention_module()
        info["attention_module"] = None if att is None else type(att).__name__
        info["loss_plain"] = loss.item()
        blob["pred_plain"] = pred.detach().numpy()
        for k, p in model.named_parameters():
            if p.grad is not None:
                blob["grad_plain/" + k] = p.grad.numpy().copy()

        # ---- eval-mode analysis helpers (on the buffers after that one step) -------------------------------------
        model.eval()
        ext = model.extract_all_activations(ob, odor_ids=ids)
        info["extract_keys"] = list(ext.keys())
        for k, v in ext.items():
            blob["extract/" + k] = v.numpy()
        blob["bottleneck_pooled"] = model.get_bottleneck_features(ob, odor_ids=ids, pool=True).numpy()
        blob["bottleneck_temporal"] = model.get_bottleneck_features(ob, odor_ids=ids, pool=False).numpy()
        for k, v in model.state_dict().items():
            if "running_" in k or "num_batches" in k:
                blob["sd_plain_after/" + k] = v.detach().numpy().copy()

        # ---- the same step with gradient checkpointing, from the same initial state -----------------------------
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        model.train()
        model.set_gradient_checkpointing(True)
        pred = model(ob, ids)
        loss = F.l1_loss(pred[..., lo:hi], tgt[..., lo:hi])
        loss.backward()
        info["loss_ckpt"] = loss.item()
        for k, p in model.named_parameters():
            if p.grad is not None:
                blob["grad_ckpt/" + k] = p.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running_" in k or "num_batches" in k:
                blob["sd_ckpt_after/" + k] = v.detach().numpy().copy()

        np.savez_compressed(os.path.join(OUT, f"introspect_{name}.npz"), **blob)
        with open(os.path.join(OUT, f"introspect_{name}.json"), "w") as f:
            json.dump(info, f, indent=1, sort_keys=True, default=lambda o: repr(o))
        print(name, "loss", info["loss_plain"], info["loss_ckpt"], "grad-norm keys", list(info["gradient_norms"]))


if __name__ == "__main__":
    main()
