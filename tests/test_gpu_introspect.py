"""Rows a15 / a16 (GPU): the reference's introspection and analysis methods of CondUNet1D (models.py:3381-3703) and
its gradient-checkpointed training step (models.py:3086-3092, 3322-3326, 3349-3353), against outputs of the
reference's own methods frozen by oracle/make_golden_introspect.py.  fp32 bars: activations <= 1e-4, gradients and
gradient / weight norms <= 1e-3 relative."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["A_std_attn_add_bn", "B_modern_add_bn_gelu", "std_crossfreq_v2"]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"introspect_{name}.npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN, f"introspect_{name}.json")) as f:
        info = json.load(f)
    return z, info


def _group(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith(prefix)}


def _model(z, info):
    import unet_amd
    model = unet_amd.CondUNet1D(**info["cfg"]).cuda()
    model.load_state_dict(_group(z, "sd/"), strict=True)
    return model


@pytest.mark.parametrize("name", NAMES)
def test_introspection_surface_matches_reference(name):
    z, info = _load(name)
    model = _model(z, info)
    ob, tgt, ids = (torch.from_numpy(z[k]).cuda() for k in ("ob", "target", "odor_ids"))
    lo, hi = (int(v) for v in z["loss_window"])
    assert model.get_layer_names() == info["layer_names"]
    assert [type(m).__name__ for m in model.get_encoder_layers()] == info["encoder_layers"]
    assert [type(m).__name__ for m in model.get_decoder_layers()] == info["decoder_layers"]
    att = model.get_attention_module()
    assert (None if att is None else type(att).__name__) == info["attention_module"]
    assert model.get_layer_parameter_counts() == info["param_counts"]
    summ = model.get_architecture_summary()
    assert summ == info["architecture_summary"]
    wn = model.get_weight_norms()
    assert list(wn) == list(_ordered(info["weight_norms"], model))
    for k, v in info["weight_norms"].items():
        assert abs(wn[k] - v) <= 1e-5 * abs(v), k

    # training step with the recorder hooks: names, order, tensors (outc = delta before the residual)
    model.train()
    seen, acts = [], {}

    def record(n, a):          # returns None: a hook's return value would replace the block's output
        seen.append(n)
        acts[n] = a.detach().clone()

    handles = model.register_activation_hooks(record)
    pred = model(ob, ids)
    for h in handles:
        h.remove()
    assert seen == info["hook_order"]
    for k, v in _group(z, "hook/").items():
        assert acts[k].shape == v.shape, k
        assert rel(acts[k], v) < 1e-4, k
    assert rel(pred, torch.from_numpy(z["pred_plain"])) < 1e-4
    loss = F.l1_loss(pred[..., lo:hi], tgt[..., lo:hi])
    assert abs(loss.item() - info["loss_plain"]) < 1e-5
    loss.backward()
    gn = model.get_gradient_norms()
    assert sorted(gn) == sorted(info["gradient_norms"])
    for k, v in info["gradient_norms"].items():
        # 1e-3 relative; mid_2 of the cross-frequency fixture is a 1e-8 gradient (gate-scaled): absolute floor
        assert abs(gn[k] - v) <= 1e-3 * abs(v) + 1e-9, (k, gn[k], v)

    # eval-mode helpers on the buffers after that one step
    model.eval()
    ext = model.extract_all_activations(ob, odor_ids=ids)
    assert list(ext) == info["extract_keys"]
    for k, v in _group(z, "extract/").items():
        assert ext[k].shape == v.shape, k
        assert rel(ext[k], v) < 1e-4, k
    assert rel(model.get_bottleneck_features(ob, odor_ids=ids, pool=True), torch.from_numpy(z["bottleneck_pooled"])) < 1e-4
    bt = model.get_bottleneck_features(ob, odor_ids=ids, pool=False)
    assert bt.shape == z["bottleneck_temporal"].shape
    assert rel(bt, torch.from_numpy(z["bottleneck_temporal"])) < 1e-4
    assert not model.outc._forward_hooks and not model.inc._forward_hooks     # helpers removed their hooks


def _ordered(d, model):
    """json sorts keys; the method's own order is inc, encoder_i, mid_j, decoder_i, outc."""
    order = ["inc"] + [f"encoder_{i}" for i in range(model.n_downsample)] + [f"mid_{j}" for j in range(len(model.mid))] + \
            [f"decoder_{i}" for i in range(model.n_downsample)] + ["outc"]
    return [k for k in order if k in d]


@pytest.mark.parametrize("name", NAMES)
def test_gradient_checkpointing_matches_reference(name):
    """set_gradient_checkpointing(True): same loss and gradients as the plain step, and the reference's buffer side
    effect (blocks are recomputed in backward, so BatchNorm running statistics take the update twice and
    num_batches_tracked advances by 2 -- frozen from the reference's own checkpointed step)."""
    z, info = _load(name)
    model = _model(z, info)
    ob, tgt, ids = (torch.from_numpy(z[k]).cuda() for k in ("ob", "target", "odor_ids"))
    lo, hi = (int(v) for v in z["loss_window"])
    model.train()
    model.set_gradient_checkpointing(True)
    assert model.gradient_checkpointing is True
    pred = model(ob, ids)
    loss = F.l1_loss(pred[..., lo:hi], tgt[..., lo:hi])
    assert abs(loss.item() - info["loss_ckpt"]) < 1e-5
    loss.backward()
    params = dict(model.named_parameters())
    plain = _group(z, "grad_plain/")
    for k, gr in _group(z, "grad_ckpt/").items():
        assert params[k].grad is not None, k
        if gr.abs().max().item() < 1e-7:
            assert params[k].grad.abs().max().item() < 1e-6, k
            continue
        assert rel(params[k].grad, gr) < 1e-3, k
        assert rel(params[k].grad, plain[k]) < 1e-3, k          # ... which are the gradients of the plain step
    sd = model.state_dict()
    for k, v in _group(z, "sd_ckpt_after/").items():
        if v.is_floating_point():
            assert rel(sd[k], v) < 1e-5, k
        else:
            assert int(sd[k]) == int(v), k
    # eval mode ignores the flag (models.py:3322: "and self.training")
    model.eval()
    with torch.no_grad():
        model(ob, ids)


def test_gradient_checkpointing_bf16_same_as_plain():
    """bf16 kernels: the checkpointed step recomputes the same kernels on the same inputs, so loss and gradients
    equal the plain step's bit for bit -- except the embedding's gradient: the plain step sums the FiLM projections'
    input gradients of all blocks in one launch, the checkpointed step block by block (fp32 summation order)."""
    z, info = _load("B_modern_add_bn_gelu")
    ob, tgt, ids = (torch.from_numpy(z[k]).cuda() for k in ("ob", "target", "odor_ids"))
    grads = []
    for ckpt in (False, True):
        model = _model(z, info)
        model.set_compute_dtype(torch.bfloat16)
        model.train()
        model.set_gradient_checkpointing(ckpt)
        loss = F.l1_loss(model(ob, ids), tgt)
        loss.backward()
        grads.append((loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert grads[0][0] == grads[1][0]
    for k, g in grads[0][1].items():
        if k == "embed.weight":
            assert rel(g, grads[1][1][k]) < 1e-6, k
            continue
        assert torch.equal(g, grads[1][1][k]), k
