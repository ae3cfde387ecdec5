"""The oracle (oracle/condunet_oracle.py) against fixtures frozen from the
reference's own models.py (tests/golden, made by oracle/make_golden.py).
fp32 CPU vs fp32 CPU with the same torch build: tolerances are tight."""
import pytest
import torch
import torch.nn.functional as F

from oracle import condunet_oracle as O
from tests._golden import Golden, model_cases


def _close(a, b, rtol=2e-5, atol=2e-6):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", model_cases())
def test_forward_backward_matches_reference(name):
    g = Golden(name)
    sd = g.state_dict()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()
              and "running_" not in k}
    run_sd = dict(sd)
    run_sd.update(leaves)
    ob = g.t("ob").requires_grad_(True)
    bn = O.BNState()
    taps = {}
    pred, bott = O.forward(run_sd, g.cfg, ob, training=True, bn=bn, taps=taps, return_bottleneck=True,
                           **g.cond_kwargs())
    _close(pred, g.t("pred"))
    _close(bott, g.t("bottleneck"))
    for k, v in g.group("act/").items():
        if k == "outc":
            continue
        _close(taps[k], v)
    loss = F.l1_loss(pred[..., g.lo:g.hi], g.t("target")[..., g.lo:g.hi])
    assert abs(loss.item() - float(g.z["loss"])) < 1e-6
    loss.backward()
    _close(ob.grad, g.t("grad_ob"), rtol=1e-4, atol=1e-7)
    grads = g.group("grad/")
    for k, gr in grads.items():
        has = bool(g.z["hasgrad/" + k])
        mine = leaves[k].grad
        if not has:
            assert mine is None or float(mine.abs().max()) == 0.0, k
            continue
        assert mine is not None, k
        _close(mine, gr, rtol=2e-4, atol=2e-7)
    # BatchNorm side effects
    for k, v in g.group("sd_after/").items():
        assert k in bn.updates, k
        _close(bn.updates[k].to(v.dtype), v)


@pytest.mark.parametrize("name", model_cases())
def test_eval_mode_matches_reference(name):
    g = Golden(name)
    sd = g.state_dict()
    sd.update(g.group("sd_after/"))
    with torch.no_grad():
        pred = O.forward(sd, g.cfg, g.t("ob"), training=False, **g.cond_kwargs())
    _close(pred, g.t("pred_eval"))


def test_zero_init_outc_is_identity_on_zscore():
    # models.py:3053-3054: outc zero-init => output == z-scored input
    cfg = dict(in_channels=4, out_channels=4, base=8, emb_dim=16, attention_type="none")
    sd = O.init_state_dict(cfg, seed=3)
    sd["outc.weight"].zero_()
    sd["outc.bias"].zero_()
    ob = torch.randn(2, 4, 128) * 3 + 1
    out = O.forward(sd, cfg, ob, odor_ids=torch.zeros(2, dtype=torch.long))
    assert torch.equal(out, O.zscore(ob))


def test_crop_and_target_norm():
    x = torch.arange(2 * 3 * 5000, dtype=torch.float32).reshape(2, 3, 5000)
    assert O.crop_to_target(x).shape[-1] == 4000
    assert O.crop_to_target(x[..., :4096]).shape[-1] == 3596
    y = O.per_channel_normalize(torch.randn(2, 3, 100))
    assert torch.allclose(y.mean(-1), torch.zeros(2, 3), atol=1e-6)
    assert torch.allclose(y.std(-1), torch.ones(2, 3), atol=1e-5)
