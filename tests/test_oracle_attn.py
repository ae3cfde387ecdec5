"""CPU: the oracle's restatement of CrossFrequencyCouplingAttentionV2 against outputs frozen from the reference
module (tests/golden/attn_cross_freq_v2.npz, made by oracle/make_golden_attn.py): output, input gradient and
every parameter gradient, fp32 <= 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "attn_cross_freq_v2.npz")
CASES = ["c64_h2_l160", "c128_h2_l200"]


GOLD_V1 = os.path.join(os.path.dirname(__file__), "golden", "attn_cross_freq_v1.npz")
CASES_V1 = ["c64_h2_l512_cs256", "c64_h2_l300_cs128"]     # chunk divides the length / ragged last chunk


def load_case(name, path=GOLD):
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[f"{name}/{k}"])
    sd = {k[len(name) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/sd/")}
    grads = {k[len(name) + 6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/grad/")}
    cfg = [int(v) for v in z[f"{name}/cfg"]]
    c, nh, l, n = cfg[:4]
    return dict(c=c, nh=nh, l=l, n=n, cs=cfg[4] if len(cfg) > 4 else None, x=t("x"), gy=t("gy"), y=t("y"), dx=t("dx"),
                sd=sd, grads=grads)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def above_floor(grads):
    """proj_k.bias shifts every score of a row by the same amount, so its true gradient is exactly zero and the
    reference holds rounding noise there: compare only tensors above 1e-5 of the largest gradient norm."""
    floor = 1e-5 * max(v.double().norm().item() for v in grads.values())
    return {k: v for k, v in grads.items() if v.double().norm().item() >= floor}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_module(name):
    from oracle import condunet_oracle as O
    g = load_case(name)
    sd = {"a." + k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    y = O.cross_freq_v2(sd, "a.", x, n_heads=g["nh"])
    assert rel(y.detach(), g["y"]) < 1e-5
    (y * g["gy"]).sum().backward()
    assert rel(x.grad, g["dx"]) < 1e-4
    assert "proj_k.bias" not in above_floor(g["grads"])
    for k, ref in above_floor(g["grads"]).items():
        assert rel(sd["a." + k].grad, ref) < 1e-4, k


@pytest.mark.parametrize("name", CASES_V1)
def test_oracle_matches_reference_v1_module(name):
    """The chunked V1 module (models.py:100-224, attention_type="cross_freq")."""
    from oracle import condunet_oracle as O
    g = load_case(name, GOLD_V1)
    sd = {"a." + k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    y = O.cross_freq_v1(sd, "a.", x, n_heads=g["nh"], chunk_size=g["cs"])
    assert rel(y.detach(), g["y"]) < 1e-5
    (y * g["gy"]).sum().backward()
    assert rel(x.grad, g["dx"]) < 1e-4
    for k, ref in above_floor(g["grads"]).items():
        assert rel(sd["a." + k].grad, ref) < 1e-4, k
