"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, the module keeps the reference's state_dict layout, geometry helpers are exact, and the
product refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

import unet_amd
from unet_amd import _lib, ops
from tests._golden import Golden, model_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "unet_hip.h")).read()
    declared = set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed from include/unet_hip.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in unet_hip.h but not exported"
    # and the Python binding table covers exactly the header
    assert declared == set(_lib.declared_symbols())
    lib.unet_abi_version.restype = ctypes.c_int
    assert lib.unet_abi_version() == 1


def test_error_convention_without_gpu():
    lib = _lib.load()
    # null descriptor: must fail with a message, not crash (validation happens before any launch)
    rc = lib.unet_conv1d_fwd(None, None, None, None, None, None, None, None)
    assert rc != 0
    assert b"null" in lib.unet_last_error()
    d = _lib.ConvDesc()
    d.dtype, d.n, d.l_in, d.c1, d.l_out, d.c_out, d.x1_ld, d.y_ld = 7, 1, 8, 8, 8, 8, 8, 8
    d.stride, d.up, d.n_taps = 1, 1, 1
    rc = lib.unet_conv1d_fwd(ctypes.byref(d), 16, None, 16, None, None, 16, None)
    assert rc != 0 and b"dtype" in lib.unet_last_error()
    d.dtype, d.stride = 0, 3
    rc = lib.unet_conv1d_fwd(ctypes.byref(d), 16, None, 16, None, None, 16, None)
    assert rc != 0 and b"stride" in lib.unet_last_error()


@pytest.mark.parametrize("name", model_cases())
def test_state_dict_layout_matches_reference(name):
    g = Golden(name)
    model = unet_amd.CondUNet1D(**g.cfg)
    ref = g.state_dict()
    mine = model.state_dict()
    assert list(mine.keys()) == list(ref.keys())  # same names in the same order
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
        assert mine[k].dtype == ref[k].dtype, k
    model.load_state_dict(ref, strict=True)


def test_constructor_defaults_and_errors():
    m = unet_amd.CondUNet1D()
    assert sum(p.numel() for p in m.parameters()) == 4_674_465
    assert len(m.state_dict()) == 103
    b = unet_amd.CondUNet1D(base=256, conv_type="modern", attention_type="none", activation="gelu")
    assert sum(p.numel() for p in b.parameters()) == 41_966_772     # config B
    assert float(m.outc.weight.abs().max()) == 0.0 and float(m.outc.bias.abs().max()) == 0.0
    for kw, msg in [(dict(norm_type="bogus"), "Unknown norm_type"), (dict(skip_type="bogus"), "Unknown skip_type"),
                    (dict(attention_type="bogus"), "Unknown attention_type"), (dict(cond_mode="bogus"), "Unknown cond_mode"),
                    (dict(activation="bogus"), "Unknown activation")]:
        with pytest.raises(ValueError, match=msg):
            unet_amd.CondUNet1D(**kw)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.randn(1, 32, 64))


def test_tap_geometry():
    g, pairs = ops.merged_dilated_taps(7, (1, 4, 16, 32))
    assert g.n_taps == 23 and len(pairs) == 28           # 28 taps, 23 distinct offsets
    assert set(g.offs) == {0, 1, 2, 3, 4, 8, 12, 16, 32, 48, 64, 96, -1, -2, -3, -4, -8, -12, -16, -32, -48, -64, -96}
    assert abs(g.alg_scale - 28 / 23) < 1e-12
    # stride-2 data gradient: taps split by parity into the two output phases
    g2 = ops.ConvGeom(g.offs, g.phases, 2, 1, g.alg_scale)
    gd, src = g2.dgrad()
    assert gd.up == 2 and gd.stride == 1 and sorted(src) == list(range(23))
    assert sum(1 for r in gd.phases if r == 1) == 4 and sum(1 for r in gd.phases if r == 0) == 19
    for o, r, j in zip(gd.offs, gd.phases, src):
        assert (r - g.offs[j]) % 2 == 0 and o == (r - g.offs[j]) // 2
    # transposed conv: o = 2i - 1 + k  <=>  dx[i] reads dy[2i + (k - 1)]
    gt, kidx = ops.geom_convt_k4s2p1()
    gtd, src = gt.dgrad()
    assert gtd.stride == 2 and gtd.up == 1
    assert sorted((o, kidx[j]) for o, j in zip(gtd.offs, src)) == [(-1, 0), (0, 1), (1, 2), (2, 3)]
    for l, k, s, d in [(4096, 7, 2, 32), (251, 3, 2, 1), (100, 7, 1, 16)]:
        ref = torch.nn.functional.conv1d(torch.zeros(1, 1, l), torch.zeros(1, 1, k), stride=s, padding=(k - 1) * d // 2,
                                         dilation=d).shape[-1]
        assert ops.conv_out_len(l, k, s, d) == ref


def test_same_seed_same_initial_weights_as_reference():
    """Parameter creation order mirrors models.py, so the init RNG stream is consumed identically;
    tests/golden/init_checksums.npz holds the reference's per-tensor sums at seed 7."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "init_checksums.npz")
    z = np.load(path, allow_pickle=False)
    for tag, kw in (("A", {}), ("B", dict(base=32, conv_type="modern", attention_type="none", activation="gelu")),
                    ("B2", dict(base=32, conv_type="modern", skip_type="concat", use_adaptive_scaling=True,
                                in_channels=16, out_channels=8))):
        torch.manual_seed(7)
        m = unet_amd.CondUNet1D(**kw)
        for k, v in m.state_dict().items():
            if v.is_floating_point():
                ref = z[f"{tag}/{k}"]
                got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
                assert np.allclose(got, ref, rtol=1e-12, atol=1e-12), (tag, k)


@pytest.mark.parametrize("name", ["A_std_attn_add_bn", "B_modern_add_bn_gelu", "std_crossfreq_v2"])
def test_introspection_names_and_norms_match_reference_on_cpu(name):
    """Row a16 without a GPU: the parameter-only part of the reference's introspection surface (models.py:3381-3627)
    against outputs of the reference's own methods (oracle/make_golden_introspect.py)."""
    import json
    import numpy as np
    import unet_amd
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gdir, f"introspect_{name}.npz"), allow_pickle=False)
    with open(os.path.join(gdir, f"introspect_{name}.json")) as f:
        info = json.load(f)
    model = unet_amd.CondUNet1D(**info["cfg"])
    model.load_state_dict({k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("sd/")}, strict=True)
    assert model.get_layer_names() == info["layer_names"]
    assert model.get_layer_parameter_counts() == info["param_counts"]
    assert model.get_architecture_summary() == info["architecture_summary"]
    assert [type(m).__name__ for m in model.get_encoder_layers()] == info["encoder_layers"]
    assert [type(m).__name__ for m in model.get_decoder_layers()] == info["decoder_layers"]
    att = model.get_attention_module()
    assert (None if att is None else type(att).__name__) == info["attention_module"]
    wn = model.get_weight_norms()
    assert sorted(wn) == sorted(info["weight_norms"])
    for k, v in info["weight_norms"].items():
        assert abs(wn[k] - v) <= 1e-6 * abs(v), k
    assert model.get_gradient_norms() == {}          # no backward yet
