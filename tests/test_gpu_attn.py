"""GPU: the band-pair attention kernels (csrc/attn.hip, through the C ABI) against
  * an fp32 torch evaluation of the same expression on the same bf16 inputs (forward, dq, dk, dv, d coupling) over
    head sizes 8..256, ragged lengths and band counts;
  * outputs frozen from the reference's CrossFrequencyCouplingAttentionV2 (tests/golden/attn_cross_freq_v2.npz):
    the whole module in fp32 (torch products, <= 1e-4 / 1e-3) and in bf16 (HIP kernels, bars below).
bf16 bars: the kernel rounds P and dS to bf16 before the second product, as every flash kernel does: forward
<= 3e-2 absolute on O(1) outputs, gradients <= 2e-2 relative L2 against the fp32 expression."""
import math

import os

import pytest
import torch

from tests.test_oracle_attn import CASES, CASES_V1, GOLD_V1, above_floor, load_case, rel

pytestmark = pytest.mark.gpu


def expression(q, k, v, coupling, nh, scale):
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


SHAPES = [(2, 64, 32, 4, 5), (1, 200, 64, 2, 3), (2, 130, 256, 4, 5), (1, 257, 512, 4, 2), (1, 300, 1024, 4, 5),
          (2, 100, 96, 4, 5), (3, 1, 64, 2, 2), (1, 65, 48, 2, 1)]


@pytest.mark.parametrize("n,l,c,nh,nb", SHAPES)
def test_kernels_match_fp32_expression(n, l, c, nh, nb):
    from unet_amd import ops
    torch.manual_seed(l * 7 + c)
    q, k, v = (torch.randn(nb, n, l, c, device="cuda").mul_(1.5).bfloat16() for _ in range(3))
    coupling = torch.softmax(torch.randn(nb, nb, device="cuda"), dim=-1)
    scale = 1.0 / math.sqrt(c // nh)
    qr, kr, vr, cr = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
    ref = expression(qr, kr, vr, cr, nh, scale)
    g = torch.randn_like(ref).bfloat16()
    ref.backward(g.float())
    qh, kh, vh, ch = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
    out = ops.cfc_attention(qh, kh, vh, ch, nh, scale)
    assert (out.float() - ref).abs().max().item() < 3e-2
    out.backward(g)
    for name, a, b in (("dq", qh.grad, qr.grad), ("dk", kh.grad, kr.grad), ("dv", vh.grad, vr.grad),
                       ("dcoupling", ch.grad, cr.grad)):
        a, b = a.double().cpu(), b.double().cpu()      # (+1e-4: with one key, dq and dk are exactly zero)
        assert (a - b).norm().item() <= 2e-2 * b.norm().item() + 1e-4, name
    # inference (nothing kept for backward) writes the same output bits
    with torch.no_grad():
        out2 = ops.cfc_attention(q, k, v, coupling, nh, scale)
    assert torch.equal(out2, out.detach())
    # C callers without a buffer for the pair outputs: summed in registers, same result up to bf16 rounding
    out3 = torch.empty_like(out2)
    from unet_amd import _lib
    _lib.check(_lib.load().unet_cfc_attn_fwd(_lib.BF16, q.data_ptr(), k.data_ptr(), v.data_ptr(), coupling.data_ptr(), n, l, c,
                                             nh, nb, scale, out3.data_ptr(), None, None,
                                             torch.cuda.current_stream().cuda_stream), "unet_cfc_attn_fwd")
    assert (out3.float() - ref).abs().max().item() < 3e-2
    assert (out3.float() - out2.float()).abs().max().item() < 4e-2


@pytest.mark.parametrize("n,l,c,nh,nb", SHAPES + [(2, 77, 32, 8, 1), (1, 40, 24, 2, 2)])
def test_fp32_kernels_match_fp64_expression(n, l, c, nh, nb):
    """csrc/attn_f32.hip (exact-fp32 MFMA, flash style) against the expression in fp64: forward <= 2e-5, gradients
    <= 1e-4 relative L2 (the fp32 bars of the model tests are 1e-4 / 1e-3); head sizes 4 .. 256, ragged lengths."""
    from unet_amd import ops
    torch.manual_seed(l * 7 + c)
    q, k, v = (torch.randn(nb, n, l, c, device="cuda").mul_(1.5) for _ in range(3))
    coupling = torch.softmax(torch.randn(nb, nb, device="cuda"), dim=-1)
    scale = 1.0 / math.sqrt(c // nh)
    qr, kr, vr, cr = (t.detach().double().requires_grad_(True) for t in (q, k, v, coupling))
    hd = c // nh
    sp = lambda t: t.view(nb, n, l, nh, hd).permute(0, 1, 3, 2, 4)
    outs = []
    for i in range(nb):
        acc = 0
        for j in range(nb):
            p = torch.softmax(sp(qr)[i] @ sp(kr)[j].transpose(-1, -2) * scale, dim=-1)
            acc = acc + cr[i, j] * (p @ sp(vr)[j])
        outs.append(acc.permute(0, 2, 1, 3).reshape(n, l, c))
    ref = torch.cat(outs, dim=2)
    g = torch.randn(n, l, nb * c, device="cuda")
    ref.backward(g.double())
    qh, kh, vh, ch = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
    out = ops.cfc_attention(qh, kh, vh, ch, nh, scale)
    assert out.dtype == torch.float32
    assert rel(out, ref) < 2e-5
    out.backward(g)
    for name, a, b in (("dq", qh.grad, qr.grad), ("dk", kh.grad, kr.grad), ("dv", vh.grad, vr.grad),
                       ("dcoupling", ch.grad, cr.grad)):
        a, b = a.double().cpu(), b.double().cpu()
        assert (a - b).norm().item() <= 1e-4 * b.norm().item() + 2e-5, name      # (one key: dq, dk are exactly zero)
    with torch.no_grad():
        out2 = ops.cfc_attention(q, k, v, coupling, nh, scale)
    assert torch.equal(out2, out.detach())


def test_scores_that_outgrow_the_first_tile():
    """The forward kernel fixes its softmax reference point at the first 64 keys of a band pair and walks the pair a
    second time when a later score exceeds it by more than 2^32: keys past the first tile that dominate by far."""
    from unet_amd import ops
    torch.manual_seed(5)
    nb, n, l, c, nh = 2, 2, 200, 128, 2
    q, k, v = (torch.randn(nb, n, l, c, device="cuda").bfloat16() for _ in range(3))
    k[:, :, 150] *= 60.0                      # one key per band, far outside the first tile's range
    k[0, :, 70] *= -45.0
    coupling = torch.softmax(torch.randn(nb, nb, device="cuda"), dim=-1)
    scale = 1.0 / math.sqrt(c // nh)
    qr, kr, vr, cr = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
    ref = expression(qr, kr, vr, cr, nh, scale)
    g = torch.randn_like(ref).bfloat16()
    ref.backward(g.float())
    qh, kh, vh, ch = (t.detach().clone().requires_grad_(True) for t in (q, k, v, coupling))
    out = ops.cfc_attention(qh, kh, vh, ch, nh, scale)
    assert torch.isfinite(out).all()
    assert (out.float() - ref).abs().max().item() < 4e-2
    out.backward(g)
    for name, a, b in (("dq", qh.grad, qr.grad), ("dk", kh.grad, kr.grad), ("dv", vh.grad, vr.grad),
                       ("dcoupling", ch.grad, cr.grad)):
        a, b = a.double().cpu(), b.double().cpu()
        assert torch.isfinite(a).all(), name
        assert (a - b).norm().item() <= 3e-2 * b.norm().item() + 1e-4, name


def test_config_b_bottleneck_size_properties():
    """At the config-B bottleneck shape (C=1024, 4 heads x 256, L=1024, 5 bands; 8 windows), where the fp32 expression is
    too slow to be the checker: (a) with a one-hot coupling matrix band i is plain attention(q_i, k_i, v_i) -- checked
    against torch's own bf16 attention; (b) the output does not depend on the order of the keys (another walk over the
    tiles, same sums); (c) gradients are finite and dq/dk/dv of the identity-coupled case match autograd through torch's
    attention (<= 3e-2 relative L2)."""
    from unet_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(11)
    nb, n, l, c, nh = 5, 8, 1024, 1024, 4
    hd = c // nh
    q, k, v = (torch.randn(nb, n, l, c, device="cuda").bfloat16() for _ in range(3))
    scale = 1.0 / math.sqrt(hd)
    eye = torch.eye(nb, device="cuda")
    qh, kh, vh = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    out = ops.cfc_attention(qh, kh, vh, eye, nh, scale)                                  # [N, L, nb*C]
    g = torch.randn_like(out)
    out.backward(g)
    sp = lambda t: t.view(nb, n, l, nh, hd).transpose(2, 3)                              # [nb, N, nh, L, hd]
    qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(sp(qr), sp(kr), sp(vr), scale=scale)            # bf16, per band
    ref_cat = ref.transpose(2, 3).reshape(nb, n, l, c).permute(1, 2, 0, 3).reshape(n, l, nb * c)
    assert (out.float() - ref_cat.float()).abs().max().item() < 3e-2
    ref_cat.backward(g)
    for name, a, b in (("dq", qh.grad, qr.grad), ("dk", kh.grad, kr.grad), ("dv", vh.grad, vr.grad)):
        assert torch.isfinite(a).all(), name
        assert rel(a.float().cpu(), b.float().cpu()) < 3e-2, name
    perm = torch.randperm(l, device="cuda")
    coupling = torch.softmax(torch.randn(nb, nb, device="cuda"), dim=-1)
    with torch.no_grad():
        o1 = ops.cfc_attention(q, k, v, coupling, nh, scale)
        o2 = ops.cfc_attention(q, k[:, :, perm].contiguous(), v[:, :, perm].contiguous(), coupling, nh, scale)
    assert (o1.float() - o2.float()).abs().max().item() < 2e-2


def test_rejects_what_the_kernel_cannot_do():
    from unet_amd import ops
    q = torch.randn(2, 1, 16, 24, device="cuda").bfloat16()         # head_dim 12: not a multiple of 8
    cp = torch.full((2, 2), 0.5, device="cuda")
    with pytest.raises(RuntimeError, match="head_dim"):
        ops.cfc_attention(q, q, q, cp, 2, 1.0)
    with pytest.raises(TypeError):                                   # mixed dtypes
        ops.cfc_attention(q.float(), q, q, cp, 2, 1.0)
    with pytest.raises(TypeError):
        ops.cfc_attention(q.half(), q.half(), q.half(), cp, 2, 1.0)
    q32 = torch.randn(2, 1, 16, 20, device="cuda")                   # fp32 head_dim 10: not a multiple of 4
    with pytest.raises(RuntimeError, match="head_dim"):
        ops.cfc_attention(q32, q32, q32, cp, 2, 1.0)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.cfc_attention(q.cpu(), q.cpu(), q.cpu(), cp.cpu(), 1, 1.0)


def _module(g, dtype):
    from unet_amd.model import CrossFrequencyCouplingAttentionV2
    m = CrossFrequencyCouplingAttentionV2(g["c"], n_heads=g["nh"]).cuda()
    m.load_state_dict(g["sd"], strict=True)
    return m


@pytest.mark.parametrize("name", CASES)
def test_module_fp32_matches_reference(name):
    g = load_case(name)
    m = _module(g, torch.float32)
    x = g["x"].cuda().requires_grad_(True)
    y = m(x)
    assert rel(y.detach().cpu(), g["y"]) < 1e-4
    (y * g["gy"].cuda()).sum().backward()
    assert rel(x.grad.cpu(), g["dx"]) < 1e-3
    grads = dict(m.named_parameters())
    for k, ref in above_floor(g["grads"]).items():
        assert rel(grads[k].grad.cpu(), ref) < 1e-3, k


@pytest.mark.parametrize("name", CASES)
def test_module_bf16_kernels_close_to_reference(name):
    """bf16 activations end to end (band signals, q/k/v, attention on the HIP kernels, out_proj): output <= 2e-2,
    input / parameter gradients <= 6e-2 relative L2 against the reference's fp32 run (coupling_weights and gate are
    a handful of numbers that are sums over everything: 3x)."""
    g = load_case(name)
    m = _module(g, torch.bfloat16)
    x = g["x"].cuda().bfloat16().requires_grad_(True)
    y = m(x)
    assert y.dtype == torch.bfloat16
    assert rel(y.detach().float().cpu(), g["y"]) < 2e-2
    (y.float() * g["gy"].cuda()).sum().backward()
    assert rel(x.grad.float().cpu(), g["dx"]) < 6e-2
    grads = dict(m.named_parameters())
    for k, ref in above_floor(g["grads"]).items():
        bar = 1.8e-1 if ref.numel() <= 25 else 6e-2
        assert rel(grads[k].grad.cpu(), ref) < bar, (k, rel(grads[k].grad.cpu(), ref))


def test_self_attention_1d_on_the_flash_kernel():
    """SelfAttention1D (models.py:75-97) at the config-A bottleneck width, peaked scores (no 1/sqrt(d) in this layer).
    bf16 on the HIP flash kernels against the module in fp32; the yardstick is the same module in bf16 with torch's
    own products (bf16 rounding of q and k moves peaked softmaxes whatever evaluates them): the kernels must be
    within 1.25x of that error + 5e-3, and the output within 2e-2."""
    from unet_amd.model import SelfAttention1D
    torch.manual_seed(4)
    m = SelfAttention1D(512).cuda()
    with torch.no_grad():
        m.gamma.fill_(0.8)
        m.query.weight.mul_(2.0)
        m.key.weight.mul_(2.0)
    x = torch.randn(2, 512, 300, device="cuda")
    g = torch.randn(2, 512, 300, device="cuda")

    def run(inp, flash):
        m.use_flash = flash
        m.zero_grad()
        inp = inp.clone().requires_grad_(True)
        y = m(inp)
        (y.float() * g).sum().backward()
        out = {"y": y.detach().float().cpu(), "dx": inp.grad.float().cpu()}
        out.update({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()})
        return out

    ref, torch16, flash16 = run(x, False), run(x.bfloat16(), False), run(x.bfloat16(), True)
    assert rel(flash16["y"], ref["y"]) < 2e-2
    floor = 1e-5 * max(v.norm().item() for v in ref.values())
    for k, v in ref.items():
        if v.norm().item() > floor:
            assert rel(flash16[k], v) <= 1.25 * rel(torch16[k], v) + 5e-3, (k, rel(flash16[k], v), rel(torch16[k], v))


@pytest.mark.parametrize("name", ["c512_l300", "c512_l1024"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_self_attention_1d_matches_reference(name, dtype):
    """Row a10 against the REFERENCE's SelfAttention1D (models.py:75-97; tests/golden/attn_self.npz, frozen by
    oracle/make_golden_attn_self.py) at the config-A bottleneck width, L = 300 and L = 1024: fp32 <= 1e-4 output,
    <= 1e-3 gradients; bf16 (the flash kernels of csrc/attn.hip) <= 2e-2 output, <= 6e-2 gradients (gamma, one
    number: 1.8e-1, as for the other attention modules)."""
    import numpy as np
    from oracle.make_golden_attn_self import CASES, build
    from oracle.make_golden_fullwidth import strided, sums
    from unet_amd.model import SelfAttention1D
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attn_self.npz"), allow_pickle=False)
    c, l, n = CASES[name]
    m, x, gy = build(SelfAttention1D, c, l, n)
    for k, v in m.state_dict().items():        # same seed -> same weights as the reference module had
        assert np.allclose(sums(v), z[f"{name}/sd/{k}"], rtol=1e-6), k
    m = m.cuda()
    x = x.cuda().to(dtype).requires_grad_(True)
    y = m(x)
    assert y.dtype == dtype

    def err(t, key):
        return rel(strided(t.detach().float().cpu()), torch.from_numpy(z[f"{name}/{key}/slice"]))

    tol_y, tol_g = (1e-4, 1e-3) if dtype == torch.float32 else (2e-2, 6e-2)
    assert err(y, "y") < tol_y
    (y.float() * gy.cuda()).sum().backward()
    assert err(x.grad, "dx") < tol_g
    norms = {k: float(z[f"{name}/grad/{k}/sums"][2]) for k, _ in m.named_parameters()}
    for k, p in m.named_parameters():
        if norms[k] < 1e-5 * max(norms.values()):
            continue        # key.bias: a shift of every score of a row, the softmax ignores it -- true gradient zero
        bar = tol_g if p.numel() > 1 or dtype == torch.float32 else 1.8e-1
        assert err(p.grad, f"grad/{k}") < bar, (k, err(p.grad, f"grad/{k}"))


def _module_v1(g):
    from unet_amd.model import CrossFrequencyCouplingAttention
    m = CrossFrequencyCouplingAttention(g["c"], n_heads=g["nh"], chunk_size=g["cs"]).cuda()
    m.load_state_dict(g["sd"], strict=True)
    return m


@pytest.mark.parametrize("name", CASES_V1)
def test_v1_module_fp32_matches_reference(name):
    """attention_type="cross_freq" (models.py:100-224): chunks as batch items; fp32 on torch products."""
    g = load_case(name, GOLD_V1)
    m = _module_v1(g)
    x = g["x"].cuda().requires_grad_(True)
    y = m(x)
    assert rel(y.detach().cpu(), g["y"]) < 1e-4
    (y * g["gy"].cuda()).sum().backward()
    assert rel(x.grad.cpu(), g["dx"]) < 1e-3
    grads = dict(m.named_parameters())
    for k, ref in above_floor(g["grads"]).items():
        assert rel(grads[k].grad.cpu(), ref) < 1e-3, k


@pytest.mark.parametrize("name", CASES_V1)
def test_v1_module_bf16_kernels_close_to_reference(name):
    """Same bars as the V2 module: output <= 2e-2, gradients <= 6e-2 (3x for coupling_weights / gate)."""
    g = load_case(name, GOLD_V1)
    m = _module_v1(g)
    x = g["x"].cuda().bfloat16().requires_grad_(True)
    y = m(x)
    assert y.dtype == torch.bfloat16
    assert rel(y.detach().float().cpu(), g["y"]) < 2e-2
    (y.float() * g["gy"].cuda()).sum().backward()
    assert rel(x.grad.float().cpu(), g["dx"]) < 6e-2
    grads = dict(m.named_parameters())
    for k, ref in above_floor(g["grads"]).items():
        bar = 1.8e-1 if ref.numel() <= 25 else 6e-2
        assert rel(grads[k].grad.cpu(), ref) < bar, (k, rel(grads[k].grad.cpu(), ref))


def test_model_with_v1_attention_builds_and_steps():
    """CondUNet1D(attention_type="cross_freq") against the oracle on the same seeded weights (fp32 <= 1e-3)."""
    import unet_amd
    from oracle import condunet_oracle as O
    cfg = dict(in_channels=8, out_channels=8, base=16, emb_dim=16, attention_type="cross_freq")
    model = unet_amd.CondUNet1D(**cfg).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        sd["mid.2.gate"].fill_(0.5)
        sd["outc.weight"].normal_(0, 0.05)
    model.load_state_dict(sd)
    model.train()
    gen = torch.Generator().manual_seed(2)
    ob = torch.randn(2, 8, 512, generator=gen)
    ids = torch.tensor([1, 3])
    pred = model(ob.cuda(), ids.cuda())
    ref = O.forward(sd, cfg, ob, training=True, odor_ids=ids)
    assert rel(pred.detach().cpu(), ref.detach()) < 1e-3
