"""End-to-end parity (GPU): unet_amd.CondUNet1D (HIP kernels through the C ABI) against the golden
fixtures frozen from the reference's models.py, and against the oracle at sizes it finishes in
seconds.  fp32 bar: <= 1e-3 relative (north star); bf16 bar stated per tensor class below."""
import pytest
import torch
import torch.nn.functional as F

from tests._golden import Golden, model_cases

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _check_bf16_grads(mine, ref, tol, small_bar=None):
    """bf16 parameter-gradient bar: rel-L2 <= tol for every tensor whose reference gradient is above
    the noise floor (1e-5 of the largest gradient norm: conv biases in front of a norm layer have an
    exactly-zero true gradient).  Tensors of <= 4 numbers (scale_weights, attention gamma / gate) are single sums
    of ~10^5..10^7 signed products that largely cancel, so a bf16-sized relative change of the factors moves them
    by far more than it moves an L2 norm over many such sums: their bar is DERIVED per tensor by the caller
    (``small_bar[k]``, see ``_bf16_sensitivity``) and falls back to 3x tol where no oracle run is affordable."""
    floor = 1e-5 * max(v.norm().item() for v in ref.values())
    bad = []
    for k, v in ref.items():
        if v.norm().item() < floor:
            continue
        r = rel(mine[k], v)
        bar = tol
        if v.numel() <= 4:
            bar = small_bar[k] if small_bar is not None else 3 * tol
        if bar is None:
            # not resolvable in bf16 (see _bf16_sensitivity): rounding the INPUTS alone moves this gradient by more
            # than 1/8 of itself, so no relative bar below 0.5 can be derived for a bf16 evaluation.  What is still
            # required: the right direction and the right order of magnitude.
            cos = float((mine[k].double().cpu() * v.double().cpu()).sum()) / max(
                float(mine[k].double().norm()) * float(v.double().norm()), 1e-300)
            if not (cos > 0.5 and r < 1.0):
                bad.append((k, r, "direction/magnitude only", cos))
            continue
        if r > bar:
            bad.append((k, r, bar))
    assert not bad, bad


def _bf16_sensitivity(O, g, G, ref_grads, tol):
    """How far the gradient of each <= 4-element tensor moves when ONLY the inputs of the computation are rounded
    to bf16 (every weight matrix and the input window; fp32 arithmetic throughout, CPU oracle): a lower bound of
    what any bf16 evaluation does to it, kernels or not, because the bf16 path also rounds every activation of the
    ~10 layers in between.  Bar for such a tensor = max(tol, 4 x that deviation); a tensor whose deviation is
    below tol/4 keeps the plain bar."""
    sd = g.state_dict()
    rounded = {k: (v.bfloat16().float() if (v.is_floating_point() and v.dim() >= 2) else v) for k, v in sd.items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in rounded.items() if v.is_floating_point() and "running_" not in k}
    run_sd = dict(rounded)
    run_sd.update(leaves)
    out = O.forward(run_sd, g.cfg, g.t("ob").bfloat16().float(), training=True, **g.cond_kwargs())
    (out * G).sum().backward()
    bars = {}
    for k, v in ref_grads.items():
        if v.numel() <= 4 and leaves[k].grad is not None:
            # a gradient that input rounding alone moves by > 12.5 % (gamma of the base-8 fixture: measured 26 %,
            # which is where round 1's observed 0.29 comes from) gets NO relative bar (None): _check_bf16_grads then
            # only requires direction and order of magnitude, and says so, instead of a bar of 1.0 that tests nothing
            dev = 4 * rel(leaves[k].grad, v)
            bars[k] = None if dev > 0.5 else max(tol, dev)
    return bars


def _build(g, dtype=None):
    import unet_amd
    model = unet_amd.CondUNet1D(**g.cfg).cuda()
    missing = model.load_state_dict(g.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    if dtype is not None:
        model.set_compute_dtype(dtype)
    return model


def _kw(g):
    return {k: v.cuda() for k, v in g.cond_kwargs().items()}


@pytest.mark.parametrize("name", model_cases())
def test_fp32_train_step_matches_reference(name):
    g = Golden(name)
    model = _build(g)
    model.train()
    acts = {}
    hooks = model.register_activation_hooks(lambda lname, o: acts.__setitem__(lname, o.detach()))
    ob = g.t("ob").cuda().requires_grad_(True)
    pred, bott = model(ob, return_bottleneck=True, **_kw(g))
    for h in hooks:
        h.remove()
    assert list(acts) == model.get_layer_names()
    assert pred.shape == g.t("pred").shape and pred.dtype == torch.float32
    assert rel(pred, g.t("pred")) < 1e-4
    assert rel(bott, g.t("bottleneck")) < 1e-4
    for k, v in g.group("act/").items():
        # fixture keys are module paths (encoders.0); the hook names are the reference's (encoder_0); the hook on
        # outc sees delta, before the residual, as the reference's does
        k = k.replace("encoders.", "encoder_").replace("decoders.", "decoder_")
        assert acts[k].shape == v.shape, k
        assert rel(acts[k], v) < 1e-4, k
    loss = F.l1_loss(pred[..., g.lo:g.hi], g.t("target").cuda()[..., g.lo:g.hi])
    assert abs(loss.item() - float(g.z["loss"])) < 1e-5
    loss.backward()
    assert rel(ob.grad, g.t("grad_ob")) < 1e-3
    params = dict(model.named_parameters())
    for k, gr in g.group("grad/").items():
        if not bool(g.z["hasgrad/" + k]):
            continue
        assert params[k].grad is not None, k
        scale = gr.abs().max().item()
        if scale < 1e-7:  # mathematically zero (e.g. a conv bias in front of a norm layer): noise on both sides
            assert params[k].grad.abs().max().item() < 1e-6, k
            continue
        assert rel(params[k].grad, gr) < 1e-3, k
    sd_now = model.state_dict()
    for k, v in g.group("sd_after/").items():
        if v.is_floating_point():
            assert rel(sd_now[k], v) < 1e-5, k
        else:
            assert int(sd_now[k]) == int(v), k


@pytest.mark.parametrize("name", model_cases())
def test_fp32_eval_matches_reference(name):
    g = Golden(name)
    model = _build(g)
    sd = g.state_dict()
    sd.update(g.group("sd_after/"))
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        pred = model(g.t("ob").cuda(), **_kw(g))
    assert rel(pred, g.t("pred_eval")) < 1e-4


# bf16 bars (kernels: bf16 storage + bf16 MFMA, fp32 accumulate; reference: fp32):
#   output rel-L2 <= 3e-2 and |loss - ref| <= 2e-2 against the reference's golden forward;
#   parameter gradients rel-L2 <= 1e-1 under a LINEAR loss <pred, G> evaluated by the oracle on the
#   same weights.  (The L1 loss of the fixtures has gradient sign(pred - target): a bf16-sized change
#   of pred flips a fraction f of the signs, which alone moves the gradient by 2*sqrt(f) -- that is a
#   property of L1 in reduced precision, not of the kernels, so it is not what this test measures.)
@pytest.mark.parametrize("name", ["A_std_attn_add_bn", "B_modern_add_bn_gelu", "B2_modern_concat", "B_modern_crop4096",
                                  "std_crossfreq_v2"])
def test_bf16_train_step_close_to_reference(name):
    from oracle import condunet_oracle as O
    g = Golden(name)
    model = _build(g, torch.bfloat16)
    model.train()
    ob = g.t("ob").cuda()
    pred = model(ob, **_kw(g))
    assert pred.dtype == torch.float32
    assert rel(pred, g.t("pred")) < 3e-2
    loss = F.l1_loss(pred[..., g.lo:g.hi], g.t("target").cuda()[..., g.lo:g.hi])
    assert abs(loss.item() - float(g.z["loss"])) < 2e-2
    G = torch.randn(pred.shape, generator=torch.Generator().manual_seed(9)) / pred.numel()
    (pred * G.cuda()).sum().backward()
    sd = g.state_dict()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    run_sd = dict(sd)
    run_sd.update(leaves)
    ref = O.forward(run_sd, g.cfg, g.t("ob"), training=True, **g.cond_kwargs())
    (ref * G).sum().backward()
    # the cross-frequency attention adds bf16 roundings of the band signals, q/k/v and 25 attention outputs at
    # head_dim 8 in this fixture (its own v-projection sees 25 bf16 attention backward passes): a 2e-1 bar
    tol = 2e-1 if g.cfg.get("attention_type") == "cross_freq_v2" else 1e-1
    refg = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    small = _bf16_sensitivity(O, g, G, refg, tol)
    _check_bf16_grads({k: p.grad for k, p in model.named_parameters()}, refg, tol, small)


def test_config_b_shape_vs_oracle():
    """Config-B structure (modern, add, gelu, batch) at base=64, T=1024, N=4: HIP fp32 and bf16 vs the
    CPU oracle on the same seeded inputs."""
    import unet_amd
    from oracle import condunet_oracle as O
    cfg = dict(in_channels=32, out_channels=32, base=64, conv_type="modern", attention_type="none", activation="gelu")
    sd = O.init_state_dict(cfg, seed=11)
    torch.manual_seed(3)
    ob, tgt = torch.randn(4, 32, 1024), torch.randn(4, 32, 1024)
    ids = torch.tensor([0, 1, 2, 3])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    run_sd = dict(sd)
    run_sd.update(leaves)
    ref = O.forward(run_sd, cfg, ob, odor_ids=ids, training=True)
    ref_loss = F.l1_loss(ref[..., 100:900], tgt[..., 100:900])
    ref_loss.backward()
    for dtype, tol_o, tol_g in ((torch.float32, 1e-4, 1e-3), (torch.bfloat16, 2e-2, 6e-2)):
        model = unet_amd.CondUNet1D(**cfg).cuda()
        model.load_state_dict(sd, strict=True)
        model.set_compute_dtype(dtype)
        model.train()
        out = model(ob.cuda(), ids.cuda())
        loss = F.l1_loss(out[..., 100:900], tgt.cuda()[..., 100:900])
        loss.backward()
        assert rel(out, ref) < tol_o, dtype
        params = dict(model.named_parameters())
        refg = {k: v.grad for k, v in leaves.items() if v.grad is not None}
        if dtype == torch.float32:
            worst = max(rel(params[k].grad, v) for k, v in refg.items() if v.norm() > 1e-7)
            assert worst < tol_g, (dtype, worst)
        else:
            _check_bf16_grads({k: p.grad for k, p in params.items()}, refg, tol_g)


def test_bias_gradients_come_from_the_norm_backward_hand_off():
    """The conv bias gradients in front of a norm / activation are taken from the column sums the norm's backward
    emits with dy (ops._offer_colsum / _take_colsum) instead of a separate pass over dy -- and they are the right
    numbers (golden gradients of every conv bias, incl. the bottleneck convs whose bias gradient is not zero)."""
    from unet_amd import ops
    g = Golden("B_modern_add_bn_gelu")
    model = _build(g)
    model.train()
    before = ops.BIAS_HINT_HITS
    pred = model(g.t("ob").cuda(), **_kw(g))
    F.l1_loss(pred[..., g.lo:g.hi], g.t("target").cuda()[..., g.lo:g.hi]).backward()
    assert ops.BIAS_HINT_HITS - before >= 7          # 5 block convs + 2 bottleneck convs
    params = dict(model.named_parameters())
    for k in ("mid.0.bias", "mid.2.bias"):
        assert rel(params[k].grad, g.t("grad/" + k)) < 1e-3, k
    assert not ops._BIAS_HINT or len(ops._BIAS_HINT) <= 64


def test_inference_reuses_packed_weights_until_a_parameter_changes():
    """Under torch.no_grad() the packed weight images are cached per conv and revalidated against the parameters'
    version counters: same output call after call with no repacking, and an in-place weight update is seen."""
    from unet_amd import ops
    g = Golden("B2_modern_concat")
    model = _build(g, torch.bfloat16)
    model.eval()
    ob = g.t("ob").cuda()
    ops.clear_pack_cache()
    with torch.no_grad():
        y0 = model(ob, **_kw(g))
        n_entries = len(ops._PACK_CACHE)
        assert n_entries >= 10
        assert any(k[0][0] == "film" for k in ops._PACK_CACHE)            # FiLM's projections are cached too
        assert any(k[0][0] == "bn_eval_affine" for k in ops._PACK_CACHE)  # ... and eval BatchNorm's coefficients
        snapshot = {k: v[1][0].data_ptr() for k, v in ops._PACK_CACHE.items()}
        y1 = model(ob, **_kw(g))
        assert torch.equal(y0, y1)
        assert {k: v[1][0].data_ptr() for k, v in ops._PACK_CACHE.items()} == snapshot     # nothing was repacked
        model.outc.weight.mul_(2.0)                                                       # bumps the version counter
        model.decoders[0].block.conv[0].scale_weights.add_(0.3)
        y2 = model(ob, **_kw(g))
        assert len(ops._PACK_CACHE) == n_entries
    assert rel(y2, y0) > 1e-3
    # and the cached path computes what the path with autograd on (no cache) computes from the same weights -- up to bf16
    # rounding: under no_grad the norm + GELU + SE scale is one fused pass (the product is rounded once), with autograd
    # the activation is stored in bf16 before the scale (rounded twice)
    with torch.enable_grad():
        y3 = model(ob, **_kw(g))
    assert rel(y3.detach(), y2) < 5e-3
    ops.clear_pack_cache()


def test_packed_weight_cache_does_not_outlive_its_model():
    """Checkpoint after checkpoint evaluated in one inference process (LOSO fold models): a freed model's blocks come
    back from the caching allocator at the same addresses with the same version counts, so the cache must key on the
    Parameter objects themselves (weak references).  Models are built, loaded and dropped one after the other with
    different weights; each must answer with its OWN weights."""
    import gc
    import unet_amd
    from unet_amd import ops
    g = Golden("B2_modern_concat")
    ob = g.t("ob").cuda()
    sd0 = g.state_dict()
    ops.clear_pack_cache()

    def run(scale, keep_cache=True):
        sd = {k: (v * scale if (v.is_floating_point() and v.dim() == 3) else v) for k, v in sd0.items()}
        sd["outc.weight"] = torch.full_like(sd0["outc.weight"], 0.05 * scale)     # outc is zero-initialised
        model = unet_amd.CondUNet1D(**g.cfg).cuda()
        model.set_compute_dtype(torch.bfloat16)
        if keep_cache:
            # fill the tensors behind the cache's back (no version bump, no load_state_dict hook): the weak
            # references alone must tell this model's parameters from the previous model's
            n_before = len(ops._PACK_CACHE)
            for k, v in model.state_dict(keep_vars=True).items():
                v.data.copy_(sd[k])
            assert len(ops._PACK_CACHE) == n_before
        else:
            model.load_state_dict(sd, strict=True)
            ops.clear_pack_cache()
        model.eval()
        with torch.no_grad():
            y = model(ob, **_kw(g)).clone()
        ptrs = sorted(p.data_ptr() for p in model.parameters())
        del model
        gc.collect()
        return y, ptrs

    want = [run(s, keep_cache=False)[0] for s in (1.0, 1.5, 0.5)]          # each with a cold cache
    ops.clear_pack_cache()
    got, ptrs = zip(*[run(s) for s in (1.0, 1.5, 0.5)])                     # cache left alone between models
    assert rel(want[1], want[0]) > 2e-3 and rel(want[2], want[0]) > 2e-3    # the weights do matter
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not ops._PACK_CACHE                                              # entries died with their parameters


def test_adaptive_scaling_kernel_statistics_match_the_differentiable_path():
    """SessionAdaptiveScaling (models.py:2043-2068, on by default in LOSO): with an input that takes no gradient the
    mean / std come from unet_ncl_moments and gamma / beta ride the output transpose (ops.ToNCLAffineFn); with an input
    that requires a gradient the statistics stay on torch's differentiable reductions.  Same output, same parameter
    gradients, and both equal the reference's (golden fixture)."""
    g = Golden("modern_adaptive_nose")
    model = _build(g)
    model.train()
    ob = g.t("ob").cuda()
    tgt = g.t("target").cuda()
    grads = []
    for need_input_grad in (False, True):
        model.zero_grad(set_to_none=True)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.momentum = 0.0                     # keep the buffers: the two passes start from the same state
        x = ob.clone().requires_grad_(need_input_grad)
        pred = model(x, **_kw(g))
        assert rel(pred, g.t("pred")) < 1e-4
        F.l1_loss(pred[..., g.lo:g.hi], tgt[..., g.lo:g.hi]).backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    scaler_keys = [k for k in grads[0] if k.startswith("adaptive_scaler.")]
    assert len(scaler_keys) >= 6
    for k in grads[0]:
        ref = g.t("grad/" + k)
        if ref.abs().max().item() < 1e-7:
            continue
        assert rel(grads[0][k], ref) < 1e-3, k
        assert rel(grads[0][k], grads[1][k]) < 1e-5, k


@pytest.mark.parametrize("name", ["B_modern_add_bn_gelu", "B2_modern_concat", "modern_group_silu_resconv_oddT"])
def test_skip_gradient_merged_in_the_data_gradient_epilogue(name):
    """A skip tensor feeds the next encoder conv and a decoder block: the decoder side parks its gradient
    (ops.StashGradFn) and the encoder conv's data-gradient kernel adds it in its epilogue instead of autograd running a
    full-tensor add.  Same gradients as with autograd's own accumulation (fp32: to rounding), and both match the
    reference (the golden tests above run with the merge on)."""
    from unet_amd import ops
    g = Golden(name)
    grads = []
    hits = []
    for fuse in (True, False):
        ops.SKIP_GRAD_FUSE = fuse
        try:
            model = _build(g)
            model.train()
            before = ops.SKIP_GRAD_HITS
            ob = g.t("ob").cuda().requires_grad_(True)
            pred = model(ob, **_kw(g))
            F.l1_loss(pred[..., g.lo:g.hi], g.t("target").cuda()[..., g.lo:g.hi]).backward()
            hits.append(ops.SKIP_GRAD_HITS - before)
            grads.append((ob.grad.clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
        finally:
            ops.SKIP_GRAD_FUSE = True
    assert hits[0] == g.cfg.get("n_downsample", 2) and hits[1] == 0
    assert rel(grads[0][0], grads[1][0]) < 1e-5
    for k, v in grads[1][1].items():
        if v.abs().max().item() < 1e-7:
            continue
        assert rel(grads[0][1][k], v) < 1e-5, k
    assert rel(grads[0][0], g.t("grad_ob")) < 1e-3


def test_training_forward_never_reads_the_inference_cache():
    """Grad mode is always off INSIDE an autograd Function's forward, so FiLM must not read "is this inference" from
    it: a differentiated call has to pack its weights afresh (the optimizer kernels update parameters without bumping
    their version counters, so a cached image would be stale from the second step on).  Weights are changed behind
    the version counters' back here; the next training forward must see them and leave no cache entry behind."""
    from unet_amd import ops
    from unet_amd.model import FiLM
    g = Golden("B_modern_add_bn_gelu")
    model = _build(g, torch.bfloat16)
    model.outc.weight.data.fill_(0.05)                     # outc is zero-initialised: make the blocks matter
    model.train()
    ob = g.t("ob").cuda()
    ops.clear_pack_cache()
    y0 = model(ob, **_kw(g)).detach().clone()
    assert not ops._PACK_CACHE
    films = [m for m in model.modules() if isinstance(m, FiLM) and hasattr(m, "q_proj")]
    assert films
    versions = [m.q_proj.weight._version for m in films]
    for m in films:
        m.q_proj.weight.data.mul_(1.5)
        m.out_proj.weight.data.mul_(-1.5)
    assert versions == [m.q_proj.weight._version for m in films]
    y1 = model(ob, **_kw(g)).detach().clone()
    assert not ops._PACK_CACHE
    assert rel(y1, y0) > 1e-3
    # ... and a cache left behind by an evaluation pass in between is not picked up by training either
    model.eval()
    with torch.no_grad():
        model(ob, **_kw(g))
    assert ops._PACK_CACHE
    model.training = True                                   # not model.train(): that clears the cache on purpose
    for m in model.modules():
        m.training = True
    for m in films:
        m.out_proj.weight.data.mul_(-1.0)
    y2 = model(ob, **_kw(g)).detach().clone()
    assert rel(y2, y1) > 1e-3
    ops.clear_pack_cache()
