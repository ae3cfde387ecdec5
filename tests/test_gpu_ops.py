"""Kernel-level parity (GPU): each HIP op through the C ABI against plain torch fp32 on the CPU.
fp32 kernels use the exact-f32 MFMA (an fmaf chain), so tolerances are ~1e-5 relative; the
bf16 kernels are checked against the same reference evaluated on bf16-rounded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from unet_amd import ops
    return ops


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _ref_conv(x_ncl, w, b, stride, dilation, k):
    return F.conv1d(x_ncl, w, b, stride=stride, padding=(k - 1) * dilation // 2, dilation=dilation)


CONV_CASES = [
    # (n, cin, cout, L, k, stride, dilation)
    (2, 8, 8, 50, 3, 1, 1),
    (2, 32, 256, 300, 3, 1, 1),
    (3, 96, 160, 257, 3, 2, 1),
    (2, 64, 64, 128, 1, 1, 1),
    (2, 70, 36, 131, 7, 1, 4),
    (2, 128, 130, 200, 7, 2, 16),
    (1, 256, 256, 384, 7, 1, 32),
    (2, 6, 4, 37, 3, 2, 1),
    # single-tap convs with >= 3 K-chunks: the persistent streaming kernel (ragged K tail, ragged tiles)
    (2, 192, 256, 300, 1, 1, 1),
    (3, 520, 136, 257, 1, 1, 1),
    (70, 256, 128, 129, 1, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, dtype):
    ops = _ops()
    n, cin, cout, L, k, stride, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr = (w.bfloat16().float() if dtype == torch.bfloat16 else w).clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = _ref_conv(xr, wr, br, stride, dil, k)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy)

    xd = x.permute(0, 2, 1).contiguous().to("cuda", dtype).requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    geom = ops.geom_conv(k, stride, dil)
    y = ops.conv(xd, wd.permute(2, 0, 1), bd, geom, yr.shape[-1])
    assert y.shape == (n, yr.shape[-1], cout)
    y.backward(dy.permute(0, 2, 1).contiguous().to("cuda", dtype))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel(y.permute(0, 2, 1), yr) < tol
    assert rel(xd.grad.permute(0, 2, 1), xr.grad) < tol
    assert rel(wd.grad, wr.grad) < tol
    assert rel(bd.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 8, 33), (2, 128, 96, 200), (1, 512, 256, 64)])
def test_conv_transpose(shape, dtype):
    ops = _ops()
    n, cin, cout, L = shape
    g = torch.Generator().manual_seed(L)
    x = torch.randn(n, cin, L, generator=g)
    w = torch.randn(cin, cout, 4, generator=g) / (cin * 2) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, 2 * L, generator=g)
    if dtype == torch.bfloat16:
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    xr, wr, br, rr = (t.clone().requires_grad_(True) for t in (x, w, b, res))
    yr = F.conv_transpose1d(xr, wr, br, stride=2, padding=1) + rr
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy)
    geom, kidx = ops.geom_convt_k4s2p1()
    xd = x.permute(0, 2, 1).contiguous().to("cuda", dtype).requires_grad_(True)
    rd = res.permute(0, 2, 1).contiguous().to("cuda", dtype).requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    y = ops.conv(xd, wd.permute(2, 1, 0)[list(kidx)], bd, geom, 2 * L, res=rd)
    y.backward(dy.permute(0, 2, 1).contiguous().to("cuda", dtype))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel(y.permute(0, 2, 1), yr) < tol
    assert rel(xd.grad.permute(0, 2, 1), xr.grad) < tol
    assert rel(wd.grad, wr.grad) < tol
    assert rel(bd.grad, br.grad) < tol
    assert rel(rd.grad.permute(0, 2, 1), rr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L", [301, 570, 1150])   # 570 / 1150: lengths that select the 192-position tiles (fwd / 2-phase dgrad)
@pytest.mark.parametrize("stride", [1, 2])
def test_merged_dilated_concat(stride, dtype, L):
    """4-branch dilated conv (models.py:752-759) on a two-source (concat) input."""
    import unet_amd
    n, c1, c2, cout = 2, 40, 24, 72
    torch.manual_seed(5)
    blk = unet_amd.MultiScaleDilatedConv1d(c1 + c2, cout, 7, stride, (1, 4, 16, 32))
    with torch.no_grad():
        blk.scale_weights.copy_(torch.tensor([0.1, 0.5, -0.3, 0.2]))
    xa, xb = torch.randn(n, c1, L), torch.randn(n, c2, L)
    if dtype == torch.bfloat16:
        xa, xb = xa.bfloat16().float(), xb.bfloat16().float()
    xar, xbr = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in blk.named_parameters()}
    wsm = torch.softmax(params["scale_weights"], 0)
    xcat = torch.cat([xar, xbr], 1)
    yr = 0
    for i, d in enumerate((1, 4, 16, 32)):
        yr = yr + wsm[i] * F.conv1d(xcat, params[f"branches.{i}.conv.weight"], params[f"branches.{i}.conv.bias"],
                                    stride=stride, padding=3 * d, dilation=d)
    dy = torch.randn(yr.shape)
    yr.backward(dy)
    blk = blk.cuda()
    xad = xa.permute(0, 2, 1).contiguous().to("cuda", dtype).requires_grad_(True)
    xbd = xb.permute(0, 2, 1).contiguous().to("cuda", dtype).requires_grad_(True)
    y = blk.forward_nlc(xad, xbd)
    y.backward(dy.permute(0, 2, 1).contiguous().to("cuda", dtype))
    tol = 3e-5 if dtype == torch.float32 else 1.5e-2
    assert rel(y.permute(0, 2, 1), yr) < tol
    assert rel(xad.grad.permute(0, 2, 1), xar.grad) < tol
    assert rel(xbd.grad.permute(0, 2, 1), xbr.grad) < tol
    for k, p in blk.named_parameters():
        assert rel(p.grad, params[k].grad) < tol, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c", [8, 6, 256, 1000])
def test_film_softmax(c, dtype):
    ops = _ops()
    n, L = 2, 150
    torch.manual_seed(c)
    q = torch.randn(n, L, c) * 2
    if dtype == torch.bfloat16:
        q = q.bfloat16().float()
    kk, v = torch.randn(n, c), torch.randn(n, c)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, kk, v))
    hr = torch.softmax(qr * kr[:, None, :], dim=-1) * vr[:, None, :]
    dh = torch.randn(hr.shape)
    if dtype == torch.bfloat16:
        dh = dh.bfloat16().float()
    hr.backward(dh)
    qd = q.to("cuda", dtype).requires_grad_(True)
    kd, vd = kk.cuda().requires_grad_(True), v.cuda().requires_grad_(True)
    h = ops.FilmSoftmaxFn.apply(qd, kd, vd)
    h.backward(dh.to("cuda", dtype))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel(h, hr) < tol
    assert rel(qd.grad, qr.grad) < tol * 2
    assert rel(kd.grad, kr.grad) < tol * 2
    assert rel(vd.grad, vr.grad) < tol * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scale_gated_add_activation(dtype):
    ops = _ops()
    n, L, c = 3, 200, 48
    torch.manual_seed(1)
    x, o = torch.randn(n, L, c), torch.randn(n, L, c)
    if dtype == torch.bfloat16:
        x, o = x.bfloat16().float(), o.bfloat16().float()
    s = torch.rand(n, c)
    xr, orr, sr = (t.clone().requires_grad_(True) for t in (x, o, s))
    yr = F.gelu(xr * sr[:, None, :]) + sr[:, None, :] * orr
    dy = torch.randn(yr.shape)
    yr.backward(dy)
    xd = x.to("cuda", dtype).requires_grad_(True)
    od = o.to("cuda", dtype).requires_grad_(True)
    sd = s.cuda().requires_grad_(True)
    t1 = ops.activation(ops.ScaleFn.apply(xd, sd), 3)
    y = ops.GatedAddFn.apply(t1, od, sd)
    y.backward(dy.to("cuda", dtype))
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert rel(y, yr) < tol
    assert rel(xd.grad, xr.grad) < tol
    assert rel(od.grad, orr.grad) < tol
    assert rel(sd.grad, sr.grad) < tol * 2


@pytest.mark.parametrize("act", ["relu", "leaky_relu", "gelu", "silu", "mish"])
def test_activations(act):
    ops = _ops()
    from unet_amd._lib import ACT_IDS
    from oracle import condunet_oracle as O
    x = torch.linspace(-6, 6, 4 * 50 * 16).reshape(4, 50, 16)
    xr = x.clone().requires_grad_(True)
    yr = O.act(act, xr)
    yr.backward(torch.ones_like(yr))
    xd = x.cuda().requires_grad_(True)
    y = ops.activation(xd, ACT_IDS[act])
    y.backward(torch.ones_like(y))
    assert (y.cpu() - yr).abs().max() < 2e-6
    assert (xd.grad.cpu() - xr.grad).abs().max() < 2e-6


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16])
def test_zscore_boundary(in_dtype):
    ops = _ops()
    from oracle import condunet_oracle as O
    torch.manual_seed(2)
    ob = (torch.randn(3, 10, 333) * 2 + 0.5)
    if in_dtype == torch.bfloat16:
        ob = ob.bfloat16().float()
    obr = ob.clone().requires_grad_(True)
    xr = O.zscore(obr)
    dx = torch.randn(xr.shape)
    xr.backward(dx)
    obd = ob.to("cuda", in_dtype).requires_grad_(True)
    x = ops.ZScoreInFn.apply(obd, torch.float32)
    x.backward(dx.permute(0, 2, 1).contiguous().cuda())
    assert rel(x.permute(0, 2, 1), xr) < 1e-5
    assert rel(obd.grad, obr.grad) < (1e-4 if in_dtype == torch.float32 else 1e-2)
    y = ops.ToNCLFn.apply(x.detach(), torch.float32)
    assert torch.equal(y, x.detach().permute(0, 2, 1))


@pytest.mark.parametrize("dims", [(2, 16, (8, 8, 8)), (64, 128, (256, 256, 256)), (70, 100, (130, 33, 64)), (5, 1024, (256,)),
                                  (130, 40, (1024, 72))])
def test_small_linear_group(dims):
    """Grouped Linear maps of one [N, I] input (FiLM k/v/gate, SE MLP; csrc/small.hip) against torch fp32: forward,
    input gradient (sum of the per-tile partials), weight and bias gradients; ragged sizes, N beyond one tile."""
    ops = _ops()
    n, n_in, outs = dims
    torch.manual_seed(n + n_in)
    x = torch.randn(n, n_in)
    acts = ["none", "sigmoid", "gelu", "none"]
    maps, ref_params = [], []
    for i, o in enumerate(outs):
        w = torch.randn(o, n_in) / n_in ** 0.5
        b = torch.randn(o) * 0.3 if i % 2 == 0 else None
        ref_params.append((w.clone().requires_grad_(True), None if b is None else b.clone().requires_grad_(True)))
        maps.append((w.cuda().requires_grad_(True), None if b is None else b.cuda().requires_grad_(True), acts[i], 0.5 + 0.25 * i))
    xr = x.clone().requires_grad_(True)
    refs = []
    for (w, b), m in zip(ref_params, maps):
        pre = m[3] * F.linear(xr, w, b)
        refs.append({"none": pre, "sigmoid": torch.sigmoid(pre), "gelu": F.gelu(pre)}[m[2]])
    gys = [torch.randn(r.shape) for r in refs]
    sum((r * g).sum() for r, g in zip(refs, gys)).backward()
    xd = x.cuda().requires_grad_(True)
    ys = ops.small_linear_group(xd, maps)
    sum((y * g.cuda()).sum() for y, g in zip(ys, gys)).backward()
    for y, r in zip(ys, refs):
        assert rel(y, r) < 2e-6
    assert rel(xd.grad, xr.grad) < 1e-5
    for (w, b), m in zip(ref_params, maps):
        assert rel(m[0].grad, w.grad) < 1e-5
        if b is not None:
            assert rel(m[1].grad, b.grad) < 1e-5


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (3, 200, 48)), (torch.float32, (2, 77, 20)),
                                         (torch.bfloat16, (4, 1024, 256)), (torch.bfloat16, (2, 130, 64))])
def test_batchnorm_act_se_node_matches_torch(dtype, shape):
    """BatchNormActSEFn (norm + GELU + squeeze-excite as one node: pool-only pass, fused apply, five-sum backward)
    against the same expression in fp64 torch autograd: models.py:801-839 (BatchNorm1d in training mode, nn.GELU,
    SqueezeExcitation1D models.py:637-662).  fp32 <= 2e-5 / 2e-4 (output / gradients); bf16: the kernel reads bf16 y
    and dz and stores bf16, the reference takes the same bf16-rounded inputs, so the bars are the bf16 store's."""
    ops = _ops()
    n, rows, c = shape
    red = max(c // 4, 8)
    g = torch.Generator(device="cuda").manual_seed(5)
    y = (torch.randn(n, rows, c, device="cuda", generator=g) * 1.5 + 0.3).to(dtype).requires_grad_(True)
    gamma = (torch.rand(c, device="cuda", generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(c, device="cuda", generator=g) * 0.2).requires_grad_(True)
    w0 = (torch.randn(red, c, device="cuda", generator=g) * 0.3).requires_grad_(True)
    w2 = (torch.randn(c, red, device="cuda", generator=g) * 0.3).requires_grad_(True)
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    dz = torch.randn(n, rows, c, device="cuda", generator=g).to(dtype)
    z = ops.BatchNormActSEFn.apply(y, gamma, beta, rm, rv, 0.1, 1e-5, ops_act_id("gelu"), w0, w2)
    z.backward(dz)

    yd, gd, bd, w0d, w2d = (t.detach().double().requires_grad_(True) for t in (y, gamma, beta, w0, w2))
    mean = yd.mean(dim=(0, 1))
    var = yd.var(dim=(0, 1), unbiased=False)
    act = F.gelu((yd - mean) / torch.sqrt(var + 1e-5) * gd + bd)
    s = torch.sigmoid(F.gelu(act.mean(dim=1) @ w0d.t()) @ w2d.t())
    zr = act * s[:, None, :]
    zr.backward(dz.double())
    to, tg = (2e-5, 2e-4) if dtype == torch.float32 else (6e-3, 2e-2)
    assert rel(z, zr) < to
    assert rel(y.grad, yd.grad) < tg
    for mine, ref in ((gamma, gd), (beta, bd), (w0, w0d), (w2, w2d)):
        assert rel(mine.grad, ref.grad) < tg
    cnt = n * rows
    assert rel(rm, 0.1 * mean) < 1e-5 and rel(rv, 0.9 + 0.1 * var * cnt / (cnt - 1)) < 1e-5


def ops_act_id(name):
    from unet_amd.model import ACT_IDS
    return ACT_IDS[name]


@pytest.mark.parametrize("shape", [(3, 700, 64, 128, 1), (2, 4096, 256, 256, 1), (2, 1000, 128, 320, 2)])
def test_conv_epilogue_statistics(shape):
    """unet_conv1d_fwd_stats: the per-tile sum / sum of squares written by the conv epilogue equal those of a pass over
    the stored bf16 output (unet_rowstats), ragged last tile and a partial channel slab included; the BatchNorm node
    picks them up (STATS_HINT_HITS) and produces the same result as with its own statistics pass."""
    ops = _ops()
    n, l, cin, cout, stride = shape
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, l, cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(3, cout, cin, device="cuda", generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(cout, device="cuda", generator=g).requires_grad_(True)
    geom = ops.geom_conv(3, stride)
    l_out = ops.conv_out_len(l, 3, stride)
    y = ops.conv(x, w, b, geom, l_out, want_stats=True)
    hint = ops._take_stats(y)
    assert hint is not None, "the tiled bf16 kernel must deliver statistics for this conv"
    s1, s2 = (t.double().sum(dim=1) for t in hint)
    r1, r2 = ops.rowstats(y.detach())
    assert torch.allclose(s1, r1, rtol=1e-5, atol=1e-3)
    assert torch.allclose(s2, r2, rtol=1e-5, atol=1e-3)

    gamma, beta = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    outs = []
    for use in (True, False):
        y = ops.conv(x, w, b, geom, l_out, want_stats=use)
        hits = ops.STATS_HINT_HITS
        rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        z = ops.BatchNormActFn.apply(y, gamma, beta, rm, rv, 0.1, 1e-5, ops_act_id("gelu"), False)
        assert (ops.STATS_HINT_HITS - hits) == (1 if use else 0)
        outs.append((z, rm, rv))
    assert rel(outs[0][0], outs[1][0]) < 1e-5
    assert rel(outs[0][1], outs[1][1]) < 1e-5 and rel(outs[0][2], outs[1][2]) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("stride", [1, 2])
def test_pack_multi_images_equal_single_packs(dtype, stride):
    """unet_conv1d_pack_multi: the forward image and both data-gradient images of a two-source conv, written by one
    launch, are bit-identical to the three unet_conv1d_pack launches they replace; and ConvFn's backward on the images
    packed in forward gives the gradients of the path that packs in backward."""
    from unet_amd import ops
    torch.manual_seed(5)
    dev = "cuda"
    c1, c2, co, k = 72, 40, 136, 7            # ragged: not multiples of the 128-row tile or the K chunk
    geom = ops.geom_conv(k, stride=stride)
    w = torch.randn(k, co, c1 + c2, device=dev)[:, :, :].contiguous()
    dt = ops._dt(torch.empty(0, dtype=dtype))
    jobs = [dict(transpose=False, tap_src=None, n_taps=k, row_lo=0, rows=co, c1=c1, c2=c2),
            ops._dgrad_pack_job(geom, w, 0, c1), ops._dgrad_pack_job(geom, w, c1, c2)]
    many = ops.pack_weights_multi(w, dt, jobs)
    for img, j in zip(many, jobs):
        assert torch.equal(img, ops.pack_weights(w, dt, **j))
    assert torch.equal(ops.pack_weights_multi(w, dt, jobs[:2])[1], many[1])
    four = ops.pack_weights_multi(w, dt, jobs + [dict(jobs[0], w=w * 2.0)])      # a job may bring its own weights
    assert torch.equal(four[3], ops.pack_weights(w * 2.0, dt, **jobs[0])) and torch.equal(four[2], many[2])
    with pytest.raises(ValueError):
        ops.pack_weights_multi(w, dt, jobs + jobs[:2])

    n, l = 3, 200
    l_out = (l + stride - 1) // stride if stride == 2 else l
    x1 = torch.randn(n, l, c1, device=dev, dtype=dtype, requires_grad=True)
    x2 = torch.randn(n, l, c2, device=dev, dtype=dtype, requires_grad=True)
    wp = w.clone().requires_grad_(True)
    l_out = ops.conv(x1, wp, None, geom, l // stride, x2=x2).shape[1]
    g = torch.randn(n, l_out, co, device=dev, dtype=dtype)
    res = {}
    for flag in (True, False):
        ops.DUAL_PACK = flag
        try:
            y = ops.conv(x1, wp, None, geom, l_out, x2=x2)
            res[flag] = (y,) + torch.autograd.grad(y, (x1, x2, wp), g)
        finally:
            ops.DUAL_PACK = True
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cin", [8, 24, 32])
@pytest.mark.parametrize("kind", ["merged", "k7s2", "k3"])
def test_small_cin_weight_gradient_and_half_chunk_forward(cin, kind):
    """Convs of at most 32 input channels (the model's first conv) take their own kernel variants: the forward kernel
    without the zero k-step and the weight-gradient kernel with four taps per group.  Both must give what the regular
    kernels give (the weight gradient bit for bit: same products, same order) and agree with an fp64 evaluation of
    the same bf16 operands."""
    import os
    from unet_amd import ops
    torch.manual_seed(7)
    dev = "cuda"
    n, l, co = 3, 333, 136
    if kind == "merged":
        g0, _ = ops.merged_dilated_taps(7, (1, 4, 16, 32))
        geom = ops.ConvGeom(g0.offs, g0.phases, 1, 1, g0.alg_scale)
    elif kind == "k7s2":
        geom = ops.geom_conv(7, stride=2)
    else:
        geom = ops.geom_conv(3)
    l_out = (l + geom.stride - 1) // geom.stride
    x = torch.randn(n, l, cin, device=dev).bfloat16()
    dy = torch.randn(n, l_out, co, device=dev).bfloat16()
    w = torch.randn(geom.n_taps, co, cin, device=dev) * 0.1
    dw = ops.conv_wgrad_raw(geom, x, None, dy, co)
    os.environ["UNET_WGRAD_NOSMALL"] = "1"
    try:
        dw_regular = ops.conv_wgrad_raw(geom, x, None, dy, co)
    finally:
        del os.environ["UNET_WGRAD_NOSMALL"]
    assert torch.equal(dw, dw_regular)
    # fp64 reference: dw[j][co][ci] = sum_{n,m} dy[n,m,co] * x[n, stride*m + off_j, ci]
    xd, dyd = x.double(), dy.double()
    ref = torch.zeros(geom.n_taps, co, cin, dtype=torch.float64, device=dev)
    for j, off in enumerate(geom.offs):
        m = torch.arange(l_out, device=dev)
        src = geom.stride * m + off
        ok = (src >= 0) & (src < l)
        xs = torch.zeros(n, l_out, cin, dtype=torch.float64, device=dev)
        xs[:, ok] = xd[:, src[ok]]
        ref[j] = torch.einsum("nmo,nmi->oi", dyd, xs)
    assert rel(dw, ref) < 1e-5
    y = ops.conv(x, w, None, geom, l_out)
    yr = torch.zeros(n, l_out, co, dtype=torch.float64, device=dev)
    wq = w.bfloat16().double()
    for j, off in enumerate(geom.offs):
        m = torch.arange(l_out, device=dev)
        src = geom.stride * m + off
        ok = (src >= 0) & (src < l)
        xs = torch.zeros(n, l_out, cin, dtype=torch.float64, device=dev)
        xs[:, ok] = xd[:, src[ok]]
        yr += torch.einsum("nmi,oi->nmo", xs, wq[j])
    assert rel(y, yr) < 4e-3


GATED_CASES = [
    # (n, c_in, c_out, L): ragged position tiles, a ragged K tail, an odd number of 128-row weight tiles, many windows
    (4, 256, 256, 500),
    (2, 1024, 1024, 256),
    (3, 200, 192, 131),
    (70, 256, 256, 129),
    (2, 512, 512, 4096),
    # exactly three K chunks: the next tile's operand request falls between a tile's stores and the wait that counts them
    (2, 192, 256, 700),
    (5, 136, 8, 300),
    (9, 192, 520, 1000),
    (64, 192, 256, 2048),       # ... with eight tiles per workgroup
]


@pytest.mark.parametrize("keep_o", [True, False])
@pytest.mark.parametrize("case", GATED_CASES)
def test_gated_epilogue_of_out_proj_equals_conv_then_gated_add(case, keep_o):
    """unet_conv1d_fwd_gated (out = res + gate * (W h + b) from the streaming kernel's epilogue) against the two
    launches it replaces, unet_conv1d_fwd + unet_gated_add_fwd: o bit-identical, out identical up to the fused
    multiply-add's last bit, and against fp32 torch on the bf16-rounded inputs."""
    ops = _ops()
    from unet_amd import _lib
    lib = _lib.load()
    n, cin, cout, L = case
    g = torch.Generator().manual_seed(sum(case))
    h = torch.randn(n, L, cin, generator=g).bfloat16().cuda()
    res = torch.randn(n, L, cout, generator=g).bfloat16().cuda()
    gate = torch.randn(n, cout, generator=g).cuda()
    w = (torch.randn(1, cout, cin, generator=g) / cin ** 0.5).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    g1 = ops.geom_conv(1)
    wp = ops.pack_weights(w, _lib.BF16, transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=cout, c1=cin, c2=0)
    o_ref = ops.conv_forward_raw(g1, h, None, wp, b, None, L, cout)
    out_ref = torch.empty_like(res)
    _lib.check(lib.unet_gated_add_fwd(_lib.BF16, res.data_ptr(), o_ref.data_ptr(), gate.data_ptr(), out_ref.data_ptr(),
                                     n, L, cout, None), "unet_gated_add_fwd")
    out = torch.full_like(res, float("nan"))
    o, done = ops.conv_forward_gated_raw(g1, h, wp, b, res, gate, out, keep_o=keep_o)
    assert done, "the streaming kernel takes every shape in this list"
    assert lib.unet_conv1d_last_variant() == 3          # UNET_VARIANT_FWD_STREAM
    torch.cuda.synchronize()
    if keep_o:
        assert torch.equal(o, o_ref)
    else:
        assert o is None
    assert not torch.isnan(out.float()).any()
    diff = (out.float() - out_ref.float()).abs()
    assert (diff <= 2.0 ** -7 * out_ref.float().abs()).all()            # at most one bf16 step apart
    assert (diff == 0).float().mean() > 0.99
    ref = res.float().cpu() + gate.cpu()[:, None, :] * (h.float().cpu() @ w[0].bfloat16().float().cpu().t() + b.cpu())
    assert rel(out, ref) < 1e-2


def test_gated_epilogue_declines_shapes_the_streaming_kernel_does_not_take():
    """Return value 1 = nothing launched (the caller runs the two-launch path): too few K chunks, fp32."""
    ops = _ops()
    from unet_amd import _lib
    g1 = ops.geom_conv(1)
    for dtype, cin in ((torch.bfloat16, 64), (torch.float32, 256)):
        h = torch.randn(2, 100, cin, device="cuda", dtype=dtype)
        res = torch.randn(2, 100, 64, device="cuda", dtype=dtype)
        gate = torch.randn(2, 64, device="cuda")
        w = torch.randn(1, 64, cin, device="cuda")
        wp = ops.pack_weights(w, _lib.BF16 if dtype == torch.bfloat16 else _lib.F32, transpose=False, tap_src=None,
                              n_taps=1, row_lo=0, rows=64, c1=cin, c2=0)
        out = torch.zeros_like(res)
        o, done = ops.conv_forward_gated_raw(g1, h, wp, None, res, gate, out, keep_o=True)
        assert not done and o is None
        assert (out == 0).all()


@pytest.mark.parametrize("case", GATED_CASES)
def test_streamed_single_tap_conv_with_residual_equals_the_tap_loop_kernel(case, monkeypatch):
    """y = conv + bias + res for single-tap convs (q_proj's data gradient with the residual path's gradient) now runs in
    the streaming kernel; it rounds once, like the tap-loop kernel it replaces, so the two agree bit for bit."""
    ops = _ops()
    from unet_amd import _lib
    lib = _lib.load()
    n, cin, cout, L = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(n, L, cin, generator=g).bfloat16().cuda()
    res = torch.randn(n, L, cout, generator=g).bfloat16().cuda()
    w = (torch.randn(1, cout, cin, generator=g) / cin ** 0.5).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    g1 = ops.geom_conv(1)
    wp = ops.pack_weights(w, _lib.BF16, transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=cout, c1=cin, c2=0)
    y = ops.conv_forward_raw(g1, x, None, wp, b, res, L, cout)
    assert lib.unet_conv1d_last_variant() == 3
    monkeypatch.setenv("UNET_STREAM_NORES", "1")
    y_tap = ops.conv_forward_raw(g1, x, None, wp, b, res, L, cout)
    assert lib.unet_conv1d_last_variant() != 3
    torch.cuda.synchronize()
    assert torch.equal(y, y_tap)
    ref = res.float().cpu() + x.float().cpu() @ w[0].bfloat16().float().cpu().t() + b.cpu()
    assert rel(y, ref) < 1e-2


def test_reduce_multi_sums_every_job_in_one_launch():
    """unet_reduce_multi: per-window sums over parts and whole-buffer column sums of several partial buffers (different
    widths, part counts and window counts) from one launch, against torch in fp64; more jobs than one launch takes
    are split over launches."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 8, 256), (64, 8, 256), (3, 5, 1000), (1, 1, 8), (7, 33, 40), (64, 16, 1024), (2, 300, 96),
              (5, 2, 512), (4, 9, 64), (6, 1, 24)]
    bufs = [torch.randn(s, generator=g).cuda() for s in shapes]
    per_window = [i % 3 != 0 for i in range(len(shapes))]
    outs = ops.reduce_multi(bufs, per_window)
    assert len(outs) == len(bufs)
    for buf, pw, out in zip(bufs, per_window, outs):
        ref = buf.double().sum(1) if pw else buf.double().sum((0, 1))
        assert out.shape == ref.shape
        assert torch.allclose(out.double(), ref, rtol=1e-6, atol=1e-5)
    with pytest.raises(ValueError):
        ops.reduce_multi([bufs[0].transpose(1, 2)], [True])
