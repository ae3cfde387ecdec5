"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

CPU restatement (plain fp32/fp64 torch ops on the host) of the reference's
``CondUNet1D`` hot path, written as *functions over a state_dict* so that it
shares no code with ``unet_amd`` (an ``nn.Module`` tree that calls the HIP
library).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file.

Pinning: ``oracle/make_golden.py`` imports the reference's own ``models.py``
in the build container, runs it on seeded inputs and freezes inputs, weights,
per-block activations, loss and gradients under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks this restatement against those files,
so the oracle is pinned by outputs of the reference itself (the reference
holds no model tests of its own).

Every function cites the reference lines (``/root/reference/models.py``) it
restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

DEFAULT_CFG = dict(  # models.py:2871-2901 (ctor defaults)
    in_channels=32, out_channels=32, base=128, n_odors=7, emb_dim=128,
    dropout=0.0, use_attention=True, attention_type="basic", norm_type="batch",
    cond_mode="cross_attn_gated", n_downsample=2, conv_type="standard",
    use_se=True, conv_kernel_size=7, dilations=(1, 4, 16, 32),
    use_adaptive_scaling=False, activation="relu", skip_type="add", n_heads=4,
)


def full_cfg(cfg: dict) -> dict:
    out = dict(DEFAULT_CFG)
    out.update(cfg)
    out["dilations"] = tuple(out["dilations"])
    return out


def channel_ladder(base: int, n_down: int) -> List[int]:
    """models.py:2982-2984 -- widths double per level, capped at 8*base."""
    return [min(base * (2 ** i), base * 8) for i in range(n_down + 1)]


def act(name: str, x: Tensor) -> Tensor:
    """models.py:59-65."""
    if name == "relu":
        return F.relu(x)
    if name == "leaky_relu":
        return F.leaky_relu(x, 0.2)
    if name == "gelu":
        return F.gelu(x)  # erf form
    if name == "silu":
        return F.silu(x)
    if name == "mish":
        return F.mish(x)
    raise ValueError(f"Unknown activation: {name}")


def group_count(norm_type: str, c: int) -> int:
    """models.py:541-551 -- 'group' = largest g<=8 dividing C; layer/rms = 1."""
    if norm_type in ("layer", "rms"):
        return 1
    g = min(8, c)
    while c % g:
        g -= 1
    return g


class BNState:
    """Running-stat side effects of training-mode BatchNorm1d collected by the
    oracle (torch defaults: eps 1e-5, momentum 0.1, unbiased var into the
    running buffer, num_batches_tracked += 1)."""

    def __init__(self):
        self.updates: Dict[str, Tensor] = {}


def norm(sd: SD, pfx: str, norm_type: str, x: Tensor, training: bool,
         bn: Optional[BNState]) -> Tensor:
    """models.py:537-555 / 801-817 (norm layer choice)."""
    if norm_type == "none":
        return x
    w, b = sd[pfx + "weight"], sd[pfx + "bias"]
    if norm_type == "batch":
        if training:
            mean = x.mean(dim=(0, 2))
            var_b = x.var(dim=(0, 2), unbiased=False)
            if bn is not None:
                n = x.shape[0] * x.shape[2]
                var_u = var_b * (n / max(n - 1, 1))
                bn.updates[pfx + "running_mean"] = (
                    0.9 * sd[pfx + "running_mean"] + 0.1 * mean.detach())
                bn.updates[pfx + "running_var"] = (
                    0.9 * sd[pfx + "running_var"] + 0.1 * var_u.detach())
                bn.updates[pfx + "num_batches_tracked"] = sd[pfx + "num_batches_tracked"] + 1
        else:
            mean, var_b = sd[pfx + "running_mean"], sd[pfx + "running_var"]
        xh = (x - mean[None, :, None]) / torch.sqrt(var_b[None, :, None] + 1e-5)
        return xh * w[None, :, None] + b[None, :, None]
    if norm_type == "instance":
        mean = x.mean(dim=2, keepdim=True)
        var = x.var(dim=2, unbiased=False, keepdim=True)
        return (x - mean) / torch.sqrt(var + 1e-5) * w[None, :, None] + b[None, :, None]
    if norm_type in ("group", "layer", "rms"):
        n, c, l = x.shape
        g = group_count(norm_type, c)
        xg = x.reshape(n, g, (c // g) * l)
        mean = xg.mean(dim=2, keepdim=True)
        var = xg.var(dim=2, unbiased=False, keepdim=True)
        xh = ((xg - mean) / torch.sqrt(var + 1e-5)).reshape(n, c, l)
        return xh * w[None, :, None] + b[None, :, None]
    raise ValueError(f"Unknown norm_type: {norm_type}")


def film(sd: SD, pfx: str, cond_mode: str, x: Tensor, emb: Optional[Tensor]) -> Tensor:
    """models.py:504-517 -- gated channel-softmax modulation; identity when
    cond_mode=='none' or no embedding is given."""
    if cond_mode == "none" or emb is None:
        return x
    c = x.shape[1]
    q = F.conv1d(x, sd[pfx + "q_proj.weight"], sd[pfx + "q_proj.bias"])
    k = F.linear(emb, sd[pfx + "k_proj.weight"], sd[pfx + "k_proj.bias"])[:, :, None]
    v = F.linear(emb, sd[pfx + "v_proj.weight"], sd[pfx + "v_proj.bias"])[:, :, None]
    a = torch.softmax(q * k * (c ** -0.5), dim=1)
    o = F.conv1d(a * v, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    g = torch.sigmoid(F.linear(emb, sd[pfx + "gate.0.weight"], sd[pfx + "gate.0.bias"]))
    return x + g[:, :, None] * o


def multi_dilated(sd: SD, pfx: str, x: Tensor, stride: int, k: int,
                  dilations: Tuple[int, ...]) -> Tensor:
    """models.py:689-699, 752-759 -- softmax(scale_weights)-weighted sum of
    parallel dilated Conv1d branches, padding (k-1)*d//2."""
    w = torch.softmax(sd[pfx + "scale_weights"], dim=0)
    y = None
    for i, d in enumerate(dilations):
        b = F.conv1d(x, sd[f"{pfx}branches.{i}.conv.weight"], sd[f"{pfx}branches.{i}.conv.bias"],
                     stride=stride, padding=(k - 1) * d // 2, dilation=d)
        y = w[i] * b if y is None else y + w[i] * b
    return y


def squeeze_excite(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """models.py:658-662."""
    y = x.mean(dim=2)
    y = F.gelu(F.linear(y, sd[pfx + "fc.0.weight"]))
    y = torch.sigmoid(F.linear(y, sd[pfx + "fc.2.weight"]))
    return x * y[:, :, None]


def conv_block(sd: SD, pfx: str, cfg: dict, x: Tensor, emb: Optional[Tensor],
               downsample: bool, training: bool, bn: Optional[BNState],
               taps: Optional[dict] = None) -> Tensor:
    """models.py:568 (ConvBlock) / 839 (ModernConvBlock):
    film(Sequential[conv, norm, GELU, (SE)](x), emb).  Dropout is p=0 in every
    parity case (K11) and is not restated."""
    stride = 2 if downsample else 1
    if cfg["conv_type"] == "modern":
        y = multi_dilated(sd, pfx + "conv.0.", x, stride, cfg["conv_kernel_size"], cfg["dilations"])
    else:
        y = F.conv1d(x, sd[pfx + "conv.0.weight"], sd[pfx + "conv.0.bias"], stride=stride, padding=1)
    if taps is not None:
        taps[pfx + "conv.0"] = y
    y = norm(sd, pfx + "conv.1.", cfg["norm_type"], y, training, bn)
    y = F.gelu(y)
    if cfg["conv_type"] == "modern" and cfg["use_se"]:
        y = squeeze_excite(sd, pfx + "conv.3.", y)
    if taps is not None:
        taps[pfx + "conv"] = y
    return film(sd, pfx + "film.", cfg["cond_mode"], y, emb)


def centre_crop_pair(x: Tensor, skip: Tensor) -> Tuple[Tensor, Tensor]:
    """models.py:613-621 / 919-927."""
    lx, ls = x.shape[-1], skip.shape[-1]
    if lx < ls:
        s = (ls - lx) // 2
        skip = skip[..., s:s + lx]
    elif lx > ls:
        s = (lx - ls) // 2
        x = x[..., s:s + ls]
    return x, skip


def up_block(sd: SD, pfx: str, cfg: dict, x: Tensor, skip: Tensor, emb: Optional[Tensor],
             training: bool, bn: Optional[BNState], taps: Optional[dict] = None) -> Tensor:
    """models.py:611-630 / 916-936."""
    x = F.conv_transpose1d(x, sd[pfx + "up.weight"], sd[pfx + "up.bias"], stride=2, padding=1)
    if taps is not None:
        taps[pfx + "up"] = x
    x, skip = centre_crop_pair(x, skip)
    if cfg["skip_type"] == "concat":
        x = torch.cat([x, skip], dim=1)
    elif cfg["skip_type"] == "add":
        if pfx + "skip_proj.weight" in sd:
            skip = F.conv1d(skip, sd[pfx + "skip_proj.weight"], sd[pfx + "skip_proj.bias"])
        x = x + skip
    else:
        raise ValueError(f"Unknown skip_type: {cfg['skip_type']}")
    return conv_block(sd, pfx + "block.", cfg, x, emb, False, training, bn, taps)


def self_attention(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """models.py:86-97."""
    q = F.conv1d(x, sd[pfx + "query.weight"], sd[pfx + "query.bias"])
    k = F.conv1d(x, sd[pfx + "key.weight"], sd[pfx + "key.bias"])
    v = F.conv1d(x, sd[pfx + "value.weight"], sd[pfx + "value.bias"])
    a = torch.softmax(torch.bmm(q.transpose(1, 2), k), dim=-1)
    o = torch.bmm(a, v.transpose(1, 2)).transpose(1, 2)
    return x + sd[pfx + "gamma"] * o


CFC_BANDS = ((1, 4), (4, 8), (8, 12), (12, 30), (30, 80))   # models.py:262-269


def cross_freq_v2(sd: SD, pfx: str, x: Tensor, n_heads: int = 4, sample_rate: float = 1000.0) -> Tensor:
    """CrossFrequencyCouplingAttentionV2.forward (models.py:342-429): FFT band split (soft sigmoid masks,
    :293-340), grouped 1x1 q/k/v projections, every query band attends to every key/value band (softmax over all
    T positions, scale 1/sqrt(head_dim)), outputs mixed by softmax(coupling_weights), 1x1 out_proj, gated residual."""
    B, C, T = x.shape
    nb, hd = len(CFC_BANDS), C // n_heads
    freqs = torch.fft.rfftfreq(T, d=1.0 / sample_rate)
    filt = torch.stack([torch.sigmoid((freqs - lo) / 2) * torch.sigmoid((hi - freqs) / 2) for lo, hi in CFC_BANDS], 0)
    xf = torch.fft.rfft(x.float(), dim=-1)
    bands = torch.fft.irfft((xf.unsqueeze(1) * filt[None, :, None, :]).reshape(B * nb, C, -1), n=T, dim=-1)
    flat = bands.reshape(B, nb * C, T).to(x.dtype)
    proj = lambda n: F.conv1d(flat, sd[pfx + n + ".weight"], sd[pfx + n + ".bias"], groups=nb) \
        .reshape(B, nb, n_heads, hd, T).permute(0, 1, 2, 4, 3)                       # [B, nb, heads, T, hd]
    Q, K, V = proj("proj_q"), proj("proj_k"), proj("proj_v")
    coupling = torch.softmax(sd[pfx + "coupling_weights"], dim=-1)
    outs = []
    for i in range(nb):
        acc = 0
        for j in range(nb):
            a = torch.softmax(torch.matmul(Q[:, i], K[:, j].transpose(-2, -1)) / math.sqrt(hd), dim=-1)
            acc = acc + coupling[i, j] * torch.matmul(a, V[:, j])                    # [B, heads, T, hd]
        outs.append(acc.permute(0, 1, 3, 2).reshape(B, C, T))
    out = F.conv1d(torch.cat(outs, 1), sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    return x + sd[pfx + "gate"] * out


def cross_freq_v1(sd: SD, pfx: str, x: Tensor, n_heads: int = 4, sample_rate: float = 1000.0, chunk_size: int = 256) -> Tensor:
    """CrossFrequencyCouplingAttention.forward (models.py:175-220), the chunked V1 module: per-band Conv1d 1x1
    projections (band_q/k/v ModuleLists), attention restricted to chunks of `chunk_size` positions (a query attends
    to the keys of its own chunk only; the last chunk may be shorter), outputs mixed by softmax(coupling_weights),
    1x1 out_proj, gated residual."""
    B, C, T = x.shape
    nb, hd = len(CFC_BANDS), C // n_heads
    freqs = torch.fft.rfftfreq(T, d=1.0 / sample_rate)
    xf = torch.fft.rfft(x.float(), dim=-1)
    bands = [torch.fft.irfft(xf * (torch.sigmoid((freqs - lo) / 2) * torch.sigmoid((hi - freqs) / 2))[None, None], n=T, dim=-1)
             .to(x.dtype) for lo, hi in CFC_BANDS]
    proj = lambda n, i: F.conv1d(bands[i], sd[f"{pfx}{n}.{i}.weight"], sd[f"{pfx}{n}.{i}.bias"]) \
        .reshape(B, n_heads, hd, T).transpose(2, 3)                                   # [B, heads, T, hd]
    coupling = torch.softmax(sd[pfx + "coupling_weights"], dim=-1)
    outs = []
    for i in range(nb):
        q = proj("band_q", i)
        acc = 0
        for j in range(nb):
            k, v = proj("band_k", j), proj("band_v", j)
            chunks = []
            for c0 in range(0, T, chunk_size):
                c1 = min(c0 + chunk_size, T)
                a = torch.softmax(torch.matmul(q[:, :, c0:c1], k[:, :, c0:c1].transpose(-2, -1)) / math.sqrt(hd), dim=-1)
                chunks.append(torch.matmul(a, v[:, :, c0:c1]))
            acc = acc + coupling[i, j] * torch.cat(chunks, dim=2)
        outs.append(acc.transpose(2, 3).reshape(B, C, T))
    out = F.conv1d(torch.cat(outs, 1), sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    return x + sd[pfx + "gate"] * out


def adaptive_scale(sd: SD, pfx: str, x: Tensor, raw: Tensor) -> Tensor:
    """models.py:2043-2068 (scalar SessionAdaptiveScaling)."""
    st = torch.cat([raw.mean(dim=-1), raw.std(dim=-1)], dim=-1)
    h = F.linear(st, sd[pfx + "predictor.0.weight"], sd[pfx + "predictor.0.bias"])
    h = F.gelu(F.layer_norm(h, h.shape[-1:], sd[pfx + "predictor.1.weight"], sd[pfx + "predictor.1.bias"]))
    h = F.linear(h, sd[pfx + "predictor.3.weight"], sd[pfx + "predictor.3.bias"])
    h = F.gelu(F.layer_norm(h, h.shape[-1:], sd[pfx + "predictor.4.weight"], sd[pfx + "predictor.4.bias"]))
    p = F.linear(h, sd[pfx + "predictor.6.weight"], sd[pfx + "predictor.6.bias"])
    c = x.shape[1]
    return x * p[:, :c, None] + p[:, c:, None]


def zscore(ob: Tensor) -> Tensor:
    """models.py:3235 -- per (window, channel); unbiased std, eps added to std."""
    return (ob - ob.mean(dim=-1, keepdim=True)) / (ob.std(dim=-1, keepdim=True) + 1e-8)


def forward(sd: SD, cfg: dict, ob: Tensor, odor_ids: Optional[Tensor] = None,
            cond_emb: Optional[Tensor] = None, training: bool = True,
            bn: Optional[BNState] = None, taps: Optional[dict] = None,
            return_bottleneck: bool = False):
    """models.py:3181-3375 restricted to the hot-path scope of SURVEY.md section 8:
    z-score -> inc/encoders -> mid -> decoders -> outc -> residual
    (-> scalar adaptive scaling)."""
    cfg = full_cfg(cfg)
    raw = ob
    x = zscore(ob)
    if cond_emb is not None:
        emb = cond_emb
    elif cfg["cond_mode"] != "none" and odor_ids is not None:
        emb = F.embedding(odor_ids, sd["embed.weight"])
    else:
        emb = None
    nd = cfg["n_downsample"]
    skips = [conv_block(sd, "inc.", cfg, x, emb, False, training, bn, taps)]
    if taps is not None:
        taps["inc"] = skips[0]
    for i in range(nd):
        skips.append(conv_block(sd, f"encoders.{i}.", cfg, skips[-1], emb, True, training, bn, taps))
        if taps is not None:
            taps[f"encoders.{i}"] = skips[-1]
    # bottleneck (models.py:3029-3041): conv k3, act, (attention), conv k3, act
    has_attn = cfg["use_attention"] and cfg["attention_type"] != "none"
    h = act(cfg["activation"], F.conv1d(skips[-1], sd["mid.0.weight"], sd["mid.0.bias"], padding=1))
    if has_attn:
        if cfg["attention_type"] == "basic":
            h = self_attention(sd, "mid.2.", h)
        elif cfg["attention_type"] == "cross_freq_v2":
            h = cross_freq_v2(sd, "mid.2.", h, cfg.get("n_heads", 4), cfg.get("sampling_rate", 1000.0))
        elif cfg["attention_type"] == "cross_freq":
            h = cross_freq_v1(sd, "mid.2.", h, cfg.get("n_heads", 4), cfg.get("sampling_rate", 1000.0))
        else:
            raise NotImplementedError("oracle covers attention_type in {none, basic, cross_freq, cross_freq_v2}")
        j = 3
    else:
        j = 2
    h = act(cfg["activation"], F.conv1d(h, sd[f"mid.{j}.weight"], sd[f"mid.{j}.bias"], padding=1))
    bott = h
    if taps is not None:
        taps["mid"] = bott
    for i in range(nd):
        h = up_block(sd, f"decoders.{i}.", cfg, h, skips[nd - i - 1], emb, training, bn, taps)
        if taps is not None:
            taps[f"decoders.{i}"] = h
    delta = F.conv1d(h, sd["outc.weight"], sd["outc.bias"], padding=1)
    if cfg["in_channels"] == cfg["out_channels"]:
        out = x + delta
    else:
        out = F.conv1d(x, sd["residual_conv.weight"], sd["residual_conv.bias"]) + delta
    if cfg["use_adaptive_scaling"]:
        out = adaptive_scale(sd, "adaptive_scaler.", out, raw)
    if return_bottleneck:
        return out, bott.mean(dim=-1)
    return out


# --------------------------------------------------------------------------
# loss / metric path (train.py)
# --------------------------------------------------------------------------

CROP = (500, 4500)  # data.py:102-115 absolute crop window


def crop_to_target(x: Tensor) -> Tensor:
    """data.py:2392-2394 -- x[..., 500:4500] (plain slicing, so T<4500 just
    truncates at T)."""
    return x[..., CROP[0]:CROP[1]]


def per_channel_normalize(x: Tensor, eps: float = 1e-6) -> Tensor:
    """train.py:476-495 -- target z-score: unbiased std clamped at eps."""
    mean = x.mean(dim=-1, keepdim=True)
    std = x.std(dim=-1, keepdim=True).clamp(min=eps)
    return (x - mean) / std


def l1_on_crop(pred: Tensor, target: Tensor, weight: float = 1.0) -> Tensor:
    """train.py:1481-1486."""
    return weight * F.l1_loss(crop_to_target(pred), crop_to_target(target))


def train_step_loss(sd: SD, cfg: dict, ob: Tensor, target: Tensor,
                    odor_ids: Optional[Tensor] = None, cond_emb: Optional[Tensor] = None,
                    bn: Optional[BNState] = None, taps: Optional[dict] = None) -> Tensor:
    """One reference training step's scalar: train.py:1405 (target z-score),
    :1462 forward, :1481-1486 L1 on the crop."""
    pred = forward(sd, cfg, ob, odor_ids=odor_ids, cond_emb=cond_emb, training=True, bn=bn, taps=taps)
    return l1_on_crop(pred, per_channel_normalize(target))


def init_state_dict(cfg: dict, seed: int = 0) -> SD:
    """Random state_dict with the reference's key/shape layout.
    Values are NOT the reference's init distribution (tests that need the
    reference's weights load them from tests/golden); this exists so the oracle
    can be run at arbitrary shapes on the GPU box."""
    cfg = full_cfg(cfg)
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def rnd(*shape, scale=None):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sc = scale if scale is not None else 1.0 / math.sqrt(max(fan, 1))
        return (torch.rand(*shape, generator=g) * 2 - 1) * sc

    e = cfg["emb_dim"]
    if cfg["cond_mode"] != "none":
        sd["embed.weight"] = torch.randn(cfg["n_odors"], e, generator=g)
    ch = channel_ladder(cfg["base"], cfg["n_downsample"])
    modern = cfg["conv_type"] == "modern"
    k = cfg["conv_kernel_size"]

    def block(pfx, cin, cout):
        if modern:
            for i, _ in enumerate(cfg["dilations"]):
                sd[f"{pfx}conv.0.branches.{i}.conv.weight"] = rnd(cout, cin, k)
                sd[f"{pfx}conv.0.branches.{i}.conv.bias"] = rnd(cout, scale=0.1)
            sd[f"{pfx}conv.0.scale_weights"] = torch.rand(len(cfg["dilations"]), generator=g)
        else:
            sd[f"{pfx}conv.0.weight"] = rnd(cout, cin, 3)
            sd[f"{pfx}conv.0.bias"] = rnd(cout, scale=0.1)
        if cfg["norm_type"] != "none":
            sd[f"{pfx}conv.1.weight"] = 1 + 0.2 * rnd(cout, scale=1.0)
            sd[f"{pfx}conv.1.bias"] = 0.2 * rnd(cout, scale=1.0)
            if cfg["norm_type"] == "batch":
                sd[f"{pfx}conv.1.running_mean"] = 0.1 * rnd(cout, scale=1.0)
                sd[f"{pfx}conv.1.running_var"] = 1 + 0.2 * rnd(cout, scale=1.0)
                sd[f"{pfx}conv.1.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
        if modern and cfg["use_se"]:
            r = max(cout // 4, 8)
            sd[f"{pfx}conv.3.fc.0.weight"] = rnd(r, cout)
            sd[f"{pfx}conv.3.fc.2.weight"] = rnd(cout, r)
        if cfg["cond_mode"] == "cross_attn_gated":
            for nm in ("q_proj", "out_proj"):
                sd[f"{pfx}film.{nm}.weight"] = rnd(cout, cout, 1)
                sd[f"{pfx}film.{nm}.bias"] = rnd(cout, scale=0.1)
            for nm in ("k_proj", "v_proj", "gate.0"):
                sd[f"{pfx}film.{nm}.weight"] = rnd(cout, e)
                sd[f"{pfx}film.{nm}.bias"] = rnd(cout, scale=0.1)

    block("inc.", cfg["in_channels"], ch[0])
    nd = cfg["n_downsample"]
    for i in range(nd):
        block(f"encoders.{i}.", ch[i], ch[i + 1])
    for i in range(nd):
        cin, cskip, cout = ch[nd - i], ch[nd - i - 1], ch[nd - i - 1]
        sd[f"decoders.{i}.up.weight"] = rnd(cin, cout, 4, scale=1.0 / math.sqrt(cin * 2))
        sd[f"decoders.{i}.up.bias"] = rnd(cout, scale=0.1)
        if cfg["skip_type"] == "concat":
            block(f"decoders.{i}.block.", cout + cskip, cout)
        else:
            if cskip != cout:
                sd[f"decoders.{i}.skip_proj.weight"] = rnd(cout, cskip, 1)
                sd[f"decoders.{i}.skip_proj.bias"] = rnd(cout, scale=0.1)
            block(f"decoders.{i}.block.", cout, cout)
    cb = ch[nd]
    has_attn = cfg["use_attention"] and cfg["attention_type"] != "none"
    sd["mid.0.weight"] = rnd(cb, cb, 3)
    sd["mid.0.bias"] = rnd(cb, scale=0.1)
    j = 2
    if has_attn and cfg["attention_type"] == "cross_freq_v2":
        nb = len(CFC_BANDS)
        for n in ("proj_q", "proj_k", "proj_v"):
            sd[f"mid.2.{n}.weight"] = rnd(nb * cb, cb, 1)
            sd[f"mid.2.{n}.bias"] = rnd(nb * cb, scale=0.1)
        sd["mid.2.out_proj.weight"] = rnd(cb, nb * cb, 1)
        sd["mid.2.out_proj.bias"] = rnd(cb, scale=0.1)
        sd["mid.2.coupling_weights"] = rnd(nb, nb, scale=0.5)
        sd["mid.2.gate"] = torch.full((1,), 0.3)
        j = 3
    elif has_attn and cfg["attention_type"] == "cross_freq":
        nb = len(CFC_BANDS)
        for n in ("band_q", "band_k", "band_v"):
            for i in range(nb):
                sd[f"mid.2.{n}.{i}.weight"] = rnd(cb, cb, 1)
                sd[f"mid.2.{n}.{i}.bias"] = rnd(cb, scale=0.1)
        sd["mid.2.out_proj.weight"] = rnd(cb, nb * cb, 1)
        sd["mid.2.out_proj.bias"] = rnd(cb, scale=0.1)
        sd["mid.2.coupling_weights"] = rnd(nb, nb, scale=0.5)
        sd["mid.2.gate"] = torch.full((1,), 0.3)
        j = 3
    elif has_attn:
        sd["mid.2.query.weight"] = rnd(cb // 8, cb, 1)
        sd["mid.2.query.bias"] = rnd(cb // 8, scale=0.1)
        sd["mid.2.key.weight"] = rnd(cb // 8, cb, 1)
        sd["mid.2.key.bias"] = rnd(cb // 8, scale=0.1)
        sd["mid.2.value.weight"] = rnd(cb, cb, 1)
        sd["mid.2.value.bias"] = rnd(cb, scale=0.1)
        sd["mid.2.gamma"] = torch.full((1,), 0.3)
        j = 3
    sd[f"mid.{j}.weight"] = rnd(cb, cb, 3)
    sd[f"mid.{j}.bias"] = rnd(cb, scale=0.1)
    sd["outc.weight"] = rnd(cfg["out_channels"], ch[0], 3)
    sd["outc.bias"] = rnd(cfg["out_channels"], scale=0.1)
    if cfg["in_channels"] != cfg["out_channels"]:
        sd["residual_conv.weight"] = rnd(cfg["out_channels"], cfg["in_channels"], 1)
        sd["residual_conv.bias"] = rnd(cfg["out_channels"], scale=0.1)
    if cfg["use_adaptive_scaling"]:
        ci, co, hd = cfg["in_channels"], cfg["out_channels"], 64
        p = "adaptive_scaler.predictor."
        sd[p + "0.weight"], sd[p + "0.bias"] = rnd(hd, 2 * ci), rnd(hd, scale=0.1)
        sd[p + "1.weight"], sd[p + "1.bias"] = torch.ones(hd), torch.zeros(hd)
        sd[p + "3.weight"], sd[p + "3.bias"] = rnd(hd, hd), rnd(hd, scale=0.1)
        sd[p + "4.weight"], sd[p + "4.bias"] = torch.ones(hd), torch.zeros(hd)
        sd[p + "6.weight"] = rnd(2 * co, hd, scale=0.05)
        sd[p + "6.bias"] = torch.cat([torch.ones(co), torch.zeros(co)])
    return sd
