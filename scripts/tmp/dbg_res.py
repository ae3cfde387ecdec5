import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unet_amd import ops, _lib
lib = _lib.load()
n, cin, cout, L = 4, 256, 256, 500
g = torch.Generator().manual_seed(1)
x = torch.randn(n, L, cin, generator=g).bfloat16().cuda()
res = torch.randn(n, L, cout, generator=g).bfloat16().cuda()
w = (torch.randn(1, cout, cin, generator=g) / cin ** 0.5).cuda()
b = (torch.randn(cout, generator=g) * 0.1).cuda()
g1 = ops.geom_conv(1)
wp = ops.pack_weights(w, _lib.BF16, transpose=False, tap_src=None, n_taps=1, row_lo=0, rows=cout, c1=cin, c2=0)
y = ops.conv_forward_raw(g1, x, None, wp, b, res, L, cout)
print("variant", lib.unet_conv1d_last_variant())
os.environ["UNET_STREAM_NORES"] = "1"
yt = ops.conv_forward_raw(g1, x, None, wp, b, res, L, cout)
print("variant", lib.unet_conv1d_last_variant())
ref = res.float() + x.float() @ w[0].bfloat16().float().t() + b
for name, t in (("stream", y), ("tap", yt)):
    e = (t.float() - ref).abs()
    print(name, "max err", e.max().item(), "bad chans", (e.amax((0, 1)) > 0.05).nonzero().flatten().tolist()[:64],
          "bad rows of window 0", (e[0].amax(1) > 0.05).nonzero().flatten().tolist()[:40])
# which part is wrong: y - res - conv?
conv = x.float() @ w[0].bfloat16().float().t() + b
d = y.float() - conv
# does d equal res of another position/channel?
bad = (y.float() - ref).abs() > 0.05
idx = bad.nonzero()[:5]
for i in idx.tolist():
    nn, m, c = i
    val = d[nn, m, c].item()
    cand = ((res[nn].float() - val).abs() < 0.02).nonzero()[:6].tolist()
    print("at", i, "y-conv =", round(val, 3), "res there", res[nn, m, c].item(), "matches res at", cand)
