import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
import unet_amd
from oracle import condunet_oracle as O
def rel(a,b):
    a,b=a.double().cpu(),b.double().cpu(); return ((a-b).norm()/b.norm().clamp_min(1e-30)).item()
cfg = dict(in_channels=32, out_channels=32, base=64, conv_type="modern", attention_type="none", activation="gelu")
sd = O.init_state_dict(cfg, seed=11)
torch.manual_seed(3)
ob, tgt = torch.randn(4, 32, 1024), torch.randn(4, 32, 1024)
ids = torch.tensor([0, 1, 2, 3])
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
run_sd = dict(sd); run_sd.update(leaves)
ref = O.forward(run_sd, cfg, ob, odor_ids=ids, training=True)
F.l1_loss(ref[..., 100:900], tgt[..., 100:900]).backward()
model = unet_amd.CondUNet1D(**cfg).cuda(); model.load_state_dict(sd); model.set_compute_dtype(torch.bfloat16); model.train()
out = model(ob.cuda(), ids.cuda())
F.l1_loss(out[..., 100:900], tgt.cuda()[..., 100:900]).backward()
print("out rel", rel(out, ref))
rows=[]
for k,p in model.named_parameters():
    g=leaves[k].grad
    if g is None: continue
    rows.append((rel(p.grad,g), k, g.norm().item()))
for r in sorted(rows, reverse=True)[:25]: print("%.4f %-50s |g|=%.3e"%r)
