"""Which torch (aten) operators still launch kernels inside one training step of config B (or, with ``eval N``, inside
one eval-mode forward of N windows under no_grad), and how many: one eager step under torch.profiler, every device
kernel launched by an aten operator grouped by operator and input shapes, plus the total number of kernels."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import unet_amd
from unet_amd import metrics
from unet_amd.optim import ClipAdamW


def main():
    torch.manual_seed(0)
    cfg = dict(in_channels=32, out_channels=32, base=256, conv_type="modern", attention_type="none", activation="gelu",
               skip_type="add", norm_type="batch", cond_mode="cross_attn_gated", n_downsample=2)
    model = unet_amd.CondUNet1D(**cfg).cuda()
    with torch.no_grad():
        model.outc.weight.normal_(0, 0.02)
    model.set_compute_dtype(torch.bfloat16)
    model.train()
    opt = ClipAdamW([{"params": list(model.parameters()), "max_grad_norm": 5.0}], lr=2e-4, weight_decay=0.0)
    ob = torch.randn(64, 32, 4096, device="cuda").bfloat16()
    tgt = torch.randn(64, 32, 4096, device="cuda")
    ids = torch.zeros(64, dtype=torch.long, device="cuda")

    if len(sys.argv) > 2 and sys.argv[1] == "eval":
        nb = int(sys.argv[2])
        model.eval()
        xe, ide = ob[:nb].contiguous(), ids[:nb].contiguous()

        def step():
            with torch.no_grad():
                model(xe, ide)
    else:
        step = None

    def train_step():
        opt.zero_grad(set_to_none=True)
        loss = metrics.l1_crop_loss(model(ob, ids), metrics.per_channel_normalize(tgt), 1.0, 500, 4500)
        loss.backward()
        opt.step()

    step = step or train_step
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.Counter()
    dur = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        if any(ch.kernels for ch in ev.cpu_children if ch.name.startswith("aten::")):
            continue                      # count the innermost operator only
        frame = next((f for f in ev.stack if "unet_amd" in f or "bench" in f or "scripts/" in f), ev.stack[0] if ev.stack else "?")
        key = (ev.name, str(ev.input_shapes)[:90], frame.strip()[-90:])
        agg[key] += len(ev.kernels)
        dur[key] += sum(k.duration for k in ev.kernels)
    total = 0
    for key, n in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
        total += n
        print(f"{n:3d} launches {dur[key]:8.1f} us  {key[0]:28s} {key[1]:90s} <- {key[2]}")
    print("torch-operator launches per step:", total)
    n_kernels = sum(1 for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    print("device kernels (and copies) in the step:", n_kernels)


if __name__ == "__main__":
    main()
