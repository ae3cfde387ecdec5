"""Row (e), GPU: the data-parallel step captured as ONE HIP graph with its RCCL collectives inside
(``GraphedTrainStep(overlap=True)``), exercised on one GPU through a one-rank RCCL group with the collectives forced
on: the average over one rank is the identity, so weights after K replays must equal those of the plain captured
step (no collectives) bit for bit -- which checks the capture of hooks, bucket packing, async all-reduce and the
deferred wait, not the xGMI transport (the driver's N>1 bench does that).

The body runs in a child process: it brings up a process group (whose watchdog thread lives as long as the process)
and captures collectives, neither of which the rest of the suite's process should inherit -- and if the capture
should ever take the process down, the child dies, not pytest."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
import torch.nn.functional as F
import unet_amd
from unet_amd.graphs import GraphedTrainStep

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29731", world_size=1, rank=0,
                        device_id=torch.device("cuda:0"))


def setup(seed=0):
    torch.manual_seed(seed)
    model = unet_amd.CondUNet1D(in_channels=8, out_channels=8, base=32, n_odors=4, emb_dim=32, n_downsample=2,
                                conv_type="modern", attention_type="cross_freq_v2", norm_type="batch",
                                cond_mode="cross_attn_gated").cuda()
    model.set_compute_dtype(torch.bfloat16)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)
    return model, opt


g = torch.Generator(device="cuda").manual_seed(1)
batches = [(torch.randn(4, 8, 512, device="cuda", generator=g), torch.randn(4, 8, 512, device="cuda", generator=g),
            torch.randint(0, 4, (4,), device="cuda", generator=g)) for _ in range(4)]
results = []
for overlap in (False, True):
    model, opt = setup()
    ob, tg, ids = batches[0]
    step = GraphedTrainStep(model, F.l1_loss, opt, [ob, ids], [tg], clip_norm=5.0, warmup=2,
                            overlap=overlap, force_collectives=overlap, bucket_bytes=256 << 10)
    if overlap:
        assert step.overlap and len(step.reducer.buckets) >= 3        # several collectives inside the graph
    losses = [step([ob, ids], [tg]).item() for ob, tg, ids in batches]
    torch.cuda.synchronize()
    results.append((losses, [p.detach().clone() for p in model.parameters()]))
assert results[0][0] == results[1][0], (results[0][0], results[1][0])
for a, b in zip(results[0][1], results[1][1]):
    assert torch.equal(a, b)
torch.cuda.synchronize()
dist.destroy_process_group()
print("DP-GRAPH-OK")
'''


def test_graph_with_collectives_equals_plain_graph():
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=110,
                         cwd=ROOT)
    assert out.returncode == 0 and "DP-GRAPH-OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
