"""Row (e), GPU: the data-parallel step captured as ONE HIP graph with its RCCL collectives inside
(``GraphedTrainStep(overlap=True)``), exercised on one GPU through a one-rank RCCL group with the collectives forced
on: the average over one rank is the identity, so weights after K replays must equal those of the plain captured
step (no collectives) bit for bit -- which checks the capture of hooks, bucket packing, async all-reduce and the
deferred wait, not the xGMI transport (the driver's N>1 bench does that)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def one_rank_rccl():
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29731", world_size=1, rank=0,
                            device_id=torch.device("cuda:0"))
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _setup(seed=0):
    import unet_amd
    torch.manual_seed(seed)
    model = unet_amd.CondUNet1D(in_channels=8, out_channels=8, base=32, n_odors=4, emb_dim=32, n_downsample=2,
                                conv_type="modern", attention_type="cross_freq_v2", norm_type="batch",
                                cond_mode="cross_attn_gated").cuda()
    model.set_compute_dtype(torch.bfloat16)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)
    return model, opt


def test_graph_with_collectives_equals_plain_graph(one_rank_rccl):
    from unet_amd.graphs import GraphedTrainStep
    g = torch.Generator(device="cuda").manual_seed(1)
    batches = [(torch.randn(4, 8, 512, device="cuda", generator=g), torch.randn(4, 8, 512, device="cuda", generator=g),
                torch.randint(0, 4, (4,), device="cuda", generator=g)) for _ in range(4)]
    results = []
    for overlap in (False, True):
        model, opt = _setup()
        ob, tg, ids = batches[0]
        step = GraphedTrainStep(model, F.l1_loss, opt, [ob, ids], [tg], clip_norm=5.0, warmup=2,
                                overlap=overlap, force_collectives=overlap, bucket_bytes=256 << 10)
        if overlap:
            assert step.overlap and len(step.reducer.buckets) >= 3        # several collectives inside the graph
        losses = []
        for ob, tg, ids in batches:
            losses.append(step([ob, ids], [tg]).item())
        torch.cuda.synchronize()
        results.append((losses, [p.detach().clone() for p in model.parameters()]))
    assert results[0][0] == results[1][0]
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)
