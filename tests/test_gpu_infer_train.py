"""GPU: sliding-window inference (device overlap-average) against the reference's own reconstruction
(tests/golden/overlap_add.npz), the Trainer step against the oracle's training-step loss, and config C
(wide ECoG, deeper encoder) against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Toy(torch.nn.Module):
    """the stand-in model oracle/make_golden_host.py ran through the reference's continuous_inference"""
    out_channels = 3

    def forward(self, x):
        ramp = torch.linspace(-1, 1, x.shape[-1]).to(x.device)   # CPU linspace, as in the generator
        return x * 0.5 + ramp * x.roll(1, dims=1)


@pytest.mark.parametrize("batch", [1, 4, 64])
def test_continuous_inference_matches_reference(batch):
    from unet_amd.infer import continuous_inference
    z = np.load(os.path.join(GD, "overlap_add.npz"))
    rec = continuous_inference(Toy(), z["signal"], 4096, 1024, batch_windows=batch)
    assert rec.shape == z["recon"].shape and rec.dtype == np.float32
    err = np.abs(rec - z["recon"]).max()
    assert err < 2e-6, err
    # uncovered tail stays exactly zero (reference behaviour)
    nw = (14000 - 4096) // 1024 + 1
    assert np.all(rec[:, (nw - 1) * 1024 + 4096:] == 0)


@pytest.mark.parametrize("mode", ["average", "last", "first"])
def test_overlap_modes_and_chunking_vs_oracle(mode):
    from oracle import metrics_oracle as MO
    from unet_amd.infer import overlap_add
    rng = np.random.default_rng(5)
    nw, c, w, s = 9, 2, 300, 70
    n = (nw - 1) * s + w + 13
    preds = rng.standard_normal((nw, c, w)).astype(np.float32)
    out = torch.zeros((c, n), device="cuda")
    pt = torch.from_numpy(preds).cuda()
    for a, b in [(0, 4), (4, 5), (5, 9)]:          # three chunks, as three ranks / batches would contribute
        overlap_add(pt[a:b], a, nw, s, out, True, False, mode)
    if mode == "average":
        overlap_add(None, 0, nw, s, out, True, True, window_size=w)
        ref = MO.overlap_average(preds, n, s)
    else:
        ref = np.zeros((c, n), dtype=np.float32)
        order = range(nw) if mode == "last" else reversed(range(nw))
        for i in order:
            ref[:, i * s:i * s + w] = preds[i]
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)


def test_short_recording_gives_zeros():
    from unet_amd.infer import continuous_inference
    rec = continuous_inference(Toy(), np.ones((3, 100), np.float32), 4096, 1024)
    assert rec.shape == (3, 100) and not rec.any()


def test_trainer_step_matches_oracle_and_learns():
    import unet_amd
    from oracle import condunet_oracle as O
    from unet_amd.trainer import Trainer
    cfg = dict(in_channels=8, out_channels=8, base=16, emb_dim=32, conv_type="modern", attention_type="none", activation="gelu")
    sd = O.init_state_dict(cfg, seed=3)
    g = torch.Generator().manual_seed(0)
    ob, tgt = torch.randn(4, 8, 4608, generator=g), torch.randn(4, 8, 4608, generator=g) * 2 + 1
    ids = torch.tensor([0, 1, 2, 3])
    ref_loss = float(O.train_step_loss(sd, cfg, ob, tgt, odor_ids=ids))
    model = unet_amd.CondUNet1D(**cfg).cuda()
    model.load_state_dict(sd, strict=True)
    tr = Trainer(model, lr=1e-3, compute_dtype=torch.float32, nan_check_every=1)
    first = float(tr.step(ob.cuda(), tgt.cuda(), ids.cuda()))
    assert abs(first - ref_loss) < 1e-4
    for _ in range(8):
        last = float(tr.step(ob.cuda(), tgt.cuda(), ids.cuda()))
    assert last < first
    res = tr.evaluate([(ob.cuda(), tgt.cuda(), ids.cuda())], fast_mode=False)
    assert set(("mse", "mae", "corr", "r2", "psd_err_db", "psd_diff_db", "loss")) <= set(res)
    assert abs(res["loss"] - res["mae"]) < 1e-12
    with pytest.raises(ValueError, match="NaN/Inf"):
        tr.step(ob.cuda() * float("nan"), tgt.cuda(), ids.cuda())


def test_fit_checkpoint_early_stop_roundtrip(tmp_path):
    """Epoch loop protocol (train.py:2434-2757): best checkpoint, history rows, early stop; the saved file loads
    back into a fresh model (strict) and reproduces the validation metrics; results JSON carries the LOSO keys."""
    import unet_amd
    from unet_amd import checkpoint as ck
    from unet_amd.trainer import Trainer
    cfg = dict(in_channels=8, out_channels=8, base=16, emb_dim=32, conv_type="modern", attention_type="none", activation="gelu")
    torch.manual_seed(1)
    model = unet_amd.CondUNet1D(**cfg).cuda()
    with torch.no_grad():
        model.outc.weight.normal_(0, 0.05)
    g = torch.Generator().manual_seed(0)
    ob, tgt = torch.randn(4, 8, 4608, generator=g).cuda(), (torch.randn(4, 8, 4608, generator=g) * 2 + 1).cuda()
    ids = torch.tensor([0, 1, 2, 3]).cuda()
    tr = Trainer(model, lr=1e-3, compute_dtype=torch.float32)
    config = {"cond_mode": "cross_attn_gated", "checkpoint_prefix": "fold0"}
    out = tr.fit(lambda epoch: [(ob, tgt, ids)] * 2, lambda: [(ob, tgt, ids)], num_epochs=3, early_stop_patience=8,
                 checkpoint_dir=tmp_path, config=config, data={"val_idx": np.arange(4)}, test_batches=lambda: [(ob, tgt, ids)])
    assert [h["epoch"] for h in out["history"]] == [1, 2, 3]
    assert {"loss", "l1_fwd", "val_loss", "val_r2", "val_corr", "val_mae", "val_mse"} <= set(out["history"][0])
    assert out["best_epoch"] == 3 and out["best_val_loss"] == out["history"][2]["val_loss"]      # it learns every epoch
    path = tmp_path / "fold0_best_model.pt"
    assert path.exists()
    saved = torch.load(path, map_location="cpu", weights_only=True)
    assert saved["epoch"] == 3 and saved["split_info"] == {"val_idx": [0, 1, 2, 3]} and saved["cond_mode"] == "cross_attn_gated"
    fresh = unet_amd.CondUNet1D(**cfg).cuda()
    assert ck.load_checkpoint(path, fresh, expected_cond_mode="cross_attn_gated") == "cross_attn_gated"
    res = Trainer(fresh, compute_dtype=torch.float32).evaluate([(ob, tgt, ids)])
    assert abs(res["loss"] - out["best_val_loss"]) < 1e-6
    js = ck.results_json(out, "condunet", 0)
    assert js["best_epoch"] == 3 and js["epochs_trained"] == 3 and js["test_avg_r2"] == out["test_metrics"]["r2"]
    assert js["best_val_r2"] == out["history"][2]["val_r2"] and js["n_parameters"] == sum(p.numel() for p in model.parameters())
    # early stop: a learning rate of zero (and no running statistics) never improves after the first epoch
    # -> stops at epoch 1 + patience
    frozen = unet_amd.CondUNet1D(**dict(cfg, norm_type="instance")).cuda()
    tr0 = Trainer(frozen, lr=0.0, compute_dtype=torch.float32)
    out0 = tr0.fit(lambda epoch: [(ob, tgt, ids)], lambda: [(ob, tgt, ids)], num_epochs=10, early_stop_patience=2)
    assert out0["best_epoch"] == 1 and len(out0["history"]) == 3


def test_config_c_wide_deep_vs_oracle():
    """BASELINE configs[2] shape family: 128 input/output channels, n_downsample=4 (channel ladder capped at
    8*base), at a base the oracle finishes in seconds."""
    import torch.nn.functional as F
    import unet_amd
    from oracle import condunet_oracle as O
    cfg = dict(in_channels=128, out_channels=128, base=32, conv_type="modern", attention_type="none", activation="gelu",
               n_downsample=4)
    sd = O.init_state_dict(cfg, seed=5)
    g = torch.Generator().manual_seed(1)
    ob, tgt = torch.randn(2, 128, 1024, generator=g), torch.randn(2, 128, 1024, generator=g)
    ids = torch.tensor([2, 5])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    run_sd = dict(sd)
    run_sd.update(leaves)
    ref = O.forward(run_sd, cfg, ob, odor_ids=ids, training=True)
    F.l1_loss(ref[..., 100:900], tgt[..., 100:900]).backward()
    model = unet_amd.CondUNet1D(**cfg).cuda()
    model.load_state_dict(sd, strict=True)
    model.train()
    out = model(ob.cuda(), ids.cuda())
    F.l1_loss(out[..., 100:900], tgt.cuda()[..., 100:900]).backward()
    rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert rel(out, ref) < 1e-4
    worst = max(rel(p.grad, leaves[k].grad) for k, p in model.named_parameters()
                if leaves[k].grad is not None and leaves[k].grad.norm() > 1e-7)
    assert worst < 1e-3, worst


def _small_film_model():
    import unet_amd
    torch.manual_seed(3)
    model = unet_amd.CondUNet1D(in_channels=8, out_channels=8, base=32, conv_type="modern", attention_type="none",
                                activation="gelu", skip_type="add", norm_type="batch", cond_mode="cross_attn_gated",
                                n_downsample=2).cuda()
    with torch.no_grad():
        model.outc.weight.normal_(0, 0.05)
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    return model


def test_graphed_inference_replays_the_eager_forward():
    """graphs.GraphedInference: one HIP graph per input shape, bit-identical to the eager eval forward (same kernels on
    the same packed weights), new inputs copied into the static buffers, a second shape gets its own graph, and a
    parameter change (version counter) drops the graphs so that the next call sees the new weights."""
    from unet_amd import ops
    from unet_amd.graphs import GraphedInference
    model = _small_film_model()
    g = GraphedInference(model)
    ops.clear_pack_cache()
    for bs in (4, 1):
        for seed in (0, 1):
            gen = torch.Generator(device="cuda").manual_seed(seed)
            x = torch.randn(bs, 8, 1024, device="cuda", generator=gen)
            ids = torch.randint(0, 3, (bs,), device="cuda", generator=gen)
            with torch.no_grad():
                want = model(x, odor_ids=ids)
            got = g(x, odor_ids=ids)
            assert torch.equal(got, want), (bs, seed)
    assert len(g._graphs) == 2
    x = torch.randn(4, 8, 1024, device="cuda")
    ids = torch.zeros(4, dtype=torch.long, device="cuda")
    before = g(x, odor_ids=ids).clone()
    with torch.no_grad():
        model.outc.weight.mul_(2.0)                     # bumps the version counter
    after = g(x, odor_ids=ids)
    assert len(g._graphs) == 1                          # dropped and re-captured
    with torch.no_grad():
        want = model(x, odor_ids=ids)
    assert torch.equal(after, want)
    assert (after.float() - before.float()).abs().max() > 0
    model.train()
    with pytest.raises(RuntimeError):
        g(x, odor_ids=ids)
    # a training phase in between (the optimizer kernels bump no version counter; model.train() drops the pack cache,
    # and with it the graphs): weights changed behind the counters are seen after the return to eval mode
    n_cap = g.captures
    model.outc.weight.data.mul_(0.5)
    model.eval()
    again = g(x, odor_ids=ids)
    assert g.captures == n_cap + 1
    with torch.no_grad():
        assert torch.equal(again, model(x, odor_ids=ids))
    assert (again.float() - after.float()).abs().max() > 0
    ops.clear_pack_cache()


def test_sliding_window_inference_with_graph_replay_equals_eager():
    """continuous_inference(graphs=...): full batches replayed as a graph, the shorter last batch eager; same
    reconstruction as the eager run, and a GraphedInference passed in is reused from recording to recording."""
    from unet_amd.graphs import GraphedInference
    from unet_amd.infer import continuous_inference
    model = _small_film_model()
    rng = np.random.default_rng(2)
    sig = rng.standard_normal((8, 1024 * 12 + 100)).astype(np.float32)
    n_windows = (sig.shape[1] - 1024) // 512 + 1
    kw = {"odor_ids": torch.arange(n_windows, device="cuda") % 3}
    want = continuous_inference(model, sig, 1024, 512, batch_windows=8, model_kwargs=kw)
    got = continuous_inference(model, sig, 1024, 512, batch_windows=8, model_kwargs=kw, graphs=True)
    assert n_windows % 8 != 0                           # there IS a ragged last batch
    assert np.array_equal(got, want)
    g = GraphedInference(model)
    a = continuous_inference(model, sig, 1024, 512, batch_windows=8, model_kwargs=kw, graphs=g)
    b = continuous_inference(model, sig * 0.5, 1024, 512, batch_windows=8, model_kwargs=kw, graphs=g)
    assert len(g._graphs) == 1 and g.captures == 1      # captured once, replayed for both recordings
    assert np.array_equal(a, want)
    assert np.array_equal(b, continuous_inference(model, sig * 0.5, 1024, 512, batch_windows=8, model_kwargs=kw))
    with pytest.raises(ValueError):
        continuous_inference(_small_film_model(), sig, 1024, 512, graphs=g)
