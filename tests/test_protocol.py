"""Checkpoint / early-stop / results-JSON protocol (unet_amd/checkpoint.py) against the layout frozen from
the reference's save_checkpoint / load_checkpoint (tests/golden/checkpoint_layout.json) and the protocol
oracle (oracle/protocol_oracle.py)."""
import json
import math
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import protocol_oracle as PO
from unet_amd import checkpoint as ck

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_layout.json")))


def _tiny():
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    model(torch.randn(5, 4)).sum().backward()
    opt.step()
    return model, opt


def test_checkpoint_layout_matches_reference(tmp_path):
    model, opt = _tiny()
    config = {"cond_mode": "cross_attn_gated", "lr": 1e-3, "checkpoint_prefix": "fold3", "base_channels": 256,
              "_session_matcher": object()}          # live objects are not data: left out
    data = {"split_info": {"test_sessions": ["s1"], "n_val": 2}, "val_idx": np.arange(3), "test_idx": np.array([7, 8]),
            "train_idx": [0, 1, 2]}
    ck.save_checkpoint(model, opt, 7, tmp_path / "sub" / "a.pt", config=config, data=data)
    got = torch.load(tmp_path / "sub" / "a.pt", map_location="cpu", weights_only=True)
    assert sorted(got.keys()) == GOLD["keys"]
    assert got["epoch"] == GOLD["epoch"] and got["cond_mode"] == GOLD["cond_mode"]
    assert got["config"] == GOLD["config"] and got["split_info"] == GOLD["split_info"]
    assert list(got["model"].keys()) == GOLD["model_keys"]
    assert sorted(got["optimizer"].keys()) == GOLD["optimizer_keys"]
    assert sorted(got["optimizer"]["state"][0].keys()) == GOLD["optimizer_state_keys"]
    ck.save_checkpoint(model, opt, 1, tmp_path / "b.pt")
    got = torch.load(tmp_path / "b.pt", map_location="cpu", weights_only=True)
    assert sorted(got.keys()) == GOLD["keys_minimal"] and got["cond_mode"] == GOLD["cond_mode_minimal"]
    assert ck.checkpoint_name(config) == "fold3_best_model.pt" and ck.checkpoint_name(None) == "best_model.pt"


def test_load_checkpoint_behaviour(tmp_path):
    model, opt = _tiny()
    ck.save_checkpoint(model, opt, 7, tmp_path / "a.pt", config={"cond_mode": "cross_attn_gated"})
    m2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    o2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    wrapped = type("W", (), {"module": m2})()            # DDP-style wrapper: hasattr(model, 'module')
    assert ck.load_checkpoint(tmp_path / "a.pt", wrapped, expected_cond_mode="cross_attn_gated", optimizer=o2) == GOLD["load_returns"]
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values())) == GOLD["loaded_equal"]
    assert torch.equal(o2.state_dict()["state"][0]["exp_avg"], opt.state_dict()["state"][0]["exp_avg"])
    with pytest.raises(ValueError) as e:
        ck.load_checkpoint(tmp_path / "a.pt", m2, expected_cond_mode="film")
    assert str(e.value) == GOLD["mismatch_error"]
    with pytest.raises(FileNotFoundError) as e:
        ck.load_checkpoint(tmp_path / "missing.pt", m2)
    assert str(e.value).replace(str(tmp_path), "<dir>") == GOLD["missing_error"]
    # strict loading (train.py:545-550): a model with other keys is refused
    with pytest.raises(RuntimeError):
        ck.load_checkpoint(tmp_path / "a.pt", torch.nn.Sequential(torch.nn.Linear(4, 3)))


def test_early_stopping_matches_oracle():
    rnd = random.Random(3)
    seqs = [[1.0, 0.9, 0.9, 0.95, 0.8, 0.81, 0.82, 0.83], [float("inf"), float("inf"), 1.0], [float("nan"), 1.0, float("nan"), 0.5],
            [0.5] * 12, []]
    seqs += [[rnd.choice([0.1, 0.2, 0.3, 0.4]) + rnd.random() * 0.01 * rnd.choice([0, 1]) for _ in range(rnd.randint(1, 40))]
             for _ in range(50)]
    for seq in seqs:
        for patience in (1, 3, 8):
            want = PO.early_stop_trace(seq, patience)
            es = ck.EarlyStopping(patience)
            got = []
            for epoch, v in enumerate(seq, start=1):
                is_best = es.update(es.sync_val_loss(v, "cpu"), epoch)
                stop = es.should_stop("cpu")
                got.append((is_best, es.best_val_loss, es.best_epoch, es.patience_counter, stop))
                if stop:
                    break
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert g[0] == w[0] and g[2:] == w[2:]
                assert g[1] == w[1] or (math.isnan(g[1]) and math.isnan(w[1]))


def test_results_json_matches_oracle(tmp_path):
    rnd = random.Random(5)
    for trial in range(20):
        n = rnd.randint(0, 6)
        history = []
        for e in range(1, n + 1):
            tm = {"loss": rnd.random(), "l1_fwd": rnd.random()}
            vm = {"loss": rnd.random(), "r2": rnd.random(), "corr": rnd.random(), "mae": rnd.random()}
            per = {"sessA": {"r2": rnd.random(), "corr": rnd.random(), "loss": rnd.random()},
                   "s_corr_x": {"r2": rnd.random()}} if trial % 2 else None
            h = ck.history_entry(e, tm, vm, per)
            if trial % 3 == 0:
                h["gate_mean"] = rnd.random()
            history.append(h)
        results = {"best_val_loss": rnd.random(), "best_epoch": rnd.randint(0, n + 1), "history": history,
                   "test_metrics": rnd.choice([{}, {"r2": 0.3, "corr": 0.5, "loss": 0.7}, {"test_avg_r2": 0.0, "r2": 0.25,
                                                "test_avg_corr": 0.6, "test_avg_delta": 0.1, "matched_r2": 0.2}]),
                   "n_parameters": 123, "total_time": 4.5}
        fold = rnd.choice([None, 0, 3])
        assert ck.results_json(results, "condunet", fold) == PO.results_payload(results, "condunet", fold)
    ck.write_results_json(results, tmp_path / "o" / "r.json", "condunet", 2)
    back = json.load(open(tmp_path / "o" / "r.json"))
    assert back["fold"] == 2 and back["completed_successfully"] is True and back["epochs_trained"] == len(history)


def test_history_entry_layout():
    h = ck.history_entry(3, {"loss": 1.0}, {"loss": 2.0, "r2": 0.5}, {"m1": {"r2": 0.1, "corr": 0.2}})
    assert h == {"epoch": 3, "loss": 1.0, "val_loss": 2.0, "val_r2": 0.5, "val_m1_r2": 0.1, "val_m1_corr": 0.2}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stop_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    es = ck.EarlyStopping(patience=2)
    # ranks see different shard losses; the averaged loss drives every rank identically, and the stop decision is rank 0's
    mine = [[1.0, 0.8, 0.9, 0.7, 0.9, 0.9], [1.0, 0.6, 0.9, 1.1, 0.9, 0.9]][rank]
    trace = []
    for epoch, v in enumerate(mine, start=1):
        v = es.sync_val_loss(v, "cpu")
        best = es.update(v, epoch)
        if rank == 1:
            es.patience_counter = 0            # a rank whose counter drifted still stops with rank 0
        stop = es.should_stop("cpu")
        trace.append((round(v, 6), best, stop))
        if stop:
            break
    q.put((rank, trace, es.best_epoch))
    dist.barrier()
    dist.destroy_process_group()


def test_early_stop_is_collective_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (t, b)) for r, t, b in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0]                               # same averaged losses, same stop epoch
    assert [v for v, _, _ in res[0][0]] == [1.0, 0.7, 0.9, 0.9]
    assert res[0][0][-1][2] is True and len(res[0][0]) == 4 and res[0][1] == 2
