"""Bit-exact window indexing and fold assignment (no GPU): unet_amd.windows / unet_amd.folds against
tests/golden/host_logic.json, which oracle/make_golden_host.py froze from the reference's own classes
(SlidingWindowDataset, ContinuousLFPDataset, MultiSessionContinuousDataset, ECoGDataset,
ContinuousInference.process_continuous, SessionSplitter, load_or_create_session_splits)."""
import json
import os

import numpy as np
import pytest

from unet_amd import folds, windows

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic.json")))


def test_window_counts_and_starts_match_reference():
    for c in G["window_cases"]:
        n, w, s = c["n"], c["w"], c["s"]
        assert windows.n_windows_phase5(n, w, s) == c["phase5_len"], c
        assert windows.n_windows_dataset(n, w, s) == c["data_len"], c
        assert windows.coverage(n, w, s) == c["coverage"], c
        ni = windows.n_windows_inference(n, w, s)
        assert [windows.window_start(i, s) for i in range(ni)] == c["infer_starts"], c


def test_multisession_and_ecog_window_tables():
    assert [list(t) for t in windows.multisession_mapping([12000, 4000, 20001], 5000, 2500)] == G["multisession_mapping"]
    assert [list(t) for t in windows.ecog_windows([11000, 30000], 5000, 5000)] == G["ecog_windows"]
    for n, sp in G["loso_split_points"]:
        assert windows.loso_temporal_split_point(n, 0.3) == sp


def test_window_shards_partition_the_recording():
    for total, world in [(119, 8), (5, 8), (0, 4), (64, 1), (13, 4)]:
        sh = windows.shard_windows(total, world, 5000, 2500)
        assert sum(s.n_windows for s in sh) == total
        nxt = 0
        for s in sh:
            if s.n_windows:
                assert s.first_window == nxt
                nxt += s.n_windows
        assert max(s.n_windows for s in sh) - min(s.n_windows for s in sh) <= 1


def test_session_splitter_matches_reference():
    g = G["session_splitter"]
    sp = folds.SessionSplitter(np.array(g["session_ids"]), 0.1, 0.2, 42)
    a = sp.intra_session()
    assert (a.train_indices, a.val_indices, a.test_indices) == (g["intra"]["train"], g["intra"]["val"], g["intra"]["test"])
    b = sp.inter_session()
    assert (b.train_indices, b.val_indices, b.test_indices) == (g["inter"]["train"], g["inter"]["val"], g["inter"]["test"])
    assert b.train_sessions == g["inter"]["train_sessions"] and b.test_sessions == g["inter"]["test_sessions"]
    c = sp.leave_one_session_out(3)
    assert (c.train_indices, c.val_indices, c.test_indices) == (g["loso3"]["train"], g["loso3"]["val"], g["loso3"]["test"])
    assert [int(s) for s in c.train_sessions] == g["loso3"]["train_sessions"]


def test_session_splits_match_reference():
    sid = np.array(G["session_splitter"]["session_ids"])
    names = {i: f"sess{i:02d}" for i in range(7)}
    tr, va, te, _ = folds.session_splits(sid, n_test_sessions=1, n_val_sessions=2, seed=42, idx_to_session=names)
    g = G["session_split_random"]
    assert (tr.tolist(), va.tolist(), te.tolist()) == (g["train"], g["val"], g["test"])
    tr, va, te, _ = folds.session_splits(sid, seed=45, idx_to_session=names, test_sessions=["sess04"])
    g = G["session_split_explicit"]
    assert (tr.tolist(), va.tolist(), te.tolist()) == (g["train"], g["val"], g["test"])
    tr, va, te, per = folds.session_splits(sid, n_val_sessions=2, seed=7, idx_to_session=names, no_test_set=True,
                                           separate_val_sessions=True)
    g = G["session_split_noval"]
    assert tr.tolist() == g["train"] and te.tolist() == g["test"] and va.tolist() == g["val"]
    assert {k: v.tolist() for k, v in per.items()} == g["val_per_session"]
    with pytest.raises(ValueError, match="not found"):
        folds.session_splits(sid, idx_to_session=names, test_sessions=["nope"])


def test_loso_schedule_seeds_and_guard():
    sessions = ["s141208", "s141209", "s160819", "s170608"]
    jobs = folds.loso_jobs(sessions, seed=42, n_seeds=3)
    assert len(jobs) == 12
    assert [(j.fold_idx, j.seed_idx) for j in jobs] == [(f, s) for f in range(4) for s in range(3)]
    assert [j.run_seed for j in jobs] == [42 + f * 3 + s for f in range(4) for s in range(3)]   # LOSO/runner.py:619
    assert jobs[4].test_session == "s141209" and "s141209" not in jobs[4].train_sessions
    assert jobs[4].result_name == "fold_1_s141209_seed1_results.json"                            # LOSO/runner.py:1128
    assert [j.fold_idx for j in folds.loso_jobs(sessions, folds_to_run=[2, 0], completed_folds=[0])] == [2, 2, 2]
    q = folds.assign_jobs(jobs, 8)
    assert [len(x) for x in q] == [2, 2, 2, 2, 1, 1, 1, 1]
    assert sorted((j.fold_idx, j.seed_idx) for x in q for j in x) == sorted((j.fold_idx, j.seed_idx) for j in jobs)
    with pytest.raises(RuntimeError, match="DATA LEAKAGE"):
        folds.check_data_leakage(0, "a", ["a", "b"])
    with pytest.raises(RuntimeError, match="No training"):
        folds.check_data_leakage(0, "a", [])
    agg = folds.aggregate_seed_results([0.5, 0.3, None], [0.7, 0.6, 0.8], [1.0, float("inf"), 2.0])
    assert abs(agg["val_r2"] - 0.4) < 1e-12 and abs(agg["val_loss"] - 1.5) < 1e-12 and abs(agg["std_r2"] - 0.1) < 1e-12
    f = folds.aggregate_folds([0.4, 0.6], [0.0, 0.5], [1.0, 2.0])
    assert abs(f["mean_r2"] - 0.5) < 1e-12 and abs(f["std_r2"] - 0.1) < 1e-12 and f["mean_corr"] == 0.5
