"""Fold / split assignment of the reference -- bit-exact (same RNG streams, same orders).

    LOSO/runner.py:577,580,619,1122-1128   LOSO job list: folds in session order x n_seeds, seed formula
    LOSO/runner.py:425-469                 leakage guard
    LOSO/runner.py:979-1039, LOSO/config.py:456-475   seed / fold aggregation
    data.py:524-735                        load_or_create_session_splits (np.random.default_rng)
    phase_four/data_splitter.py:74-235     SessionSplitter (legacy np.random.RandomState)

and the placement of LOSO jobs on the GPUs of one node (the reference runs them serially, one
``train.py`` subprocess each; jobs are independent, so they are dealt round-robin to per-GPU queues
while ids, order and seeds stay exactly the reference's).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------------------------
# LOSO schedule
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class LOSOJob:
    fold_idx: int
    seed_idx: int
    test_session: str
    train_sessions: Tuple[str, ...]
    run_seed: int
    result_name: str


def check_data_leakage(fold_idx: int, test_session: str, train_sessions: Sequence[str], session_label: str = "session"):
    """LOSO/runner.py:425-469 (same conditions, same exception type)."""
    if test_session in train_sessions:
        raise RuntimeError(
            f"CRITICAL DATA LEAKAGE DETECTED in fold {fold_idx}!\n"
            f"  Test {session_label} '{test_session}' found in training set!\n"
            f"  Training {session_label}s: {list(train_sessions)}\n"
            f"  This would invalidate all results. Aborting.")
    if not test_session:
        raise RuntimeError(f"Empty test {session_label} in fold {fold_idx}. This indicates a bug in session assignment.")
    if len(train_sessions) == 0:
        raise RuntimeError(f"No training {session_label}s in fold {fold_idx}. Cannot train without any training data.")


def loso_jobs(all_sessions: Sequence[str], seed: int = 42, n_seeds: int = 3,
              folds_to_run: Optional[Sequence[int]] = None, completed_folds: Sequence[int] = ()) -> List[LOSOJob]:
    """The (fold, seed) jobs in the reference's order (LOSO/runner.py:1103-1128): fold ``i`` holds out
    ``all_sessions[i]``; ``run_seed = seed + fold_idx * n_seeds + seed_idx`` (:619); the results file is
    ``fold_{i}_{session}_seed{j}_results.json`` (:1128)."""
    n_folds = len(all_sessions)
    if folds_to_run is not None:
        remaining = [f for f in folds_to_run if f not in completed_folds]
    else:
        remaining = [i for i in range(n_folds) if i not in completed_folds]
    jobs = []
    for fold_idx in remaining:
        test = all_sessions[fold_idx]
        train = tuple(s for s in all_sessions if s != test)
        check_data_leakage(fold_idx, test, train)
        for seed_idx in range(n_seeds):
            jobs.append(LOSOJob(fold_idx, seed_idx, test, train, seed + fold_idx * n_seeds + seed_idx,
                                f"fold_{fold_idx}_{test}_seed{seed_idx}_results.json"))
    return jobs


def assign_jobs(jobs: Sequence[LOSOJob], n_gpus: int) -> List[List[LOSOJob]]:
    """Per-GPU queues: job k (reference order) goes to GPU k mod n_gpus; each queue keeps reference order."""
    queues: List[List[LOSOJob]] = [[] for _ in range(n_gpus)]
    for k, j in enumerate(jobs):
        queues[k % n_gpus].append(j)
    return queues


def aggregate_seed_results(r2: Sequence[Optional[float]], corr: Sequence[Optional[float]],
                           loss: Sequence[Optional[float]]) -> Dict[str, float]:
    """LOSO/runner.py:999-1010."""
    r2v = [v for v in r2 if v is not None]
    cv = [v for v in corr if v is not None]
    lv = [v for v in loss if v is not None and v != float("inf")]
    return dict(
        val_r2=float(np.mean(r2v)) if r2v else 0.0,
        val_corr=float(np.mean(cv)) if cv else 0.0,
        val_loss=float(np.mean(lv)) if lv else float("inf"),
        std_r2=float(np.std(r2v)) if len(r2v) > 1 else 0.0,
        std_corr=float(np.std(cv)) if len(cv) > 1 else 0.0)


def aggregate_folds(fold_r2: Sequence[float], fold_corr: Sequence[float], fold_loss: Sequence[float]) -> Dict[str, float]:
    """LOSO/config.py:456-475 (population std; correlations equal to 0.0 are treated as missing)."""
    out = dict(mean_r2=float(np.mean(fold_r2)), std_r2=float(np.std(fold_r2)),
               mean_loss=float(np.mean(fold_loss)), std_loss=float(np.std(fold_loss)))
    cv = [c for c in fold_corr if c != 0.0]
    if cv:
        out.update(mean_corr=float(np.mean(cv)), std_corr=float(np.std(cv)))
    return out


# ---------------------------------------------------------------------------------------------
# data.py session splits
# ---------------------------------------------------------------------------------------------
def session_splits(session_ids: np.ndarray, n_test_sessions: int = 1, n_val_sessions: int = 1, seed: int = 42,
                   idx_to_session: Optional[Dict[int, str]] = None, test_sessions: Optional[List[str]] = None,
                   val_sessions: Optional[List[str]] = None, no_test_set: bool = False,
                   separate_val_sessions: bool = False):
    """data.py:524-735 with ``force_recreate=True`` (what LOSO passes, LOSO/runner.py:697): hold out whole
    sessions; ``np.random.default_rng(seed)``; shuffle sessions (random mode only), then shuffle train, val,
    test index arrays in that order (:716-718); per-session val indices are cut from the SHUFFLED val array
    (:725-733; returned in split_info["val_idx_per_session"] by the reference).  Returns
    (train_idx, val_idx, test_idx, per_session_val | None)."""
    session_ids = np.asarray(session_ids)
    rng = np.random.default_rng(seed)
    unique_sessions = np.unique(session_ids)
    n_sessions = len(unique_sessions)
    session_to_idx = {name: idx for idx, name in idx_to_session.items()} if idx_to_session is not None else {}
    explicit = test_sessions is not None or val_sessions is not None
    if explicit:
        def to_ids(names, what):
            ids = set()
            for name in names or []:
                if name not in session_to_idx:
                    raise ValueError(f"{what} session '{name}' not found. Available: {list(session_to_idx.keys())}")
                ids.add(session_to_idx[name])
            return ids
        test_ids, val_ids = to_ids(test_sessions, "Test"), to_ids(val_sessions, "Val")
        train_ids = set(unique_sessions.tolist()) - test_ids - val_ids
        if len(train_ids) == 0:
            raise ValueError("No sessions left for training after specifying test/val sessions!")
    else:
        n_holdout = n_val_sessions if no_test_set else (n_test_sessions + n_val_sessions)
        if n_holdout >= n_sessions:
            raise ValueError(f"Cannot hold out {n_holdout} sessions from only {n_sessions} total sessions. "
                             f"Need at least 1 for training.")
        shuffled = unique_sessions.copy()
        rng.shuffle(shuffled)
        if no_test_set:
            test_ids = set()
            val_ids = set(shuffled[:n_val_sessions].tolist())
            train_ids = set(shuffled[n_val_sessions:].tolist())
        else:
            test_ids = set(shuffled[:n_test_sessions].tolist())
            val_ids = set(shuffled[n_test_sessions:n_test_sessions + n_val_sessions].tolist())
            train_ids = set(shuffled[n_test_sessions + n_val_sessions:].tolist())
    all_idx = np.arange(len(session_ids))
    train_idx = all_idx[np.isin(session_ids, list(train_ids))]
    test_idx = all_idx[np.isin(session_ids, list(test_ids))]
    val_idx = all_idx[np.isin(session_ids, list(val_ids))]
    if set(train_idx.tolist()) & set(val_idx.tolist()) or set(train_idx.tolist()) & set(test_idx.tolist()) \
            or set(val_idx.tolist()) & set(test_idx.tolist()):
        raise ValueError("BUG: overlapping split indices")
    rng.shuffle(train_idx)
    rng.shuffle(val_idx)
    rng.shuffle(test_idx)
    if separate_val_sessions:
        per = {}
        for sid in val_ids:
            name = idx_to_session[sid] if idx_to_session is not None else str(sid)
            per[name] = val_idx[session_ids[val_idx] == sid]
        return train_idx, val_idx, test_idx, per
    return train_idx, val_idx, test_idx, None


# ---------------------------------------------------------------------------------------------
# phase_four SessionSplitter
# ---------------------------------------------------------------------------------------------
@dataclass
class SessionSplit:
    train_indices: List[int]
    val_indices: List[int]
    test_indices: List[int]
    train_sessions: List[Any] = field(default_factory=list)
    test_sessions: List[Any] = field(default_factory=list)


class SessionSplitter:
    """phase_four/data_splitter.py:74-235 (np.random.RandomState(random_state), re-seeded per call)."""

    def __init__(self, session_ids, val_ratio: float = 0.1, test_ratio: float = 0.2, random_state: int = 42):
        self.session_ids = np.asarray(session_ids)
        self.val_ratio, self.test_ratio, self.random_state = val_ratio, test_ratio, random_state
        self.unique_sessions = np.unique(self.session_ids)
        self.session_to_indices: Dict[Any, List[int]] = {}
        for idx, s in enumerate(self.session_ids):
            self.session_to_indices.setdefault(s, []).append(idx)

    def intra_session(self) -> SessionSplit:
        rng = np.random.RandomState(self.random_state)
        tr, va, te = [], [], []
        for s in self.unique_sessions:
            idx = np.array(self.session_to_indices[s])
            rng.shuffle(idx)
            n = len(idx)
            n_test = max(1, int(n * self.test_ratio))
            n_val = max(1, int(n * self.val_ratio))
            n_train = n - n_test - n_val
            tr.extend(idx[:n_train].tolist())
            va.extend(idx[n_train:n_train + n_val].tolist())
            te.extend(idx[n_train + n_val:].tolist())
        return SessionSplit(tr, va, te, self.unique_sessions.tolist(), self.unique_sessions.tolist())

    def inter_session(self) -> SessionSplit:
        rng = np.random.RandomState(self.random_state)
        sessions = self.unique_sessions.copy()
        rng.shuffle(sessions)
        n_test = max(1, int(len(sessions) * self.test_ratio))
        test_s, tv = sessions[:n_test], sessions[n_test:]
        n_val = max(1, int(len(tv) * self.val_ratio))
        val_s, train_s = tv[:n_val], tv[n_val:]
        gather = lambda ss: [i for s in ss for i in self.session_to_indices[s]]
        return SessionSplit(gather(train_s), gather(val_s), gather(test_s), train_s.tolist(), test_s.tolist())

    def leave_one_session_out(self, test_session) -> SessionSplit:
        rng = np.random.RandomState(self.random_state)
        train_s = [s for s in self.unique_sessions if s != test_session]
        rng.shuffle(train_s)
        if len(train_s) > 1:
            val_s, train_s = [train_s[0]], train_s[1:]
        else:
            val_s = []
        gather = lambda ss: [i for s in ss for i in self.session_to_indices[s]]
        return SessionSplit(gather(train_s), gather(val_s), list(self.session_to_indices[test_session]),
                            list(train_s), [test_session])
