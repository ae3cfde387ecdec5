"""Data-parallel training step, evaluation and LOSO fold scheduling on the HIP path.

Reproduces the per-batch semantics of the reference's ``train_epoch`` (train.py:1395-1560) and the pooled
metrics of ``evaluate`` (train.py:776-1291) for the model in scope:

    pcx = per_channel_normalize(pcx)                     train.py:1405
    pred = model(ob, odor | cond_emb=...)                train.py:1462-1464
    loss = weight_l1 * F.l1_loss(crop(pred), crop(pcx))  train.py:1481-1486
    loss.backward()  (bucketed RCCL all-reduce overlapped, replaces DDP)   train.py:1546, 2161-2181
    clip_grad_norm_(5.0); AdamW.step(); zero_grad(set_to_none)              train.py:1553-1560

One process per GPU (``torchrun``); ``Trainer`` picks up ``torch.distributed`` if it is initialised, unless it is
built with ``process_group=dp.NO_GROUP`` -- the mode for jobs that share the process group but not the model
(fold-parallel LOSO, ``run_loso``): no weight broadcast, no gradient / metric / early-stop collectives, every rank
writes its own checkpoint.
The NaN/Inf abort of train.py:1543 is kept but checked on the device and read every ``nan_check_every``
steps instead of syncing the host every step: when it raises, up to ``nan_check_every`` updates have already been
applied to non-finite gradients, i.e. the weights are invalid and the caller reloads the last checkpoint
(``nan_check_every=1`` restores the reference's check-before-every-update cost and semantics: the check then runs
before ``optimizer.step()``).

``use_graph=True`` replays one captured HIP graph per step, so everything that shapes the step is frozen at capture:
batch shapes (a different shape raises; run a last partial batch through a second, eager ``Trainer`` or drop it),
constructor options that change control flow (dropout > 0, BatchNorm ``momentum=None``).  The learning rate is NOT
frozen: it lives in a device tensor, change it with ``set_lr``.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import metrics
from .checkpoint import EarlyStopping, checkpoint_name, history_entry, save_checkpoint
from .dp import (NO_GROUP, BucketedGradAllReduce, broadcast_parameters, coalesced_grad_mean, group_active, group_rank,
                 group_size)
from .folds import LOSOJob, aggregate_folds, aggregate_seed_results, assign_jobs, loso_jobs

GRAD_CLIP = 5.0                                             # train.py:235
DEFAULT_BETAS = (0.7595905764360957, 0.920298282605139)     # train.py:186-188


class Trainer:
    def __init__(self, model: torch.nn.Module, lr: float = 2e-4, betas=DEFAULT_BETAS, weight_decay: float = 0.0,
                 weight_l1: float = 1.0, grad_clip: float = GRAD_CLIP, compute_dtype: Optional[torch.dtype] = torch.bfloat16,
                 use_graph: bool = False, bucket_bytes: int = 64 << 20, nan_check_every: int = 50, process_group=None,
                 cond_encoder: Optional[torch.nn.Module] = None, noise_augmentor: Optional[torch.nn.Module] = None,
                 loss_fn: Optional[Callable] = None, normalize_target: Optional[Callable] = None,
                 optimizer: Optional[str] = None):
        """``optimizer``: "native" (``optim.ClipAdamW``: clip + AdamW as three HIP launches, the default on a GPU) or
        "torch" (``clip_grad_norm_`` + ``torch.optim.AdamW``, the default on CPU).
        ``loss_fn(pred, target)`` / ``normalize_target(target)`` default to the reference's step
        (``weight_l1 * l1_crop_loss`` on the HIP kernels after ``per_channel_normalize``); the host-logic tests
        pass plain torch callables so that the data-parallel protocol runs on CPU under gloo."""
        self.model = model
        self._loss_fn = loss_fn
        self._normalize = normalize_target or metrics.per_channel_normalize
        self.noise_augmentor = noise_augmentor    # train.py:2310-2328; applied to the input only, in train mode
        self.cond_encoder = cond_encoder          # train.py:2026-2030 (--conditioning spectro_temporal)
        if compute_dtype is not None and hasattr(model, "set_compute_dtype"):
            model.set_compute_dtype(compute_dtype)
        self.pg = process_group
        self.world = group_size(process_group)
        broadcast_parameters(model, process_group=process_group)
        if use_graph:       # the captured optimizer step reads lr from memory: schedulers keep working (set_lr)
            lr = torch.tensor(float(lr), dtype=torch.float32, device=next(model.parameters()).device)
        groups = [{"params": list(model.parameters()), "lr": lr}]
        if cond_encoder is not None:
            broadcast_parameters(cond_encoder, process_group=process_group)
            if compute_dtype is not None and hasattr(cond_encoder, "set_compute_dtype"):
                cond_encoder.set_compute_dtype(compute_dtype)
            groups.append({"params": list(cond_encoder.parameters()), "lr": lr, "name": "cond_encoder"})  # train.py:2225-2227
        self.weight_l1, self.grad_clip = weight_l1, grad_clip
        self.use_graph = use_graph
        if optimizer is None:
            optimizer = "native" if next(model.parameters()).is_cuda else "torch"
        if optimizer == "native":     # K17: clip + AdamW in three HIP launches (unet_amd/optim.py); the model's clip runs
            from .optim import ClipAdamW   # inside step(), the encoder's stays in front of its gradient average
            groups[0]["max_grad_norm"] = grad_clip
            for g in groups[1:]:
                g["max_grad_norm"] = None
            self.opt = ClipAdamW(groups, lr=lr, betas=betas, weight_decay=weight_decay)
        elif optimizer == "torch":
            self.opt = torch.optim.AdamW(groups, lr=lr, betas=betas, weight_decay=weight_decay, fused=True,
                                         capturable=use_graph)
        else:
            raise ValueError(f"optimizer must be 'native' or 'torch', got {optimizer!r}")
        # (a module argument lets the reducer ask for the model's late-gradient parameters: the hoisted FiLM maps)
        self.reducer = None if use_graph else BucketedGradAllReduce(model, bucket_bytes, process_group)
        self._graphed = None
        self._bucket_bytes = bucket_bytes
        self.nan_check_every = nan_check_every
        self._steps = 0
        self._bad = None

    def set_lr(self, lr: float):
        """Learning-rate change that also reaches a captured step (warm-up, cosine, ReduceLROnPlateau of the
        reference's ``train()``): with ``use_graph`` the rate is a device tensor the graph reads on every replay."""
        for g in self.opt.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(float(lr))
            else:
                g["lr"] = float(lr)

    def _cond(self, ob: torch.Tensor) -> Optional[torch.Tensor]:
        """train.py:1409-1414: the encoder runs in eval() (running BatchNorm statistics, no dropout) even while
        its weights train."""
        enc = self.cond_encoder
        if enc is None:
            return None
        was_training = enc.training
        enc.eval()
        emb = enc(ob)
        enc.train(was_training)
        return emb

    def _loss(self, pred, target):
        if self._loss_fn is not None:
            return self._loss_fn(pred, target)
        return metrics.l1_crop_loss(pred, target, self.weight_l1)

    def step(self, ob: torch.Tensor, target: torch.Tensor, odor_ids: Optional[torch.Tensor] = None,
             cond_emb: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step on a batch of windows; returns the (device) loss."""
        self.model.train()
        if self.noise_augmentor is not None:      # train.py:1401-1402 (host-side coin flip: stays outside the graph)
            self.noise_augmentor.train()
            ob = self.noise_augmentor(ob)
        tgt = self._normalize(target)
        if self.use_graph:
            cond = [cond_emb] if cond_emb is not None else ([odor_ids] if odor_ids is not None else [])
            if self._graphed is None:
                from .graphs import GraphedTrainStep
                if cond_emb is not None:
                    fwd = lambda o, e: self.model(o, cond_emb=e)
                elif self.cond_encoder is not None:
                    fwd = lambda o, *i: self.model(o, *i, cond_emb=self._cond(o))
                else:
                    fwd = lambda o, *i: self.model(o, *i)
                extra = [self.cond_encoder] if (self.cond_encoder is not None and cond_emb is None) else []
                # N > 1: ONE graph with the bucket all-reduces inside it, overlapped with the captured backward.  With a
                # conditioning encoder in the step (its per-rank clip precedes its own small all-reduce) the
                # two-graph form is the one that exists.
                self._graphed = GraphedTrainStep(self.model, self._loss, self.opt, [ob] + cond, [tgt],
                                                 clip_norm=self.grad_clip, process_group=self.pg,
                                                 bucket_bytes=self._bucket_bytes, forward=fwd, extra_modules=extra,
                                                 overlap=self.world > 1 and not extra)
            loss = self._graphed([ob] + cond, [tgt])
        else:
            self.reducer.zero_grad()
            if self.cond_encoder is not None:
                for p in self.cond_encoder.parameters():
                    p.grad = None
                if cond_emb is None:
                    cond_emb = self._cond(ob)
            pred = self.model(ob, odor_ids, cond_emb=cond_emb) if cond_emb is not None else self.model(ob, odor_ids)
            loss = self._loss(pred, tgt)
            if self.nan_check_every == 1 and not bool(torch.isfinite(loss.detach())):
                raise ValueError("NaN/Inf detected in loss")      # train.py:1543-1544: before backward, weights intact
            loss.backward()
            self.reducer.finish()
            if not getattr(self.opt, "clips_gradients", False):
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
            if self.cond_encoder is not None:                      # train.py:1556-1558: clip per rank, then average
                torch.nn.utils.clip_grad_norm_(self.cond_encoder.parameters(), self.grad_clip)
                coalesced_grad_mean(list(self.cond_encoder.parameters()), self.pg)
            self.opt.step()
        bad = ~torch.isfinite(loss.detach())
        self._bad = bad if self._bad is None else (self._bad | bad)
        self._steps += 1
        if self._steps % self.nan_check_every == 0:
            self.check_finite()
        return loss

    def check_finite(self):
        if self._bad is not None and bool(self._bad):
            raise ValueError("NaN/Inf detected in loss")          # train.py:1543-1544
        self._bad = None

    @torch.no_grad()
    def evaluate(self, batches: Iterable[Sequence[torch.Tensor]], fast_mode: bool = True,
                 sampling_rate: float = 1000.0) -> Dict[str, float]:
        """Pooled metrics of train.py:776-1291 over an iterable of (ob, target, odor_ids | None) batches."""
        self.model.eval()
        stats = None
        psd_err, psd_diff, diag = [], [], None
        for batch in batches:
            ob, target = batch[0], batch[1]
            ids = batch[2] if len(batch) > 2 else None
            tgt = metrics.per_channel_normalize(target)
            emb = self._cond(ob)
            pred = self.model(ob, ids, cond_emb=emb) if emb is not None else self.model(ob, ids)
            pc, tc = metrics.crop_to_target_torch(pred).float(), metrics.crop_to_target_torch(tgt).float()
            if stats is None:
                stats = metrics.PooledStats(ob.device)
            stats.update(pc, tc)
            if not fast_mode:
                psd_err.append(metrics.psd_error_db_torch(pc, tc, fs=sampling_rate))
                psd_diff.append(metrics.psd_diff_db_torch(pc, tc, fs=sampling_rate))
                if diag is None:
                    diag = metrics.psd_diagnostics_torch(pc, tc, fs=sampling_rate)
        if stats is None:       # no batch on this rank: still join the collective the other ranks are in
            stats = metrics.PooledStats(next(self.model.parameters()).device)
        stats.all_reduce(self.pg)
        if float(stats.sums[0]) == 0.0:
            return {"mse": 0.0, "mae": 0.0, "corr": 0.0, "r2": 0.0}           # train.py:1061-1068
        out = stats.finalize()
        if psd_err:
            out["psd_err_db"] = float(torch.stack(psd_err).mean())
            out["psd_diff_db"] = float(torch.stack(psd_diff).mean())
            out["psd_diagnostics"] = diag
        out["loss"] = self.weight_l1 * out["mae"]                               # train.py:1271-1282
        return out


    def train_epoch(self, batches: Iterable[Sequence[torch.Tensor]]) -> Dict[str, float]:
        """train.py:1395-1580: mean of the per-batch losses, accumulated on the device (one host read per epoch)."""
        total, n = None, 0
        for batch in batches:
            ob, target = batch[0], batch[1]
            ids = batch[2] if len(batch) > 2 else None
            loss = self.step(ob, target, ids).detach()
            total = loss.clone() if total is None else total + loss
            n += 1
        self.check_finite()
        if n == 0:
            return {"loss": 0.0}                                              # train.py:1568-1574
        l1 = float(total.item()) / n
        return {"loss": l1, "l1_fwd": l1}

    def fit(self, train_batches: Callable[[int], Iterable], val_batches: Callable[[], Iterable], num_epochs: int,
            early_stop_patience: int = 8, checkpoint_dir=None, config: Optional[dict] = None, data: Optional[dict] = None,
            test_batches: Optional[Callable[[], Iterable]] = None) -> Dict[str, object]:
        """Epoch loop of train.py:2434-2757 for the path in scope: train, validate (pooled metrics), average the
        val loss over ranks, keep the best checkpoint (strict <), append the history row, stop after
        ``early_stop_patience`` epochs without improvement (rank 0's decision, broadcast).  ``train_batches(epoch)``
        and ``val_batches()`` return iterables of ``(ob, target[, odor_ids])`` device batches.  Returns what
        ``train()`` returns (train.py:3290-3299), ready for ``checkpoint.results_json``."""
        import time
        from pathlib import Path
        t0 = time.time()
        dev = next(self.model.parameters()).device
        stopper = EarlyStopping(early_stop_patience, self.pg)
        history: List[dict] = []
        for epoch in range(1, num_epochs + 1):
            train_metrics = self.train_epoch(train_batches(epoch))
            val_metrics = self.evaluate(val_batches())
            val_loss = val_metrics.get("loss", val_metrics.get("mae", float("inf")))
            val_loss = stopper.sync_val_loss(val_loss, dev)
            val_metrics["loss"] = val_loss
            if stopper.update(val_loss, epoch) and checkpoint_dir is not None:
                save_checkpoint(self.model, self.opt, epoch, Path(checkpoint_dir) / checkpoint_name(config), config=config,
                                data=data, process_group=self.pg)
            if group_active(self.pg):
                dist.barrier(group=self.pg)
            history.append(history_entry(epoch, train_metrics, val_metrics))
            if stopper.should_stop(dev):
                break
        test_metrics = self.evaluate(test_batches(), fast_mode=False) if test_batches is not None else {}
        return {"best_val_loss": stopper.best_val_loss, "best_epoch": stopper.best_epoch, "test_metrics": test_metrics,
                "history": history, "model": self.model, "reverse_model": None,
                "n_parameters": sum(p.numel() for p in self.model.parameters() if p.requires_grad),
                "total_time": time.time() - t0}


_LOSO_CALLS = 0          # run_loso calls in this process: every rank makes the same calls, so it names a run's keys


def _group_store(process_group):
    """The key-value store behind the process group (TCPStore for torchrun jobs): ``add`` is an atomic fetch-add that
    every rank can reach without entering a collective -- what a shared job counter needs."""
    try:
        from torch.distributed import distributed_c10d as c10d
        return c10d._get_default_store()
    except Exception:
        return None


def run_loso(all_sessions: Sequence[str], run_fold: Callable[[LOSOJob], Dict[str, float]], seed: int = 42,
             n_seeds: int = 3, folds_to_run: Optional[Sequence[int]] = None, process_group=None,
             schedule: str = "dynamic") -> Dict[str, object]:
    """Fold-parallel LOSO (LOSO/runner.py:1046-1177 runs the same jobs serially, one subprocess each).

    Every rank builds the SAME job list (reference order and seeds: ``run_seed = seed + fold * n_seeds + seed_idx``,
    LOSO/runner.py:619) and calls ``run_fold(job)`` -> ``{"val_r2", "val_corr", "val_loss"}`` (the test_avg_r2 /
    test_avg_corr / test_avg_delta of the results JSON, LOSO/runner.py:831-833) for the jobs it gets:

    * ``schedule="dynamic"`` (default): a shared counter in the process group's store hands out the jobs in
      reference order, one at a time, to whichever rank asks next -- folds stop early at very different epochs
      (patience 15), and a rank whose folds ended early pulls the next job instead of idling.  Which rank runs a job
      does not enter its result (the job carries its own seed).  Results are published in the same store as they are
      produced and collected by every rank at the end, so no rank waits inside a collective while others train.
    * ``schedule="static"``: job k goes to rank k mod world (``folds.assign_jobs``); results travel in one
      ``all_gather_object``.  Also what a group without a reachable store falls back to.

    Aggregation as in the reference: mean over seeds per fold, then mean +- std over folds.

    Ranks run DIFFERENT jobs (and possibly different numbers of them), so ``run_fold`` must not enter a collective
    of ``process_group``: build its ``Trainer`` with ``process_group=dp.NO_GROUP`` (also what ``EarlyStopping`` /
    ``save_checkpoint`` / ``PooledStats.all_reduce`` / ``infer.continuous_inference`` take), and give every job its
    own checkpoint directory.
    """
    global _LOSO_CALLS
    if schedule not in ("dynamic", "static"):
        raise ValueError(f"schedule must be 'dynamic' or 'static', got {schedule!r}")
    world = group_size(process_group)
    rank = group_rank(process_group)
    jobs = loso_jobs(all_sessions, seed, n_seeds, folds_to_run)
    store = _group_store(process_group) if (world > 1 and schedule == "dynamic") else None
    _LOSO_CALLS += 1
    if store is not None:
        import json
        from datetime import timedelta
        members = "-".join(map(str, dist.get_process_group_ranks(process_group or dist.group.WORLD)))
        prefix = f"unet_amd/loso/{members}/{_LOSO_CALLS}"          # sub-groups running their own LOSO do not collide
        ran = []
        while True:
            k = int(store.add(f"{prefix}/next", 1)) - 1          # atomic: every index is handed out exactly once
            if k >= len(jobs):
                break
            j = jobs[k]
            store.set(f"{prefix}/res/{k}", json.dumps(run_fold(j)))
            ran.append(k)
        keys = [f"{prefix}/res/{k}" for k in range(len(jobs))]
        if keys:
            store.wait(keys, timedelta(days=30))                 # other ranks' last jobs may still be training
        local = [((j.fold_idx, j.seed_idx), json.loads(store.get(key))) for j, key in zip(jobs, keys)]
    else:
        mine = assign_jobs(jobs, world)[rank]
        local = [((j.fold_idx, j.seed_idx), run_fold(j)) for j in mine]
        if world > 1:
            gathered: List[object] = [None] * world
            dist.all_gather_object(gathered, local, group=process_group)
            local = [x for part in gathered for x in part]
    res = dict(local)
    folds = sorted({j.fold_idx for j in jobs})
    per_fold = {}
    for f in folds:
        rs = [res[(f, s)] for s in range(n_seeds) if (f, s) in res]
        per_fold[f] = aggregate_seed_results([r.get("val_r2") for r in rs], [r.get("val_corr") for r in rs],
                                             [r.get("val_loss") for r in rs])
    summary = aggregate_folds([per_fold[f]["val_r2"] for f in folds], [per_fold[f]["val_corr"] for f in folds],
                              [per_fold[f]["val_loss"] for f in folds])
    return {"jobs": jobs, "per_fold": per_fold, "summary": summary}
