"""Data-parallel gradient exchange (unet_amd/dp.py) on 2 CPU ranks over gloo: bucketed, hook-driven
all-reduce must equal the average of the per-rank gradients (what DDP computes, train.py:2161)
including parameters that receive no gradient on a step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dp import BucketedGradAllReduce, broadcast_parameters
    model = _model()
    unused = torch.nn.Parameter(torch.ones(5))           # never used in forward
    params = list(model.parameters()) + [unused]
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                              # ranks start different ...
    broadcast_parameters(model)                          # ... and are made equal to rank 0
    red = BucketedGradAllReduce(params, bucket_bytes=600)  # tiny buckets -> several messages
    assert len(red.buckets) > 2
    out = []
    for step in range(2):
        red.zero_grad()
        torch.manual_seed(100 + 10 * step + rank)
        x = torch.randn(4, 6)
        model(x).pow(2).sum().backward()
        red.finish()
        out.append([p.grad.numpy().copy() for p in params])
    q.put((rank, out, [p.detach().numpy().copy() for p in model.parameters()]))  # by value
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_mean_of_rank_gradients():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, grads, weights = q.get(timeout=120)
        res[rank] = ([[torch.from_numpy(g) for g in st] for st in grads], [torch.from_numpy(w) for w in weights])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds the same weights and the same reduced gradients
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    for step in range(2):
        for a, b in zip(res[0][0][step], res[1][0][step]):
            assert torch.allclose(a, b, atol=1e-7)
    # and they equal the mean of single-process gradients
    for step in range(2):
        expect = None
        for rank in range(world):
            model = _model()
            torch.manual_seed(100 + 10 * step + rank)
            x = torch.randn(4, 6)
            model(x).pow(2).sum().backward()
            g = [p.grad.clone() for p in model.parameters()]
            expect = g if expect is None else [e + gi for e, gi in zip(expect, g)]
        expect = [e / world for e in expect]
        got = res[0][0][step]
        for e, gi in zip(expect, got[:-1]):
            assert torch.allclose(e, gi, atol=1e-6)
        assert float(got[-1].abs().max()) == 0.0      # unused parameter: zeros, still reduced


def test_single_process_is_a_no_op():
    from unet_amd.dp import BucketedGradAllReduce
    model = _model()
    red = BucketedGradAllReduce(model.parameters())
    red.zero_grad()
    model(torch.randn(3, 6)).sum().backward()
    red.finish()
    ref = _model()
    torch.manual_seed(0)
    assert all(p.grad is not None for p in model.parameters())
    assert red.world == 1


def _loso_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.trainer import run_loso
    ran = []

    def run_fold(job):
        ran.append((job.fold_idx, job.seed_idx))
        return {"val_r2": 0.1 * job.fold_idx + 0.01 * job.seed_idx, "val_corr": 0.5, "val_loss": float(job.run_seed)}

    out = run_loso(["a", "b", "c"], run_fold, seed=42, n_seeds=2, schedule="static")
    q.put((rank, ran, out["summary"], {k: v["val_r2"] for k, v in out["per_fold"].items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_fold_parallel_loso_two_ranks():
    """LOSO jobs dealt to 2 ranks: every (fold, seed) runs exactly once, both ranks aggregate the same
    result, equal to the serial (reference-order) aggregation."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loso_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, ran, summary, per = q.get(timeout=120)
        res[rank] = (ran, summary, per)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][0] + res[1][0]) == [(f, s) for f in range(3) for s in range(2)]
    assert res[0][0] == [(0, 0), (1, 0), (2, 0)] and res[1][0] == [(0, 1), (1, 1), (2, 1)]
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert abs(res[0][2][1] - 0.105) < 1e-12
    assert abs(res[0][1]["mean_r2"] - 0.105) < 1e-12
    assert abs(res[0][1]["mean_loss"] - (42 + 43 + 44 + 45 + 46 + 47) / 6) < 1e-12


def _loso_dynamic_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.trainer import run_loso
    ran = []

    def run_fold(job):
        ran.append((job.fold_idx, job.seed_idx))
        time.sleep(0.6 if rank == 1 else 0.05)            # rank 1's folds "train" 12x longer (no early stop)
        return {"val_r2": 0.1 * job.fold_idx + 0.01 * job.seed_idx, "val_corr": 0.5, "val_loss": float(job.run_seed)}

    dist.barrier()
    out = run_loso(["a", "b", "c", "d"], run_fold, seed=42, n_seeds=2)        # 8 jobs, dynamic (default)
    ran_first, ran[:] = list(ran), []
    out2 = run_loso(["a", "b"], run_fold, seed=1, n_seeds=1)                    # a second run: fresh counter
    q.put((rank, ran_first, list(ran), out["summary"], {k: v["val_r2"] for k, v in out["per_fold"].items()},
           out2["summary"]))
    dist.barrier()
    dist.destroy_process_group()


def test_fold_parallel_loso_dynamic_queue_two_ranks():
    """Default schedule: jobs are pulled from a shared counter in reference order.  Every (fold, seed) runs exactly
    once, the rank whose jobs finish early takes more of them, each rank's own sequence stays in reference order, and
    both ranks aggregate the serial result."""
    from unet_amd.trainer import run_loso
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loso_dynamic_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        got = q.get(timeout=120)
        res[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    jobs8 = [(f, s) for f in range(4) for s in range(2)]
    assert sorted(res[0][0] + res[1][0]) == jobs8                  # first run: each of the 8 jobs exactly once
    assert sorted(res[0][1] + res[1][1]) == [(0, 0), (1, 0)]       # second run: its own counter
    for rank in range(world):
        seq = [jobs8.index(j) for j in res[rank][0]]
        assert seq == sorted(seq)                                  # reference order within a rank
    assert len(res[0][0]) > len(res[1][0])        # the fast rank pulled more jobs: no idle GPU behind a static split
    serial = run_loso(["a", "b", "c", "d"], lambda job: {"val_r2": 0.1 * job.fold_idx + 0.01 * job.seed_idx,
                                                         "val_corr": 0.5, "val_loss": float(job.run_seed)},
                      seed=42, n_seeds=2)
    for rank in range(world):
        assert res[rank][2] == serial["summary"]
        assert res[rank][3] == {k: v["val_r2"] for k, v in serial["per_fold"].items()}
    assert res[0][4] == res[1][4]


def _coalesce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dp import coalesced_grad_mean
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.LayerNorm(7), torch.nn.Linear(7, 2))
    extra = torch.nn.Parameter(torch.ones(3))               # no gradient on any rank: sent as zeros, comes back as zeros
    torch.manual_seed(50 + rank)
    m(torch.randn(6, 5)).square().sum().backward()
    local = [p.grad.clone() for p in m.parameters()]
    coalesced_grad_mean(list(m.parameters()) + [extra])
    q.put((rank, [g.numpy().copy() for g in local], [p.grad.numpy().copy() for p in m.parameters()],
           extra.grad is not None and float(extra.grad.abs().max()) == 0.0))
    dist.barrier()
    dist.destroy_process_group()


def test_coalesced_grad_mean_two_ranks():
    """One flattened all-reduce == the reference's per-parameter sync_gradients_manual (train.py:738-758)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_coalesce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, local, reduced, none_kept = q.get(timeout=120)
        res[rank] = (local, reduced, none_kept)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i in range(len(res[0][0])):
        want = (res[0][0][i] + res[1][0][i]) / 2
        assert torch.allclose(torch.from_numpy(res[0][1][i]), torch.from_numpy(want), atol=1e-7)
        assert (res[0][1][i] == res[1][1][i]).all()
    assert res[0][2] and res[1][2]


# ---------------------------------------------------------------------------------------------------------------
# fold-parallel LOSO with a real Trainer per job: jobs must not join the WORLD collectives (dp.NO_GROUP)
# ---------------------------------------------------------------------------------------------------------------
class _TinyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))

    def forward(self, x, odor_ids=None):
        return self.net(x)


def _fold_job(job, ckpt_dir=None):
    """One LOSO job on a tiny torch model: Trainer steps, early-stop bookkeeping, a checkpoint.  Deterministic in
    the job's seed, so a serial run and a fold-parallel run must produce the same weights."""
    import torch.nn.functional as F
    from unet_amd.checkpoint import EarlyStopping, save_checkpoint
    from unet_amd.dp import NO_GROUP
    from unet_amd.trainer import Trainer
    torch.manual_seed(job.run_seed)
    model = _TinyModel()
    tr = Trainer(model, lr=1e-2, compute_dtype=None, process_group=NO_GROUP, loss_fn=F.l1_loss,
                 normalize_target=lambda t: t)
    assert tr.world == 1 and tr.reducer.world == 1
    stopper = EarlyStopping(2, NO_GROUP)
    g = torch.Generator().manual_seed(1000 + job.run_seed)
    losses = []
    for epoch in range(1, 4 + job.fold_idx):          # different folds run different numbers of steps
        x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
        loss = float(tr.step(x, y))
        losses.append(loss)
        val = stopper.sync_val_loss(loss, x.device)
        if stopper.update(val, epoch) and ckpt_dir is not None:
            save_checkpoint(model, tr.opt, epoch, os.path.join(ckpt_dir, f"f{job.fold_idx}s{job.seed_idx}.pt"),
                            process_group=NO_GROUP)
        if stopper.should_stop(x.device):
            break
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    return {"val_r2": float(flat.sum()), "val_corr": float(flat.abs().sum()), "val_loss": losses[-1]}


def _loso_trainer_worker(rank, world, port, q, ckpt_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.trainer import run_loso
    out = run_loso(["a", "b", "c"], lambda job: _fold_job(job, ckpt_dir), seed=7, n_seeds=1)   # 3 jobs on 2 ranks
    q.put((rank, out["per_fold"], out["summary"]))
    dist.barrier()
    dist.destroy_process_group()


def test_fold_parallel_loso_with_real_trainers_equals_serial(tmp_path):
    """Uneven job counts (3 jobs, 2 ranks), every job building its own Trainer / EarlyStopping / checkpoint with
    dp.NO_GROUP: no job may broadcast weights, average gradients or wait for the other rank (it would deadlock or
    mix folds).  Each rank's results equal the serial run's, and every job wrote its own checkpoint."""
    from unet_amd.folds import loso_jobs
    from unet_amd.trainer import run_loso
    serial = run_loso(["a", "b", "c"], _fold_job, seed=7, n_seeds=1)      # no process group here: the reference order
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loso_trainer_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, per_fold, summary = q.get(timeout=180)
        res[rank] = (per_fold, summary)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        assert res[rank][0] == serial["per_fold"], rank
        assert res[rank][1] == serial["summary"], rank
    jobs = loso_jobs(["a", "b", "c"], 7, 1, None)
    assert sorted(os.listdir(tmp_path)) == sorted(f"f{j.fold_idx}s{j.seed_idx}.pt" for j in jobs)


# ---------------------------------------------------------------------------------------------------------------
# sliding-window inference: ranks that SHARE a recording (default group) and ranks that do not (dp.NO_GROUP)
# ---------------------------------------------------------------------------------------------------------------
def _overlap_average(preds, first, total, stride, n_samples, out):
    """numpy restatement of the overlap accumulation (phase_five/sliding_window.py:238-251), raw sums only"""
    for i, p in enumerate(preds):
        s = (first + i) * stride
        out[:, s:s + p.shape[1]] += p


def _stub_inference(signal, scale, window_size, stride, process_group):
    """The host logic of infer.continuous_inference (plan_windows -> per-rank window range -> raw sums ->
    merge_seams -> division by the geometric count) around a stub model ``x -> scale * x + 1`` on CPU tensors."""
    import numpy as np
    from unet_amd.infer import merge_seams, plan_windows
    n_ch, n_samples = signal.shape
    total, shard, world = plan_windows(n_samples, window_size, stride, process_group)
    out = np.zeros((n_ch, n_samples), dtype=np.float64)
    preds = [scale * signal[:, (shard.first_window + i) * stride:(shard.first_window + i) * stride + window_size] + 1.0
             for i in range(shard.n_windows)]
    _overlap_average(preds, shard.first_window, total, stride, n_samples, out)
    t = torch.from_numpy(out)
    merge_seams(t, process_group)
    counts = np.zeros(n_samples)
    for i in range(total):
        counts[i * stride:i * stride + window_size] += 1
    return t.numpy() / np.maximum(counts, 1), shard, world


def _infer_worker(rank, world, port, q):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dp import NO_GROUP
    rng = np.random.default_rng(0)
    shared = rng.standard_normal((3, 1037))
    own = np.random.default_rng(10 + rank).standard_normal((3, 900 + 50 * rank))     # this rank's fold recording
    # 1. fold-parallel: every rank runs ITS fold's model over ITS recording, lengths differ; nothing may be exchanged
    solo, shard, w = _stub_inference(own, 2.0 + rank, 100, 50, NO_GROUP)
    assert w == 1 and shard.first_window == 0
    # 2. the ranks share one recording
    both, shard2, w2 = _stub_inference(shared, 3.0, 100, 30, None)
    q.put((rank, solo, both, (shard2.first_window, shard2.n_windows), w2))
    dist.barrier()
    dist.destroy_process_group()


def test_sliding_window_inference_group_handling_two_ranks():
    """infer.plan_windows / merge_seams: under dp.NO_GROUP a rank processes its whole recording alone (different
    recordings and window counts per rank: a collective would hang or mix folds); with the default group two ranks
    split one recording's windows and one all-reduce of the raw sums gives the serial result on both."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_infer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, solo, both, shard, w2 = q.get(timeout=120)
        res[rank] = (solo, both, shard, w2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shared = np.random.default_rng(0).standard_normal((3, 1037))
    serial_shared, _, w = _stub_inference(shared, 3.0, 100, 30, None)          # no process group here: one rank
    assert w == 1
    for rank in range(world):
        own = np.random.default_rng(10 + rank).standard_normal((3, 900 + 50 * rank))
        serial_own, _, _ = _stub_inference(own, 2.0 + rank, 100, 50, None)
        assert np.array_equal(res[rank][0], serial_own), rank
        assert np.allclose(res[rank][1], serial_shared, rtol=0, atol=1e-12), rank
        assert res[rank][3] == 2
    total = (1037 - 100) // 30 + 1
    assert res[0][2][0] == 0 and res[0][2][0] + res[0][2][1] == res[1][2][0] and res[1][2][0] + res[1][2][1] == total


# ---------------------------------------------------------------------------------------------------------------
# hoisted (late-gradient) parameters travel in the tail bucket
# ---------------------------------------------------------------------------------------------------------------
class _HoistedBlock(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 6)
        self.map = torch.nn.Linear(3, 6)          # the block's embedding map (FiLM k/v/gate in CondUNet1D)


class _HoistedModel(torch.nn.Module):
    """Blocks whose embedding maps are evaluated for ALL blocks in front of the first block, as CondUNet1D.forward
    does with the FiLM projections: their gradients are produced at the very end of backward although they are
    registered inside their blocks."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.blocks = torch.nn.ModuleList([_HoistedBlock() for _ in range(4)])

    def late_grad_parameters(self):
        return [p for b in self.blocks for p in b.map.parameters()]

    def forward(self, x, emb):
        pre = torch.cat([b.map(emb) for b in self.blocks], dim=1)      # ONE node for every block's map
        pre = pre.tanh().split(6, dim=1)
        for b, g in zip(self.blocks, pre):
            x = torch.tanh(b.lin(x)) * g
        return x


def _hoist_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dp import BucketedGradAllReduce
    out = {}
    for tag, use_late in (("late", True), ("plain", False)):
        model = _HoistedModel()
        late = model.late_grad_parameters()
        # a module argument makes the reducer ask the model; an explicit empty list switches the tail placement off
        red = BucketedGradAllReduce(model, bucket_bytes=400, tail_bytes=0, late_params=None if use_late else [])
        log = []
        launch = red._launch

        def spy(b, launch=launch, red=red, log=log, late=late):
            if not red._launched[b]:
                log.append((b, sum(p.grad is None for p in late)))      # late gradients still missing at launch time
            launch(b)

        red._launch = spy
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        for step in range(2):
            red.zero_grad()
            torch.manual_seed(7 + 10 * step + rank)
            model(torch.randn(5, 6), torch.randn(5, 3)).pow(2).sum().backward()
            red.finish()
            opt.step()
        tail = len(red._groups) - 1 if use_late else None
        out[tag] = (log, tail, [id(p) for p in red._groups[-1]] == [id(p) for p in reversed(late)] if use_late else None,
                    [p.detach().numpy().copy() for p in model.parameters()])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_hoisted_parameters_ride_the_tail_bucket_and_equal_the_serial_mean_step():
    """With the FiLM hoist on in a data-parallel job (late_grad_parameters), every block's bucket must launch while
    backward is still running -- i.e. before the hoisted maps' gradients exist -- and the step must equal the serial
    step on the mean gradient."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hoist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, out = q.get(timeout=120)
        res[rank] = out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        log, tail, tail_is_late, _ = res[rank]["late"]
        assert tail_is_late
        steps = [log[:len(log) // 2], log[len(log) // 2:]]
        for st in steps:
            assert [b for b, _ in st] == list(range(tail + 1))          # reverse-registration order, tail last
            assert all(missing == 8 for b, missing in st[:-1])          # launched before ANY late gradient existed
            assert st[-1][1] == 0
        plain = res[rank]["plain"][0]
        # without the tail placement most buckets hold some block's map and wait for the end of backward
        half = plain[:len(plain) // 2]
        assert sum(missing < 8 for _, missing in half[:-1]) >= len(half) // 2
    # serial reference: mean gradient of the two ranks' batches, same SGD steps
    model = _HoistedModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for step in range(2):
        grads = None
        for rank in range(world):
            model.zero_grad()
            torch.manual_seed(7 + 10 * step + rank)
            model(torch.randn(5, 6), torch.randn(5, 3)).pow(2).sum().backward()
            g = [p.grad.clone() for p in model.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(model.parameters(), grads):
            p.grad = g / world
        opt.step()
    for tag in ("late", "plain"):
        for rank in range(world):
            for p, w in zip(model.parameters(), res[rank][tag][3]):
                assert torch.allclose(p.detach(), torch.from_numpy(w), atol=1e-6), (tag, rank)
