"""K17: optim.ClipAdamW against clip_grad_norm_ + torch.optim.AdamW (train.py:1553-1559) on the GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BETAS = (0.7595905764360957, 0.920298282605139)      # train.py:186-188


def _params(seed, shapes, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]


SHAPES = [(256, 32, 7), (256,), (1,), (3,), (512, 256, 1), (37, 5), (1024, 130), (2, 3, 5, 7), (8193,), (4096, 33)]


@pytest.mark.parametrize("max_norm,grad_scale", [(5.0, 30.0), (5.0, 1e-3), (None, 1.0)])
@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_clip_adamw_matches_torch(max_norm, grad_scale, weight_decay):
    from unet_amd.optim import ClipAdamW
    dev = "cuda"
    pa, pb = _params(0, SHAPES, dev), _params(0, SHAPES, dev)
    ref = torch.optim.AdamW(pa, lr=2e-4, betas=BETAS, weight_decay=weight_decay)
    opt = ClipAdamW(pb, lr=2e-4, betas=BETAS, weight_decay=weight_decay, max_grad_norm=max_norm)
    g = torch.Generator(device="cpu").manual_seed(1)
    for it in range(4):
        grads = [torch.randn(p.shape, generator=g).to(dev) * grad_scale for p in pa]
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        if it == 2:                      # a parameter without a gradient is skipped (its moments do not move)
            pa[3].grad = None
            pb[3].grad = None
        total = torch.nn.utils.clip_grad_norm_(pa, max_norm) if max_norm else torch.linalg.vector_norm(
            torch.stack([torch.linalg.vector_norm(p.grad) for p in pa if p.grad is not None]))
        ref.step()
        opt.step()
        assert abs(float(opt.total_norm()) - float(total)) <= 2e-6 * float(total)
        for q, gr in zip(pb, grads):     # the gradient itself is left as it was (not scaled in place)
            if q.grad is not None:
                assert torch.equal(q.grad, gr)
        for p, q in zip(pa, pb):
            # one AdamW step moves a weight by ~lr; both sides round the same fp32 expression differently at most
            assert float((p - q).detach().abs().max()) <= 2e-4 * 1e-4 + 1e-6 * float(p.detach().abs().max()), it
        for p, q in zip(pa, pb):
            sa, sb = ref.state[p], opt.state[q]
            if "exp_avg" in sa:
                torch.testing.assert_close(sb["exp_avg"], sa["exp_avg"], rtol=2e-5, atol=1e-7 * grad_scale)
                torch.testing.assert_close(sb["exp_avg_sq"], sa["exp_avg_sq"], rtol=4e-5, atol=1e-9 * grad_scale ** 2)


def test_state_dict_round_trip_with_torch_adamw():
    """A torch.optim.AdamW state (what the reference's checkpoints hold, train.py:653-657) loads into ClipAdamW and
    continues the same trajectory; ClipAdamW's own state loads back into torch.optim.AdamW."""
    from unet_amd.optim import ClipAdamW
    dev = "cuda"
    shapes = SHAPES[:6]
    pa, pb = _params(2, shapes, dev), _params(2, shapes, dev)
    ref = torch.optim.AdamW(pa, lr=1e-3, betas=BETAS, weight_decay=0.0)
    g = torch.Generator(device="cpu").manual_seed(3)

    def grads_for(ps):
        gs = [torch.randn(p.shape, generator=g).to(dev) for p in ps]
        return gs

    for _ in range(3):
        gs = grads_for(pa)
        for p, gr in zip(pa, gs):
            p.grad = gr
        ref.step()
    with torch.no_grad():
        for p, q in zip(pa, pb):
            q.copy_(p)
    opt = ClipAdamW(pb, lr=1e-3, betas=BETAS, weight_decay=0.0)
    import copy
    opt.load_state_dict(copy.deepcopy(ref.state_dict()))     # (torch hands out its live `step` tensors: decouple them)
    gs = grads_for(pa)
    for p, q, gr in zip(pa, pb, gs):
        p.grad = gr.clone()
        q.grad = gr.clone()
    ref.step()
    opt.step()
    assert float(opt.state[pb[0]]["step"]) == 4.0
    for p, q in zip(pa, pb):
        assert float((p - q).detach().abs().max()) <= 1e-3 * 1e-4 + 1e-6 * float(p.detach().abs().max())
    back = torch.optim.AdamW(_params(2, shapes, dev), lr=1e-3, betas=BETAS, weight_decay=0.0)
    back.load_state_dict(copy.deepcopy(opt.state_dict()))
    for p, q in zip(back.param_groups[0]["params"], pb):
        torch.testing.assert_close(back.state[p]["exp_avg"], opt.state[q]["exp_avg"], rtol=0, atol=0)
        assert float(back.state[p]["step"]) == 4.0


def test_more_tensors_than_one_table_and_graph_capture():
    """300 tensors (two sub-tables) stepped inside a captured HIP graph: the step counter, the clip factor and the
    learning rate are read from device memory at replay time."""
    from unet_amd.optim import ClipAdamW
    dev = "cuda"
    shapes = [(17 + (i % 13), 3 + (i % 5)) for i in range(300)]
    pa, pb = _params(4, shapes, dev), _params(4, shapes, dev)
    lr = torch.tensor(1e-3, device=dev)
    ref = torch.optim.AdamW(pa, lr=1e-3, betas=BETAS, weight_decay=0.0)
    opt = ClipAdamW(pb, lr=lr, betas=BETAS, weight_decay=0.0, max_grad_norm=1.0)
    static = [torch.zeros_like(p) for p in pb]
    for q, s in zip(pb, static):
        q.grad = s
    g = torch.Generator(device="cpu").manual_seed(5)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step()                       # builds the tables outside the capture (zero gradients: only step += 1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()
    for it in range(3):
        grads = [torch.randn(p.shape, generator=g).to(dev) * 3 for p in pa]
        for p, s, gr in zip(pa, static, grads):
            p.grad = gr.clone()
            s.copy_(gr)
        if it == 2:
            lr.fill_(5e-4)
            ref.param_groups[0]["lr"] = 5e-4
        torch.nn.utils.clip_grad_norm_(pa, 1.0)
        ref.step()
        graph.replay()
    torch.cuda.synchronize()
    # the reference side has taken 3 steps, ours 1 (zero-gradient) + 3: bias corrections differ, so compare the
    # moments (independent of the step count) exactly and the weights loosely
    for p, q in zip(pa, pb):
        torch.testing.assert_close(opt.state[q]["exp_avg"], ref.state[p]["exp_avg"], rtol=3e-5, atol=1e-7)
    assert float(opt.state[pb[0]]["step"]) == 4.0        # one eager step + three replays (the capture itself runs nothing)


def test_loading_a_torch_adamw_state_keeps_the_clip_and_the_device_lr():
    """torch's Optimizer.load_state_dict replaces every param_group by the saved one: a reference checkpoint
    (torch.optim.AdamW: no ``max_grad_norm`` key, float ``lr``) must neither switch the clip off nor replace the
    device-scalar learning rate a captured step reads through its pointer."""
    import copy
    from unet_amd.optim import ClipAdamW
    dev = "cuda"
    shapes = SHAPES[:6]
    pa, pb = _params(6, shapes, dev), _params(6, shapes, dev)
    ref = torch.optim.AdamW(pa, lr=3e-4, betas=BETAS, weight_decay=0.0)
    g = torch.Generator(device="cpu").manual_seed(7)
    for p in pa:
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    ref.step()
    with torch.no_grad():
        for p, q in zip(pa, pb):
            q.copy_(p)
    lr = torch.tensor(1e-3, device=dev)
    opt = ClipAdamW(pb, lr=lr, betas=BETAS, weight_decay=0.0, max_grad_norm=5.0)
    opt.load_state_dict(copy.deepcopy(ref.state_dict()))
    group = opt.param_groups[0]
    assert group["max_grad_norm"] == 5.0
    assert group["lr"] is lr and abs(float(lr) - 3e-4) < 1e-10          # same tensor, loaded value
    grads = [torch.randn(p.shape, generator=g).to(dev) * 50.0 for p in pa]    # norm far above 5
    for p, q, gr in zip(pa, pb, grads):
        p.grad = gr.clone()
        q.grad = gr.clone()
    total = torch.nn.utils.clip_grad_norm_(pa, 5.0)
    assert float(total) > 50.0
    ref.step()
    opt.step()
    assert abs(float(opt.total_norm()) - float(total)) <= 2e-6 * float(total)
    for p, q in zip(pa, pb):                   # the step taken is the CLIPPED one
        torch.testing.assert_close(opt.state[q]["exp_avg"], ref.state[p]["exp_avg"], rtol=2e-5, atol=1e-7)
        assert float((p - q).detach().abs().max()) <= 3e-4 * 1e-4 + 1e-6 * float(p.detach().abs().max())
    # a group-level max_grad_norm saved by ClipAdamW itself wins over the constructor's
    opt2 = ClipAdamW(_params(6, shapes, dev), lr=1e-3, betas=BETAS, weight_decay=0.0, max_grad_norm=1.0)
    opt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    assert opt2.param_groups[0]["max_grad_norm"] == 5.0
