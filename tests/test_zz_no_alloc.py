"""The allocation-free entry point ``unet_conv1d_wgrad_noalloc`` (include/unet_hip.h; no counterpart in the
reference: it is for callers that cannot allocate device memory around the call) against the existing entry point
``unet_conv1d_wgrad``, through the C ABI, on two calls in a row, the second with a smaller input.

Both run the same kernels with the same split count, so the bar is bit-exact (and both are checked against an
fp32 ``F.conv1d`` weight gradient on the GPU at the bf16 bar of tests/test_gpu_ops.py) -- except where the wide
single-tap kernel's slabs do not fit the built-in scratch and the entry point takes the paired kernel instead
(last test: same result to summation order).  torch only provides the device memory of inputs and outputs here; the
entry point itself is given no workspace.  "Allocates nothing" is checked against the DEVICE's free memory
(hipMemGetInfo), which a hipMalloc inside the library would move; torch's allocator counters would not see one."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (n, cin, cout, L, k, stride, dilation, dtype): each case is smaller than the one before it; the last is fp32
CASES = [
    (4, 128, 128, 512, 7, 1, 4, torch.bfloat16),
    (3, 64, 72, 300, 3, 2, 1, torch.bfloat16),
    (2, 40, 24, 130, 3, 1, 1, torch.float32),
]


def _wgrad_both(case, seed, warm=True, expect_fit=True):
    from unet_amd import _lib, ops
    lib = _lib.load()
    n, cin, cout, L, k, stride, dil, dt = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    geom = ops.geom_conv(k, stride, dil)
    l_out = ops.conv_out_len(L, k, stride, dil)
    x = torch.randn(n, L, cin, generator=g).to(dt).to(dev)
    dy = torch.randn(n, l_out, cout, generator=g).to(dt).to(dev)
    d = ops._desc(geom, x, None, dy, None)
    need = int(lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d)))
    assert need > 0 and (need <= int(lib.unet_conv1d_wgrad_noalloc_max_bytes())) == expect_fit
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # existing entry point: caller-provided workspace
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dw_ref = torch.full((geom.n_taps, cout, cin), float("nan"), dtype=torch.float32, device=dev)
    _lib.check(lib.unet_conv1d_wgrad(C.byref(d), x.data_ptr(), None, dy.data_ptr(), dw_ref.data_ptr(), ws.data_ptr(),
                                     need, st), "unet_conv1d_wgrad")
    # allocation-free entry point: inputs and outputs only
    dw = torch.full_like(dw_ref, float("nan"))
    if warm:     # first use on a device resolves the scratch symbol (and may load code): not part of the property
        _lib.check(lib.unet_conv1d_wgrad_noalloc(C.byref(d), x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), st),
                   "unet_conv1d_wgrad_noalloc")
        dw.fill_(float("nan"))
    torch.cuda.synchronize()
    before, before_free = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    _lib.check(lib.unet_conv1d_wgrad_noalloc(C.byref(d), x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), st),
               "unet_conv1d_wgrad_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == before_free          # hipMemGetInfo: no hipMalloc / hipFree in the call
    # fp32 torch reference of the same weight gradient ([Cout, Cin, K] -> [K, Cout, Cin])
    w = torch.zeros(cout, cin, k, device=dev, requires_grad=True)
    y = F.conv1d(x.float().permute(0, 2, 1), w, None, stride=stride, padding=(k - 1) * dil // 2, dilation=dil)
    (gw,) = torch.autograd.grad(y, w, dy.float().permute(0, 2, 1))
    return dw, dw_ref, gw.permute(2, 0, 1)


def test_noalloc_wgrad_matches_existing_entry_point_twice_in_a_row():
    for i, case in enumerate(CASES):           # the second and third calls are smaller than the one before them
        dw, dw_ref, gw = _wgrad_both(case, seed=10 + i)
        assert torch.isfinite(dw).all()
        assert torch.equal(dw, dw_ref), f"case {case}: noalloc differs from unet_conv1d_wgrad"
        bar = 2e-2 if case[-1] == torch.bfloat16 else 1e-4
        err = ((dw - gw).norm() / gw.norm()).item()
        assert err < bar, f"case {case}: rel err {err} vs fp32 F.conv1d"


def test_noalloc_falls_back_to_the_paired_kernel_when_the_wide_kernels_slabs_do_not_fit():
    """C = 256 single-tap conv over 256 position chunks: unet_conv1d_wgrad with a full workspace runs conv_wgrad3 on
    its single 256 x 256 tile with 256 splits (64 MiB of slabs), the allocation-free entry point's 32 MiB hold only the
    paired kernel's 64 splits -- another summation order: equal at the fp32 bar, both within the bf16 bar of an fp32
    F.conv1d weight gradient."""
    case = (4, 256, 256, 4096, 1, 1, 1, torch.bfloat16)
    dw, dw_ref, gw = _wgrad_both(case, seed=21, expect_fit=False)
    assert torch.isfinite(dw).all()
    assert ((dw - dw_ref).norm() / dw_ref.norm()).item() < 1e-5
    assert ((dw - gw).norm() / gw.norm()).item() < 2e-2
    assert ((dw_ref - gw).norm() / gw.norm()).item() < 2e-2


def test_noalloc_rejects_a_conv_beyond_the_scratch():
    from unet_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    geom = ops.geom_conv(7, 1, 1)
    n, L, c = 1, 64, 1536                       # 7 * 1536 * 1536 * 4 B = 66 MB of slabs per split > 32 MiB
    x = torch.zeros(n, L, c, dtype=torch.bfloat16, device=dev)
    dy = torch.zeros(n, L, c, dtype=torch.bfloat16, device=dev)
    d = ops._desc(geom, x, None, dy, None)
    assert int(lib.unet_conv1d_wgrad_workspace_bytes(C.byref(d))) > int(lib.unet_conv1d_wgrad_noalloc_max_bytes())
    dw = torch.zeros((7, c, c), dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.unet_conv1d_wgrad_noalloc(C.byref(d), x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), st)
    assert rc != 0
    assert b"scratch" in lib.unet_last_error()
    torch.cuda.synchronize()
    assert float(dw.abs().max()) == 0.0          # nothing was launched
