"""ccnet_conv_gemm_bf16_noalloc (include/ccnet_hip.h): the allocation-free variant of ccnet_conv_gemm_bf16 for callers that
cannot call hipMalloc.  Called through the C ABI; its output must equal the existing entry point's bit for bit (same plan,
same kernels, the split-K partial sums in the library's static device array instead of a caller workspace), on two calls in
a row, the second with a smaller input (the array keeps the first call's contents)."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(M, K, Cout, dev, seed, res=False):
    from ccnet_amd import lib as _lib
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    r = torch.randn(M, Cout, generator=g).to(torch.bfloat16).to(dev) if res else None
    out = torch.zeros(M, Cout, dtype=torch.bfloat16, device=dev)
    p = _lib.ConvParams()
    p.x1 = x.data_ptr(); p.w = w.data_ptr(); p.bias = b.data_ptr(); p.out = out.data_ptr()
    if r is not None:
        p.res1 = r.data_ptr(); p.ldres1 = Cout
    p.N, p.Hin, p.Win, p.Hout, p.Wout = 1, M, 1, M, 1
    p.C1 = K; p.Cout = Cout; p.R = 1; p.stride = 1; p.ups = 1; p.ldx1 = K; p.ldw = K; p.ldo = Cout; p.scale = 1.0; p.batch = 1
    return p, out, (x, w, b, r)


def test_noalloc_conv_matches_workspace_entry_point(cuda):
    from ccnet_amd import lib as _lib
    l = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    scratch_bytes = 96 << 20
    assert l.ccnet_noalloc_init() == 0  # the array's address, looked up outside any capture
    # first a large split-K problem, then a smaller one: the second call sees the first call's leftovers in the array
    for (M, K, Cout, res, seed) in [(2048, 11520, 1280, True, 1), (512, 5760, 640, False, 2), (4096, 320, 320, True, 3)]:
        p, out, keep = _problem(M, K, Cout, cuda, seed, res)
        need = l.ccnet_conv_workspace_bytes(C.byref(p))
        if K > 320:
            assert 0 < need <= scratch_bytes, (M, K, Cout, need)  # these shapes are planned with K slices
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=cuda)
        p.workspace = ws.data_ptr() if need else None
        p.workspace_bytes = need
        assert l.ccnet_conv_gemm_bf16(C.byref(p), st) == 0
        ref = out.clone()
        out.zero_()
        p.workspace = None
        p.workspace_bytes = 0
        torch.cuda.synchronize()
        before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        assert l.ccnet_conv_gemm_bf16_noalloc(C.byref(p), st) == 0
        # neither torch's allocator nor the driver (a hipMalloc inside the library would not show in torch's counter) gave memory out
        assert torch.cuda.memory_allocated() == before and torch.cuda.mem_get_info()[0] == free_before
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (M, K, Cout, (out.float() - ref.float()).abs().max().item())
        x, w, b, r = keep
        full = x.float() @ w.float().T + b
        if r is not None:
            full = full + r.float()
        assert (out.float() - full).abs().max().item() <= 3e-2 * max(1.0, full.abs().max().item())


def test_noalloc_conv_rejects_oversized_problem(cuda):
    from ccnet_amd import lib as _lib
    l = _lib.load()
    p, out, keep = _problem(8192, 11520, 1280, cuda, 4)
    l.ccnet_set_option(3, 4)  # force 4 K slices: 4 * 8192 * 1280 * 4 B = 160 MiB of partial sums > 96 MiB
    try:
        assert l.ccnet_conv_workspace_bytes(C.byref(p)) > (96 << 20)
        rc = l.ccnet_conv_gemm_bf16_noalloc(C.byref(p), torch.cuda.current_stream().cuda_stream)
    finally:
        l.ccnet_set_option(3, 0)
    assert rc == -5 and b"scratch" in l.ccnet_strerror(rc)
    torch.cuda.synchronize()
    assert out.abs().max().item() == 0.0  # nothing was launched
