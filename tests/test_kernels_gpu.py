"""Per-kernel parity: every HIP kernel, called through the C ABI, against the same op in plain PyTorch fp32
on the same (bf16-rounded) inputs.  Tolerances are bf16-output tolerances: |err| <= atol + rtol*|ref| with
rtol = 2^-7 (one bf16 ulp is 2^-8 relative) unless a test states otherwise.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2.0 ** -7


def _ops():
    from ccnet_amd import ops
    return ops


def _rand_act(N, H, W, C, dev, seed, scale=1.0):
    from ccnet_amd.ops import Act
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = (torch.randn(N * H * W, C, generator=g) * scale).to(torch.bfloat16).to(dev)
    return Act(t, N, H, W, C)


def _nchw(a, C=None):
    C = C or a.C
    return a.t[:, :C].float().reshape(a.N, a.H, a.W, C).permute(0, 3, 1, 2).contiguous()


def _close(out, ref, atol, rtol=RTOL, what=""):
    err = (out - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = (err > bound)
    assert not bad.any(), f"{what}: max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g}), " \
                          f"{bad.sum().item()} / {bad.numel()} outside tolerance"


def _pack_w(w):  # [Cout, Cin, R, R] fp32 -> bf16 [Cout, R*R*Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(torch.bfloat16)


CONV_CASES = [
    # N, H, W, C1, C2, Cout, R, stride, ups
    (2, 16, 16, 64, 0, 128, 3, 1, 1),
    (2, 16, 16, 128, 0, 320, 3, 1, 1),
    (1, 7, 9, 64, 0, 72, 3, 1, 1),       # ragged M and a partial cout tile
    (2, 16, 16, 64, 0, 64, 3, 2, 1),     # stride 2
    (2, 8, 8, 64, 0, 64, 3, 1, 2),       # nearest x2 upsample folded into the gather
    (2, 8, 8, 128, 64, 128, 3, 1, 1),    # channel concat of two sources
    (2, 8, 8, 64, 128, 64, 1, 1, 1),     # 1x1 over a concat
    (4, 32, 32, 320, 0, 320, 3, 1, 1),   # G1 level-0 shape at reduced size -> 128x128 / 128x64 tiles
    (32, 8, 8, 1280, 0, 1280, 1, 1, 1),
    (3, 5, 5, 64, 0, 8, 1, 1, 1),
    (32, 16, 16, 64, 0, 2560, 1, 1, 1),  # 32 x 8 tiles of 256 x 320: the windowed tile walk (4 pixel panels x 8 cout tiles per XCD)
    (8, 32, 32, 128, 0, 5120, 1, 1, 1),  # 32 x 16 tiles, two windows per row of cout tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_gemm(cuda, case):
    ops = _ops()
    N, H, W, C1, C2, Cout, R, stride, ups = case
    x1 = _rand_act(N, H, W, C1, cuda, 1)
    x2 = _rand_act(N, H, W, C2, cuda, 2) if C2 else None
    g = torch.Generator().manual_seed(3)
    Cin = C1 + C2
    w = torch.randn(Cout, Cin, R, R, generator=g) / math.sqrt(Cin * R * R)
    b = torch.randn(Cout, generator=g)
    wp = _pack_w(w).to(cuda)
    out = ops.conv(x1, wp, b.to(cuda), Cout, R=R, stride=stride, ups=ups, x2=x2)
    xin = _nchw(x1) if x2 is None else torch.cat([_nchw(x1), _nchw(x2)], 1)
    if ups == 2:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wp.float().reshape(Cout, R, R, Cin).permute(0, 3, 1, 2).to(cuda), b.to(cuda), stride=stride,
                   padding=R // 2)
    assert (out.H, out.W) == tuple(ref.shape[2:])
    _close(_nchw(out), ref, atol=2e-2, what=f"conv {case}")


@pytest.mark.parametrize("case", [(16, 64, True, 0, 320, 0, 320), (17, 64, False, 0, 320, 0, 320), (16, 64, True, 64, 320, 0, 320),
                                  (20, 60, True, 0, 320, 0, 320), (16, 64, False, 0, 960, 0, 320), (17, 64, True, 0, 960, 0, 320),
                                  (16, 64, False, 0, 2560, 2, 320), (18, 64, False, 64, 2560, 2, 320),
                                  (17, 63, True, 0, 320, 0, 320), (17, 63, False, 0, 960, 0, 320), (17, 63, False, 0, 2560, 2, 320),
                                  (32, 32, False, 0, 1920, 0, 640), (33, 31, False, 64, 1920, 0, 640), (34, 32, False, 0, 5120, 2, 640),
                                  (35, 31, False, 0, 5120, 2, 640),  # 20 cout groups: one round of 12 row ranges, ragged rows
                                  (32, 32, True, 0, 1280, 0, 640)])
def test_pointwise_streaming_kernel(cuda, case):
    """1x1 layers with 320 input channels at the 64x64 level (o-proj / proj_in / proj_out / to_q, qkv, ff.linear1 +
    GEGLU) and the wide ones with 640 at the 32x32 level (qkv, ff.linear1 + GEGLU) run on the weight-stationary streaming kernel (weights in registers, pixel rows through an LDS ring): bit-identical
    to the tiled kernel (ccnet_set_option(8, 0)), within tolerance of the fp32 product.  Cases: with / without residual, row
    counts that are not a whole number of chunks per block or of 32-row chunks (67473 rows), inputs that are column views of wider rows, several cout groups,
    GEGLU."""
    ops = _ops()
    from ccnet_amd import lib as L
    from ccnet_amd.ops import Act
    N, H, with_res, extra, Cout, act, Cin = case
    g = torch.Generator().manual_seed(71)
    M = N * H * H
    Co = Cout // 2 if act == 2 else Cout
    wide = (torch.randn(M, Cin + extra, generator=g)).to(torch.bfloat16).to(cuda)
    x = Act(wide[:, extra:], N, H, H, Cin)
    res = Act((torch.randn(M, Co + extra, generator=g)).to(torch.bfloat16).to(cuda)[:, :Co], N, H, H, Co) if with_res else None
    w = (torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin)).to(torch.bfloat16)
    b = torch.randn(Cout, generator=g)
    wk, bk = (ops.geglu_interleave(w), ops.geglu_interleave(b)) if act == 2 else (w, b)
    wk, bk = wk.contiguous().to(cuda), bk.contiguous().to(cuda)
    out = ops.conv(x, wk, bk, Cout, R=1, pad=0, res1=res, act=act)
    assert L.load().ccnet_set_option(8, 0) == 0 and L.load().ccnet_set_option(9, 0) == 0  # the tiled kernel, 32x32x16 MFMA
    try:
        tiled = ops.conv(x, wk, bk, Cout, R=1, pad=0, res1=res, act=act)
    finally:
        L.load().ccnet_set_option(8, 31)
        L.load().ccnet_set_option(9, 1)
    assert torch.equal(out.t, tiled.t), f"streaming vs tiled kernel differ: max {(out.t.float() - tiled.t.float()).abs().max().item()}"
    ref = x.t.float() @ w.float().T.to(cuda) + b.to(cuda)
    if act == 2:
        ref = ref[:, :Co] * F.gelu(ref[:, Co:])
    if with_res:
        ref = ref + res.t.float()
    _close(out.t.float(), ref, atol=3e-2, what=f"pointwise streaming {case}")


@pytest.mark.parametrize("case", [(65536, 0, 320), (65536, 64, 960), (67473, 0, 960), (131072, 0, 320), (65690, 0, 320)])
def test_pointwise_streaming_layernorm_inline(cuda, case):
    """LayerNorm -> Linear with the row statistics taken INSIDE the streaming kernel (colsum without rowstat: norm1 -> qkv and
    norm2 -> attn2.to_q of the 320-channel level): against fp32 LayerNorm + Linear, and against the form that feeds the same
    folded GEMM with ccnet_row_stats_bf16.  Rows include large offsets (the in-kernel sums are shifted by the row's first
    element), a constant row (variance 0) and ragged row counts (65690 rows: the last block's range is one partial chunk); a shape
    the streaming kernel does not take is refused."""
    ops = _ops()
    from ccnet_amd import lib as L
    M, extra, Cout = case
    g = torch.Generator().manual_seed(91)
    xs = torch.randn(M, 320 + extra, generator=g)
    xs[1::7] += 30.0   # offset rows: E[x^2] - E[x]^2 around 900 would lose the variance of 1 without the shift
    xs[5::11] *= 4.0
    xs[3] = 2.5        # constant row
    wide = xs.to(torch.bfloat16).to(cuda)
    x = wide[:, extra:]
    gamma, beta = 1 + 0.2 * torch.randn(320, generator=g), 0.3 * torch.randn(320, generator=g)
    w = torch.randn(Cout, 320, generator=g) / math.sqrt(320)
    b = torch.randn(Cout, generator=g)
    wf, bf, cs = (t.to(cuda) for t in ops.fold_layernorm(w, b, gamma, beta))
    assert ops.ln_inline_ok(M, 320, Cout)
    out = ops.linear(x, wf, bf, Cout, 320, colsum=cs)
    fed = ops.linear(x, wf, bf, Cout, 320, rowstat=ops.row_stats(x, 320), colsum=cs)
    ref = F.layer_norm(x.float(), (320,), gamma.to(cuda), beta.to(cuda)) @ w.to(cuda).T + b.to(cuda)
    _close(out.float(), ref, atol=6e-2, what=f"layernorm in the streaming kernel {case}")
    d = (out.float() - fed.float()).abs()
    assert d.max().item() <= 6e-2 and (d > 0).float().mean().item() < 0.05, f"vs row_stats-fed GEMM: max {d.max().item()}"
    assert not ops.ln_inline_ok(4096, 320, Cout)
    with pytest.raises(L.CcnetError):
        ops.linear(x[:4096], wf, bf, Cout, 320, colsum=cs)


@pytest.mark.parametrize("case", CONV_CASES + [(2, 64, 64, 320, 0, 320, 3, 1, 1), (4, 32, 32, 640, 320, 640, 3, 1, 1),
                                               (8, 16, 16, 1280, 0, 1280, 3, 1, 1), (32, 8, 8, 1280, 0, 1280, 3, 1, 1)])
def test_conv_gemm_mfma_16x16x32_instances_match_32x32x16(cuda, case):
    """Plain and split-K launches run on the v_mfma_f32_16x16x32_bf16 instances of their tile configuration
    (conv_gemm16_kernel: the chip holds a higher clock with that shape); ccnet_set_option(9, 0) puts them back on the 32x32x16
    kernel.  Same products, another order of fp32 sums inside the matrix core: agreement to bf16 rounding, with residual,
    second output, SiLU, scale and the per-image row vector in the epilogue, ragged rows and partial cout tiles included."""
    ops = _ops()
    from ccnet_amd import lib as L
    N, H, W, C1, C2, Cout, R, stride, ups = case
    x1 = _rand_act(N, H, W, C1, cuda, 11)
    x2 = _rand_act(N, H, W, C2, cuda, 12) if C2 else None
    g = torch.Generator().manual_seed(13)
    Cin = C1 + C2
    wp = _pack_w(torch.randn(Cout, Cin, R, R, generator=g) / math.sqrt(Cin * R * R)).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    Ho, Wo = H * ups // stride, W * ups // stride
    rv = torch.randn(N, Cout, generator=g).to(cuda)
    r1, r2 = _rand_act(N, Ho, Wo, Cout, cuda, 14), _rand_act(N, Ho, Wo, Cout, cuda, 15)

    def run():
        plain = ops.conv(x1, wp, b, Cout, R=R, stride=stride, ups=ups, x2=x2)
        full, full2 = ops.conv(x1, wp, b, Cout, R=R, stride=stride, ups=ups, x2=x2, rowvec=rv, res1=r1, res2=r2, act=1, scale=0.7)
        return plain.t.float(), full.t.float(), full2.t.float()
    new = run()
    assert L.load().ccnet_set_option(9, 0) == 0
    try:
        old = run()
    finally:
        L.load().ccnet_set_option(9, 1)
    for nm, a_, b_ in zip(("plain", "epilogue", "second output"), new, old):
        _close(a_, b_, atol=2e-2, what=f"16x16x32 vs 32x32x16 {nm} {case}")


def test_conv_epilogue_full(cuda):
    """bias + per-image rowvec + SiLU + scale + residual + dual output in one launch."""
    ops = _ops()
    N, H, W, C, Cout = 3, 8, 8, 128, 192
    x = _rand_act(N, H, W, C, cuda, 5)
    r1 = _rand_act(N, H, W, Cout, cuda, 6)
    r2 = _rand_act(N, H, W, Cout, cuda, 7)
    g = torch.Generator().manual_seed(8)
    w = torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, Cout, generator=g).to(cuda)
    wp = _pack_w(w).to(cuda)
    out, out2 = ops.conv(x, wp, b.to(cuda), Cout, R=3, rowvec=rv, act=1, scale=0.7, res1=r1, res2=r2)
    ref = F.conv2d(_nchw(x), wp.float().reshape(Cout, 3, 3, C).permute(0, 3, 1, 2), b.to(cuda), padding=1)
    ref = F.silu(ref + rv[:, :, None, None]) * 0.7 + _nchw(r1)
    _close(_nchw(out), ref, atol=2e-2, what="epilogue out")
    _close(_nchw(out2), ref + _nchw(r2), atol=3e-2, what="epilogue out2")


def test_conv_split_k(cuda):
    """Deep-K layer with few output tiles: K slices + reduce pass (workspace from ccnet_conv_workspace_bytes) must
    give the unsplit result, with the whole epilogue (bias, rowvec, residual, dual output) applied by the reduce."""
    import ctypes as C
    from ccnet_amd import lib as _lib
    ops = _ops()
    N, H, W, C1, C2, Cout = 2, 16, 16, 640, 640, 320   # M = 512, K = 11520: 2 tiles of 256x320
    x1 = _rand_act(N, H, W, C1, cuda, 11)
    x2 = _rand_act(N, H, W, C2, cuda, 12)
    r1 = _rand_act(N, H, W, Cout, cuda, 13)
    r2 = _rand_act(N, H, W, Cout, cuda, 14)
    g = torch.Generator().manual_seed(15)
    Cin = C1 + C2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, Cout, generator=g).to(cuda)
    wp = _pack_w(w).to(cuda)
    l = _lib.load()
    l.ccnet_set_option(3, 1)   # forced unsplit
    o_ref, o2_ref = ops.conv(x1, wp, b.to(cuda), Cout, R=3, x2=x2, rowvec=rv, scale=0.9, res1=r1, res2=r2)
    for S in (0, 3, 8, 13, 32):  # heuristic (splits this shape), forced slice counts (32 is lowered to 13: 180 K steps, >= 13 per slice)
        l.ccnet_set_option(3, S)
        out, out2 = ops.conv(x1, wp, b.to(cuda), Cout, R=3, x2=x2, rowvec=rv, scale=0.9, res1=r1, res2=r2)
        _close(out.t.float(), o_ref.t.float(), atol=2e-2, what=f"split-K {S} out")
        _close(out2.t.float(), o2_ref.t.float(), atol=3e-2, what=f"split-K {S} out2")
    l.ccnet_set_option(3, 0)
    ref = F.conv2d(torch.cat([_nchw(x1), _nchw(x2)], 1), wp.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2),
                   b.to(cuda), padding=1)
    ref = (ref + rv[:, :, None, None]) * 0.9 + _nchw(r1)
    _close(_nchw(out), ref, atol=2e-2, what="split-K vs conv2d")


def test_conv_without_workspace_runs_unsplit(cuda):
    """A caller that passes workspace = NULL (graph capture, foreign runtimes) gets the unsplit plan, not CCNET_EWS; a
    workspace that is too small is still an error."""
    import ctypes as C
    from ccnet_amd import lib as _lib
    ops = _ops()
    l = _lib.load()
    M, K, Cout = 512, 11520, 1280
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    w = (torch.randn(Cout, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(cuda)
    ref = ops.linear(x, w, None, Cout, K)  # allocates the workspace the plan asks for
    out = torch.empty(M, Cout, dtype=torch.bfloat16, device=cuda)
    p = _lib.ConvParams()
    p.x1 = x.data_ptr(); p.w = w.data_ptr(); p.out = out.data_ptr()
    p.N, p.Hin, p.Win, p.Hout, p.Wout = 1, M, 1, M, 1
    p.C1 = K; p.Cout = Cout; p.R = 1; p.stride = 1; p.ups = 1; p.ldx1 = K; p.ldw = K; p.ldo = Cout; p.scale = 1.0; p.batch = 1
    need = l.ccnet_conv_workspace_bytes(C.byref(p))
    assert need > 0  # this shape is planned with K slices
    assert l.ccnet_conv_gemm_bf16(C.byref(p), torch.cuda.current_stream().cuda_stream) == 0
    _close(out.float(), ref.float(), atol=2e-2, what="NULL workspace (unsplit) vs split-K")
    small = torch.empty(1024, dtype=torch.float32, device=cuda)
    p.workspace = small.data_ptr(); p.workspace_bytes = 4096
    assert l.ccnet_conv_gemm_bf16(C.byref(p), torch.cuda.current_stream().cuda_stream) == -4  # CCNET_EWS
    torch.cuda.synchronize()


def test_gemm_batched(cuda):
    ops = _ops()
    Bt, M, Nn, K = 3, 100, 72, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randn(Bt, M, K, generator=g).to(torch.bfloat16).to(cuda)
    w = torch.randn(Bt, Nn, K, generator=g).to(torch.bfloat16).to(cuda)
    out = torch.empty(Bt, M, Nn, dtype=torch.bfloat16, device=cuda)
    ops.gemm_batched(x, w, out, M, Nn, K, Bt, K, K, Nn, M * K, Nn * K, M * Nn)
    ref = torch.bmm(x.float(), w.float().transpose(1, 2))
    _close(out.float(), ref, atol=0.15, what="batched gemm")


GN_CASES = [(2, 16, 16, 320, 0), (2, 8, 8, 1280, 1280), (1, 7, 7, 256, 0), (2, 32, 32, 128, 0), (3, 4, 4, 640, 320),
            (2, 64, 64, 320, 0),
            # one-pass kernel (channels per group % 8 == 0): XCD-grouped images, two sources, several vectors per thread
            (8, 16, 16, 1280, 0), (16, 32, 32, 640, 640), (2, 16, 16, 2560, 0), (9, 8, 8, 1280, 0), (8, 32, 32, 1280, 640),
            # multi-group one-pass kernel (10 / 20 / 30 channels per group: a block takes 4 / 2 / 4 groups): the 64x64 and 32x32
            # levels of 320 / 640-channel tensors, two sources whose boundary is a slice boundary, a ragged image count
            (8, 64, 64, 320, 0), (3, 64, 64, 640, 0), (8, 32, 32, 320, 0), (2, 64, 64, 320, 320), (5, 16, 16, 960, 0),
            (2, 24, 24, 320, 0)]


@pytest.mark.parametrize("case", GN_CASES)
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm(cuda, case, silu):
    ops = _ops()
    N, H, W, C1, C2 = case
    x1 = _rand_act(N, H, W, C1, cuda, 21, scale=2.0)
    x1.t += 0.5
    x2 = _rand_act(N, H, W, C2, cuda, 22) if C2 else None
    C = C1 + C2
    g = torch.Generator().manual_seed(23)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(C, generator=g)).to(cuda)
    out = ops.groupnorm(x1, gamma, beta, 32, 1e-5, silu, x2=x2)
    xin = _nchw(x1) if x2 is None else torch.cat([_nchw(x1), _nchw(x2)], 1)
    ref = F.group_norm(xin, 32, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    _close(_nchw(out), ref, atol=1e-2, what=f"groupnorm {case}")


def test_groupnorm_one_pass_agrees_with_two_kernel_path(cuda):
    from ccnet_amd import lib as _lib
    ops = _ops()
    l = _lib.load()
    x1 = _rand_act(8, 16, 16, 640, cuda, 24, scale=3.0)
    x2 = _rand_act(8, 16, 16, 640, cuda, 25)
    g = torch.Generator().manual_seed(26)
    gamma, beta = (1 + 0.2 * torch.randn(1280, generator=g)).to(cuda), (0.2 * torch.randn(1280, generator=g)).to(cuda)
    try:
        l.ccnet_set_option(6, 0)
        two = ops.groupnorm(x1, gamma, beta, 32, 1e-5, True, x2=x2).t.clone()
        l.ccnet_set_option(6, 2)
        one = ops.groupnorm(x1, gamma, beta, 32, 1e-5, True, x2=x2).t
        again = ops.groupnorm(x1, gamma, beta, 32, 1e-5, True, x2=x2).t
    finally:
        l.ccnet_set_option(6, 2)
    assert torch.equal(one, again)                      # deterministic
    assert (one.float() - two.float()).abs().max().item() <= 4e-2  # bf16 outputs of two summation orders: <= 1-2 ulp


@pytest.mark.parametrize("rows,C", [(100, 320), (64, 640), (33, 1280), (10, 512), (7, 768)])
def test_layernorm(cuda, rows, C):
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(rows, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(cuda)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(C, generator=g)).to(cuda)
    out = ops.layernorm(x, C, gamma, beta, 1e-5)
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    _close(out.float(), ref, atol=1e-2, what="layernorm")


@pytest.mark.parametrize("rows,C,N,geglu", [(300, 320, 960, False), (4096, 640, 640, False), (1000, 1280, 2560, True),
                                            (8192, 320, 2560, True), (77, 64, 192, False), (70000, 320, 320, False)])
def test_layernorm_folded_into_linear(cuda, rows, C, N, geglu):
    """Linear(LayerNorm(x)) as ONE GEMM on the un-normalised rows (ccnet_row_stats_bf16 + the rowstat / colsum operands of
    ccnet_conv_gemm_bf16) against fp32 F.layer_norm -> F.linear (-> a * gelu(gate)); rows with large means and very
    different scales, non-trivial gamma / beta, with and without a bias, every tile configuration the dispatch picks."""
    ops = _ops()
    g = torch.Generator().manual_seed(33 + rows)
    x = torch.randn(rows, C, generator=g) * (0.2 + 4.0 * torch.rand(rows, 1, generator=g)) + 3.0 * torch.randn(rows, 1, generator=g)
    xb = x.to(torch.bfloat16).to(cuda)
    w = torch.randn(N, C, generator=g) / math.sqrt(C)
    b = 0.3 * torch.randn(N, generator=g) if geglu else None
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    st = ops.row_stats(xb, C, 1e-5)
    xf = xb.float()
    assert (st[:, 0] - xf.mean(1)).abs().max().item() < 1e-4
    assert ((st[:, 1] - (xf.var(1, unbiased=False) + 1e-5).rsqrt()).abs() * xf.std(1)).max().item() < 1e-3
    wf, bf, cs = ops.fold_layernorm(w, b, gamma, beta)
    ref = F.linear(F.layer_norm(xf, (C,), gamma.to(cuda), beta.to(cuda), 1e-5), w.to(cuda), None if b is None else b.to(cuda))
    if geglu:
        wf, bf, cs = ops.geglu_interleave(wf), ops.geglu_interleave(bf), ops.geglu_interleave(cs)
        a, gate = ref.chunk(2, dim=-1)
        ref = a * F.gelu(gate)
    out = ops.linear(xb, wf.to(cuda), bf.to(cuda), N, C, act=2 if geglu else 0, rowstat=st, colsum=cs.to(cuda))
    assert tuple(out.shape) == tuple(ref.shape)
    _close(out.float(), ref, atol=3e-2, what=f"LayerNorm folded into linear {rows}x{C}->{N}")
    # and against the unfused pair of kernels (same bf16 inputs, one rounding fewer on the folded side)
    two = ops.linear(ops.layernorm(xb, C, gamma.to(cuda), beta.to(cuda)),
                     (ops.geglu_interleave(w) if geglu else w).to(torch.bfloat16).to(cuda),
                     None if b is None else ops.geglu_interleave(b).to(cuda), N, C, act=2 if geglu else 0)
    _close(out.float(), two.float(), atol=5e-2, what="folded vs LayerNorm kernel + linear")


@pytest.mark.parametrize("rows,C,with_res,expect_parts", [(70000, 320, False, 2), (70000, 320, True, 2), (32768, 640, True, 4),
                                                          (8192, 1280, False, 8), (8192, 1280, True, 8), (300, 320, True, 0)])
def test_row_statistics_from_the_producing_layer(cuda, rows, C, with_res, expect_parts):
    """ccnet_conv_gemm_bf16(stat_out): the (mean, M2) partials that a layer's epilogue writes for the LayerNorm that follows
    it (proj_in -> norm1, to_out + x -> norm2 / norm3 of a BasicTransformerBlock) against the statistics of its own bf16
    output, and the next GEMM consuming them (rowstat_parts) against LayerNorm kernel + linear.  Rows with large means; the
    last case has no statistics instance for its plan and must say so (None) instead of failing."""
    ops = _ops()
    g = torch.Generator().manual_seed(5 + rows + C)
    x = (torch.randn(rows, C, generator=g)).to(torch.bfloat16).to(cuda)
    w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(torch.bfloat16).to(cuda)
    b = (0.5 * torch.randn(C, generator=g) + 2.0).to(cuda)  # a row mean well away from zero
    res = ((torch.randn(rows, C, generator=g) * (0.2 + 3.0 * torch.rand(rows, 1, generator=g))
            + 3.0 * torch.randn(rows, 1, generator=g)).to(torch.bfloat16).to(cuda)) if with_res else None
    out, st = ops.linear(x, w, b, C, C, res1=res, emit_stats=True)
    plain = ops.linear(x, w, b, C, C, res1=res)
    assert torch.equal(out, plain), "the statistics instance must write the same output as the plain one"
    if expect_parts == 0:
        assert st is None
        return
    assert st is not None and tuple(st.shape) == (rows, expect_parts, 2)
    of = out.float()
    span = C // expect_parts
    mean = st[:, :, 0].mean(1)
    m2 = st[:, :, 1].sum(1) + span * ((st[:, :, 0] - mean[:, None]) ** 2).sum(1)
    var = m2 / C
    # statistics are taken from the fp32 values on their way out: they differ from those of the rounded output by the
    # rounding noise only (measured: mean within 5e-3 at |x| up to 20, variance within 0.9 % on the worst of 70000 rows)
    assert (mean - of.mean(1)).abs().max().item() < 1e-2
    assert ((var - of.var(1, unbiased=False)).abs() / of.var(1, unbiased=False)).max().item() < 2e-2
    gamma, beta = (1.0 + 0.2 * torch.randn(C, generator=g)), 0.3 * torch.randn(C, generator=g)
    w2 = torch.randn(2 * C, C, generator=g) / math.sqrt(C)
    wf, bf, cs = ops.fold_layernorm(w2, None, gamma, beta)
    got = ops.linear(out, wf.to(cuda), bf.to(cuda), 2 * C, C, rowstat=st, colsum=cs.to(cuda))
    ref = F.linear(F.layer_norm(of, (C,), gamma.to(cuda), beta.to(cuda), 1e-5), w2.to(cuda))
    _close(got.float(), ref, atol=3e-2, what=f"producer statistics -> folded LayerNorm {rows}x{C}")
    two = ops.linear(ops.layernorm(out, C, gamma.to(cuda), beta.to(cuda)), w2.to(torch.bfloat16).to(cuda), None, 2 * C, C)
    _close(got.float(), two.float(), atol=5e-2, what="producer statistics vs LayerNorm kernel + linear")


def test_geglu(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(50, 2 * 1280, generator=g) * 2).to(torch.bfloat16).to(cuda)
    out = ops.geglu(x, 1280)
    a, gg = x.float().chunk(2, dim=-1)
    _close(out.float(), a * F.gelu(gg), atol=1e-2, what="geglu")


ATT_CASES = [(2, 5, 256, 256), (1, 10, 1024, 1024), (2, 20, 64, 77), (1, 5, 4096, 4096), (2, 3, 49, 49),
             (1, 2, 784, 77), (3, 1, 200, 130)]


@pytest.mark.parametrize("case", ATT_CASES)
def test_attention_d64(cuda, case):
    ops = _ops()
    B, H, Sq, Skv = case
    g = torch.Generator().manual_seed(51)
    # fused-QKV style views: q/k/v are column slices of wider row matrices
    qkv = torch.randn(B * Sq, 3 * H * 64, generator=g).to(torch.bfloat16).to(cuda)
    kv = torch.randn(B * Skv, 2 * H * 64, generator=g).to(torch.bfloat16).to(cuda)
    q = qkv[:, : H * 64]
    k, v = kv[:, : H * 64], kv[:, H * 64:]
    scale = 1.0 / 8.0
    o = ops.attention_d64(q, k, v, B, H, Sq, Skv, scale)
    qf = q.float().reshape(B, Sq, H, 64).permute(0, 2, 1, 3)
    kf = k.float().reshape(B, Skv, H, 64).permute(0, 2, 1, 3)
    vf = v.float().reshape(B, Skv, H, 64).permute(0, 2, 1, 3)
    p = torch.softmax(qf @ kf.transpose(-1, -2) * scale, dim=-1)
    ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * Sq, H * 64)
    _close(o.float(), ref, atol=1.5e-2, rtol=2.0 ** -6, what=f"attention {case}")


@pytest.mark.parametrize("case", [(1, 2, 128, 64), (2, 1, 192, 128), (1, 3, 320, 192), (2, 2, 64, 320), (1, 1, 576, 1024),
                                  (1, 2, 1024, 256)])
def test_attention_d64_long_key_instance(cuda, case, mode=2):
    """The 64-queries-per-wave instance (LDS-DMA ring of three K/V tiles) forced onto shapes that exercise its edges: one,
    two and three key tiles (ring start-up / drain), query counts that leave waves of the last block without rows; a late
    dominant key (rescale branch)."""
    ops = _ops()
    from ccnet_amd import lib as L
    B, H, Sq, Skv = case
    g = torch.Generator().manual_seed(53)
    qkv = torch.randn(B * Sq, 3 * H * 64, generator=g)
    kv = torch.randn(B * Skv, 2 * H * 64, generator=g) * 1.5
    kv[Skv - 5, :64] = qkv[3, H * 64: H * 64 + 64] * 4.0  # spike for query 3 of head 0 in the last key tile
    qkv, kv = qkv.to(torch.bfloat16).to(cuda), kv.to(torch.bfloat16).to(cuda)
    q, k, v = qkv[:, H * 64: 2 * H * 64], kv[:, : H * 64], kv[:, H * 64:]
    assert L.load().ccnet_set_option(7, mode) == 0
    try:
        o = ops.attention_d64(q, k, v, B, H, Sq, Skv, 0.125)
        assert L.load().ccnet_set_option(7, 1) == 0
        o_ref_kernel = ops.attention_d64(q, k, v, B, H, Sq, Skv, 0.125)
    finally:
        L.load().ccnet_set_option(7, 0)
    qf = q.float().reshape(B, Sq, H, 64).permute(0, 2, 1, 3)
    kf = k.float().reshape(B, Skv, H, 64).permute(0, 2, 1, 3)
    vf = v.float().reshape(B, Skv, H, 64).permute(0, 2, 1, 3)
    p = torch.softmax(qf @ kf.transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * Sq, H * 64)
    _close(o.float(), ref, atol=1.5e-2, rtol=2.0 ** -6, what=f"attention (mode {mode}) {case}")
    _close(o.float(), o_ref_kernel.float(), atol=1.5e-2, rtol=2.0 ** -6, what=f"attention: instances agree (mode {mode}) {case}")


@pytest.mark.parametrize("case", [(2, 5, 256, 256), (1, 2, 128, 64), (1, 3, 320, 192), (1, 1, 576, 1024), (2, 2, 200, 77)])
def test_attention_d64_prescaled(cuda, case):
    """Q that already carries scale * log2(e): the long-key instance takes the row maximum off inside the matrix core (an extra
    MFMA against -m kept bf16-exact), shorter / ragged shapes run the round-1 kernel with a unit scale.  Reference: softmax of
    the same (bf16) scaled Q in the log2 domain.  A late dominant key and a first tile far below zero exercise both
    directions of the reference maximum."""
    ops = _ops()
    B, H, Sq, Skv = case
    g = torch.Generator().manual_seed(57)
    c = math.log2(math.e) / 8.0
    q = (torch.randn(B * Sq, H * 64, generator=g) * c)
    k = torch.randn(B * Skv, H * 64, generator=g) * 1.5
    v = torch.randn(B * Skv, H * 64, generator=g)
    k[Skv - 5, :64] = q[3, :64] / c * 4.0            # spike for query 3 of head 0 in the last key tile
    k[:64, :64] -= q[7, :64] / c * 3.0               # query 7: every score of the first tile far below zero
    qb, kb, vb = (t.to(torch.bfloat16).to(cuda) for t in (q, k, v))
    o = ops.attention_d64(qb, kb, vb, B, H, Sq, Skv, 0.0, prescaled=True)
    qf = qb.float().reshape(B, Sq, H, 64).permute(0, 2, 1, 3)
    kf = kb.float().reshape(B, Skv, H, 64).permute(0, 2, 1, 3)
    vf = vb.float().reshape(B, Skv, H, 64).permute(0, 2, 1, 3)
    p = torch.softmax(qf @ kf.transpose(-1, -2) * math.log(2.0), dim=-1)   # exp2(s) / sum exp2(s)
    ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * Sq, H * 64)
    _close(o.float(), ref, atol=1.5e-2, rtol=2.0 ** -6, what=f"prescaled attention {case}")


def test_attention_rescale_branch(cuda):
    """Online-softmax rescale: a key in a LATE tile dominates, so the running max jumps after O has mass."""
    ops = _ops()
    B, H, Sq, Skv = 1, 1, 64, 256
    g = torch.Generator().manual_seed(52)
    q = torch.randn(Sq, 64, generator=g)
    k = torch.randn(Skv, 64, generator=g) * 0.1
    v = torch.randn(Skv, 64, generator=g)
    k[200] = q[3] * 4.0  # spike for query 3 in the 4th key tile
    k[70] = q[40] * 3.0
    qb, kb, vb = (t.to(torch.bfloat16).to(cuda) for t in (q, k, v))
    o = ops.attention_d64(qb, kb, vb, B, H, Sq, Skv, 0.125)
    p = torch.softmax(qb.float() @ kb.float().T * 0.125, dim=-1)
    ref = p @ vb.float()
    _close(o.float(), ref, atol=1.5e-2, rtol=2.0 ** -6, what="attention rescale")


@pytest.mark.parametrize("case", [(2, 1, 256, 256, 512), (2, 1, 196, 77, 768), (1, 2, 49, 49, 128),
                                  # one image, 64x64 tokens, 512 channels (SD-VAE mid-block attention on one 512^2 slice): the
                                  # P.V product has M = 4096, K = 4096 and is planned with K slices -> needs the workspace
                                  (1, 1, 4096, 4096, 512)])
def test_attention_generic_head_dim(cuda, case):
    """Head dims beyond the flash kernel (repo-YAML geometry: one head of 512 / 768 channels): batched score GEMM ->
    masked row softmax -> P.V GEMM, against fp32 softmax attention; ragged token counts and the 77-token context."""
    ops = _ops()
    B, H, Sq, Skv, d = case
    g = torch.Generator().manual_seed(57)
    qkv = torch.randn(B * Sq, 3 * H * d, generator=g).to(torch.bfloat16).to(cuda)
    kv = torch.randn(B * Skv, 2 * H * d, generator=g).to(torch.bfloat16).to(cuda)
    q, k, v = qkv[:, : H * d], kv[:, : H * d], kv[:, H * d:]
    scale = 1.0 / math.sqrt(d)
    o = ops.attention_generic(q, k, v, B, H, Sq, Skv, d, scale)
    qf = q.float().reshape(B, Sq, H, d).permute(0, 2, 1, 3)
    kf = k.float().reshape(B, Skv, H, d).permute(0, 2, 1, 3)
    vf = v.float().reshape(B, Skv, H, d).permute(0, 2, 1, 3)
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf
    ref = ref.permute(0, 2, 1, 3).reshape(B * Sq, H * d)
    _close(o.float(), ref, atol=3e-2, what=f"generic attention {case}")


def test_softmax_rows(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(53)
    x = (torch.randn(37, 196, generator=g) * 3).to(torch.bfloat16).to(cuda)
    out = ops.softmax_rows(x, 196, 0.5)
    _close(out.float(), torch.softmax(x.float() * 0.5, -1), atol=2e-3, what="softmax rows")
    xp = torch.cat([x, torch.full((37, 60), 9.0, dtype=torch.bfloat16, device=cuda)], 1).contiguous()
    outp = ops.softmax_rows(xp, 196, 0.5, cols_pad=256)   # padded columns come out as zeros
    _close(outp[:, :196].float(), torch.softmax(x.float() * 0.5, -1), atol=2e-3, what="softmax rows (padded)")
    assert outp[:, 196:].abs().max().item() == 0.0


def test_timestep_embedding_and_linear(cuda):
    ops = _ops()
    dim, B = 320, 6
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    t = torch.tensor([980.0], device=cuda)
    emb = ops.timestep_embedding(t, B, dim, freqs.to(cuda))
    args = torch.tensor([980.0])[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1).expand(B, -1)
    assert (emb.cpu() - ref).abs().max() < 2e-4
    g = torch.Generator().manual_seed(61)
    w = torch.randn(1280, dim, generator=g) / math.sqrt(dim)
    b = torch.randn(1280, generator=g)
    add = torch.randn(B, 1280, generator=g)
    y = ops.linear_f32(emb, w.to(cuda), b.to(cuda), silu_in=True, silu_out=True, add=add.to(cuda))
    yref = F.silu(F.linear(F.silu(emb.cpu()), w, b)) + add
    assert (y.cpu() - yref).abs().max() < 1e-4


def test_small_convs(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(71)
    # conv_in: 4 -> 320 with CFG-style source duplication and an added NHWC residual
    x = torch.randn(2, 4, 16, 16, generator=g).to(cuda)
    w = torch.randn(320, 4, 3, 3, generator=g) / 6
    b = torch.randn(320, generator=g)
    res = _rand_act(2, 16, 16, 320, cuda, 72)
    out = ops.conv_in_small(x, 4, w.to(cuda), b.to(cuda), 320, 320, res=res)
    ref = F.conv2d(torch.cat([x, x]), w.to(cuda), b.to(cuda), padding=1) + torch.cat([_nchw(res), _nchw(res)])
    _close(_nchw(out), ref, atol=2e-2, what="conv_in_small")
    # 1 -> 16 (padded to 64 channels), SiLU, stride 1; padded channels must be exactly zero
    c = torch.randn(2, 1, 32, 32, generator=g).to(cuda)
    w1 = torch.randn(16, 1, 3, 3, generator=g) / 3
    b1 = torch.randn(16, generator=g)
    o1 = ops.conv_in_small(c, 2, w1.to(cuda), b1.to(cuda), 16, 64, act=1)
    _close(_nchw(o1, 16), F.silu(F.conv2d(c, w1.to(cuda), b1.to(cuda), padding=1)), atol=1e-2, what="cond conv_in")
    assert o1.t[:, 16:].abs().max().item() == 0.0
    # ragged width (not a multiple of the 32-pixel strip), 3 / 8 input channels, stride 2, padded couts, image reuse n % Nsrc
    for (Cin, Cout, ldo, H, W, stride, N) in [(3, 40, 64, 10, 37, 1, 2), (8, 96, 128, 9, 70, 2, 3), (2, 64, 64, 5, 33, 2, 4)]:
        xs = torch.randn(2, Cin, H, W, generator=g).to(cuda)
        ws = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(cuda)
        bs = torch.randn(Cout, generator=g).to(cuda)
        o = ops.conv_in_small(xs, N, ws, bs, Cout, ldo, stride=stride, act=1)
        refx = torch.cat([xs, xs])[:N] if N <= 4 else None
        ref = F.silu(F.conv2d(torch.cat([xs, xs, xs])[:N] if N == 3 else refx, ws, bs, stride=stride, padding=1))
        assert (o.H, o.W) == (ref.shape[2], ref.shape[3])
        _close(_nchw(o, Cout), ref, atol=1e-2, what=f"conv_in_small {Cin}->{Cout} stride {stride}")
        assert o.t[:, Cout:].abs().max().item() == 0.0 if ldo > Cout else True
    # conv_out: 320 -> 4 to fp32 NCHW
    xa = _rand_act(2, 16, 16, 320, cuda, 73)
    wo = (torch.randn(4, 320, 3, 3, generator=g) / math.sqrt(2880))
    bo = torch.randn(4, generator=g)
    wop = wo.permute(0, 2, 3, 1).reshape(4, 9, 320).contiguous().to(torch.bfloat16).to(cuda)
    y = ops.conv_out_small(xa, wop, bo.to(cuda), 4)
    yref = F.conv2d(_nchw(xa), wop.float().reshape(4, 3, 3, 320).permute(0, 3, 1, 2), bo.to(cuda), padding=1)
    assert (y - yref).abs().max() < 2e-3
    # AEKL-style output convs: few input channels (several pixel strips per wave), ragged width, 1 / 3 / 8 couts
    for (Cin, Cout, H, W, N) in [(128, 1, 24, 20, 3), (64, 3, 9, 13, 2), (512, 8, 5, 8, 1), (192, 2, 7, 31, 2),
                                 (72, 2, 6, 9, 1), (320, 4, 16, 70, 1), (128, 5, 3, 97, 2)]:
        xa = _rand_act(N, H, W, Cin, cuda, 74 + Cin)
        wo = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
        bo = torch.randn(Cout, generator=g)
        wop = wo.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().to(torch.bfloat16).to(cuda)
        y = ops.conv_out_small(xa, wop, bo.to(cuda), Cout)
        yref = F.conv2d(_nchw(xa), wop.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bo.to(cuda), padding=1)
        assert y.shape == yref.shape and (y - yref).abs().max() < 2e-3, (Cin, Cout, H, W)


def test_cfg_scheduler_step_and_converters(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(81)
    B = 3
    mo = torch.randn(2 * B, 4, 8, 8, generator=g).to(cuda)
    x = torch.randn(B, 4, 8, 8, generator=g).to(cuda)
    a_t, a_p = 0.3, 0.45
    coef = torch.tensor([a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5, 0.0, 0.0, 2.0, 0.0], device=cuda)
    xs = x.clone()
    x0 = torch.empty_like(x)
    ops.cfg_scheduler_step(mo, 2, 7.0, coef, xs, x0_out=x0)
    u, t = mo.chunk(2)
    v = u + 7.0 * (t - u)
    x0r = a_t ** 0.5 * x - (1 - a_t) ** 0.5 * v
    eps = a_t ** 0.5 * v + (1 - a_t) ** 0.5 * x
    ref = a_p ** 0.5 * x0r + (1 - a_p) ** 0.5 * eps
    assert (xs - ref).abs().max() < 1e-5 and (x0 - x0r).abs().max() < 1e-5
    # layout converters + uint8 map
    y = torch.randn(2, 70, 9, 9, generator=g).to(cuda)
    a = ops.nchw_to_act(y, 128)
    assert a.t[:, 70:].abs().max().item() == 0.0
    back = ops.act_to_nchw(a, 70)
    assert (back - y.to(torch.bfloat16).float()).abs().max().item() == 0.0


@pytest.mark.parametrize("kind", ["ddim", "ddpm"])
@pytest.mark.parametrize("ptype", ["epsilon", "sample", "v_prediction"])
@pytest.mark.parametrize("clip", [False, True])
def test_cfg_modes_and_scheduler_variants_match_oracle(cuda, kind, ptype, clip):
    """Every branch of the fused CFG + scheduler kernel against oracle/schedulers.py + oracle.sampler.cfg_combine:
    guidance None -> text (mode 0), 0 -> uncond (mode 1; the reference's canonical script runs guidance_scale=0.0,
    src/bash/INF_CTRL_CCNET_P1.sh:21), g -> uncond + g (text - uncond) (mode 2), no CFG batch (mode 3, the plain
    scheduler.step of util.py:906); DDIM and DDPM; clip_sample on / off; epsilon / sample / v_prediction.  fp32 both sides:
    1e-5 relative to the largest |reference| value."""
    import ccnet_amd.schedulers as ps
    import oracle.schedulers as osch
    from oracle.sampler import cfg_combine
    sched = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0205, schedule="scaled_linear_beta",
                 prediction_type=ptype, clip_sample=clip)
    P, O = (ps.DDIMScheduler, osch.DDIMScheduler) if kind == "ddim" else (ps.DDPMScheduler, osch.DDPMScheduler)
    p, o = P(**sched), O(**sched)
    p.set_timesteps(50)
    o.set_timesteps(50)
    assert p.timesteps.tolist() == o.timesteps.tolist()
    g = torch.Generator().manual_seed(5)
    B = 3
    for t in (980, 500, 20, 0):
        mo = torch.randn(2 * B, 4, 8, 8, generator=g)
        x = 0.7 * torch.randn(B, 4, 8, 8, generator=g)
        nz = torch.randn(B, 4, 8, 8, generator=g) if kind == "ddpm" else None
        kw = {} if nz is None else {"noise": nz.to(cuda)}
        okw = {} if nz is None else {"noise": nz}
        for gs in (None, 0, 0.0, 1.0, 7.0):
            prev, x0 = p.step_cfg(mo.to(cuda), t, x.to(cuda), gs, **kw)
            rprev, rx0 = o.step(cfg_combine(mo, gs), t, x, **okw)
            for a, b, nm in ((prev, rprev, "prev"), (x0, rx0, "x0")):
                err = (a.cpu() - b).abs().max().item()
                assert err <= 1e-5 * max(1.0, b.abs().max().item()), (kind, ptype, clip, t, gs, nm, err)
            if clip:
                assert x0.abs().max().item() <= 1.0
        prev, x0 = p.step(mo[:B].to(cuda), t, x.to(cuda), **kw)      # mode 3: the model output is the prediction
        rprev, rx0 = o.step(mo[:B], t, x, **okw)
        assert (prev.cpu() - rprev).abs().max().item() <= 1e-5 * max(1.0, rprev.abs().max().item())
        assert (x0.cpu() - rx0).abs().max().item() <= 1e-5 * max(1.0, rx0.abs().max().item())


def _ref_u8(a, f32_scale=False):
    """The reference's numpy recipe, util.py:1162-1173 + 1402-1414 (float64) / 1432-1435 (float32 array in)."""
    import numpy as np
    a = a if f32_scale else a.astype(np.float64)
    scaled = np.array((a - (-1.0)) / float(1.0 - (-1.0)), dtype=float)
    x = np.clip(0.0 + scaled * (1.0 - 0.0), 0.0, 1.0)
    return np.rint(x * 255).astype(np.uint8)[:, 0]


def test_to_uint8_bit_exact(cuda):
    """Integer edge: array_equal against the float64 numpy recipe, including exact .5 ties (rint = ties to
    even), values one float32 ulp either side of every tie, out-of-range values and both channel counts."""
    import numpy as np
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    k = np.arange(0, 255, dtype=np.float64)
    ties = ((k + 0.5) / 255.0 * 2.0 - 1.0).astype(np.float32)  # closest float32 to every tie point
    near = np.concatenate([ties, np.nextafter(ties, np.float32(2)), np.nextafter(ties, np.float32(-2)),
                           np.array([-1.0, 1.0, -1.5, 1.5, 0.0, -0.0, 1e-8, -1e30, 1e30, np.inf, -np.inf], np.float32)])
    # exact ties in float64: x = (2 k + 1) / 255 - 1 is not a float32 in general; 0.5/255 multiples that are: scale by 2^-n
    exact = np.array([(m / 256.0) * 2.0 - 1.0 for m in range(257)], np.float32)  # (x+1)/2*255 = m*255/256: .5 ties at m=128
    vals = np.concatenate([near, exact, (torch.randn(4096, generator=g) * 0.8).numpy()])
    vals = np.concatenate([vals, np.zeros((-len(vals)) % 64, np.float32)]).astype(np.float32)
    for C in (1, 3):
        a = np.stack([vals.reshape(1, -1, 64)] * C, axis=1)  # [1, C, H, 64]; only channel 0 is mapped
        a[:, 1:] = 0.25
        x = torch.from_numpy(a).to(cuda)
        for f32 in (False, True):
            got = ops.to_uint8(x, f32_scale=f32).cpu().numpy()
            assert got.dtype == np.uint8 and np.array_equal(got, _ref_u8(a, f32)), (C, f32)
    img = torch.randn(2, 1, 512, 512, generator=g)
    assert np.array_equal(ops.to_uint8(img.to(cuda)).cpu().numpy(), _ref_u8(img.numpy()))


@pytest.mark.parametrize("rows,C", [(300, 320), (4096, 640), (64, 1280)])
def test_linear1_geglu_fused(cuda, rows, C):
    """MLPBlock.linear1 + GEGLU in one launch (act=2) == a * gelu(gate) of the plain linear."""
    ops = _ops()
    g = torch.Generator().manual_seed(91)
    x = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(cuda)
    w = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).to(torch.bfloat16)
    b = 0.3 * torch.randn(8 * C, generator=g)
    out = ops.linear(x, ops.geglu_interleave(w).to(cuda), ops.geglu_interleave(b).to(cuda), 8 * C, C, act=2)
    assert tuple(out.shape) == (rows, 4 * C)
    a, gate = F.linear(x.float(), w.float().to(cuda), b.to(cuda)).chunk(2, dim=-1)
    _close(out.float(), a * F.gelu(gate), atol=2e-2, what="linear1+geglu")


def test_loader_and_writer_edges(cuda, tmp_path):
    """Bicubic resize + intensity map against F.interpolate (the arithmetic MONAI's Resized(mode="bicubic") runs), and
    the PIL / PIL_comparison writer against the reference's numpy recipe (util.py:1162-1173,1396-1447)."""
    import numpy as np
    from PIL import Image
    from ccnet_amd import io as cio
    ops = _ops()
    g = torch.Generator().manual_seed(91)
    for (H0, W0, H, W) in [(64, 64, 64, 64), (37, 53, 64, 48), (300, 280, 128, 128), (96, 80, 40, 56)]:
        u8 = torch.randint(0, 256, (2, H0, W0), generator=g, dtype=torch.uint8)
        for tr in (False, True):
            src = u8.transpose(1, 2) if tr else u8
            ref = F.interpolate(src[:, None].float().to(cuda), size=(H, W), mode="bicubic", align_corners=False)
            ref = (ref / 255.0 * 2.0 - 1.0).clamp(-1.0, 1.0)
            out = ops.preprocess_u8(u8.to(cuda), H, W, transpose=tr)
            assert out.shape == ref.shape and (out - ref).abs().max().item() < 2e-5, (H0, W0, H, W, tr)
    # PNG round trip through load_images: same-size images come back as x/127.5 - 1 of the transposed array
    a = np.random.RandomState(3).randint(0, 256, (48, 32), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "src.png")
    x = cio.load_images([tmp_path / "src.png"], 32, 48, device=cuda)
    assert torch.allclose(x[0, 0].cpu(), torch.from_numpy(a.T.astype(np.float32)) / 127.5 - 1.0, atol=1e-6)
    # writer
    smp = torch.rand(2, 3, 16, 24, generator=g) * 2.4 - 1.2
    cond = torch.rand(2, 1, 16, 24, generator=g) * 2 - 1
    real = torch.rand(2, 1, 16, 24, generator=g) * 2 - 1
    paths = cio.save_controlnet_inference_sample(tmp_path / "run", ["a.png", "b.png"], smp.to(cuda), cond.to(cuda), real.to(cuda))
    assert [p.parent.name for p in paths] == ["PIL", "PIL", "PIL_comparison", "PIL_comparison"]

    # integer outputs: equal to the reference's numpy recipe bit for bit (float64 for the synthetic slice, float32 scaling
    # for cond / images, which reach interval_mapping as float32 arrays: util.py:1402-1414 vs 1432-1435)
    s8, c8, r8 = _ref_u8(smp.numpy()), _ref_u8(cond.numpy(), True), _ref_u8(real.numpy(), True)
    for i, nm in enumerate(["a.png", "b.png"]):
        got = np.asarray(Image.open(tmp_path / "run" / "PIL" / nm))
        assert got.shape == (16, 24) and np.array_equal(got, s8[i])
        trip = np.asarray(Image.open(tmp_path / "run" / "PIL_comparison" / nm))
        assert trip.shape == (16, 72) and np.array_equal(trip[:, :24], c8[i]) and np.array_equal(trip[:, 48:], r8[i])
        assert np.array_equal(trip[:, 24:48], got)


@pytest.mark.parametrize("rows,C,geglu", [(8192, 320, True), (8256, 320, True), (16384, 640, True), (24576, 1280, False),
                                          (70000, 320, False)])
def test_persistent_tile_loop_matches_one_tile_per_block(cuda, rows, C, geglu):
    """Persistent launches (one block per CU walking tiles, next tile's stages issued before the epilogue, stores of a
    tile draining under the next tile's first K-steps behind counted vmcnt waits) must give bit-identical results to
    the same kernel launched one block per tile -- full and ragged tile grids, repeated to give a race a chance."""
    from ccnet_amd import lib as _lib
    ops = _ops()
    l = _lib.load()
    g = torch.Generator().manual_seed(97)
    x = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(cuda)
    Cout = 8 * C if geglu else (1280 if C == 1280 else 3 * C)
    w = (torch.randn(Cout, C, generator=g) / math.sqrt(C)).to(torch.bfloat16).to(cuda)
    b = (0.3 * torch.randn(Cout, generator=g)).to(cuda)
    if geglu:
        w, b = ops.geglu_interleave(w), ops.geglu_interleave(b)
    try:
        l.ccnet_set_option(4, 0)
        ref = ops.linear(x, w, b, Cout, C, act=2 if geglu else 0).clone()
        l.ccnet_set_option(4, 1)
        for _ in range(6):
            out = ops.linear(x, w, b, Cout, C, act=2 if geglu else 0)
            assert torch.equal(out, ref)
    finally:
        l.ccnet_set_option(4, 1)
    if not geglu:  # and against torch
        _close(out.float(), F.linear(x.float(), w.float(), b), atol=3e-2, what="persistent linear")


def test_conv_dispatch_sweep(cuda):
    """Seeded sweep over shapes that land on every branch of the tile / split-K dispatch (small and ragged M, couts that
    are multiples of 8 / 128 / 256 / 320, short and deep K, residual / rowvec / GEGLU / scale variants) against F.conv2d."""
    import random
    ops = _ops()
    rnd = random.Random(1234)
    g = torch.Generator().manual_seed(1235)
    shapes = []
    for _ in range(28):
        R = rnd.choice([1, 1, 3])
        C1 = 64 * rnd.choice([1, 2, 5, 10, 20])
        C2 = 64 * rnd.choice([0, 0, 0, 1, 5])
        Cout = rnd.choice([8, 40, 64, 128, 256, 320, 512, 640, 960, 1280, 2560])
        H = rnd.choice([3, 8, 13, 16, 32])
        N = rnd.choice([1, 2, 5, 8, 32])
        if N * H * H * (C1 + C2 + Cout) > 6e7 or R * R * (C1 + C2) * Cout > 3e7:
            N, H = min(N, 2), min(H, 16)
        mode = rnd.choice(["plain", "plain", "res", "rowvec", "scale", "geglu"])
        shapes.append((N, H, C1, C2, Cout, R, mode))
    shapes += [(32, 8, 1280, 0, 1280, 3, "res"), (32, 8, 2560, 0, 1280, 1, "plain"), (8, 32, 640, 0, 1280, 1, "res"),
               (2, 64, 320, 0, 2560, 1, "geglu"), (32, 16, 1280, 0, 1280, 1, "res"), (32, 16, 1280, 0, 3840, 1, "plain"),
               # 128x320 tiles: residual-prefetch persistent instance (K = 320 with a residual, full and ragged M, two column
               # tiles) and the one-round instance at M = 8192 without a residual
               (32, 64, 320, 0, 320, 1, "res"), (5, 115, 320, 0, 320, 1, "res"), (16, 64, 320, 0, 640, 1, "res"),
               (32, 16, 1280, 0, 1280, 1, "plain"), (32, 16, 640, 0, 1280, 1, "scale")]
    for (N, H, C1, C2, Cout, R, mode) in shapes:
        x1 = _rand_act(N, H, H, C1, cuda, 11)
        x2 = _rand_act(N, H, H, C2, cuda, 12) if C2 else None
        Cin = C1 + C2
        if mode == "geglu":
            if Cout % 64 or R != 1:
                mode = "plain"
        w = (torch.randn(Cout, Cin * R * R, generator=g) / math.sqrt(Cin * R * R)).to(torch.bfloat16).to(cuda)
        b = torch.randn(Cout, generator=g).to(cuda)
        xin = _nchw(x1) if x2 is None else torch.cat([_nchw(x1), _nchw(x2)], 1)
        ref = F.conv2d(xin, w.float().reshape(Cout, R, R, Cin).permute(0, 3, 1, 2), b, padding=R // 2)
        what = f"sweep N={N} H={H} C1={C1} C2={C2} Cout={Cout} R={R} {mode}"
        if mode == "plain":
            out = ops.conv(x1, w, b, Cout, R=R, x2=x2)
        elif mode == "res":
            r1 = _rand_act(N, H, H, Cout, cuda, 13)
            out = ops.conv(x1, w, b, Cout, R=R, x2=x2, res1=r1)
            ref = ref + _nchw(r1)
        elif mode == "rowvec":
            rv = torch.randn(N, Cout, generator=g).to(cuda)
            out = ops.conv(x1, w, b, Cout, R=R, x2=x2, rowvec=rv, act=1)
            ref = F.silu(ref + rv[:, :, None, None])
        elif mode == "scale":
            out = ops.conv(x1, w, b, Cout, R=R, x2=x2, scale=0.37)
            ref = ref * 0.37
        else:
            out = ops.conv(x1, ops.geglu_interleave(w), ops.geglu_interleave(b), Cout, R=R, x2=x2, act=2)
            a, gate = ref.chunk(2, dim=1)
            ref = a * F.gelu(gate)
            assert out.C == Cout // 2
        _close(_nchw(out, ref.shape[1]), ref, atol=3e-2, what=what)


# ------------------------------------------------------------------ text tower kernels
@pytest.mark.parametrize("case", [(2, 2, 77), (1, 3, 200), (3, 1, 128), (2, 2, 5), (1, 2, 300)])
def test_causal_attention_matches_masked_softmax(case):
    B, H, S = case
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(700 + S)
    qkv = torch.randn(B * S, 3 * H * 64, generator=g).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :H * 64], qkv[:, H * 64:2 * H * 64], qkv[:, 2 * H * 64:]
    out = ops.attention_d64_causal(q, k, v, B, H, S, 0.125).float()
    qf, kf, vf = (t.float().view(B, S, H, 64).transpose(1, 2) for t in (q, k, v))
    mask = torch.full((S, S), float("-inf"), device=dev).triu(1)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * 0.125 + mask, -1) @ vf).transpose(1, 2).reshape(B * S, H * 64)
    _close(out, ref, atol=2e-2, what=f"causal attention {case}")
    # the first query sees only the first key: its output is V[0] exactly (bf16 round trip aside)
    first = out.view(B, S, H * 64)[:, 0]
    _close(first, v.float().view(B, S, H * 64)[:, 0], atol=1e-6, rtol=2.0 ** -8, what="first row")


@pytest.mark.parametrize("quick", [False, True])
def test_gelu_modes(quick):
    ops = _ops()
    dev = torch.device("cuda:0")
    x = (torch.randn(77 * 3, 512 + 64, generator=torch.Generator().manual_seed(71)) * 3).to(torch.bfloat16).to(dev)
    out = ops.gelu(x, 512, quick).float()
    xf = x[:, :512].float()
    ref = xf * torch.sigmoid(1.702 * xf) if quick else F.gelu(xf)
    _close(out, ref, atol=1e-3, what="gelu")


def test_embed_tokens():
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(72)
    tok, pos = torch.randn(500, 128, generator=g).to(dev), torch.randn(77, 128, generator=g).to(dev)
    ids = torch.randint(0, 500, (4 * 77,), generator=g).to(dev)
    out = ops.embed_tokens(ids, 77, tok, pos).float()
    ref = (tok[ids] + pos.repeat(4, 1)).to(torch.bfloat16).float()
    assert torch.equal(out, ref)


def _ff_inputs(M, extra, cuda, seed=123):
    g = torch.Generator().manual_seed(seed)
    C_, Hd = 320, 1280
    xs = torch.randn(M, C_ + extra, generator=g)
    xs[1::7] += 20.0   # rows with a large mean (two-pass statistics)
    xs[5::11] *= 3.0
    x = xs.to(torch.bfloat16).to(cuda)[:, extra:]
    gamma, beta = 1 + 0.2 * torch.randn(C_, generator=g), 0.3 * torch.randn(C_, generator=g)
    w1 = torch.randn(2 * Hd, C_, generator=g) / math.sqrt(C_)
    b1 = 0.5 * torch.randn(2 * Hd, generator=g)
    w2 = torch.randn(C_, Hd, generator=g) * (0.5 / math.sqrt(Hd))
    b2 = 0.5 * torch.randn(C_, generator=g)
    return x, gamma.to(cuda), beta.to(cuda), w1.to(cuda), b1.to(cuda), w2.to(cuda), b2.to(cuda)


@pytest.mark.parametrize("case", [(128, 0), (677, 0), (4096 + 37, 64), (32768, 0), (131072, 0)])
def test_ff_fused_320(cuda, case):
    """x + linear2(GEGLU(linear1(LayerNorm(x)))) in one kernel (ccnet_ff_geglu_fused_bf16; SURVEY Appendix A.3: norm3 +
    MLPBlock(act="GEGLU") + residual) at the G1 64x64-level width: against the fp32 chain F.layer_norm -> F.linear ->
    a * gelu(g) -> F.linear -> + x on the same bf16-rounded operands, and against the three-launch path it replaces
    (LayerNorm kernel, linear1 + GEGLU epilogue, linear2 + residual: same rounding points, so nearly every element agrees to
    an ulp).  Row counts: one tile, ragged (not a multiple of the 128-row tile or of 32), a column view of wider rows, several
    tiles per workgroup (131072 rows = 1024 tiles over 256 workgroups)."""
    ops = _ops()
    M, extra = case
    x, gamma, beta, w1, b1, w2, b2 = _ff_inputs(M, extra, cuda)
    pack = ops.ff_pack(w1, b1, w2)
    assert pack is not None and pack[0].shape == (80, 15360)
    out = ops.ff_fused(x, 320, pack, b2, gamma, beta, 1e-5)
    w1b, w2b = w1.to(torch.bfloat16), w2.to(torch.bfloat16)
    ln = F.layer_norm(x.float(), (320,), gamma, beta, 1e-5)
    h = ln.to(torch.bfloat16).float() @ w1b.float().T + b1
    hid = (h[:, :1280] * F.gelu(h[:, 1280:])).to(torch.bfloat16).float()
    ref = hid @ w2b.float().T + b2 + x.float()
    _close(out.float(), ref, atol=3e-2, what=f"fused feed-forward {case}")
    t = ops.layernorm(x, 320, gamma, beta)
    gl = ops.linear(t, ops.geglu_interleave(w1b).contiguous(), ops.geglu_interleave(b1).contiguous(), 2560, 320, act=2)
    un = ops.linear(gl, w2b.contiguous(), b2, 320, 1280, res1=x)
    d = (out.float() - un.float()).abs()
    scale = un.float().abs().max().item()
    assert d.max().item() <= 2.0 ** -6 * scale, f"vs the three-launch path: max {d.max().item()} (|out| max {scale})"
    assert (d > 0).float().mean().item() < 0.25


@pytest.mark.parametrize("case", [(128, 128, 77), (4096 * 2, 4096, 77), (4096 + 200, 4096, 80), (128 * 600 - 37, 128, 17),
                                  (32 * 4096, 4096, 77)])
def test_xattn_fused_320(cuda, case):
    """x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(64)) V) in one kernel (ccnet_xattn_fused_bf16; SURVEY Appendix A.3: norm2 +
    CrossAttention + residual) at the G1 64x64-level width, 5 heads of 64: against the fp32 chain on the same bf16-rounded
    operands, and against the three launches it replaces (LayerNorm + to_q with the scale folded in, flash kernel, to_out +
    residual).  Cases: one tile; two images; ragged rows with 80 keys (no masked key); one image per tile, so that a workgroup
    reloads K / V between its tiles, with few keys; the full CFG batch (several tiles per workgroup inside one image)."""
    ops = _ops()
    M, HW, S = case
    images = -(-M // HW)
    g = torch.Generator().manual_seed(1234 + M)
    xs = torch.randn(M, 320, generator=g)
    xs[1::7] += 10.0
    x = xs.to(torch.bfloat16).to(cuda)
    gamma, beta = (1 + 0.2 * torch.randn(320, generator=g)).to(cuda), (0.3 * torch.randn(320, generator=g)).to(cuda)
    wq = (torch.randn(320, 320, generator=g) * (2.0 / math.sqrt(320))).to(cuda)
    wo = (torch.randn(320, 320, generator=g) / math.sqrt(320)).to(cuda)
    bo = (0.5 * torch.randn(320, generator=g)).to(cuda)
    kv = torch.randn(images * S, 640, generator=g).to(torch.bfloat16).to(cuda)
    pack = ops.xattn_pack(wq, None, wo, 5)
    assert pack is not None and pack[0].shape == (20, 10240)
    img = ops.xattn_kv_image(kv, images, S, 320)
    assert img.shape == (images, 51200)
    out = ops.xattn_fused(x, 320, 5, pack, bo, gamma, beta, img, S, HW, 1e-5)
    # fp32 reference on the same rounded operands (the kernel rounds to_q's weight after folding the scale in)
    qs = math.log2(math.e) / 8.0
    wqb = (wq * qs).to(torch.bfloat16).float() / qs
    ln = F.layer_norm(x.float(), (320,), gamma, beta, 1e-5).to(torch.bfloat16).float()
    q = (ln @ wqb.T).to(torch.bfloat16).float()
    ref = torch.empty(M, 320, device=cuda)
    for i in range(images):
        r0, r1 = i * HW, min((i + 1) * HW, M)
        k = kv[i * S:(i + 1) * S, :320].float().reshape(S, 5, 64).permute(1, 0, 2)
        v = kv[i * S:(i + 1) * S, 320:].float().reshape(S, 5, 64).permute(1, 0, 2)
        qi = q[r0:r1].reshape(-1, 5, 64).permute(1, 0, 2)
        p_ = torch.softmax(qi @ k.transpose(1, 2) / 8.0, dim=-1)
        ref[r0:r1] = (p_ @ v).permute(1, 0, 2).reshape(-1, 320)
    ref = ref.to(torch.bfloat16).float() @ wo.to(torch.bfloat16).float().T + bo + x.float()
    _close(out.float(), ref, atol=4e-2, what=f"fused cross-attention {case}")
    if M % HW == 0:
        t = ops.layernorm(x, 320, gamma, beta)
        q3 = ops.linear(t, (wq * qs).to(torch.bfloat16).contiguous(), None, 320, 320)
        o3 = ops.attention_d64(q3, kv[:, :320], kv[:, 320:], images, 5, HW, S, 1.0 / 8.0, prescaled=True)
        un = ops.linear(o3, wo.to(torch.bfloat16).contiguous(), bo, 320, 320, res1=x)
        d = (out.float() - un.float()).abs()
        scale = un.float().abs().max().item()
        assert d.max().item() <= 2.0 ** -5 * scale, f"vs the three-launch path: max {d.max().item()} (|out| max {scale})"


def test_xattn_fused_unsupported_shape(cuda):
    ops = _ops()
    assert ops.xattn_pack(torch.zeros(640, 640), None, torch.zeros(640, 640), 10) is None
    assert ops.xattn_kv_image(torch.zeros(2 * 96, 640, dtype=torch.bfloat16), 2, 96, 320) is None


def test_ff_fused_unsupported_shape(cuda):
    ops = _ops()
    assert ops.ff_pack(torch.zeros(2 * 2560, 640), torch.zeros(2 * 2560), torch.zeros(640, 2560)) is None
