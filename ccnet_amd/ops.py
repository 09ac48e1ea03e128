"""Tensor-level wrappers over the C ABI (include/ccnet_hip.h).

PyTorch is used only as the owner of device memory and of the HIP stream; every op below enqueues one or two
hand-written gfx950 kernels on `torch.cuda.current_stream()`.  Activations travel as `Act`: a 2-D bf16
tensor [N*H*W, C] (NHWC, "pixel rows") plus its geometry.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import lib as _l

BF16 = torch.bfloat16


@dataclass
class Act:
    t: torch.Tensor  # [N*H*W, ld] bf16, row stride t.stride(0)
    N: int
    H: int
    W: int
    C: int

    @property
    def ld(self) -> int:
        return self.t.stride(0)

    @property
    def rows(self) -> int:
        return self.N * self.H * self.W


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _l.CcnetError("ccnet_amd ops run on the GPU only (tensor is on %s); there is no CPU path" % t.device)


def new_act(N, H, W, Cch, device) -> Act:
    return Act(torch.empty((N * H * W, Cch), dtype=BF16, device=device), N, H, W, Cch)


def conv(x: Act, w: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, R: int = 3, stride: int = 1,
         pad: Optional[int] = None, ups: int = 1, x2: Optional[Act] = None, rowvec: Optional[torch.Tensor] = None,
         res1: Optional[Act] = None, scale: float = 1.0, act: int = 0, res2: Optional[Act] = None,
         out_hw: Optional[Tuple[int, int]] = None, out: Optional[Act] = None,
         rowstat: Optional[torch.Tensor] = None, colsum: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
         emit_stats: bool = False, out2: Optional[Act] = None):
    """Implicit-GEMM conv / linear. w: bf16 [Cout, R*R*(C1+C2)]. Returns Act, or (Act, Act2) when res2 is given
    (`out` / `out2`: write into these instead of fresh tensors, e.g. row ranges of a larger one).
    rowstat / colsum: LayerNorm folded into the layer (include/ccnet_hip.h: x holds the un-normalised rows, w and bias the
    folded parameters of `fold_layernorm`, rowstat = row_stats(x) [rows, 2], or a producer's partials [rows, P, 2]).
    colsum alone: the kernel takes the statistics itself (only where `ln_inline_ok`).
    emit_stats: returns (Act, stats) with stats = the per-row (mean, M2) partials [rows, P, 2] of the output, taken in the
    epilogue, or None when the layer's plan cannot emit them (the caller then runs row_stats on the output)."""
    lib = _l.load()
    _need_cuda(x.t, w)
    if pad is None:
        pad = R // 2
    Hl, Wl = x.H * ups, x.W * ups
    if out_hw is None:
        Ho = (Hl + 2 * pad - R) // stride + 1
        Wo = (Wl + 2 * pad - R) // stride + 1
    else:
        Ho, Wo = out_hw
    cout_out = Cout // 2 if act == 2 else Cout  # GEGLU epilogue halves the width
    if out is None:
        out = new_act(x.N, Ho, Wo, cout_out, x.t.device)
    if res2 is None:
        out2 = None
    elif out2 is None:
        out2 = new_act(x.N, Ho, Wo, cout_out, x.t.device)
    p = _l.ConvParams()
    p.x1 = x.t.data_ptr(); p.x2 = _ptr(x2.t) if x2 is not None else None
    p.w = w.data_ptr(); p.bias = _ptr(bias); p.rowvec = _ptr(rowvec)
    p.res1 = _ptr(res1.t) if res1 is not None else None
    p.res2 = _ptr(res2.t) if res2 is not None else None
    p.out = out.t.data_ptr(); p.out2 = _ptr(out2.t) if out2 is not None else None
    p.N, p.Hin, p.Win, p.Hout, p.Wout = x.N, x.H, x.W, Ho, Wo
    p.C1 = x.C; p.C2 = x2.C if x2 is not None else 0
    p.Cout = Cout; p.R = R; p.stride = stride; p.pad = pad; p.ups = ups
    p.ldx1 = x.ld; p.ldx2 = x2.ld if x2 is not None else 0
    p.ldw = w.stride(0); p.ldo = out.ld; p.ldo2 = out2.ld if out2 is not None else 0
    p.ldres1 = res1.ld if res1 is not None else 0
    p.ldres2 = res2.ld if res2 is not None else 0
    p.ld_rowvec = rowvec.stride(0) if rowvec is not None else 0
    p.act = act; p.scale = scale; p.batch = 1
    if rowstat is not None and colsum is None:
        raise _l.CcnetError("rowstat needs colsum")
    if rowstat is None and colsum is not None:
        # LayerNorm statistics taken inside the streaming pointwise kernel (ln_inline_ok says which layers it takes)
        if colsum.dtype != torch.float32 or colsum.numel() < Cout:
            raise _l.CcnetError("colsum must be fp32 [>= Cout]")
        p.colsum = colsum.data_ptr(); p.ln_eps = ln_eps
    if rowstat is not None:
        parts = rowstat.shape[1] if rowstat.dim() == 3 else 0
        if rowstat.dtype != torch.float32 or rowstat.numel() != 2 * out.rows * max(parts, 1) or colsum.dtype != torch.float32 \
                or colsum.numel() < Cout or not rowstat.is_contiguous() or (parts and x.C % parts):
            raise _l.CcnetError("rowstat must be fp32 [rows, 2] or [rows, P, 2] with P | C1, and colsum fp32 [>= Cout]")
        p.rowstat = rowstat.data_ptr(); p.colsum = colsum.data_ptr(); p.rowstat_parts = parts; p.ln_eps = ln_eps
    if w.shape[1] != R * R * (p.C1 + p.C2):
        raise _l.CcnetError(f"weight K {w.shape[1]} != R*R*(C1+C2) = {R * R * (p.C1 + p.C2)}")
    if res1 is not None and res1.rows != out.rows:
        raise _l.CcnetError("res1 rows mismatch")
    if rowvec is not None and (rowvec.shape[0] != x.N or rowvec.shape[1] < Cout or rowvec.dtype != torch.float32):
        raise _l.CcnetError("rowvec must be fp32 [N, >=Cout]")
    stats = None
    if emit_stats:
        if out2 is not None:
            raise _l.CcnetError("emit_stats and res2 do not combine")
        parts = lib.ccnet_conv_stat_parts(C.byref(p))
        if parts in (2, 4, 8):  # what a consuming GEMM merges (rowstat_parts)
            stats = torch.empty((out.rows, parts, 2), dtype=torch.float32, device=x.t.device)
            p.stat_out = stats.data_ptr()
    _launch_conv(p, x.t.device, "conv_gemm")
    if emit_stats:
        return out, stats
    return (out, out2) if out2 is not None else out


def _launch_conv(p, device, what: str) -> None:
    """ccnet_conv_gemm_bf16 with the split-K workspace the library asks for (every caller goes through here: a deep-K
    problem with few tiles -- also a batch-1 P.V product of the generic attention path -- is planned with K slices)."""
    lib = _l.load()
    wsb = lib.ccnet_conv_workspace_bytes(C.byref(p))
    ws = None
    if wsb:
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=device)  # split-K partial sums
        p.workspace = ws.data_ptr(); p.workspace_bytes = wsb
    _l.check(lib.ccnet_conv_gemm_bf16(C.byref(p), _stream()), what)


def gemm_batched(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, M: int, Nn: int, K: int, batch: int,
                 ldx: int, ldw: int, ldo: int, sx: int, sw: int, so: int):
    """out[z][m][n] = sum_k x[z][m][k] * w[z][n][k] (bf16, K % 64 == 0) -- attention fallback for head dim != 64."""
    lib = _l.load()
    p = _l.ConvParams()
    p.x1 = x.data_ptr(); p.w = w.data_ptr(); p.out = out.data_ptr()
    p.N, p.Hin, p.Win, p.Hout, p.Wout = 1, M, 1, M, 1
    p.C1 = K; p.C2 = 0; p.Cout = Nn; p.R = 1; p.stride = 1; p.pad = 0; p.ups = 1
    p.ldx1 = ldx; p.ldw = ldw; p.ldo = ldo; p.scale = 1.0; p.batch = batch
    p.sx1 = sx; p.sw = sw; p.so = so
    _launch_conv(p, x.device, "gemm_batched")


def linear(x2d: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, Cin: int,
           res1: Optional[torch.Tensor] = None, scale: float = 1.0, act: int = 0,
           rowstat: Optional[torch.Tensor] = None, colsum: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
           emit_stats: bool = False):
    """Row-matrix convenience over conv(): x2d [rows, >=Cin] bf16 -> [rows, Cout] (Cout/2 with act=2, GEGLU); with
    emit_stats (tensor, stats-or-None) as conv()."""
    rows = x2d.shape[0]
    xa = Act(x2d, 1, rows, 1, Cin)
    ra = Act(res1, 1, rows, 1, Cout) if res1 is not None else None
    r = conv(xa, w, bias, Cout, R=1, pad=0, res1=ra, scale=scale, act=act, rowstat=rowstat, colsum=colsum, ln_eps=ln_eps,
             emit_stats=emit_stats)
    return (r[0].t, r[1]) if emit_stats else r.t


def ln_inline_ok(rows: int, Cin: int, Cout: int) -> bool:
    """Whether linear(x, w, b, Cout, Cin, colsum=...) WITHOUT rowstat is served: the layers that the weight-stationary streaming
    kernel takes with the LayerNorm statistics computed in the kernel (gemm_conv.hip try_pw320: 320 input channels, whole
    320-cout groups, enough rows to fill the chip); any other shape returns CCNET_EUNSUP from the library."""
    return Cin == 320 and Cout % 320 == 0 and Cout // 320 <= 32 and rows >= 65536


def row_stats(x2d: torch.Tensor, Cc: int, eps: float = 1e-5) -> torch.Tensor:
    """fp32 [rows, 2] = (mean, 1 / sqrt(var + eps)) over the first Cc columns of every row: the statistics of the LayerNorm
    that `linear(..., rowstat=, colsum=)` applies inside the GEMM."""
    lib = _l.load()
    _need_cuda(x2d)
    out = torch.empty((x2d.shape[0], 2), dtype=torch.float32, device=x2d.device)
    _l.check(lib.ccnet_row_stats_bf16(x2d.data_ptr(), x2d.stride(0), x2d.shape[0], Cc, eps, out.data_ptr(), _stream()),
             "row_stats")
    return out


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """Parameters of `Linear(LayerNorm(x))` as one GEMM on the un-normalised x: (w_folded bf16 [N, K] = gamma o W, bias_folded
    fp32 [N] = bias + W beta, colsum fp32 [N] = row sums of the ROUNDED folded weights -- what the kernel subtracts mean
    times).  w fp32 / bf16 [N, K] (nn.Linear layout); gamma, beta [K]."""
    w32 = w.detach().float()
    wf = (w32 * gamma.detach().float()[None, :]).to(BF16).contiguous()
    bf = w32 @ beta.detach().float()
    if bias is not None:
        bf = bf + bias.detach().float()
    return wf, bf.contiguous(), wf.float().sum(dim=1).contiguous()


def geglu_interleave(w: torch.Tensor) -> torch.Tensor:
    """Reorder the rows of MLPBlock.linear1 (weight [2H, C] or bias [2H]; value half first, gate half second) into
    alternating 8-row blocks [a0..7, g0..7, a8..15, g8..15, ...] as the act=2 epilogue expects (the 8 couts that a lane pair
    of the MFMA accumulator layout owns per register quad: value and gate of an output meet in one lane)."""
    H = w.shape[0] // 2
    a, g = w[:H], w[H:]
    sh = (H // 8, 8) + tuple(w.shape[1:])
    return torch.stack([a.reshape(sh), g.reshape(sh)], dim=1).reshape(w.shape).contiguous()


def ff_pack(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor):
    """MLPBlock(act="GEGLU") weights in the staging order of ccnet_ff_geglu_fused_bf16 (include/ccnet_hip.h), or None when
    the kernel does not cover the shape.  w1 [2H, C] / b1 [2H] (value half first, gate half second), w2 [C, H].
    Returns (w_packed bf16 [H/16, 15360], b1 interleaved fp32 [2H])."""
    H2, Cc = w1.shape
    H = H2 // 2
    if _l.load().ccnet_ff_packed_bytes(Cc, H) == 0:
        return None
    dev = w1.device
    nb = H // 16
    w1i = geglu_interleave(w1.detach().float()).to(BF16).reshape(nb, 32, Cc // 8, 8)  # [block][row][16-byte unit][8]
    r = torch.arange(32, device=dev)
    pos = torch.arange(Cc // 8, device=dev)
    src = pos[None, :] ^ ((r[:, None] >> 1) & 7)                                        # position p of row r holds unit p ^ swz(r)
    img1 = w1i[:, r[:, None], src]                                                      # [nb, 32, units, 8]
    t = torch.arange(8, device=dev)
    hmap = torch.stack([torch.where(t < 4, 4 * hh + t, 4 + 4 * hh + t) for hh in (0, 1)])  # [2, 8]
    w2b = w2.detach().to(BF16).reshape(Cc, nb, 16)                                      # [cout][block][hidden in block]
    img2 = w2b[:, :, hmap].permute(1, 2, 0, 3)                                          # [nb, 2, cout, 8]
    img2 = torch.roll(img2, 1, 0)   # linear2 runs one hidden block behind linear1: block j's stage carries W2 of block j - 1 (mod nb)
    wp = torch.cat([img1.reshape(nb, -1), img2.reshape(nb, -1)], dim=1).contiguous()
    return wp, geglu_interleave(b1.detach().float()).contiguous()


class FfParams(C.Structure):
    """Mirror of `ccnet_ff_params` (include/ccnet_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("w_packed", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int32), ("C", C.c_int32), ("hidden", C.c_int32),
                ("ldx", C.c_int32), ("ldo", C.c_int32), ("eps", C.c_float)]


def ff_fused(x2d: torch.Tensor, Cc: int, pack, b2: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
             eps: float = 1e-5) -> torch.Tensor:
    """x + linear2(GEGLU(linear1(LayerNorm(x)))) in one kernel; `pack` from ff_pack (not None)."""
    lib = _l.load()
    _need_cuda(x2d, pack[0])
    out = torch.empty((x2d.shape[0], Cc), dtype=BF16, device=x2d.device)
    p = FfParams()
    p.x = x2d.data_ptr(); p.w_packed = pack[0].data_ptr(); p.b1 = pack[1].data_ptr(); p.b2 = b2.data_ptr()
    p.gamma = gamma.data_ptr(); p.beta = beta.data_ptr(); p.out = out.data_ptr()
    p.M = x2d.shape[0]; p.C = Cc; p.hidden = pack[1].numel() // 2; p.ldx = x2d.stride(0); p.ldo = out.stride(0); p.eps = eps
    _l.check(lib.ccnet_ff_geglu_fused_bf16(C.byref(p), _stream()), "ff_geglu_fused")
    return out


def xattn_pack(wq: torch.Tensor, bq: Optional[torch.Tensor], wo: torch.Tensor, heads: int):
    """CrossAttention weights in the staging order of ccnet_xattn_fused_bf16 (include/ccnet_hip.h), or None when the kernel does
    not cover the shape.  wq [C, C] (to_q, no transpose), bq [C] or None, wo [C, C] (to_out[0]).  Returns (w_packed bf16
    [20, 10240], bq fp32 [C] in the packed row order): the softmax scale and log2 e are folded into to_q before its rounding."""
    Cc = wq.shape[0]
    if _l.load().ccnet_xattn_packed_bytes(Cc, heads) == 0 or wq.shape[1] != Cc or tuple(wo.shape) != (Cc, Cc):
        return None
    dev = wq.device
    hd = Cc // heads
    qs = math.log2(math.e) / math.sqrt(hd)
    # accumulator register order -> natural head-dim order: position p of a 16-row group holds row pi[p] of that group
    pi = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15], device=dev)
    rows = (torch.arange(Cc, device=dev) // 16) * 16 + pi[torch.arange(Cc, device=dev) % 16]
    wqp = (wq.detach().float() * qs)[rows].to(BF16).reshape(Cc // 32, 32, Cc // 8, 8)      # [block][row][16-byte unit][8]
    r = torch.arange(32, device=dev)
    pos = torch.arange(Cc // 8, device=dev)
    src = pos[None, :] ^ ((r[:, None] >> 1) & 7)                                           # position p of row r holds unit p ^ swz(r)
    img_q = wqp[:, r[:, None], src]                                                        # [10, 32, units, 8]
    t = torch.arange(8, device=dev)
    hmap = torch.stack([torch.where(t < 4, 4 * hh + t, 4 + 4 * hh + t) for hh in (0, 1)])  # [2, 8]
    wob = wo.detach().to(BF16).reshape(Cc, Cc // 16, 16)                                   # [cout][k-step][dim in k-step]
    img_o = wob[:, :, hmap].permute(1, 2, 0, 3)                                            # [k-step (head-major), 2, cout, 8]
    wp = torch.cat([img_q.reshape(-1), img_o.reshape(-1)]).reshape(-1, 10240).contiguous()
    bqp = (bq.detach().float() * qs)[rows].contiguous() if bq is not None else torch.zeros(Cc, dtype=torch.float32, device=dev)
    return wp, bqp


def xattn_kv_image(kv: torch.Tensor, images: int, S: int, Cc: int) -> Optional[torch.Tensor]:
    """The prompt's keys / values ([images * S, 2 C]: K | V, as to_k | to_v write them) as the per-image LDS images of
    ccnet_xattn_fused_bf16: [80 keys][C] with swizzled 16-byte units, then V^T as [5 k-steps][2][C][8] with the keys of a k-step
    in the order in which a lane holds the probabilities.  Once per prompt batch.  None: S > 80."""
    if S > 80:
        return None
    dev = kv.device
    k = torch.zeros((images, 80, Cc), dtype=BF16, device=dev)
    v = torch.zeros((images, 80, Cc), dtype=BF16, device=dev)
    k[:, :S] = kv[:, :Cc].reshape(images, S, Cc)
    v[:, :S] = kv[:, Cc:2 * Cc].reshape(images, S, Cc)
    r = torch.arange(80, device=dev)
    pos = torch.arange(Cc // 8, device=dev)
    src = pos[None, :] ^ ((r[:, None] >> 1) & 7)
    kimg = k.reshape(images, 80, Cc // 8, 8)[:, r[:, None], src]                           # [images, 80, units, 8]
    t = torch.arange(8, device=dev)
    hmap = torch.stack([torch.where(t < 4, 4 * hh + t, 4 + 4 * hh + t) for hh in (0, 1)])  # [2, 8]
    keys = 16 * torch.arange(5, device=dev)[:, None, None] + hmap[None]                    # [5, 2, 8]
    vimg = v[:, keys].permute(0, 1, 2, 4, 3)                                               # [images, 5, 2, C, 8]
    return torch.cat([kimg.reshape(images, -1), vimg.reshape(images, -1)], dim=1).contiguous()


class XattnParams(C.Structure):
    """Mirror of `ccnet_xattn_params` (include/ccnet_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("w_packed", C.c_void_p), ("bq", C.c_void_p), ("bo", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("kv_image", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int32), ("C", C.c_int32),
                ("heads", C.c_int32), ("keys", C.c_int32), ("rows_per_image", C.c_int32), ("images", C.c_int32),
                ("ldx", C.c_int32), ("ldo", C.c_int32), ("eps", C.c_float)]


def xattn_fused(x2d: torch.Tensor, Cc: int, heads: int, pack, bo: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                kv_image: torch.Tensor, S: int, rows_per_image: int, eps: float = 1e-5,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x + to_out(attention(to_q(LayerNorm(x)), K, V)) in one kernel; `pack` from xattn_pack, `kv_image` from xattn_kv_image (the
    images that rows 0, rows_per_image, ... of x attend to).  Raises CcnetUnsupported for shapes the kernel does not cover."""
    lib = _l.load()
    _need_cuda(x2d, pack[0], kv_image)
    if out is None:
        out = torch.empty((x2d.shape[0], Cc), dtype=BF16, device=x2d.device)
    p = XattnParams()
    p.x = x2d.data_ptr(); p.w_packed = pack[0].data_ptr(); p.bq = pack[1].data_ptr(); p.bo = bo.data_ptr()
    p.gamma = gamma.data_ptr(); p.beta = beta.data_ptr(); p.kv_image = kv_image.data_ptr(); p.out = out.data_ptr()
    p.M = x2d.shape[0]; p.C = Cc; p.heads = heads; p.keys = S; p.rows_per_image = rows_per_image; p.images = kv_image.shape[0]
    p.ldx = x2d.stride(0); p.ldo = out.stride(0); p.eps = eps
    _l.check(lib.ccnet_xattn_fused_bf16(C.byref(p), _stream()), "xattn_fused")
    return out


def groupnorm(x: Act, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
              x2: Optional[Act] = None) -> Act:
    lib = _l.load()
    _need_cuda(x.t)
    Cc = x.C + (x2.C if x2 is not None else 0)
    out = new_act(x.N, x.H, x.W, Cc, x.t.device)
    hw = x.H * x.W
    wsb = lib.ccnet_groupnorm_workspace_bytes(x.N, hw, Cc, groups)
    ws = torch.empty((max(wsb, 4) + 3) // 4, dtype=torch.float32, device=x.t.device)
    _l.check(lib.ccnet_groupnorm_bf16(x.t.data_ptr(), x.ld, x.C, _ptr(x2.t) if x2 is not None else None,
                                      x2.ld if x2 is not None else 0, x2.C if x2 is not None else 0, x.N, hw, groups,
                                      eps, gamma.data_ptr(), beta.data_ptr(), int(silu), out.t.data_ptr(), out.ld,
                                      ws.data_ptr(), ws.numel() * 4, _stream()), "groupnorm")
    return out


def layernorm(x2d: torch.Tensor, Cc: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    lib = _l.load()
    _need_cuda(x2d)
    out = torch.empty((x2d.shape[0], Cc), dtype=BF16, device=x2d.device)
    _l.check(lib.ccnet_layernorm_bf16(x2d.data_ptr(), x2d.stride(0), x2d.shape[0], Cc, eps, gamma.data_ptr(),
                                      beta.data_ptr(), out.data_ptr(), out.stride(0), _stream()), "layernorm")
    return out


def geglu(x2d: torch.Tensor, Cc: int) -> torch.Tensor:
    lib = _l.load()
    out = torch.empty((x2d.shape[0], Cc), dtype=BF16, device=x2d.device)
    _l.check(lib.ccnet_geglu_bf16(x2d.data_ptr(), x2d.stride(0), x2d.shape[0], Cc, out.data_ptr(), out.stride(0),
                                  _stream()), "geglu")
    return out


def attention_d64(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Sq: int, Skv: int,
                  scale: float, prescaled: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B*Sq, >=H*64] (views allowed), k/v [B*Skv, >=H*64] -> o [B*Sq, H*64] (`out`: write there).  prescaled: q already
    carries scale * log2(e) (ATTN_PRESCALE, folded into the query projection by the caller); `scale` is then ignored."""
    lib = _l.load()
    _need_cuda(q, k, v)
    o = out if out is not None else torch.empty((B * Sq, H * 64), dtype=BF16, device=q.device)
    if prescaled:
        _l.check(lib.ccnet_attention_d64_prescaled_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                                        v.stride(0), o.data_ptr(), o.stride(0), B, H, Sq, Skv, _stream()),
                 "attention_d64_prescaled")
        return o
    _l.check(lib.ccnet_attention_d64_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                          v.stride(0), o.data_ptr(), o.stride(0), B, H, Sq, Skv, scale, _stream()),
             "attention_d64")
    return o


def attention_d64_causal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, S: int, scale: float) -> torch.Tensor:
    """Causal self-attention (key j <= query i): q/k/v [B*S, >=H*64] (views allowed) -> o [B*S, H*64]."""
    lib = _l.load()
    _need_cuda(q, k, v)
    o = torch.empty((B * S, H * 64), dtype=BF16, device=q.device)
    _l.check(lib.ccnet_attention_d64_causal_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                                 v.stride(0), o.data_ptr(), o.stride(0), B, H, S, scale, _stream()),
             "attention_d64_causal")
    return o


def gelu(x2d: torch.Tensor, Cc: int, quick: bool = False) -> torch.Tensor:
    """erf GELU (or x * sigmoid(1.702 x) with quick=True) over the first Cc columns of bf16 rows."""
    lib = _l.load()
    _need_cuda(x2d)
    out = torch.empty((x2d.shape[0], Cc), dtype=BF16, device=x2d.device)
    _l.check(lib.ccnet_gelu_bf16(x2d.data_ptr(), x2d.stride(0), x2d.shape[0], Cc, int(quick), out.data_ptr(),
                                 out.stride(0), _stream()), "gelu")
    return out


def embed_tokens(ids: torch.Tensor, seq: int, tok: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """ids int64 [T] (T = B*seq), tok fp32 [vocab, C], pos fp32 [>=seq, C] -> bf16 [T, C] = tok[ids] + pos[t % seq]."""
    lib = _l.load()
    _need_cuda(ids, tok, pos)
    T, Cc = ids.numel(), tok.shape[1]
    out = torch.empty((T, Cc), dtype=BF16, device=ids.device)
    _l.check(lib.ccnet_embed_tokens_bf16(ids.data_ptr(), T, seq, tok.shape[0], tok.data_ptr(), pos.data_ptr(), Cc,
                                         out.data_ptr(), out.stride(0), _stream()), "embed_tokens")
    return out


def softmax_rows(x2d: torch.Tensor, cols: int, scale: float, cols_pad: Optional[int] = None) -> torch.Tensor:
    """Row softmax of scale * x over the first `cols` columns, in place capable; [cols, cols_pad) is zero-filled."""
    lib = _l.load()
    out = torch.empty_like(x2d)
    _l.check(lib.ccnet_softmax_rows_bf16(x2d.data_ptr(), x2d.stride(0), x2d.shape[0], cols, cols_pad or cols, scale,
                                         out.data_ptr(), out.stride(0), _stream()), "softmax_rows")
    return out


def attention_generic(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Sq: int, Skv: int, d: int,
                      scale: float) -> torch.Tensor:
    """Attention for head dims the flash kernel does not cover (d % 64 == 0, e.g. one head of 512 / 768 channels):
    scores = scale * Q.K^T (batched MFMA GEMM) -> row softmax -> P.V (batched MFMA GEMM on V^T).  The scores are
    materialised in bf16, as the reference materialises them in fp16 (CrossAttention._attention)."""
    lib = _l.load()
    _need_cuda(q, k, v)
    if d % 64:
        raise _l.CcnetError(f"attention head dim {d} is not a multiple of 64")
    dev = q.device
    Sp = (Skv + 63) // 64 * 64  # K of the P.V product
    o = torch.empty((B * Sq, H * d), dtype=BF16, device=dev)
    kp = torch.zeros((B, Sp, d), dtype=BF16, device=dev)  # zero rows beyond Skv: their scores are masked by the softmax
    vt = torch.empty((B, d, Sp), dtype=BF16, device=dev)
    scores = torch.empty((B, Sq, Sp), dtype=BF16, device=dev)
    for h in range(H):
        qh, kh, vh = q[:, h * d:(h + 1) * d], k[:, h * d:(h + 1) * d], v[:, h * d:(h + 1) * d]
        kp[:, :Skv].copy_(kh.unflatten(0, (B, Skv)))
        _l.check(lib.ccnet_transpose_pad_bf16(vh.data_ptr(), vh.stride(0), Skv, d, Skv * vh.stride(0), vt.data_ptr(), Sp,
                                              Sp, d * Sp, B, _stream()), "transpose_pad")
        p = _l.ConvParams()  # scores[b] = scale * Q[b] . Kp[b]^T
        p.x1 = qh.data_ptr(); p.w = kp.data_ptr(); p.out = scores.data_ptr()
        p.N, p.Hin, p.Win, p.Hout, p.Wout = 1, Sq, 1, Sq, 1
        p.C1 = d; p.C2 = 0; p.Cout = Sp; p.R = 1; p.stride = 1; p.pad = 0; p.ups = 1
        p.ldx1 = qh.stride(0); p.ldw = d; p.ldo = Sp; p.scale = scale; p.batch = B
        p.sx1 = Sq * qh.stride(0); p.sw = Sp * d; p.so = Sq * Sp
        _launch_conv(p, dev, "attention scores")
        s2 = scores.view(B * Sq, Sp)
        _l.check(lib.ccnet_softmax_rows_bf16(s2.data_ptr(), Sp, B * Sq, Skv, Sp, 1.0, s2.data_ptr(), Sp, _stream()),
                 "softmax_rows")
        oh = o[:, h * d:(h + 1) * d]
        p = _l.ConvParams()  # O[b] = P[b] . V[b]  (weights = V^T rows)
        p.x1 = scores.data_ptr(); p.w = vt.data_ptr(); p.out = oh.data_ptr()
        p.N, p.Hin, p.Win, p.Hout, p.Wout = 1, Sq, 1, Sq, 1
        p.C1 = Sp; p.C2 = 0; p.Cout = d; p.R = 1; p.stride = 1; p.pad = 0; p.ups = 1
        p.ldx1 = Sp; p.ldw = Sp; p.ldo = o.stride(0); p.scale = 1.0; p.batch = B
        p.sx1 = Sq * Sp; p.sw = d * Sp; p.so = Sq * o.stride(0)
        _launch_conv(p, dev, "attention P.V")
    return o


def timestep_embedding(t: torch.Tensor, B: int, dim: int, freqs: torch.Tensor) -> torch.Tensor:
    lib = _l.load()
    _need_cuda(t, freqs)
    out = torch.empty((B, dim), dtype=torch.float32, device=t.device)
    _l.check(lib.ccnet_timestep_embedding(t.data_ptr(), t.numel(), B, dim, freqs.data_ptr(), out.data_ptr(),
                                          _stream()), "timestep_embedding")
    return out


def linear_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], silu_in: bool = False,
               silu_out: int = 0, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _l.load()
    _need_cuda(x, w)
    B, K = x.shape
    Nout = w.shape[0]
    out = torch.empty((B, Nout), dtype=torch.float32, device=x.device)
    _l.check(lib.ccnet_linear_f32(x.data_ptr(), x.stride(0), B, K, w.data_ptr(), _ptr(bias), Nout, int(silu_in),
                                  int(silu_out), _ptr(add), add.stride(0) if add is not None else 0, out.data_ptr(),
                                  out.stride(0), _stream()), "linear_f32")
    return out


def conv_in_small(x: torch.Tensor, N: int, w: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, ldo: int,
                  stride: int = 1, act: int = 0, res: Optional[Act] = None) -> Act:
    """x fp32 NCHW [Nsrc, Cin<=8, H, W]; image n of the N outputs reads x[n % Nsrc]."""
    lib = _l.load()
    _need_cuda(x, w)
    Nsrc, Cin, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out = new_act(N, Ho, Wo, ldo, x.device)
    out.C = ldo
    _l.check(lib.ccnet_conv_in_small(x.data_ptr(), Nsrc, N, Cin, H, W, w.data_ptr(), _ptr(bias), Cout, stride, act,
                                     _ptr(res.t) if res is not None else None, res.N if res is not None else 0,
                                     res.ld if res is not None else 0, out.t.data_ptr(), out.ld, Ho, Wo, _stream()),
             "conv_in_small")
    return out


def conv_out_small(x: Act, w: torch.Tensor, bias: Optional[torch.Tensor], Cout: int) -> torch.Tensor:
    lib = _l.load()
    out = torch.empty((x.N, Cout, x.H, x.W), dtype=torch.float32, device=x.t.device)
    _l.check(lib.ccnet_conv_out_small(x.t.data_ptr(), x.ld, x.N, x.H, x.W, x.C, w.data_ptr(), _ptr(bias), Cout,
                                      out.data_ptr(), _stream()), "conv_out_small")
    return out


def cfg_scheduler_step(model_out: torch.Tensor, cfg_mode: int, guidance: float, coef: torch.Tensor,
                       sample: torch.Tensor, noise: Optional[torch.Tensor] = None,
                       x0_out: Optional[torch.Tensor] = None) -> None:
    """In-place update of `sample` (fp32, [B, ...]); model_out fp32 [2B or B, ...]."""
    lib = _l.load()
    _need_cuda(model_out, coef, sample)
    B = sample.shape[0]
    per = sample.numel() // B
    _l.check(lib.ccnet_cfg_scheduler_step(model_out.data_ptr(), cfg_mode, float(guidance), coef.data_ptr(),
                                          _ptr(noise), sample.data_ptr(), _ptr(x0_out), B, per, _stream()),
             "cfg_scheduler_step")


def nchw_to_act(x: torch.Tensor, ldo: Optional[int] = None) -> Act:
    lib = _l.load()
    _need_cuda(x)
    N, Cc, H, W = x.shape
    ldo = ldo or Cc
    x = x.contiguous().float()
    out = new_act(N, H, W, ldo, x.device)
    _l.check(lib.ccnet_nchw_f32_to_nhwc_bf16(x.data_ptr(), N, Cc, H * W, out.t.data_ptr(), out.ld, _stream()),
             "nchw_to_nhwc")
    return out


def act_to_nchw(a: Act, Cc: Optional[int] = None) -> torch.Tensor:
    lib = _l.load()
    Cc = Cc or a.C
    out = torch.empty((a.N, Cc, a.H, a.W), dtype=torch.float32, device=a.t.device)
    _l.check(lib.ccnet_nhwc_bf16_to_nchw_f32(a.t.data_ptr(), a.ld, a.N, Cc, a.H * a.W, out.data_ptr(), _stream()),
             "nhwc_to_nchw")
    return out


def to_uint8(x: torch.Tensor, f32_scale: bool = False) -> torch.Tensor:
    """interval_mapping -> rint(* 255) -> uint8 of channel 0, bit-exact with the reference's numpy recipe: float64 for the
    synthetic slice (util.py:1402-1414), `f32_scale=True` for the float32 (x + 1) / 2 of the comparison panels (:1432-1435)."""
    lib = _l.load()
    _need_cuda(x)
    N, Cc, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty((N, H, W), dtype=torch.uint8, device=x.device)
    _l.check(lib.ccnet_to_uint8(x.data_ptr(), N, Cc, H * W, int(f32_scale), out.data_ptr(), _stream()), "to_uint8")
    return out


def preprocess_u8(img: torch.Tensor, H: int, W: int, transpose: bool = False) -> torch.Tensor:
    """uint8 [N, H0, W0] -> bicubic resize to (H, W) -> [0,255] -> [-1,1] clipped; fp32 [N, 1, H, W] (util.py:123-128)."""
    lib = _l.load()
    _need_cuda(img)
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise _l.CcnetError("preprocess_u8 expects a uint8 [N, H0, W0] tensor")
    img = img.contiguous()
    N, H0, W0 = img.shape
    out = torch.empty((N, 1, H, W), dtype=torch.float32, device=img.device)
    _l.check(lib.ccnet_preprocess_u8_bicubic(img.data_ptr(), N, H0, W0, int(transpose), out.data_ptr(), H, W, _stream()),
             "preprocess_u8_bicubic")
    return out


def pointwise_nchw(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode: int) -> torch.Tensor:
    lib = _l.load()
    _need_cuda(x, w)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
    _l.check(lib.ccnet_pointwise_nchw_f32(x.data_ptr(), N, Cin, H * W, w.data_ptr(), _ptr(bias), Cout, mode,
                                          out.data_ptr(), _stream()), "pointwise_nchw")
    return out


def axpy_f32(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    lib = _l.load()
    _need_cuda(a, b, c)
    a, b, c = a.contiguous().float(), b.contiguous().float(), c.contiguous().float()
    out = torch.empty_like(a)
    _l.check(lib.ccnet_axpy_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.numel(), out.data_ptr(), _stream()),
             "axpy_f32")
    return out


_prof_on = False


def prof_enable(on: bool) -> None:
    """Bracket every conv launch with HIP events (bench.py roofline pass); incompatible with graph capture."""
    global _prof_on
    _prof_on = bool(on)
    _l.load().ccnet_prof_enable(int(on))


def prof_is_enabled() -> bool:
    return _prof_on


def prof_collect():
    ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
    _l.check(_l.load().ccnet_prof_collect(C.byref(ms), C.byref(fl), C.byref(n)), "prof_collect")
    return ms.value, fl.value, n.value


def prof_shapes(max_rows: int = 512):
    """Per-shape (M, Cout, K, R, flags, launches, ms) rows of the recorded conv launches (before prof_collect)."""
    buf = (C.c_double * (7 * max_rows))()
    n = _l.load().ccnet_prof_shapes(buf, max_rows)
    if n < 0:
        _l.check(n, "prof_shapes")
    return [tuple(buf[7 * i + j] for j in range(7)) for i in range(n)]
