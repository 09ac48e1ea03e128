"""Training step of the ControlNet with a frozen UNet on the MI355X kernels.

Reference: `src/python/training/training_functions.py:888-1043` (`train_epoch_controlnet`): encode the target with the
frozen stage 1, `scheduler.add_noise` (:931), ControlNet forward (:933), frozen UNet forward with the residuals (:943),
target = `get_velocity` / noise (:958-962), (SNR-weighted) MSE (:991-1014), `loss.backward()`, gradient clipping and the
optimizer step (:1018-1032).  Upstream's backward is torch autograd over monai-generative's modules; here it is a tape of
hand-written passes over the same HIP kernels as the sampler:

* the UNet's encoder and middle block see neither the ControlNet nor any trainable parameter, so they run once through the
  sampler's own forward (`DiffusionModelUNet.encode_phase`) and take no part in the backward;
* the ControlNet and the UNet decoder run op by op under `Tape`, which records for every op the closure that turns the
  output gradient into input and parameter gradients.  Input gradients of conv / linear layers are the forward
  implicit-GEMM kernel over dY with transposed, tap-flipped weights; weight gradients are the same kernel on
  [channel][pixel] operands (dY^T x im2col(X)^T); GroupNorm / LayerNorm / softmax / GEGLU / SiLU have their own backward
  kernels (csrc/train.hip).  Activations and activation gradients are bf16 (as under upstream's autocast), statistics,
  sums and parameter gradients fp32.

There is no CPU or autograd fallback: without the HIP library every call raises `CcnetError`.
"""
from __future__ import annotations

import logging

import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .lib import CcnetError
from .ops import Act

BF16 = torch.bfloat16


_OPERAND_LIMIT = 0x7fffffff  # bytes one operand of ccnet_conv_gemm_bf16 may span (gemm_conv.hip: 32-bit buffer offsets)


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Var:
    """An activation on the tape: value `a` (Act: bf16 [rows, C] rows + NHWC geometry), gradient `g` (bf16 tensor
    [rows, C], allocated on first use) and whether a gradient is wanted at all."""

    __slots__ = ("a", "g", "rg")

    def __init__(self, a: Act, rg: bool):
        self.a, self.g, self.rg = a, None, rg

    def grad_buf(self) -> torch.Tensor:
        if self.g is None:
            self.g = torch.zeros((self.a.rows, self.a.C), dtype=BF16, device=self.a.t.device)
        return self.g

    def accumulate(self, d: torch.Tensor, col0: int = 0) -> None:
        """g[:, col0:col0 + d.shape[1]] += d (d bf16 [rows, n], any row stride)."""
        if not self.rg:
            return
        n = d.shape[1]
        if self.g is None and col0 == 0 and n == self.a.C and d.is_contiguous():
            self.g = d
            return
        g = self.grad_buf()[:, col0:col0 + n]
        _l.check(_l.load().ccnet_axpby_bf16(g.data_ptr(), g.stride(0), d.data_ptr(), d.stride(0), d.shape[0], n, 1.0, 1.0,
                                            _stream()), "axpby")


def _v2d(t: torch.Tensor, rg: bool) -> Var:
    return Var(Act(t, 1, t.shape[0], 1, t.shape[1]), rg)


# ------------------------------------------------------------------------------------------------ low-level helpers
def _transpose_pad(x: torch.Tensor, rows: int, cols: int, rows_pad: int) -> torch.Tensor:
    """bf16 [rows, >= cols] (row stride free) -> [cols, rows_pad], zero beyond `rows`."""
    out = torch.empty((cols, rows_pad), dtype=BF16, device=x.device)
    _l.check(_l.load().ccnet_transpose_pad_bf16(x.data_ptr(), x.stride(0), rows, cols, 0, out.data_ptr(), rows_pad, rows_pad, 0,
                                                1, _stream()), "transpose_pad")
    return out


def _gemm_nt(a: torch.Tensor, b: torch.Tensor, scale: float = 1.0, nb: int = 1) -> torch.Tensor:
    """fp32 out[i][j] = scale * sum_k a[i][k] b[j][k]; a [Ra, K], b [Rb, K] bf16 (row strides free), Rb % 8 == 0.  nb > 1: the
    contraction is cut into nb equal slices of K / nb (a multiple of 64) that run as one batched launch and are summed in fp32
    -- a weight gradient contracts over up to millions of pixels into a few tiles, far too few for the kernel's own split-K."""
    Ra, K = a.shape
    Rb = b.shape[0]
    if nb <= 1:
        return ops.conv(Act(a, 1, Ra, 1, K), b, None, Rb, R=1, pad=0, scale=scale).t.float()
    Kc = K // nb
    out = torch.empty((nb, Ra, Rb), dtype=BF16, device=a.device)
    p = _l.ConvParams()
    p.x1 = a.data_ptr(); p.w = b.data_ptr(); p.out = out.data_ptr()
    p.N, p.Hin, p.Win, p.Hout, p.Wout = 1, Ra, 1, Ra, 1
    p.C1 = Kc; p.C2 = 0; p.Cout = Rb; p.R = 1; p.stride = 1; p.pad = 0; p.ups = 1
    p.ldx1 = a.stride(0); p.ldw = b.stride(0); p.ldo = Rb; p.scale = scale; p.batch = nb
    p.sx1 = Kc; p.sw = Kc; p.so = Ra * Rb
    ops._launch_conv(p, a.device, "weight-gradient GEMM")
    return out.sum(0, dtype=torch.float32)  # (bf16 partials, fp32 accumulation: no fp32 copy of the partials)


def _plan_wgrad(M: int, Ra: int, Rb: int) -> Tuple[int, int]:
    """(padded contraction length, slices) for a weight-gradient GEMM [Ra, M] x [Rb, M]^T: enough (tile, slice) pairs to fill
    the chip, every slice at least 4096 deep."""
    Mp = _pad64(M)
    tiles = -(-Ra // 256) * -(-Rb // 256)
    nb = max(1, min(1024 // tiles, Mp // 4096))
    if nb <= 1:
        return Mp, 1
    Kc = _pad64(-(-Mp // nb))
    return Kc * nb, nb


def _colsum(y: torch.Tensor, nseg: int, x: Optional[torch.Tensor] = None, rowstat: Optional[torch.Tensor] = None):
    """fp32 [nseg, C] column sums of the bf16 rows y per segment of rows; with x / rowstat also sum of y * xhat."""
    rows, Cc = y.shape
    rps = rows // nseg
    nsplit = max(1, min(16, rps // 256))
    o = torch.empty((nsplit, nseg, Cc), dtype=torch.float32, device=y.device)
    o2 = torch.empty_like(o) if x is not None else None
    _l.check(_l.load().ccnet_colsum_bf16(y.data_ptr(), y.stride(0), nseg, rps, Cc, x.data_ptr() if x is not None else None,
                                         x.stride(0) if x is not None else 0,
                                         rowstat.data_ptr() if rowstat is not None else None, nsplit, o.data_ptr(),
                                         o2.data_ptr() if o2 is not None else None, _stream()), "colsum")
    return (o.sum(0), o2.sum(0)) if x is not None else o.sum(0)


def _dgrad_weight(w: torch.Tensor, R: int, cache: dict) -> torch.Tensor:
    """Packed forward weight bf16 [Cout, R*R*Cin] -> the weight of the input-gradient convolution [Cin, R*R*Cout]
    (taps flipped, channels transposed)."""
    key = (w.data_ptr(), R, w._version)
    hit = cache.get(key)
    if hit is None:
        co = w.shape[0]
        ci = w.shape[1] // (R * R)
        hit = w.view(co, R, R, ci).flip(1, 2).permute(3, 1, 2, 0).reshape(ci, R * R * co).contiguous()
        cache[key] = hit
    return hit


class Tape:
    """Forward ops that record their backward.  `grads` collects parameter gradients (fp32, in the parameter's own layout)
    keyed by the nn.Parameter."""

    def __init__(self):
        self.steps: List[Callable[[], None]] = []
        self.grads: Dict[nn.Parameter, torch.Tensor] = {}
        self.wcache: dict = {}

    # ---- parameter gradients
    def add_grad(self, p: Optional[nn.Parameter], g: torch.Tensor) -> None:
        if p is None or not p.requires_grad:
            return
        g = g.to(torch.float32).reshape(p.shape)
        if p in self.grads:
            self.grads[p] += g
        else:
            self.grads[p] = g.clone()

    def backward(self) -> None:
        for fn in reversed(self.steps):
            fn()
        self.steps.clear()

    # ---- conv / linear
    def conv(self, x: Var, w: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, R: int = 3, stride: int = 1, ups: int = 1,
             x2: Optional[Var] = None, rowvec: Optional[torch.Tensor] = None, drowvec: Optional[torch.Tensor] = None,
             res1: Optional[Var] = None, scale: float = 1.0,
             params: Optional[Tuple[Sequence[Tuple[nn.Parameter, int, int]], Optional[Sequence[Tuple[nn.Parameter, int, int]]]]] = None,
             conv_shape: Optional[Tuple[int, int]] = None) -> Var:
        """y = scale * (conv(x | x2) + bias + rowvec[image]) + res1.  `params` = (weights, biases): lists of
        (parameter, first packed cout row, rows) -- several when the packed matrix concatenates layers (to_q | to_k | to_v).
        `conv_shape` = (cout, cin) of the un-padded nn.Conv2d weight when the packed one is channel-padded.
        `drowvec` (fp32 [N, Cout] view) receives the gradient of the rowvec operand."""
        pad = R // 2
        y = ops.conv(x.a, w, bias, Cout, R=R, stride=stride, pad=pad, ups=ups, x2=x2.a if x2 is not None else None,
                     rowvec=rowvec, res1=res1.a if res1 is not None else None, scale=scale)
        rg_in = x.rg or (x2 is not None and x2.rg)
        out = Var(y, rg_in or params is not None or drowvec is not None or (res1 is not None and res1.rg))
        if not out.rg:
            return out

        def bw():
            dy = out.g
            if dy is None:
                return
            dya = Act(dy, y.N, y.H, y.W, Cout)
            if res1 is not None:
                res1.accumulate(dy)
            want_b = params is not None and params[1] is not None
            if want_b or drowvec is not None:
                per_image = _colsum(dy, y.N) * scale  # [N, Cout]
                if drowvec is not None:
                    drowvec.copy_(per_image)
                if want_b:
                    tot = per_image.sum(0)
                    for p, r0, nr in params[1]:
                        self.add_grad(p, tot[r0:r0 + nr])
            if params is not None:
                Ct = x.a.C + (x2.a.C if x2 is not None else 0)
                # dW = dY^T . im2col(X): both operands in [channel][pixel] layout, images taken in chunks so that the
                # transposed im2col stays under 1 GiB (the conditioning embedding's first layers live at 512^2)
                hw_o, hw_i = y.H * y.W, x.a.H * x.a.W
                per_image = max(R * R * Ct, Cout) * hw_o * 2
                nchunk = max(1, min(x.a.N, (1 << 30) // per_image))
                dw = None
                for n0 in range(0, x.a.N, nchunk):
                    n1 = min(x.a.N, n0 + nchunk)
                    M = (n1 - n0) * hw_o
                    Mp, nb = _plan_wgrad(M, Cout, R * R * Ct)
                    dyT = _transpose_pad(dy[n0 * hw_o:n1 * hw_o], M, Cout, Mp)
                    xc = torch.empty((R * R * Ct, Mp), dtype=BF16, device=dy.device)
                    x1t = x.a.t[n0 * hw_i:n1 * hw_i]
                    x2t = x2.a.t[n0 * hw_i:n1 * hw_i] if x2 is not None else None
                    _l.check(_l.load().ccnet_im2col_t_bf16(
                        x1t.data_ptr(), x.a.ld, x.a.C, x2t.data_ptr() if x2t is not None else None,
                        x2.a.ld if x2 is not None else 0, x2.a.C if x2 is not None else 0, n1 - n0, x.a.H, x.a.W, y.H, y.W, R,
                        stride, pad, ups, xc.data_ptr(), Mp, _stream()), "im2col_t")
                    part = _gemm_nt(dyT, xc, scale=scale, nb=nb)  # [Cout, R*R*Ct]
                    dw = part if dw is None else dw + part
                for p, r0, nr in params[0]:
                    g = dw[r0:r0 + nr]
                    if p.dim() == 4:  # nn.Conv2d weight [co, ci, R, R] <- packed [co_pad][R][R][ci_pad]
                        co, ci = conv_shape if conv_shape is not None else (nr, Ct)
                        g = g.view(nr, R, R, Ct)[:co, :, :, :ci].permute(0, 3, 1, 2)
                    self.add_grad(p, g)
            if rg_in:
                wt = _dgrad_weight(w, R, self.wcache)
                Ct = wt.shape[0]
                if stride == 2:
                    # dY zero-stuffed to the input resolution, images in chunks that stay under the conv kernel's 2 GiB
                    # operand limit (the conditioning embedding's 512^2 -> 256^2 layer at batch 32: 2.1 GB in one piece)
                    hw_o, hw_i = y.H * y.W, x.a.H * x.a.W
                    nchunk = max(1, min(x.a.N, (_OPERAND_LIMIT - 2 * Cout) // (hw_i * Cout * 2)))
                    dxa = ops.new_act(x.a.N, x.a.H, x.a.W, Ct, dy.device)
                    for n0 in range(0, x.a.N, nchunk):
                        n1 = min(x.a.N, n0 + nchunk)
                        z = ops.new_act(n1 - n0, x.a.H, x.a.W, Cout, dy.device)
                        dyc = dy[n0 * hw_o:n1 * hw_o]
                        _l.check(_l.load().ccnet_resample_bwd_bf16(dyc.data_ptr(), dy.stride(0), z.t.data_ptr(), z.ld, n1 - n0,
                                                                   x.a.H, x.a.W, y.H, y.W, Cout, 1, _stream()), "zero_stuff")
                        ops.conv(z, wt, None, Ct, R=R, pad=pad, scale=scale,
                                 out=Act(dxa.t[n0 * hw_i:n1 * hw_i], n1 - n0, x.a.H, x.a.W, Ct))
                        del z
                elif ups == 2:
                    du = ops.conv(dya, wt, None, Ct, R=R, pad=pad, scale=scale)  # at the upsampled resolution
                    dxa = ops.new_act(x.a.N, x.a.H, x.a.W, Ct, dy.device)
                    _l.check(_l.load().ccnet_resample_bwd_bf16(du.t.data_ptr(), du.ld, dxa.t.data_ptr(), dxa.ld, x.a.N, x.a.H,
                                                               x.a.W, 0, 0, Ct, 0, _stream()), "sum_pool")
                else:
                    dxa = ops.conv(dya, wt, None, Ct, R=R, pad=pad, scale=scale)
                if x2 is None:
                    x.accumulate(dxa.t)
                else:
                    x.accumulate(dxa.t[:, :x.a.C])
                    x2.accumulate(dxa.t[:, x.a.C:])

        self.steps.append(bw)
        return out

    def linear(self, x: Var, w: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, res1: Optional[Var] = None,
               params=None) -> Var:
        """Row-matrix layer: the 1x1 case of conv() (the NHWC geometry of x rides along unchanged)."""
        return self.conv(x, w, bias, Cout, R=1, res1=res1, params=params)

    # ---- normalisation
    def groupnorm(self, x: Var, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
                  x2: Optional[Var] = None, params: Optional[Tuple[nn.Parameter, nn.Parameter]] = None) -> Var:
        y = ops.groupnorm(x.a, gamma, beta, groups, eps, silu, x2=x2.a if x2 is not None else None)
        out = Var(y, x.rg or (x2 is not None and x2.rg) or params is not None)
        if not out.rg:
            return out

        def bw():
            dy = out.g
            if dy is None:
                return
            C1, C2 = x.a.C, x2.a.C if x2 is not None else 0
            dx = torch.empty((dy.shape[0], C1 + C2), dtype=BF16, device=dy.device)
            dp = torch.empty((x.a.N, C1 + C2, 2), dtype=torch.float32, device=dy.device) if params is not None else None
            dx2 = dx[:, C1:] if C2 else None
            _l.check(_l.load().ccnet_groupnorm_bwd_bf16(
                x.a.t.data_ptr(), x.a.ld, C1, x2.a.t.data_ptr() if x2 is not None else None, x2.a.ld if x2 is not None else 0, C2,
                dy.data_ptr(), dy.stride(0), x.a.N, x.a.H * x.a.W, groups, eps, gamma.data_ptr(), beta.data_ptr(), int(silu),
                dx.data_ptr(), dx.stride(0), dx2.data_ptr() if dx2 is not None else None, dx.stride(0),
                dp.data_ptr() if dp is not None else None, _stream()), "groupnorm_bwd")
            if params is not None:
                s = dp.sum(0)
                self.add_grad(params[0], s[:, 0])
                self.add_grad(params[1], s[:, 1])
            if x2 is None:
                x.accumulate(dx)
            else:
                x.accumulate(dx[:, :C1])
                x2.accumulate(dx[:, C1:])

        self.steps.append(bw)
        return out

    def layernorm(self, x: Var, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                  params: Optional[Tuple[nn.Parameter, nn.Parameter]] = None) -> Var:
        Cc = x.a.C
        y = ops.layernorm(x.a.t, Cc, gamma, beta, eps)
        out = Var(Act(y, x.a.N, x.a.H, x.a.W, Cc), x.rg or params is not None)
        if not out.rg:
            return out

        def bw():
            dy = out.g
            if dy is None:
                return
            rows = dy.shape[0]
            dx = torch.empty((rows, Cc), dtype=BF16, device=dy.device)
            rs = torch.empty((rows, 2), dtype=torch.float32, device=dy.device) if params is not None else None
            _l.check(_l.load().ccnet_layernorm_bwd_bf16(x.a.t.data_ptr(), x.a.ld, dy.data_ptr(), dy.stride(0), rows, Cc, eps,
                                                        gamma.data_ptr(), dx.data_ptr(), dx.stride(0),
                                                        rs.data_ptr() if rs is not None else None, _stream()), "layernorm_bwd")
            if params is not None:
                db, dg = _colsum(dy, 1, x=x.a.t, rowstat=rs)
                self.add_grad(params[0], dg[0])
                self.add_grad(params[1], db[0])
            x.accumulate(dx)

        self.steps.append(bw)
        return out

    # ---- activations
    def silu(self, x: Var) -> Var:
        a = x.a
        y = torch.empty((a.rows, a.C), dtype=BF16, device=a.t.device)
        _l.check(_l.load().ccnet_silu_bf16(a.t.data_ptr(), a.ld, None, 0, a.rows, a.C, 0, y.data_ptr(), y.stride(0), _stream()),
                 "silu")
        out = Var(Act(y, a.N, a.H, a.W, a.C), x.rg)
        if not out.rg:
            return out

        def bw():
            if out.g is None:
                return
            dx = torch.empty_like(y)
            _l.check(_l.load().ccnet_silu_bf16(a.t.data_ptr(), a.ld, out.g.data_ptr(), out.g.stride(0), a.rows, a.C, 1,
                                               dx.data_ptr(), dx.stride(0), _stream()), "silu_bwd")
            x.accumulate(dx)

        self.steps.append(bw)
        return out

    def geglu(self, h: Var) -> Var:
        a = h.a
        Cc = a.C // 2
        y = ops.geglu(a.t, Cc)
        out = Var(Act(y, a.N, a.H, a.W, Cc), h.rg)
        if not out.rg:
            return out

        def bw():
            if out.g is None:
                return
            dh = torch.empty((a.rows, 2 * Cc), dtype=BF16, device=y.device)
            _l.check(_l.load().ccnet_geglu_bwd_bf16(a.t.data_ptr(), a.ld, out.g.data_ptr(), out.g.stride(0), a.rows, Cc,
                                                    dh.data_ptr(), dh.stride(0), _stream()), "geglu_bwd")
            h.accumulate(dh)

        self.steps.append(bw)
        return out

    # ---- attention (head dim 64)
    def attention(self, q: Var, q_off: int, kv: Var, k_off: int, v_off: int, B: int, H: int, T: int, S: int, scale: float) -> Var:
        """softmax(scale Q K^T) V per (image, head); q = q.a.t[:, q_off : q_off + 64 H] ([B*T] rows), k / v columns of
        kv.a.t ([B*S] rows).  Forward: the flash kernel, keeping the row log-sum-exp.  Backward: ccnet_attention_d64_bwd_bf16
        (P recomputed tile by tile from Q, K and the log-sum-exp; one pass for dQ, one for dK / dV; written straight into the
        column slices of the operands' gradient buffers)."""
        d = 64
        inner = H * d
        lib = _l.load()
        qt, kt, vt = q.a.t[:, q_off:q_off + inner], kv.a.t[:, k_off:k_off + inner], kv.a.t[:, v_off:v_off + inner]
        dev = qt.device
        o = torch.empty((B * T, inner), dtype=BF16, device=dev)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=dev)
        _l.check(lib.ccnet_attention_d64_lse_bf16(qt.data_ptr(), qt.stride(0), kt.data_ptr(), kt.stride(0), vt.data_ptr(),
                                                  vt.stride(0), o.data_ptr(), o.stride(0), B, H, T, S, scale, lse.data_ptr(),
                                                  _stream()), "attention_d64_lse")
        out = Var(Act(o, q.a.N, q.a.H, q.a.W, inner), q.rg or kv.rg)
        if not out.rg:
            return out

        def bw():
            do = out.g
            if do is None:
                return
            delta = torch.empty((B, H, T), dtype=torch.float32, device=dev)
            dq = q.grad_buf()[:, q_off:q_off + inner] if q.rg else None
            dkv = kv.grad_buf() if kv.rg else None
            dk = dkv[:, k_off:k_off + inner] if dkv is not None else None
            dv = dkv[:, v_off:v_off + inner] if dkv is not None else None
            _l.check(lib.ccnet_attention_d64_bwd_bf16(
                qt.data_ptr(), qt.stride(0), kt.data_ptr(), kt.stride(0), vt.data_ptr(), vt.stride(0), o.data_ptr(), o.stride(0),
                do.data_ptr(), do.stride(0), lse.data_ptr(), delta.data_ptr(),
                dq.data_ptr() if dq is not None else None, dq.stride(0) if dq is not None else 0,
                dk.data_ptr() if dk is not None else None, dk.stride(0) if dk is not None else 0,
                dv.data_ptr() if dv is not None else None, dv.stride(0) if dv is not None else 0, B, H, T, S, scale, _stream()),
                "attention_d64_bwd")

        self.steps.append(bw)
        return out


# ================================================================================================ network lowering
class _Net:
    """Per-network view used by the training lowering: the sampler's packed weights + the parameters behind them."""

    def __init__(self, net, trainable: bool):
        self.net = net
        self.pk = net._pk or net._pack()
        self.trainable = trainable

    def cparams(self, conv: nn.Conv2d):
        if not self.trainable:
            return None
        return ([(conv.weight, 0, conv.weight.shape[0])], [(conv.bias, 0, conv.bias.shape[0])] if conv.bias is not None else None)

    def lparams(self, *linears: nn.Linear):
        if not self.trainable:
            return None
        ws, bs, r = [], [], 0
        for l in linears:
            n = l.weight.shape[0]
            ws.append((l.weight, r, n))
            if l.bias is not None:
                bs.append((l.bias, r, n))
            r += n
        return (ws, bs or None)

    def nparams(self, norm):
        return (norm.weight, norm.bias) if self.trainable else None


def _resnet(T: Tape, nt: _Net, r, x: Var, tall: torch.Tensor, dtall: Optional[torch.Tensor], x2: Optional[Var] = None) -> Var:
    net, pk = nt.net, nt.pk
    p = pk["res"][id(r)]
    g, eps = net.norm_num_groups, net.norm_eps
    off, cout = pk["tproj_off"][id(r)], p["cout"]
    h = T.groupnorm(x, p["n1"][0], p["n1"][1], g, eps, True, x2=x2, params=nt.nparams(r.norm1))
    h = T.conv(h, p["c1"][0], p["c1"][1], cout, R=3, rowvec=tall[:, off:off + cout],
               drowvec=dtall[:, off:off + cout] if dtall is not None else None, params=nt.cparams(r.conv1.conv))
    h = T.groupnorm(h, p["n2"][0], p["n2"][1], g, eps, True, params=nt.nparams(r.norm2))
    if p["skip"] is None:
        skip = x
    else:
        skip = T.conv(x, p["skip"][0], p["skip"][1], cout, R=1, x2=x2, params=nt.cparams(r.skip_connection.conv))
    return T.conv(h, p["c2"][0], p["c2"][1], cout, R=3, res1=skip, params=nt.cparams(r.conv2.conv))


def _attn(T: Tape, nt: _Net, a, x: Var, ctx: Var, S: int) -> Var:
    net, pk = nt.net, nt.pk
    p = pk["att"][id(a)]
    b = a.transformer_blocks[0]
    inner, heads, hc = p["inner"], p["heads"], p["head_ch"]
    if hc != 64:
        raise CcnetError(f"training lowering: attention head dim {hc} (only 64, the SD-2.1 geometry, has a backward)")
    B, Tq = x.a.N, x.a.H * x.a.W
    scale = 1.0 / math.sqrt(hc)
    tp = pk.setdefault("_train", {})
    key = id(a)
    if key not in tp:  # GEGLU linear in its plain row order (the sampler's copy is interleaved for the fused epilogue)
        tp[key] = (b.ff.linear1.weight.detach().to(BF16).contiguous(), b.ff.linear1.bias.detach().float().contiguous())
    ff1w, ff1b = tp[key]
    h = T.groupnorm(x, p["norm"][0], p["norm"][1], net.norm_num_groups, net.transformer_norm_eps, False, params=nt.nparams(a.norm))
    t = T.conv(h, p["pin"][0], p["pin"][1], inner, R=1, params=nt.cparams(a.proj_in.conv))
    n1 = T.layernorm(t, *p["ln1"], params=nt.nparams(b.norm1))
    qkv = T.linear(n1, p["qkv1"], None, 3 * inner, params=nt.lparams(b.attn1.to_q, b.attn1.to_k, b.attn1.to_v))
    o = T.attention(qkv, 0, qkv, inner, 2 * inner, B, heads, Tq, Tq, scale)
    t = T.linear(o, p["o1"][0], p["o1"][1], inner, res1=t, params=nt.lparams(b.attn1.to_out[0]))
    n2 = T.layernorm(t, *p["ln2"], params=nt.nparams(b.norm2))
    q = T.linear(n2, p["q2"], None, inner, params=nt.lparams(b.attn2.to_q))
    kv = T.linear(ctx, p["kv2"], None, 2 * inner, params=nt.lparams(b.attn2.to_k, b.attn2.to_v))
    o = T.attention(q, 0, kv, 0, inner, B, heads, Tq, S, scale)
    t = T.linear(o, p["o2"][0], p["o2"][1], inner, res1=t, params=nt.lparams(b.attn2.to_out[0]))
    n3 = T.layernorm(t, *p["ln3"], params=nt.nparams(b.norm3))
    hh = T.linear(n3, ff1w, ff1b, 8 * inner, params=nt.lparams(b.ff.linear1))
    gl = T.geglu(hh)
    t = T.linear(gl, p["ff2"][0], p["ff2"][1], inner, res1=t, params=nt.lparams(b.ff.linear2))
    return T.conv(t, p["pout"][0], p["pout"][1], x.a.C, R=1, res1=x, params=nt.cparams(a.proj_out.conv))


def _embedding(T: Tape, nt: _Net, timesteps, class_labels, B: int):
    """The fp32 time / class embedding path with its backward (tiny [B, 4 c0] tensors: the three linears run on
    ccnet_linear_f32 in both directions, the SiLUs between them are element-wise torch on [B, 4 c0]).
    Returns (tall [B, sum of resnet couts], dtall zero buffer that the resnets' conv1 fill)."""
    net, pk = nt.net, nt.pk
    dev = pk["freqs"].device
    t = timesteps.to(device=dev, dtype=torch.float32).reshape(-1)
    c0 = net.block_out_channels[0]
    t_emb = ops.timestep_embedding(t, B, c0, pk["freqs"])
    h_pre = ops.linear_f32(t_emb, pk["te0"][0], pk["te0"][1])
    h = torch.nn.functional.silu(h_pre)
    e = ops.linear_f32(h, pk["te2"][0], pk["te2"][1])
    c1_pre = c1 = cl = None
    if net.num_class_embeds is not None:
        if class_labels is None:
            raise ValueError("class_labels should be provided when num_class_embeds > 0")
        if "class_table" in pk:
            raise CcnetError("training lowering: nn.Embedding class conditioning has no backward here")
        cl = class_labels.to(device=dev, dtype=torch.float32).reshape(B, 1).contiguous()
        c1_pre = ops.linear_f32(cl, pk["ce0"][0], pk["ce0"][1])
        c1 = torch.nn.functional.silu(c1_pre)
        e = e + ops.linear_f32(c1, pk["ce2"][0], pk["ce2"][1])
    semb = torch.nn.functional.silu(e)
    tall = ops.linear_f32(semb, pk["tproj_w"], pk["tproj_b"])
    if not nt.trainable:
        return tall, None
    dtall = torch.zeros_like(tall)

    def dsilu(x):
        s = torch.sigmoid(x)
        return s * (1 + x * (1 - s))

    def lin_bw(dy, x, w):  # y = x w^T + b: (dx, dw, db)
        dx = ops.linear_f32(dy, w.t().contiguous(), None)
        # dw = dy^T x contracts over the batch: bf16 operands (as upstream's autocast), the batch zero-padded to one K step
        Bp = _pad64(dy.shape[0])
        a = torch.zeros((dy.shape[1], Bp), dtype=BF16, device=dy.device)
        b = torch.zeros((_pad64(x.shape[1]) if x.shape[1] % 8 else x.shape[1], Bp), dtype=BF16, device=dy.device)
        a[:, :dy.shape[0]] = dy.t()
        b[:x.shape[1], :x.shape[0]] = x.t()
        dw = _gemm_nt(a, b)[:, :x.shape[1]]
        return dx, dw, dy.sum(0)

    def bw():
        rs = net._resnets_in_order()
        dsemb, dw, db = lin_bw(dtall, semb, pk["tproj_w"])
        o = 0
        for r in rs:
            T.add_grad(r.time_emb_proj.weight, dw[o:o + r.cout])
            T.add_grad(r.time_emb_proj.bias, db[o:o + r.cout])
            o += r.cout
        de = dsemb * dsilu(e)
        dh, dw, db = lin_bw(de, h, pk["te2"][0])
        T.add_grad(net.time_embed[2].weight, dw)
        T.add_grad(net.time_embed[2].bias, db)
        _, dw, db = lin_bw(dh * dsilu(h_pre), t_emb, pk["te0"][0])
        T.add_grad(net.time_embed[0].weight, dw)
        T.add_grad(net.time_embed[0].bias, db)
        if c1 is not None:
            dc1, dw, db = lin_bw(de, c1, pk["ce2"][0])
            T.add_grad(net.class_embedding[2].weight, dw)
            T.add_grad(net.class_embedding[2].bias, db)
            _, dw, db = lin_bw(dc1 * dsilu(c1_pre), cl, pk["ce0"][0])
            T.add_grad(net.class_embedding[0].weight, dw)
            T.add_grad(net.class_embedding[0].bias, db)

    T.steps.append(bw)
    return tall, dtall


def _conv_in(T: Tape, nt: _Net, x: torch.Tensor, B: int, w, b, conv: nn.Conv2d, cout: int, ldo: int, res: Optional[Var]) -> Var:
    """conv_in of the trunk / first conv of the conditioning embedding: fp32 NCHW input with <= 8 channels."""
    y = ops.conv_in_small(x, B, w, b, cout, ldo, res=res.a if res is not None else None)
    out = Var(y, nt.trainable or (res is not None and res.rg))
    if not out.rg:
        return out

    def bw():
        dy = out.g
        if dy is None:
            return
        if res is not None:
            res.accumulate(dy)
        if nt.trainable:
            M = dy.shape[0]
            Mp, nb = _plan_wgrad(M, ldo, 64)
            Nsrc, Cin, H, W = x.shape
            dyT = _transpose_pad(dy, M, ldo, Mp)
            xc = torch.empty((64, Mp), dtype=BF16, device=dy.device)
            _l.check(_l.load().ccnet_im2col_t_nchw_f32(x.data_ptr(), Nsrc, B, Cin, H, W, 1, xc.data_ptr(), Mp, 64, _stream()),
                     "im2col_t_nchw")
            dw = _gemm_nt(dyT, xc, nb=nb)[:cout, :9 * Cin]  # rows k = tap * Cin + ci
            T.add_grad(conv.weight, dw.view(cout, 3, 3, Cin).permute(0, 3, 1, 2))
            T.add_grad(conv.bias, _colsum(dy, 1)[0][:cout])

    T.steps.append(bw)
    return out


def _cond_embedding(T: Tape, nt: _Net, cond: torch.Tensor) -> Var:
    net, pk = nt.net, nt.pk
    ce = net.controlnet_cond_embedding
    c = cond.contiguous().float()
    e0 = net.cond_channels[0]
    h = _conv_in(T, nt, c, c.shape[0], pk["ce_in"][0], pk["ce_in"][1], ce.conv_in.conv, e0, _pad64(e0), None)
    h = T.silu(h)
    for blk, (w, b, stride, cop) in zip(ce.blocks, pk["ce_blocks"]):
        conv = blk.conv
        h = T.conv(h, w, b, cop, R=3, stride=stride, params=nt.cparams(conv), conv_shape=(conv.out_channels, conv.in_channels))
        h = T.silu(h)
    w, b = pk["ce_out"]
    conv = ce.conv_out.conv
    return T.conv(h, w, b, net.block_out_channels[0], R=3, params=nt.cparams(conv),
                  conv_shape=(conv.out_channels, conv.in_channels))


def _trunk(T: Tape, nt: _Net, h: Var, tall, dtall, ctx: Var, S: int):
    """Down blocks + middle block of a trunk; returns (h, skips without the conv_in output)."""
    net, pk = nt.net, nt.pk
    skips: List[Var] = []
    for blk in net.down_blocks:
        has_att = hasattr(blk, "attentions")
        for j, r in enumerate(blk.resnets):
            h = _resnet(T, nt, r, h, tall, dtall)
            if has_att:
                h = _attn(T, nt, blk.attentions[j], h, ctx, S)
            skips.append(h)
        if blk.downsampler is not None:
            w, b = pk["down"][id(blk)]
            h = T.conv(h, w, b, h.a.C, R=3, stride=2, params=nt.cparams(blk.downsampler.op.conv))
            skips.append(h)
    m = net.middle_block
    h = _resnet(T, nt, m.resnet_1, h, tall, dtall)
    h = _attn(T, nt, m.attention, h, ctx, S)
    h = _resnet(T, nt, m.resnet_2, h, tall, dtall)
    return h, skips


def controlnet_unet_forward(T: Tape, controlnet, unet, noisy_e: torch.Tensor, timesteps: torch.Tensor, cond: torch.Tensor,
                            context: torch.Tensor, class_labels: Optional[torch.Tensor] = None,
                            conditioning_scale: float = 1.0) -> Tuple[torch.Tensor, Var]:
    """sample_from_controlnet + sample_from_ldm (util.py:934-962, 1043-1092) under the tape: returns the UNet's prediction
    (fp32 NCHW) and the variable whose gradient seeds the backward (see `backward_from_prediction`)."""
    if not noisy_e.is_cuda:
        raise CcnetError("ccnet_amd training runs on the GPU only (no CPU fallback)")
    cn, un = _Net(controlnet, True), _Net(unet, False)
    x = noisy_e.contiguous().float()
    B = x.shape[0]
    dev = x.device
    # frozen UNet: encoder + middle block without a tape (they depend on neither the ControlNet nor a trainable parameter)
    ust = unet.encode_phase(x, timesteps, context, class_labels)
    # ControlNet under the tape
    tall, dtall = _embedding(T, cn, timesteps, class_labels, B)
    ctx_t, S = controlnet._context(context, B)
    ctx = _v2d(ctx_t, False)
    ce = _cond_embedding(T, cn, cond)
    c0 = controlnet.block_out_channels[0]
    w, b = cn.pk["conv_in"]
    h = _conv_in(T, cn, x, B, w, b, controlnet.conv_in.conv, c0, c0, ce)
    s0 = h
    h, skips = _trunk(T, cn, h, tall, dtall, ctx, S)
    skips = [s0] + skips
    sc = float(conditioning_scale)
    if len(skips) != len(ust["skips"]):
        raise ValueError(f"UNet state holds {len(ust['skips'])} skips, the ControlNet produces {len(skips)}")
    zc = controlnet.controlnet_down_blocks
    # the zero-convolutions write unet_skip + scale * zero_conv(controlnet_skip) (their res1 operand), as in the sampler
    down = [T.conv(s, zw, zb, s.a.C, R=1, scale=sc, res1=Var(u, False), params=cn.cparams(zconv))
            for s, (zw, zb), u, zconv in zip(skips, cn.pk["zero"], ust["skips"], zc)]
    zw, zb = cn.pk["zero_mid"]
    mid = T.conv(h, zw, zb, h.a.C, R=1, scale=sc, res1=Var(ust["h"], False), params=cn.cparams(controlnet.controlnet_mid_block))
    # UNet decoder under the tape (input gradients only)
    upk = ust["pk"]
    utall = ust["tall"]
    uctx = _v2d(unet._context(context, B)[0], False)
    h = mid
    sk = list(down)
    for blk in unet.up_blocks:
        has_att = hasattr(blk, "attentions")
        for j, r in enumerate(blk.resnets):
            s = sk.pop()
            h = _resnet(T, un, r, h, utall, None, x2=s)
            if has_att:
                h = _attn(T, un, blk.attentions[j], h, uctx, S)
        if blk.upsampler is not None:
            uw, ub = upk["up"][id(blk)]
            h = T.conv(h, uw, ub, h.a.C, R=3, ups=2)
    h = T.groupnorm(h, upk["out_norm"][0], upk["out_norm"][1], unet.norm_num_groups, unet.norm_eps, True)
    pred = ops.conv_out_small(h.a, upk["out_conv"][0], upk["out_conv"][1], unet.out_channels)
    return pred, h


def backward_from_prediction(T: Tape, unet, h: Var, dpred: torch.Tensor) -> None:
    """Seed the tape with d(loss)/d(prediction) (fp32 NCHW [B, Cout <= 8, H, W]) through the UNet's output convolution
    (3x3, C0 -> Cout; its input gradient is the implicit-GEMM kernel over dpred zero-padded to 64 channels) and run it."""
    pk = unet._pk
    w = pk["out_conv"][0]  # bf16 [Cout][9][C0]
    co, _, c0 = w.shape
    B, _, H, W = dpred.shape
    dp = torch.zeros((B, 64, H, W), dtype=torch.float32, device=dpred.device)
    dp[:, :co] = dpred
    dya = ops.nchw_to_act(dp, 64)
    key = ("out_conv", w.data_ptr(), w._version)
    wt = T.wcache.get(key)
    if wt is None:
        wp = torch.zeros((64, 3, 3, c0), dtype=BF16, device=w.device)
        wp[:co] = w.view(co, 3, 3, c0)
        wt = wp.flip(1, 2).permute(3, 1, 2, 0).reshape(c0, 9 * 64).contiguous()
        T.wcache[key] = wt
    h.accumulate(ops.conv(dya, wt, None, c0, R=3).t)
    T.backward()


# ================================================================================================ the training step
def compute_snr(scheduler, timesteps: torch.Tensor) -> torch.Tensor:
    """diffusers.training_utils.compute_snr restated (training_functions.py:1003): (alpha / sigma)^2 per timestep."""
    ac = scheduler.alphas_cumprod.to(timesteps.device)[timesteps.long()].float()
    return ac / (1.0 - ac)


def controlnet_loss_and_grads(controlnet, unet, scheduler, e: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                              cond: torch.Tensor, prompt_embeds: torch.Tensor, acquisition_time: Optional[torch.Tensor] = None,
                              conditioning_scale: float = 1.0, snr_gamma: Optional[float] = None):
    """One forward + backward of train_epoch_controlnet's step body (training_functions.py:930-1018) for latents `e`:
    returns (loss, {parameter: fp32 gradient}) over the ControlNet's parameters."""
    noisy_e = scheduler.add_noise(original_samples=e, noise=noise, timesteps=timesteps)
    class_labels = acquisition_time
    T = Tape()
    pred, hvar = controlnet_unet_forward(T, controlnet, unet, noisy_e, timesteps, cond, prompt_embeds, class_labels,
                                         conditioning_scale)
    if scheduler.prediction_type == "v_prediction":
        target = scheduler.get_velocity(e, noise, timesteps)
    elif scheduler.prediction_type == "epsilon":
        target = noise
    else:
        raise ValueError(f"prediction_type {scheduler.prediction_type!r} has no training target upstream")
    diff = pred.float() - target.float()
    B = diff.shape[0]
    per = diff[0].numel()
    if snr_gamma is None:
        loss = (diff * diff).mean()
        dpred = diff * (2.0 / diff.numel())
    else:
        snr = compute_snr(scheduler, timesteps)
        wts = torch.minimum(snr, torch.full_like(snr, float(snr_gamma)))
        wts = wts / snr if scheduler.prediction_type == "epsilon" else wts / (snr + 1)
        loss = ((diff * diff).reshape(B, -1).mean(1) * wts).mean()
        dpred = diff * (2.0 / (per * B)) * wts.view(B, 1, 1, 1)
    backward_from_prediction(T, unet, hvar, dpred)
    return loss, T.grads


def train_step(controlnet, unet, scheduler, optimizer, e, noise, timesteps, cond, prompt_embeds, acquisition_time=None,
               conditioning_scale: float = 1.0, snr_gamma: Optional[float] = None, clip_grad_norm_by: Optional[float] = None,
               clip_grad_norm_or_value: str = "norm"):
    """The step body of train_epoch_controlnet (training_functions.py:920-1037) on latents: zero_grad, forward, loss,
    backward, clipping as `clip_grad` does it (util.py:1176-1206: global norm, or element-wise 'value'), then the optimizer
    step under GradScaler.step's rule (training_functions.py:1027-1029): when the loss or any gradient is inf / NaN the update is
    SKIPPED -- parameters, optimizer moments and the packed weights stay as they are -- and the step is logged.  There is no
    loss scale on this path (bf16 activations), so only the skip rule of the scaler remains.  Parameters the backward did not
    reach keep `grad = None` (the optimizer leaves them untouched, as with autograd).  The parameters live in the nn.Module as
    upstream, so any torch optimizer built over `controlnet.parameters()` applies.  Returns (loss, gradient norm before
    clipping -- None when nothing was clipped)."""
    if clip_grad_norm_or_value not in ("norm", "value"):
        raise ValueError(f"clip_grad_at {clip_grad_norm_or_value} not supported. Use 'norm' or 'value'.")  # util.py:1190
    optimizer.zero_grad(set_to_none=True)
    loss, grads = controlnet_loss_and_grads(controlnet, unet, scheduler, e, noise, timesteps, cond, prompt_embeds,
                                            acquisition_time, conditioning_scale, snr_gamma)
    with_grad = []
    for p in controlnet.parameters():
        if p.requires_grad:
            p.grad = grads.get(p)  # None where the tape recorded no gradient
            if p.grad is not None:
                with_grad.append(p)
    # one fused finite-ness test over the loss and every gradient (what GradScaler.unscale_ records as found_inf)
    norms = torch._foreach_norm([p.grad for p in with_grad]) if with_grad else []
    total = torch.linalg.vector_norm(torch.stack([n.float() for n in norms])) if norms else loss.new_zeros(())
    finite = bool(torch.isfinite(loss).item()) and bool(torch.isfinite(total).item())
    norm = None
    if not finite:
        logging.getLogger(__name__).warning("train_step: non-finite loss or gradient (loss %s, gradient norm %s): optimizer step "
                                            "skipped", float(loss), float(total))
        return loss, norm
    if clip_grad_norm_by is not None:
        if clip_grad_norm_or_value == "norm":
            norm = torch.nn.utils.clip_grad_norm_(with_grad, clip_grad_norm_by)
        else:
            norm = total
            torch.nn.utils.clip_grad_value_(with_grad, clip_grad_norm_by)
    optimizer.step()
    controlnet._invalidate()
    return loss, norm


class Stage1Wrapper(nn.Module):
    """util.py:1473-1492: the frozen stage 1 as the training loop calls it -- images -> a sampled latent z = mu + sigma * eps
    (`stack_input_to_rgb` is the diffusers SD-VAE branch: grey images stacked to three channels, posterior sample)."""

    def __init__(self, model: nn.Module, stack_input_to_rgb: bool = False) -> None:
        super().__init__()
        self.model = model
        self.stack_input_to_rgb = stack_input_to_rgb

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[1] == 1 and self.stack_input_to_rgb:
            x = torch.cat((x, x, x), dim=1)
            return self.model.encode(x).latent_dist.sample()
        z_mu, z_sigma = self.model.encode(x)
        return self.model.sampling(z_mu, z_sigma)


def train_epoch_controlnet(controlnet, diffusion, stage1, scheduler, text_encoder, loader, optimizer, device, epoch: int = 0,
                           writer=None, scaler=None, scale_factor: float = 1.0, cond_on_acq_times: bool = False,
                           clip_grad_norm_by: Optional[float] = None, clip_grad_norm_or_value: str = "norm",
                           controlnet_conditioning_scale: float = 1.0, snr_gamma: Optional[float] = None, lr_scheduler=None):
    """training_functions.py:888-1043 with the same arguments (drop-in for the reference's loop; `writer` and `scaler` are
    accepted and unused: there is no fp16 loss scaling on this path -- activations are bf16 -- and sample logging is the
    sampler's `preview_controlnet_sample`).  Batches are dicts with the reference's keys (training_functions.py:18-23).
    Returns the list of per-step losses."""
    if clip_grad_norm_or_value not in ("norm", "value"):
        raise ValueError(f"clip_grad_at {clip_grad_norm_or_value} not supported. Use 'norm' or 'value'.")  # util.py:1190
    losses = []
    for x in loader:
        images = x["target"].to(device)
        reports = x["report"].to(device)
        cond = x["source"].to(device)
        acq = x["acquisition_time"].to(device).unsqueeze(0) if cond_on_acq_times else None
        timesteps = torch.randint(0, scheduler.num_train_timesteps, (images.shape[0],), device=device).long()
        e = stage1(images) * scale_factor
        prompt_embeds = text_encoder(reports.squeeze(1))[0]
        noise = torch.randn_like(e)
        loss, _ = train_step(controlnet, diffusion, scheduler, optimizer, e, noise, timesteps, cond, prompt_embeds, acq,
                             controlnet_conditioning_scale, snr_gamma, clip_grad_norm_by, clip_grad_norm_or_value)
        if lr_scheduler is not None:
            lr_scheduler.step()
        losses.append(float(loss))
    return losses


@torch.no_grad()
def eval_controlnet(controlnet, diffusion, stage1, scheduler, text_encoder, loader, device, step: int = 0, writer=None,
                    scale_factor: float = 1.0, break_at_iteration: int = -1, cond_on_acq_times: bool = False,
                    sample: bool = True, controlnet_conditioning_scale: float = 1.0) -> float:
    """training_functions.py:1046-1176 with the same arguments: the validation pass of the ControlNet.  Per batch a random
    timestep per image, latents from stage 1, the prompt through the text tower, one ControlNet + UNet forward on the HIP path
    (`sample_from_controlnet` / `sample_from_ldm`, no tape), MSE against the scheduler's target (:1126-1133); the batch means
    are weighted by the batch size and divided by `len(loader.dataset)` (:1139,1146 -- by the images seen for a loader without
    a dataset).  `break_at_iteration > 0` stops after that many + 1 batches (:1141).  With `sample` the last batch's first
    pair goes through the sampler (`preview_controlnet_sample`, the sampling part of `log_controlnet_samples`, :1157-1175) and,
    when `writer` has `add_image`, conditioning image / target / sample are logged at `step`; the matplotlib figure of the
    reference stays with the caller.  Returns the mean loss."""
    import torch.nn.functional as F

    from .sampling import preview_controlnet_sample, sample_from_controlnet, sample_from_ldm
    controlnet.eval()
    total, seen, counter = 0.0, 0, 0
    images = cond = reports = acq = e = None
    for x in loader:
        images = x["target"].to(device)
        reports = x["report"].to(device)
        cond = x["source"].to(device)
        acq = x["acquisition_time"].to(device).unsqueeze(0) if cond_on_acq_times else None
        timesteps = torch.randint(0, scheduler.num_train_timesteps, (images.shape[0],), device=device).long()
        e = stage1(images) * scale_factor
        prompt_embeds = text_encoder(reports.squeeze(1))[0]
        noise = torch.randn_like(e)
        noisy_e = scheduler.add_noise(original_samples=e, noise=noise, timesteps=timesteps)
        down, mid = sample_from_controlnet(controlnet=controlnet, noisy_e=noisy_e, controlnet_cond=cond, timesteps=timesteps,
                                           prompt_embeds=prompt_embeds, class_labels=acq,
                                           conditioning_scale=controlnet_conditioning_scale)
        noise_pred = sample_from_ldm(model=diffusion, noisy_e=noisy_e, timesteps=timesteps, prompt_embeds=prompt_embeds,
                                     down_block_res_samples=tuple(down), mid_block_res_sample=mid, class_labels=acq)
        if scheduler.prediction_type == "v_prediction":
            target = scheduler.get_velocity(e, noise, timesteps)
        elif scheduler.prediction_type == "epsilon":
            target = noise
        else:
            raise ValueError(f"prediction_type {scheduler.prediction_type!r} has no training target upstream")
        loss = F.mse_loss(noise_pred.float(), target.float())
        total += float(loss) * images.shape[0]
        seen += images.shape[0]
        if break_at_iteration > 0 and counter == break_at_iteration:
            break
        counter += 1
    if seen == 0:
        raise ValueError("eval_controlnet: the loader yielded no batch")
    total /= len(loader.dataset) if hasattr(loader, "dataset") else seen
    if sample:
        acq0 = acq.reshape(-1)[:1] if acq is not None else None
        res = preview_controlnet_sample(cond[:1], controlnet, diffusion, stage1, text_encoder, scheduler,
                                        spatial_shape=tuple(e.shape[1:]), device=device, scale_factor=scale_factor,
                                        acquisition_time=acq0, text_condition=reports[0].reshape(1, -1),
                                        controlnet_conditioning_scale=controlnet_conditioning_scale)
        if writer is not None and hasattr(writer, "add_image"):
            to01 = lambda t: ((t.float().clamp(-1, 1) + 1) * 0.5)[0, :1].cpu()  # noqa: E731
            writer.add_image("validation/controlnet_cond", to01(cond), step)
            writer.add_image("validation/target", to01(images), step)
            writer.add_image("validation/sample", to01(res["image"]), step)
    return total
