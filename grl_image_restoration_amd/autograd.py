"""Autograd wiring of the HIP kernels for the training path (BASELINE config 5).

The reference trains by plain autograd through its PyTorch modules (engines/base.py:221-236).  Here every contraction of
the forward AND the backward pass runs in libgrl_hip.so:

  LinearFn     y = x W^T + b      fwd grl_linear_fwd | dx = grl_linear_fwd on (dy, W) | dW = grl_gemm_tn(dy, x)
  Conv3x3Fn    3x3 conv (pad 1)   fwd grl_conv3x3_fwd | dx = grl_conv3x3_fwd on (dy, flipped W^T) | dW = grl_gemm_tn (9 taps)
  AttentionFn  cosine attention   fwd grl_attention_fwd (+ log-sum-exp) | grl_attention_bwd (dq, dk, dv, dtable)

and the element-wise glue between them (LayerNorm, GELU, L2 normalisation, logit scale, CPB-MLP, squeeze-excite, residuals,
pixel shuffle) is ordinary differentiable torch code on the GPU (model.py: ``GRL._forward_train``).

Gradient range: the kernels contract fp16 operands.  An L1 loss over a 256x256 output produces gradients of ~1e-6, below the
fp16 normal range, so the backward contractions multiply their gradient operand by a power of two on its way to fp16 and
divide the product by it again (``a_scale`` / ``out_scale`` / ``g_scale`` of the C ABI): every Function receives and returns
true-valued fp32 gradients.  The factor is chosen once per backward pass from the largest gradient entering the network
(``GradScaleTop``: one host read per step).
"""
import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops

_SIZES = (64, 96, 128, 192, 256, 384, 576, 768, 1152)   # widths both as N (n-tile chunks of 4/6/8) and as K (k-steps) of grl_linear_fwd


def pad_width(n: int) -> int:
    for s in _SIZES:
        if n <= s:
            return s
    raise ValueError(f"layer width {n} exceeds the linear kernel's largest shape")


class _State:
    scale = 1.0          # gradient operand scale of the current backward pass (power of two)
    target = 64.0        # the largest incoming gradient is brought to about this magnitude


def grad_scale() -> float:
    return _State.scale


class GradScaleTop(torch.autograd.Function):
    """Identity on the network output; in backward it fixes the gradient scale of the pass from max|dL/dy|."""

    @staticmethod
    def forward(ctx, y):
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        amax = float(dy.abs().max())
        _State.scale = 2.0 ** math.floor(math.log2(_State.target / amax)) if amax > 0 and math.isfinite(amax) else 1.0
        return dy


def _padded(x: torch.Tensor, width: int) -> torch.Tensor:
    return x if x.shape[1] == width and x.is_contiguous() else F.pad(x, (0, width - x.shape[1])).contiguous()


class LinearFn(torch.autograd.Function):
    """y[M, N] = x[M, K] w[N, K]^T + b  (fp32 in / out, fp16 operands in the kernel)."""

    @staticmethod
    def forward(ctx, x, w, b):
        M, K = x.shape
        N = w.shape[0]
        Kp, Np = pad_width(K), pad_width(N)
        xp = _padded(x.detach().float(), Kp)
        wp = torch.zeros(Np, Kp, dtype=ops.GEMM_DTYPE, device=x.device)
        wp[:N, :K] = w.detach()
        bp = torch.zeros(Np, dtype=torch.float32, device=x.device)
        if b is not None:
            bp[:N] = b.detach()
        y = ops.linear(xp, wp, bp, out_dtype=torch.float32)
        ctx.save_for_backward(xp, wp)
        ctx.dims = (M, K, N, Kp, Np, b is not None)
        return y[:, :N]

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        M, K, N, Kp, Np, has_b = ctx.dims
        s = grad_scale()
        dyp = _padded(dy.float(), Np)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = wp.t().contiguous()                                   # [Kp, Np]: rows = input channels
            dx = ops.linear(dyp, wt, torch.zeros(Kp, dtype=torch.float32, device=dy.device), out_dtype=torch.float32,
                            a_scale=s, out_scale=1.0 / s)[:, :K]
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(dyp, xp, Np, Kp, a_scale=s, out_scale=1.0 / s)[0, :N, :K]
        if has_b and ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db


class Conv3x3Fn(torch.autograd.Function):
    """3x3 convolution, stride 1, zero pad 1, on channels-last token matrices x[B*H*W, Cin] -> [B*H*W, Cout]."""

    @staticmethod
    def forward(ctx, x, w, b, B, H, W):
        Cout, Cin = w.shape[:2]
        CinP, CoutP = (Cin + 31) // 32 * 32, (Cout + 15) // 16 * 16
        xp = _padded(x.detach().float(), CinP)
        wpk = ops.pack_conv_weight(w.detach(), CinP, CoutP)
        bpk = ops.pack_conv_bias(b.detach(), CoutP)
        y = ops.conv3x3(xp, wpk, bpk, B, H, W)
        ctx.save_for_backward(xp, w.detach())
        ctx.dims = (B, H, W, Cin, Cout, CinP, CoutP)
        return y[:, :Cout]

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        B, H, W, Cin, Cout, CinP, CoutP = ctx.dims
        s = grad_scale()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # data gradient = the same convolution with the taps flipped and the channel roles swapped
            gin, gout = (Cout + 31) // 32 * 32, (Cin + 15) // 16 * 16
            wt = ops.pack_conv_weight(w.flip(2, 3).transpose(0, 1).contiguous(), gin, gout)
            dyp = _padded(dy.float(), gin)
            dx = ops.conv3x3(dyp, wt, torch.zeros(gout, dtype=torch.float32, device=dy.device), B, H, W, x_scale=s, out_scale=1.0 / s)[:, :Cin]
        if ctx.needs_input_grad[1]:
            n8 = (Cout + 7) // 8 * 8
            dyp = _padded(dy.float(), n8)
            c = ops.gemm_tn(dyp, xp, n8, CinP, taps=9, hw=(H, W), a_scale=s, out_scale=1.0 / s)      # [9, n8, CinP]
            dw = c[:, :Cout, :Cin].reshape(3, 3, Cout, Cin).permute(2, 3, 0, 1).contiguous()
        if ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db, None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T + bias (+ mask)) v over every window; operands are head planes [nh, tokens, 32] (fp32 here, fp16 in the
    kernels): q = normalised * scale * log2e, k = normalised (slot 31 := 1.0 by this function), v raw (slot ``d`` := 1.0 by this
    function); ``table`` from tables.kernel_table.  Returns fp32 planes [nh, q_tokens, 32]."""

    @staticmethod
    def forward(ctx, q, k, v, table, geo):
        # geo: dict(q=(Himg, Wimg, wh, ww, shy, shx), k=(...), B, nh, d, masked, floor)
        nh, d = geo["nh"], geo["d"]
        q16 = q.detach().to(ops.PLANE_DTYPE)
        k16 = k.detach().to(ops.PLANE_DTYPE)
        v16 = v.detach().to(ops.PLANE_DTYPE)
        ones = d if d < 32 else -1
        one31 = d <= 30
        if one31:
            k16[..., 31] = 1.0
        if ones >= 0:
            v16[..., ones] = 1.0
        o = torch.empty(nh, q.shape[1], 32, dtype=torch.float32, device=q.device)
        lse = torch.empty(nh, q.shape[1], dtype=torch.float32, device=q.device)
        TG = ops.TokenGrid
        tab = table.detach().contiguous()
        ops.attention(TG(q16, 0, *geo["q"]), TG(k16, 0, *geo["k"]), TG(v16, 0, *geo["k"]), TG(o, 0, *geo["q"]), B=geo["B"], nh=nh,
                      table=tab, masked=geo["masked"], ones_col=ones, head_dim=d, k_one31=one31,
                      lazy_floor=geo["floor"] if one31 else None, lse=lse)
        ctx.save_for_backward(q16, k16, v16, o, lse, tab)
        ctx.geo = geo
        return o

    @staticmethod
    def backward(ctx, d_o):
        q16, k16, v16, o, lse, tab = ctx.saved_tensors
        geo = ctx.geo
        nh, d = geo["nh"], geo["d"]
        TG = ops.TokenGrid
        dq, dk, dv, dtab = ops.attention_bwd(TG(q16, 0, *geo["q"]), TG(k16, 0, *geo["k"]), TG(v16, 0, *geo["k"]), TG(o, 0, *geo["q"]),
                                             d_o.float().contiguous(), lse, B=geo["B"], nh=nh, table=tab, masked=geo["masked"],
                                             ones_col=d if d < 32 else -1, head_dim=d, g_scale=grad_scale())
        return dq, dk, dv, dtab, None


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


def conv3x3(x, w, b, B, H, W):
    return Conv3x3Fn.apply(x, w, b, B, H, W)
