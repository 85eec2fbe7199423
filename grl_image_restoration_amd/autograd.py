"""Autograd wiring of the HIP kernels for the training path (BASELINE config 5).

The reference trains by plain autograd through its PyTorch modules (engines/base.py:221-236).  Here every contraction of
the forward AND the backward pass runs in libgrl_hip.so:

  torch.ops.grl.linear     y = x W^T + b      fwd grl_linear_fwd | dx = grl_linear_fwd on (dy, W) | dW = grl_gemm_tn(dy, x)
  torch.ops.grl.conv3x3    3x3 conv (pad 1)   fwd grl_conv3x3_fwd | dx = grl_conv3x3_fwd on (dy, flipped W^T) | dW = grl_gemm_tn (9 taps)
  torch.ops.grl.attention  cosine attention   fwd grl_attention_fwd (+ log-sum-exp) | grl_attention_bwd (dq, dk, dv, dtable)

(torch.library custom ops with registered autograd and fake kernels)

and the element-wise glue between them (LayerNorm, GELU, L2 normalisation, logit scale, CPB-MLP, squeeze-excite, residuals,
pixel shuffle) is ordinary differentiable torch code on the GPU (model.py: ``GRL._forward_train``).

Gradient range: the kernels contract fp16 operands.  An L1 loss over a 256x256 output produces gradients of ~1e-6, below the
fp16 normal range, so the backward contractions multiply their gradient operand by a power of two on its way to fp16 and
divide the product by it again (``a_scale`` / ``out_scale`` / ``g_scale`` of the C ABI): every op receives and returns
true-valued fp32 gradients.  The factor is chosen once per backward pass from the largest gradient entering the network
(``GradScaleTop``: one host read per step).
"""
import math
import os
import threading
import weakref
from typing import Optional, Sequence


import torch
import torch.nn.functional as F

from . import _lib as L
from . import composite, ops

_SIZES = (64, 96, 128, 192, 256, 384, 576, 768, 1152)   # widths both as N (n-tile chunks of 4/6/8) and as K (k-steps) of grl_linear_fwd


def pad_width(n: int) -> int:
    for s in _SIZES:
        if n <= s:
            return s
    raise ValueError(f"layer width {n} exceeds the linear kernel's largest shape")


class _State:
    # gradient operand scale (a power of two) of the backward pass running on a device: autograd runs the backward of a device's
    # nodes on that device's own engine thread, so models on different GPUs of one process do not see each other's scale
    # (keying by thread as well would hide the value from the thread that called backward())
    scales: dict = {}
    target = 64.0        # the largest incoming gradient is brought to about this magnitude
    frozen = False       # frozen_grad_scale(): keep the current scales (no host read of max|dL/dy|: graph capture)


def _scale_key(device) -> int:
    idx = torch.device(device).index if device is not None else None
    return idx if idx is not None else torch.cuda.current_device()


class frozen_grad_scale:
    """Context: the backward passes inside keep the gradient operand scale of the last pass before it instead of reading
    max|dL/dy| back to the host -- what a step captured in a HIP graph needs (the scale is a power of two with 2^10 of headroom to
    the fp16 range: it only has to be of the right order of magnitude)."""

    def __enter__(self):
        self._prev, _State.frozen = _State.frozen, True
        return self

    def __exit__(self, *exc):
        _State.frozen = self._prev
        return False


def grad_scale(device=None) -> float:
    return _State.scales.get(_scale_key(device), 1.0)


def last_grad_scale(device=None) -> float:
    """The power-of-two gradient operand scale of the most recent (non-frozen) backward pass on ``device``: what a captured step
    froze, and what GraphedTrainStep compares against when it re-calibrates."""
    if device is None and _State.scales:
        return next(reversed(_State.scales.values()))
    return _State.scales.get(_scale_key(device), 1.0)


class GradScaleTop(torch.autograd.Function):
    """Identity on the network output; in backward it fixes the gradient scale of the pass from max|dL/dy|."""

    @staticmethod
    def forward(ctx, y):
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        ops.zero_arena_begin(dy.device)          # (the backward pass starts here: one zero fill for its accumulation buffers)
        if _State.frozen:
            return dy
        amax = float(dy.abs().max())
        _State.scales[_scale_key(dy.device)] = 2.0 ** math.floor(math.log2(_State.target / amax)) if amax > 0 and math.isfinite(amax) else 1.0
        return dy


_PAD_BLOCKS: dict = {}     # (rows, width, ones, device) -> constant fp32 block appended to a token matrix
_ZERO_VECS: dict = {}      # (n, device) -> fp32 zeros (bias operand of the data-gradient launches)
_WEIGHTS: dict = {}        # data_ptr of a weight -> (version, shape, Np, Kp, padded fp16 copy): rebuilt when the optimizer bumps the version


def _padded(x: torch.Tensor, width: int, ones: bool = False) -> torch.Tensor:
    """x [M, K] -> contiguous [M, width] with zero pad columns in ONE launch (a cat with a cached constant block; F.pad is a
    fill plus a copy).  ``ones``: the first pad column holds 1.0 -- against zero weight columns it is inert in the forward and
    in the data gradient, and in the weight-gradient contraction dy^T [x | 1 | 0] it yields the bias gradient for free."""
    M, K = x.shape
    if K == width:
        return x.contiguous()
    key = (M, width - K, ones, x.device)
    blk = _PAD_BLOCKS.get(key)
    if blk is None:
        blk = torch.zeros(M, width - K, dtype=torch.float32, device=x.device)
        if ones:
            blk[:, 0] = 1.0
        _PAD_BLOCKS[key] = blk
    return torch.cat([x, blk], 1)


def _zeros(n: int, device) -> torch.Tensor:
    z = _ZERO_VECS.get((n, device))
    if z is None:
        z = _ZERO_VECS[(n, device)] = torch.zeros(n, dtype=torch.float32, device=device)
    return z


_REGISTERED = weakref.WeakKeyDictionary()   # module -> storage addresses of its parameters (long-lived: safe cache keys)


def forget_parameters(module: torch.nn.Module) -> None:
    """Invalidates the cached padded fp16 copies of ``module``'s weights (GRL.invalidate_plan: weights changed behind autograd's back).
    Parameters that are still where they were keep their BUFFERS -- the entry is only marked stale and refreshed in place at its next
    use: a captured training step (train_graph.py) holds the addresses of those buffers and rewrites them on every replay, so
    returning them to the allocator would let a replay scribble over whatever tensor received the memory next (round 5: an eval forward
    between two replays did exactly that through invalidate_plan -- an order-dependent failure of the LR-schedule test).  Parameters that
    were REPLACED by new tensors: the copies under the addresses the module was registered with go (those addresses are free for other
    tensors now), and a registered module is registered again under its current ones."""
    old = _REGISTERED.get(module)
    cur = frozenset(p.data_ptr() for p in module.parameters())
    for ptr in old or ():
        if ptr not in cur:
            _WEIGHTS.pop(ptr, None)
    for ptr in cur:
        ent = _WEIGHTS.get(ptr)
        if isinstance(ent, list):
            ent[0], ent[3] = None, None          # no version matches: the next _padded_weight copies into the same buffer again
        elif ent is not None:
            _WEIGHTS.pop(ptr, None)
    if old is not None:
        register_parameters(module, _keep=cur)


def register_parameters(module: torch.nn.Module, _keep: frozenset = frozenset()) -> None:
    """Allow the padded fp16 copies of this module's weights to be kept between the forward and the backward of a step (and
    across steps until the optimizer changes them).  Only registered parameters are cached: a temporary tensor's address can be
    reused by another tensor with the same shape and version, a parameter's cannot while its module lives (the registration
    goes away with the module)."""
    ptrs = frozenset(p.data_ptr() for p in module.parameters())
    for ptr in ptrs:                       # a previous owner of the same addresses (a deleted model) may have left copies behind
        if ptr not in _keep:               # (forget_parameters: entries of this module it has just marked stale)
            _WEIGHTS.pop(ptr, None)
    _REGISTERED[module] = ptrs


def _is_registered(ptr: int) -> bool:
    return any(ptr in s for s in _REGISTERED.values())


def _padded_weight(w: torch.Tensor, Np: int, Kp: int, transposed: bool = False, bias: Optional[torch.Tensor] = None, want_bias: bool = False):
    """fp16 [Np, Kp] zero-padded copy of a weight (``transposed``: its [Kp, Np] transpose, the operand of the data-gradient
    launch); for registered parameters kept per parameter and refreshed when the version counter moves (an optimizer step),
    so the backward of a step finds the copy its forward made.  ``want_bias``: returns (copy, fp32 bias [Np] zero padded) -- the
    bias of the layer packed by the same launch (round 6: a cat per linear layer and step before)."""
    key = w.data_ptr()
    bver = None if bias is None else (bias.data_ptr(), bias._version)
    sig = (tuple(w.shape), Np, Kp, w.device)
    keep = _is_registered(key)
    ent = _WEIGHTS.get(key) if keep else None
    stale = ent is None or ent[0] != w._version or ent[1] != sig or (want_bias and ent[5] != bver)
    if stale:
        reuse = ent is not None and ent[1] == sig
        buf = ent[2] if reuse else ops.empty(Np, Kp, dtype=ops.GEMM_DTYPE, device=w.device)
        # a parameter that needs gradients will want the transpose in this step's backward pass: both in one launch (the buffers
        # are kept: a captured step holds their addresses, see forget_parameters)
        tbuf = ent[4] if reuse else None
        if tbuf is None and (w.requires_grad or transposed):
            tbuf = ops.empty(Kp, Np, dtype=ops.GEMM_DTYPE, device=w.device)
        bbuf = ent[6] if reuse and len(ent) > 6 else None
        if bbuf is None and want_bias:
            bbuf = ops.empty(Np, dtype=torch.float32, device=w.device)
        if w.is_cuda:
            ops.pack_linear_train(w, buf, tbuf, bias if want_bias else None, bbuf if want_bias else None)
        else:                                  # (host-side cache logic is tested on CPU tensors; nothing computes there)
            buf.zero_()
            buf[: w.shape[0], : w.shape[1]].copy_(w.detach())
            if tbuf is not None:
                tbuf.copy_(buf.t())
            if want_bias:
                bbuf.zero_()
                if bias is not None:
                    bbuf[: bias.numel()].copy_(bias.detach())
        ent = [w._version, sig, buf, tbuf, tbuf, bver if want_bias else (ent[5] if reuse and len(ent) > 5 else None), bbuf]
        if keep:
            _WEIGHTS[key] = ent
    if want_bias:
        return ent[2], ent[6]
    if not transposed:
        return ent[2]
    if ent[3] is None:
        ent[3] = ent[4] = ent[2].t().contiguous()
    return ent[3]


# ---- the contractions as torch.library custom ops (torch.ops.grl.*) with registered autograd ----------------------------
# forward = one or two C-ABI launches; backward = the C-ABI launches listed in the module docstring.  Registered ops (rather than
# bare autograd.Function) are visible to torch.compile / export, carry fake (meta) kernels and are what DDP / Lightning see.

# The padded operand [x | 1 | 0] a forward launch read is exactly what the weight-gradient contraction of its backward wants.  The
# op's implementation leaves it here; the op's setup_context -- which the dispatcher runs right after the implementation, on the same
# thread, and only when autograd is recording -- takes it into the node (one slot per thread: nothing accumulates when no graph is
# recorded).  Saves one cat per linear / convolution in the backward pass (7 per block).
_HANDOVER = threading.local()


def _leave_operand(x, xp):
    _HANDOVER.slot = (x.data_ptr(), x._version, tuple(x.shape), xp) if xp.data_ptr() != x.data_ptr() else None


def _take_operand(x):
    slot, _HANDOVER.slot = getattr(_HANDOVER, "slot", None), None
    if slot is None:
        return None
    try:                                   # (fake / meta tensors -- torch.compile tracing -- have no storage address)
        key = (x.data_ptr(), x._version, tuple(x.shape))
    except (RuntimeError, NotImplementedError):
        return None
    return slot[3] if slot[:3] == key else None


def _rows16(t: torch.Tensor) -> torch.Tensor:
    """fp32 matrix whose rows the kernels can read in 16-byte pieces where it lies: unit column stride, row stride a multiple of 4,
    16-byte aligned base (a narrow() of a wider matrix qualifies); anything else is copied."""
    t = t.float()
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


# Round 6: operands and results at their REAL widths.  The kernels used to want every token matrix padded to their channel multiple
# (C = 180 -> 192): one cat per operand on the way in, one strided copy per result on the way out -- ~650 cats and ~600 copies of a
# captured step's ~10 k nodes, 18 ms of 117.  GrlLinearArgs.a_cols / n_store, GrlConvArgs.x_cols / n_store and GrlGemmTnArgs.b_ones
# (ABI 22) let the loaders supply the pad columns as constants and the epilogues skip them, for widths that are multiples of 4
# (16-byte pieces); other widths (the 90-wide anchor projection, the 45-wide CAB bottleneck) keep the padded path.
# GRL_REAL_WIDTHS=0: the padded path everywhere (A/B timing, and the reference the tests compare the new path with).
def _real(n: int, npad: int) -> bool:
    return n % 4 == 0 and n < npad and _REAL_WIDTHS[0]


_REAL_WIDTHS = [os.environ.get("GRL_REAL_WIDTHS", "1") != "0"]
_F16_HANDOVER = [os.environ.get("GRL_F16_HANDOVER", "1") != "0"]     # linear layers: fp16 operand copies for the weight gradient


@torch.library.custom_op("grl::linear", mutates_args=())
def linear_op(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], one_col: int = -1, gelu_in: bool = False) -> torch.Tensor:
    """y[M, N] = x[M, K] w[N, K]^T + b  (fp32 in / out, fp16 operands in grl_linear_fwd).  ``one_col`` >= 0: the caller's promise
    that column one_col of x holds 1.0 (against a zero weight column) -- the bias gradient is then read off the weight-gradient
    contraction, as with the pad column elsewhere, instead of a separate column sum of dy.  ``gelu_in``: y = gelu(x) w^T + b -- fc2 of the
    Mlp on fc1's pre-activation (swin_v1_block.py:37-43): the loader applies the GELU on its way to fp16 and the data-gradient launch
    multiplies by gelu'(x) in its epilogue, so neither the activation nor its adjoint is a launch (or a tensor) of its own."""
    M, K = x.shape
    N = w.shape[0]
    Kp, Np = pad_width(K), pad_width(N)
    wp, bp = _padded_weight(w, Np, Kp, bias=b, want_bias=True)     # (weight, its transpose and the padded bias: one launch per step)
    if _real(K, Kp):
        xa, kw = _rows16(x.detach()), dict(a_cols=K, a_one=True)      # (the ones column meets a zero weight column)
        _leave_operand(x, xa)
    else:
        xa, kw = _padded(x.detach().float(), Kp, ones=True), {}
        _leave_operand(x, xa)
    if gelu_in:
        kw["a_gelu"] = True
    if w.requires_grad and _F16_HANDOVER[0] and K % 4 == 0:
        # the fp16 operand [x | 1 | 0] the kernel contracts, kept for the weight gradient INSTEAD of x: grl_gemm_tn then reads 2-byte
        # values from 16-byte aligned rows once per output tile instead of converting the fp32 matrix every time (and the saved
        # activation is half the size)
        x16 = ops.empty(M, Kp, dtype=ops.GEMM_DTYPE, device=x.device)
        kw["a16_out"] = x16
        _leave_operand(x, x16)
    if _real(N, Np):
        return ops.linear(xa, wp, bp, out_dtype=torch.float32, n_store=N, **kw)
    y = ops.linear(xa, wp, bp, out_dtype=torch.float32, **kw)
    return y if N == Np else y[:, :N].contiguous()


@linear_op.register_fake
def _(x, w, b, one_col=-1, gelu_in=False):
    return x.new_empty(x.shape[0], w.shape[0], dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, w, b, one_col, gelu_in = inputs
    ctx.one_col = int(one_col)
    ctx.gelu_in = bool(gelu_in)
    xp = _take_operand(x)
    # the operand the forward launch read -- [x | 1 | 0], or x itself where the kernels take real widths -- is all the backward needs
    # of x: saved INSTEAD of x (both were kept, doubling the saved activation of every layer whose width is not a kernel
    # width -- C = 180 in all GRL-Base blocks)
    ctx.x_shape, ctx.padded = tuple(x.shape), xp is not None and xp.shape[1] != x.shape[1]
    ctx.x16 = xp is not None and xp.dtype == ops.GEMM_DTYPE
    if ctx.gelu_in:        # the pre-activation for gelu' in the backward pass, and the operand the launch contracted (fp16 gelu(x), if handed over)
        ctx.save_for_backward(xp if ctx.x16 else x, w, x)
    else:
        ctx.save_for_backward(xp if xp is not None else x, w)
    ctx.has_b = b is not None


def _linear_backward(ctx, dy):
    if ctx.gelu_in:
        xs, w, pre = ctx.saved_tensors
        pre = _rows16(pre.detach())
        if not ctx.x16:                        # (no fp16 operand was handed over: the weight gradient needs gelu(x) as a tensor)
            xs = F.gelu(pre)
    else:
        xs, w = ctx.saved_tensors
        pre = None
    M, K = ctx.x_shape
    N = w.shape[0]
    Kp, Np = pad_width(K), pad_width(N)
    rk, rn = _real(K, Kp), _real(N, Np)
    s = grad_scale(dy.device)
    if rn:
        dya, dkw = _rows16(dy), dict(a_cols=N)
    else:
        dya, dkw = _padded(dy.float(), Np), {}
    dx = dw = db = None
    want_b = ctx.has_b and ctx.needs_input_grad[2]
    need_w = ctx.needs_input_grad[1] or (want_b and Kp > K)
    dy16 = None
    if ctx.needs_input_grad[0]:
        wt = _padded_weight(w, Np, Kp, transposed=True)            # [Kp, Np]: rows = input channels
        if need_w and ctx.x16:                                     # the fp16 (scaled) dy of this launch feeds the weight gradient below
            dy16 = dkw["a16_out"] = ops.empty(M, Np, dtype=ops.GEMM_DTYPE, device=dy.device)
        if pre is not None:                                        # ... times gelu'(x), in the launch's epilogue
            dkw.update(epi=L.EPI_GELU_GRAD, resid=pre if (rk or K == Kp) else _padded(pre, Kp))
        if rk:
            dx = ops.linear(dya, wt, _zeros(Kp, dy.device), out_dtype=torch.float32, a_scale=s, out_scale=1.0 / s, n_store=K, **dkw)
        else:
            dx = ops.linear(dya, wt, _zeros(Kp, dy.device), out_dtype=torch.float32, a_scale=s, out_scale=1.0 / s, **dkw)[:, :K]
    if need_w and ctx.x16:
        # fp16 operands on both sides: xs = [x | 1 | 0] (Kp wide, the ones column routes the bias gradient), dy16 = s * dy (Np wide)
        ga = dy16 if dy16 is not None else dya
        Ng = N if (N % 4 == 0 and (dy16 is not None or rn or N == Np)) else Np       # (rows beyond N: zero pad columns of the operand)
        if Kp > K:
            full, cb = ops.gemm_tn(ga, xs, Ng, K, a_scale=s, out_scale=1.0 / s, b_ones=True)
            db = cb[:N] if want_b else None
        else:
            full = ops.gemm_tn(ga, xs, Ng, K, a_scale=s, out_scale=1.0 / s)
            if want_b and 0 <= ctx.one_col < K:
                db = full[0][:N, ctx.one_col]                        # a column of x that holds 1.0 (the caller's promise)
        if ctx.needs_input_grad[1]:
            dw = full[0][:N]
    elif need_w:
        if ctx.padded:
            xb = xs                                                  # [x | 1 | 0] of the forward launch
        elif rk:
            xb = _rows16(xs.detach())                                # x at its real width; the ones column is virtual
        else:
            xb = _padded(xs.detach().float(), Kp, ones=True)
        virt = xb.shape[1] == K and Kp > K and rk                    # bias gradient from the virtual ones column
        Ng, Kg = (N if rn else Np), (K if xb.shape[1] == K else Kp)
        if virt and want_b:
            full, cb = ops.gemm_tn(dya, xb, Ng, Kg, a_scale=s, out_scale=1.0 / s, b_ones=True)
            db = cb[:N]
        else:
            full = ops.gemm_tn(dya, xb, Ng, Kg, a_scale=s, out_scale=1.0 / s)
            if want_b and Kg > K:
                db = full[0][:N, K]                                  # the ones column of [x | 1 | 0]: sum over the rows of dy
            elif want_b and 0 <= ctx.one_col < K:
                db = full[0][:N, ctx.one_col]                        # a column of x that holds 1.0 (the caller's promise)
        if ctx.needs_input_grad[1]:
            dw = full[0][:N, :K]                                     # (contiguous -- adopted as .grad without a copy -- when Kg == K)
    if want_b and db is None:
        db = dy.float().sum(0)
    return dx, dw, db, None, None


linear_op.register_autograd(_linear_backward, setup_context=_linear_setup)


def _conv_pads(Cin: int, Cout: int):
    CinP, CoutP = (Cin + 31) // 32 * 32, (Cout + 15) // 16 * 16
    keep192 = CinP == 192 and CoutP == 192        # csrc/conv192.hip's shape (stage / after-body convolutions): it reads padded rows
    return CinP, CoutP, keep192


@torch.library.custom_op("grl::conv3x3", mutates_args=())
def conv3x3_op(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """3x3 convolution, stride 1, zero pad 1, on channels-last token matrices x[B*H*W, Cin] -> [B*H*W, Cout] (grl_conv3x3_fwd)."""
    Cout, Cin = w.shape[:2]
    CinP, CoutP, keep192 = _conv_pads(Cin, Cout)
    wp, bp = ops.pack_conv_train(w, b, CoutP, CinP)           # one launch (round 6; torch chain: pack_conv_weight + pack_conv_bias)
    if _real(Cin, CinP) and not keep192:
        xa, kw = _rows16(x.detach()), dict(x_cols=Cin)
    else:
        xa, kw = _padded(x.detach().float(), CinP, ones=True), {}   # (the ones column meets zero weight columns; see _padded)
    _leave_operand(x, xa)
    if _real(Cout, CoutP) and not keep192:
        return ops.conv3x3(xa, wp, bp, B, H, W, n_store=Cout, **kw)
    y = ops.conv3x3(xa, wp, bp, B, H, W, **kw)
    return y if Cout == CoutP else y[:, :Cout].contiguous()


@conv3x3_op.register_fake
def _(x, w, b, B, H, W):
    return x.new_empty(x.shape[0], w.shape[0], dtype=torch.float32)


def _conv_setup(ctx, inputs, output):
    x, w, b, B, H, W = inputs
    xp = _take_operand(x)
    ctx.save_for_backward(xp if xp is not None else x, w)      # (see _linear_setup)
    ctx.padded = xp is not None and xp.shape[1] != x.shape[1]
    ctx.bhw = (B, H, W)


def _conv_backward(ctx, dy):
    xs, w = ctx.saved_tensors
    B, H, W = ctx.bhw
    Cout, Cin = w.shape[:2]
    CinP, CoutP, keep192 = _conv_pads(Cin, Cout)
    s = grad_scale(dy.device)
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        # data gradient = the same convolution with the taps flipped and the channel roles swapped
        gin, gout = (Cout + 31) // 32 * 32, (Cin + 15) // 16 * 16
        g192 = gin == 192 and gout == 192
        wt, _ = ops.pack_conv_train(w, None, gout, gin, flip_t=True)
        if _real(Cout, gin) and not g192:
            dya, kw = _rows16(dy), dict(x_cols=Cout)
        else:
            dya, kw = _padded(dy.float(), gin), {}
        if _real(Cin, gout) and not g192:
            dx = ops.conv3x3(dya, wt, _zeros(gout, dy.device), B, H, W, x_scale=s, out_scale=1.0 / s, n_store=Cin, **kw)
        else:
            dx = ops.conv3x3(dya, wt, _zeros(gout, dy.device), B, H, W, x_scale=s, out_scale=1.0 / s, **kw)[:, :Cin]
    want_b = ctx.needs_input_grad[2]
    if ctx.needs_input_grad[1] or (want_b and CinP > Cin):
        n8 = (Cout + 7) // 8 * 8
        rn = Cout % 4 == 0 and _REAL_WIDTHS[0]
        dya = _rows16(dy) if rn else _padded(dy.float(), n8)
        Ng = Cout if rn else n8
        if ctx.padded:
            xb = xs
        elif _real(Cin, CinP) and not keep192:
            xb = _rows16(xs.detach())
        else:
            xb = _padded(xs.detach().float(), CinP, ones=True)
        virt = xb.shape[1] == Cin and CinP > Cin
        Kg = Cin if xb.shape[1] == Cin else CinP
        if virt and want_b:
            c, cb = ops.gemm_tn(dya, xb, Ng, Kg, taps=9, hw=(H, W), a_scale=s, out_scale=1.0 / s, b_ones=True)   # [9, Ng, Kg], [Ng]
            db = cb[:Cout]
        else:
            c = ops.gemm_tn(dya, xb, Ng, Kg, taps=9, hw=(H, W), a_scale=s, out_scale=1.0 / s)
            if want_b and Kg > Cin:
                db = c[4, :Cout, Cin]          # centre tap against the ones column of [x | 1 | 0]: sum of dy over all pixels
        if ctx.needs_input_grad[1]:
            dw = c[:, :Cout, :Cin].reshape(3, 3, Cout, Cin).permute(2, 3, 0, 1).contiguous()
    if want_b and db is None:
        db = dy.float().sum(0)
    return dx, dw, db, None, None, None


conv3x3_op.register_autograd(_conv_backward, setup_context=_conv_setup)


def _attn_operands(q, k, v, d, prepared, have=(None, None, None)):
    """fp16 copies of the head planes for the kernels.  ``prepared``: the caller has already put the constants into the pad
    columns (GRL._to_planes: 1.0 in k's slot 31 and in v's column d) -- otherwise two index fills per call.  ``have``: copies the
    caller already made (of prepared planes)."""
    q16 = have[0] if have[0] is not None else q.detach().to(ops.PLANE_DTYPE)
    k16 = have[1] if have[1] is not None else k.detach().to(ops.PLANE_DTYPE)
    v16 = have[2] if have[2] is not None else v.detach().to(ops.PLANE_DTYPE)
    if not prepared:
        if d <= 30 and have[1] is None:
            k16[..., 31] = 1.0          # partner of the kernel's running softmax offset (q slot 31)
        if d < 32 and have[2] is None:
            v16[..., d] = 1.0           # ones column: the softmax denominator falls out of the PV product
    return q16, k16, v16


@torch.library.custom_op("grl::attention", mutates_args=())
def attention_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, table: torch.Tensor, floor: torch.Tensor, qgeo: Sequence[int],
                 kgeo: Sequence[int], B: int, nh: int, d: int, masked: bool, prepared: bool = False,
                 q16: Optional[torch.Tensor] = None, k16: Optional[torch.Tensor] = None,
                 v16: Optional[torch.Tensor] = None,
                 token_major: bool = False) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """softmax(q k^T + bias (+ mask)) v over every window (grl_attention_fwd); operands are fp32 head planes [nh, tokens, 32]
    (fp16 in the kernel): q = normalised * scale * log2e, k normalised, v raw; ``table`` from tables.kernel_table; ``floor`` from
    tables.lazy_floor; qgeo / kgeo = (Himg, Wimg, wh, ww, shy, shx).  Returns (fp32 planes [nh, q_tokens, 32], log2-sum-exp2,
    and the fp16 operand planes the kernel ran on -- kept for the backward pass instead of converting them again).
    ``q16`` / ``k16`` / ``v16``: fp16 copies of (prepared) q / k / v the caller already has -- GRL._block_train converts the planes of
    a whole block in one launch instead of three per attention call; the matching outputs are empty then (an op's outputs must not
    alias its inputs) and the backward takes the operands from the inputs.  ``token_major``: the output as a token matrix
    [q_tokens, nh * 32] instead of head planes -- what the projection behind the attention reads (round 6: the planes of the two
    branches were concatenated, permuted and sliced into it, and the gradient took the same way back)."""
    have = (q16, k16, v16)
    q16, k16, v16 = _attn_operands(q, k, v, d, prepared, have)
    o = ops.empty(*((q.shape[1], nh * 32) if token_major else (nh, q.shape[1], 32)), dtype=torch.float32, device=q.device)
    lse = ops.empty(nh, q.shape[1], dtype=torch.float32, device=q.device)
    TG = ops.TokenGrid
    ops.attention(TG(q16, 0, *qgeo), TG(k16, 0, *kgeo), TG(v16, 0, *kgeo), TG(o, 0, *qgeo), B=B, nh=nh, table=table.detach().contiguous(),
                  masked=masked, ones_col=d if d < 32 else -1, head_dim=d, k_one31=d <= 30, lazy_floor=floor if d <= 30 else None, lse=lse)
    ret = lambda h, t: t if h is None else t.new_empty(0)
    return o, lse, ret(have[0], q16), ret(have[1], k16), ret(have[2], v16)


@attention_op.register_fake
def _(q, k, v, table, floor, qgeo, kgeo, B, nh, d, masked, prepared=False, q16=None, k16=None, v16=None, token_major=False):
    h = lambda t, g: t.new_empty(t.shape if g is None else (0,), dtype=ops.PLANE_DTYPE)
    return (q.new_empty(*((q.shape[1], nh * 32) if token_major else (nh, q.shape[1], 32)), dtype=torch.float32),
            q.new_empty(nh, q.shape[1], dtype=torch.float32),
            h(q, q16), h(k, k16), h(v, v16))


def _attn_setup(ctx, inputs, output):
    q, k, v, table, floor, qgeo, kgeo, B, nh, d, masked, prepared, q16_in, k16_in, v16_in, token_major = inputs
    o, lse, q16, k16, v16 = output
    q16, k16, v16 = (i if i is not None else t for i, t in zip((q16_in, k16_in, v16_in), (q16, k16, v16)))
    ctx.save_for_backward(q16, k16, v16, table, o, lse)
    ctx.geo = (tuple(qgeo), tuple(kgeo), B, nh, d, masked)
    # only `o` carries a gradient: without this autograd hands the backward a zero tensor per unused output (lse and up to three
    # fp16 operand planes of 6 MB each, per call -- ~300 fills per training step)
    ctx.set_materialize_grads(False)


def _attn_backward(ctx, d_o, d_lse, d_q16, d_k16, d_v16):
    q16, k16, v16, table, o, lse = ctx.saved_tensors
    qgeo, kgeo, B, nh, d, masked = ctx.geo
    if d_o is None:                      # (set_materialize_grads(False): the output took no part in the loss)
        return (None,) * 16
    TG = ops.TokenGrid
    d_o = _rows16(d_o) if d_o.dim() == 2 else d_o.float().contiguous()      # (token-major: a column block of the projection's input gradient)
    dq, dk, dv, dtab = ops.attention_bwd(TG(q16, 0, *qgeo), TG(k16, 0, *kgeo), TG(v16, 0, *kgeo), TG(o, 0, *qgeo), d_o,
                                         lse, B=B, nh=nh, table=table.detach().contiguous(), masked=masked, ones_col=d if d < 32 else -1,
                                         head_dim=d, g_scale=grad_scale(d_o.device))
    return dq, dk, dv, dtab, None, None, None, None, None, None, None, None, None, None, None, None


attention_op.register_autograd(_attn_backward, setup_context=_attn_setup)


class HeadPlanesFn(torch.autograd.Function):
    """x [T, S_in, nh, d] -> the fp32 head planes of S_out slots (differentiable) and their fp16 copies (ops.head_planes /
    head_planes_bwd, csrc/planes.hip).  ``scale`` [S_out, nh]; ``src`` / ``raw`` / ``one_cols``: tuples per output slot."""

    @staticmethod
    def forward(ctx, x, scale, src, raw, one_cols, write32=True):
        xc, sc = x.detach().float().contiguous(), scale.detach().float().contiguous()
        out32, out16 = ops.head_planes(xc, sc, src, raw, one_cols, write32)
        ctx.save_for_backward(xc, sc)
        ctx.cfg = (tuple(src), tuple(raw), tuple(one_cols))
        p32, p16 = out32.unbind(0), out16.unbind(0)
        ctx.mark_non_differentiable(*p16)
        ctx.set_materialize_grads(False)     # (no zero tensors for the fp16 copies / unused planes: head_planes_bwd takes None)
        return (*p32, *p16)

    @staticmethod
    def backward(ctx, *grads):
        x, scale = ctx.saved_tensors
        src, raw, one_cols = ctx.cfg
        S = len(src)
        want = [ctx.needs_input_grad[1] and not r for r in raw]
        dx, dscale = ops.head_planes_bwd(x, scale, src, raw, one_cols, list(grads[:S]), want)
        return dx, (dscale if ctx.needs_input_grad[1] else None), None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of a token matrix [M, n] (ops.layernorm_train / layernorm_bwd, csrc/ln_train.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xc = x.detach().float().contiguous()
        y, mean, rstd = ops.layernorm_train(xc, gamma, beta, eps)
        ctx.save_for_backward(xc, mean, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(dy.float().contiguous(), x, mean, rstd, gamma)
        return dx, dg, db, None


class LayerNormResFn(torch.autograd.Function):
    """resid + alpha * row_scale[image] * LayerNorm(x): the post-norm residual of a block (efficient.py:543-556) in the LayerNorm
    launches (csrc/ln_train.hip).  ``row_scale``: DropPath keep mask, one entry per image (no gradient), or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, resid, row_scale, rows_per_image, alpha):
        xc = _rows16(x.detach())
        y, mean, rstd = ops.layernorm_train(xc, gamma, beta, eps, resid=_rows16(resid.detach()), row_scale=row_scale,
                                            rows_per_image=rows_per_image, alpha=alpha)
        ctx.save_for_backward(xc, mean, rstd, gamma, row_scale)
        ctx.cfg = (rows_per_image, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, row_scale = ctx.saved_tensors
        rpi, alpha = ctx.cfg
        dyc = _rows16(dy)
        dx, dg, db = ops.layernorm_bwd(dyc, x, mean, rstd, gamma, row_scale=row_scale, rows_per_image=rpi, alpha=alpha)
        return dx, dg, db, None, (dy if ctx.needs_input_grad[4] else None), None, None, None


def layer_norm_residual(resid, x, gamma, beta, eps: float, row_scale, rows_per_image: int, alpha: float):
    """resid + alpha * row_scale[row // rows_per_image] * F.layer_norm(x) on token matrices [M, n] (row_scale None: 1)."""
    n = x.shape[-1]
    if x.is_cuda and x.dim() == 2 and n % 4 == 0 and n <= 256 and not ops.deterministic() and alpha != 0.0:
        return LayerNormResFn.apply(x, gamma, beta, eps, resid, row_scale, rows_per_image, alpha)
    t = F.layer_norm(x, (n,), gamma, beta, eps)
    if row_scale is None:
        return torch.add(resid, t, alpha=alpha)
    return torch.addcmul(resid.view(-1, rows_per_image, n), t.view(-1, rows_per_image, n), row_scale.view(-1, 1, 1), value=alpha).view_as(resid)


def layer_norm(x, gamma, beta, eps: float = 1e-5):
    """F.layer_norm(x, (n,), gamma, beta, eps) on a token matrix; GPU tensors with n % 4 == 0, n <= 256 take the HIP kernels (fp32
    atomics in the column sums of dgamma / dbeta: GRL_DETERMINISTIC=1 keeps torch's kernels)."""
    n = x.shape[-1]
    if x.is_cuda and x.dim() == 2 and n % 4 == 0 and n <= 256 and not ops.deterministic():
        return LayerNormFn.apply(x, gamma, beta, eps)
    return F.layer_norm(x, (n,), gamma, beta, eps)


class FanOut(torch.autograd.Function):
    """n aliases of x for n consumers; the backward pass adds their gradients in ONE launch (ops.sum_tensors) where autograd would add
    them pairwise as they arrive."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = ops.sum_tensors(gs[:4])
        for g in gs[4:]:
            out = out + g
        return out, None


def fan_out(x, n: int):
    if x.is_cuda and x.requires_grad and os.environ.get("GRL_FAN_OUT", "1") != "0":
        return FanOut.apply(x, n)
    return (x,) * n


class PadGradMask(torch.autograd.Function):
    """Identity forward (a view, no launch); the gradient is multiplied by ``mask`` (broadcast over the last dimension).  For a tensor
    whose pad columns already hold the constants its consumer wants -- the anchors -> stripe attention output used as the values of the
    reverse direction: column d is the softmax denominator over itself = 1.0, column 31 is 0 -- but must not carry gradient."""

    @staticmethod
    def forward(ctx, y, mask):
        ctx.save_for_backward(mask)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class SeMlpFn(torch.autograd.Function):
    """sigmoid(W2 relu(W1 pool + b1) + b2) on the pooled channel means [B, C] (ops.se_mlp / se_mlp_bwd, csrc/se_train.hip)."""

    @staticmethod
    def forward(ctx, pool, w1, b1, w2, b2):
        pc = pool.detach().float().contiguous()
        gate, hidden = ops.se_mlp(pc, w1, b1, w2, b2)
        ctx.save_for_backward(pc, gate, hidden, w1, w2)
        return gate

    @staticmethod
    def backward(ctx, d_gate):
        pool, gate, hidden, w1, w2 = ctx.saved_tensors
        d_pool, d_w1, d_b1, d_w2, d_b2 = ops.se_mlp_bwd(d_gate, pool, gate, hidden, w1, w2)
        return d_pool, d_w1.view_as(w1), d_b1, d_w2.view_as(w2), d_b2


class SeResidualFn(torch.autograd.Function):
    """x1 + u * gate(u) with gate = sigmoid(W2 relu(W1 mean_rows(u) + b1) + b2) per image: ChannelAttention on the CAB's conv output u
    [M, C] and the block's residual (mixed_attn_block.py:956-967, mixed_attn_block_efficient.py:548) -- pool, MLP, apply: three launches
    forward; three backward (csrc/se_train.hip).  As torch code: mean, the MLP chain, addcmul forward; three multiplies, a reduction,
    the MLP's adjoints, the mean's expand / divide and a gradient add backward."""

    @staticmethod
    def forward(ctx, x1, u, w1, b1, w2, b2, rows_per_image):
        xc, uc = _rows16(x1.detach()), _rows16(u.detach())
        pool = ops.se_colsum(uc, rows_per_image, 1.0 / rows_per_image)
        gate, hidden = ops.se_mlp(pool, w1, b1, w2, b2)
        y = ops.se_apply(uc, gate, rows_per_image, f=xc)
        ctx.save_for_backward(uc, pool, gate, hidden, w1, w2)
        ctx.rpi = rows_per_image
        return y

    @staticmethod
    def backward(ctx, dy):
        u, pool, gate, hidden, w1, w2 = ctx.saved_tensors
        rpi = ctx.rpi
        dyc = _rows16(dy)
        d_gate = ops.se_colsum(dyc, rpi, 1.0, f=u)
        d_pool, d_w1, d_b1, d_w2, d_b2 = ops.se_mlp_bwd(d_gate, pool, gate, hidden, w1, w2)
        d_u = ops.se_apply(dyc, gate, rpi, h=d_pool, k=1.0 / rpi)
        return dy, d_u, d_w1.view_as(w1), d_b1, d_w2.view_as(w2), d_b2, None


def se_residual(x1, u, w1, b1, w2, b2, rows_per_image: int):
    """x1 + u * ChannelAttention-gate(u) on token matrices [B * rows_per_image, C]; w1 [Cmid, C], w2 [C, Cmid]."""
    C_ = u.shape[1]
    if u.is_cuda and ops.se_mlp_ok(u[:1], w1) and C_ % 4 == 0 and os.environ.get("GRL_SE_KERNEL", "1") != "0":
        return SeResidualFn.apply(x1, u, w1, b1, w2, b2, rows_per_image)
    B = u.shape[0] // rows_per_image
    gate = se_gate(u.view(B, rows_per_image, C_).mean(dim=1), w1, b1, w2, b2)
    return torch.addcmul(x1.view(B, rows_per_image, C_), u.view(B, rows_per_image, C_), gate.unsqueeze(1)).view_as(x1)


def se_gate(pool, w1, b1, w2, b2):
    """The CAB's squeeze-excite gate from the pooled means (mixed_attn_block.py:956-963); w1 [Cmid, C], w2 [C, Cmid]."""
    if pool.is_cuda and ops.se_mlp_ok(pool, w1) and os.environ.get("GRL_SE_KERNEL", "1") != "0":
        return SeMlpFn.apply(pool, w1, b1, w2, b2)
    return torch.sigmoid(F.linear(F.relu(F.linear(pool, w1, b1)), w2, b2))


class CpbTableFn(torch.autograd.Function):
    """Bias tables of G AffineTransforms (ops.cpb_table / cpb_table_bwd, csrc/cpb.hip): differentiable w.r.t. the CPB-MLP weights,
    no [G, rows, 512] hidden layer in memory.  ``coords`` is a constant of the geometry."""

    @staticmethod
    def forward(ctx, coords, w1, b1, w2, rows4):
        ctx.save_for_backward(coords, w1, b1, w2)
        return ops.cpb_table(coords, w1, b1, w2, rows4)

    @staticmethod
    def backward(ctx, d_out):
        coords, w1, b1, w2 = ctx.saved_tensors
        d_w1, d_b1, d_w2 = ops.cpb_table_bwd(coords, w1, b1, w2, d_out.float().contiguous())
        return None, d_w1, d_b1, d_w2, None


def cpb_tables(coords, w1, b1, w2, idx):
    """[G, nh, rows4] kernel-domain bias tables from stacked CPB-MLP weights.  GPU: CpbTableFn; CPU tensors, head counts the kernel is
    not instantiated for and GRL_DETERMINISTIC=1 (the kernel's cross-workgroup reduction uses fp32 atomics): the torch expression
    (``idx``: the reversed-row gather with the pad entries pointing at row 0)."""
    rows = coords.shape[0]
    if w1.is_cuda and w2.shape[1] in ops.CPB_HEADS and w2.shape[2] == 512 and not ops.deterministic():
        return CpbTableFn.apply(coords, w1, b1, w2, int(idx.numel()))
    # layer 1 has K = 2: two broadcast multiply-adds instead of a GEMM;  h [G, rows, 512]
    h = F.relu(torch.addcmul(torch.addcmul(b1.unsqueeze(1), coords[:, 0].view(1, rows, 1), w1[:, :, 0].unsqueeze(1)),
                             coords[:, 1].view(1, rows, 1), w1[:, :, 1].unsqueeze(1)))
    return torch.sigmoid(torch.bmm(w2, h.transpose(1, 2))).index_select(2, idx) * (16.0 * 1.4426950408889634)


class AttentionFn:
    """Call-compatible front of torch.ops.grl.attention: ``AttentionFn.apply(q, k, v, table, geo)`` with
    geo = dict(q=(Himg, Wimg, wh, ww, shy, shx), k=(...), B, nh, d, masked, floor)."""

    @staticmethod
    def apply(q, k, v, table, geo):
        tm = bool(geo.get("token_major", False))
        if not q.is_cuda:
            o = composite.attention(q, k, v, table, list(geo["q"]), list(geo["k"]), geo["B"], geo["nh"], geo["d"], bool(geo["masked"]))
            return o.permute(1, 0, 2).reshape(o.shape[1], -1) if tm else o
        q16, k16, v16 = geo.get("f16", (None, None, None))
        return attention_op(q, k, v, table, geo["floor"], list(geo["q"]), list(geo["k"]), geo["B"], geo["nh"], geo["d"], bool(geo["masked"]),
                            bool(geo.get("prepared", False)), q16, k16, v16, tm)[0]


def linear(x, w, b=None, one_col: int = -1, gelu_in: bool = False):
    if not x.is_cuda:                      # CPU tensors: the composite torch path (composite.py; never a CUDA tensor)
        return composite.linear(F.gelu(x) if gelu_in else x, w, b)
    return linear_op(x, w, b, one_col, gelu_in)


def conv3x3(x, w, b, B, H, W):
    if not x.is_cuda:
        return composite.conv3x3(x, w, b, B, H, W)
    return conv3x3_op(x, w, b, B, H, W)
