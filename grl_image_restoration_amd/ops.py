"""Tensor-level wrappers over the C ABI (include/grl_hip.h).

These functions only validate tensors and fill the argument structs; all arithmetic happens in
the HIP kernels.  Inputs must be CUDA (ROCm) tensors -- there is no CPU path.
"""
import ctypes as C
import os
import dataclasses
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib as L

GEMM_DTYPE = torch.float16  # operand / intermediate type of the linear and conv kernels (csrc/common.h: gemm_t)
PLANE_DTYPE = torch.float16  # q / k / v / anchor head planes (attention operands)
_KIND = {torch.float32: L.DT_F32, torch.float16: L.DT_F16}


# ---- uninitialised workspace ------------------------------------------------------------------------------------------
# Every output / workspace tensor of the C-ABI wrappers (and of autograd.py / model.py) comes from here.  GRL_POISON=1 fills it with
# NaN (floating types) or 0x7f bytes first: a kernel that reads an element neither it nor a predecessor wrote turns up as a NaN
# in the parity tests instead of depending on what the allocator handed out (zeros in a fresh process, another process's leftovers
# in memory the driver recycled -- round 5's order-dependent failure of the captured training step).  The fill is an ordinary
# launch on the current stream, so it is captured into a HIP graph with the step and re-poisons the buffer on every replay.
_POISON = os.environ.get("GRL_POISON", "0") == "1"


def set_poison(flag: bool) -> bool:
    """Switch the poison mode at run time (tests); returns the previous setting."""
    global _POISON
    prev, _POISON = _POISON, bool(flag)
    return prev


def _poison(t: torch.Tensor) -> torch.Tensor:
    if _POISON and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0x7F)
    return t


def deterministic() -> bool:
    """GRL_DETERMINISTIC=1: bit-identical training gradients run to run.  The cross-workgroup reductions of the backward pass --
    the M slabs of the weight-gradient GEMM, the bias-table gradient of the attention backward -- are accumulated as 64-bit fixed
    point with integer atomics (integer addition commutes, fp32 addition does not), and the attention backward does not split
    its launches (grl_hip.h: GrlGemmTnArgs.c_fix, GrlAttnBwdArgs.d_table_fix).  Slower; read per call."""
    return os.environ.get("GRL_DETERMINISTIC", "0") == "1"


# ---- one zero fill per backward pass -----------------------------------------------------------------------------------------
# The backward wrappers need ~700 small zeroed fp32 buffers per training step -- the destinations of atomics: weight gradients (the M
# slabs of grl_gemm_tn), bias-table gradients, replicated column sums -- and a fill launch each was 2 ms of a 90-ms captured step.  They
# are cut from ONE arena that is zeroed by one launch when the backward pass starts (autograd.GradScaleTop.backward calls
# zero_arena_begin): sized by what the previous pass used; a pass that needs more falls back to torch.zeros for the rest and the next
# arena is larger.  The slices become gradients (``.grad`` of the parameters): the arena lives as long as any of them.
# GRL_ZERO_ARENA=0: torch.zeros per buffer.
_ARENA: dict = {}            # device index -> [buffer, next free element, elements requested in this pass, autograd graph-task id]
_ARENA_SIZE: dict = {}       # device index -> elements the last complete pass requested
_ARENA_ON = os.environ.get("GRL_ZERO_ARENA", "1") != "0"


def _graph_task() -> int:
    """Id of the autograd backward pass this thread is executing (-1: none)."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def zero_arena_begin(device) -> None:
    """Start of a backward pass on ``device``: one zeroed arena for its small accumulation buffers (scoped to THIS pass by the
    autograd graph-task id: a request from anywhere else -- a forward pass, a direct call, another backward -- gets torch.zeros)."""
    task = _graph_task()
    if not _ARENA_ON or device.type != "cuda" or task < 0:
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    prev = _ARENA.get(idx)
    if prev is not None:
        _ARENA_SIZE[idx] = max(prev[2], 1)
    n = _ARENA_SIZE.get(idx, 0)
    buf = torch.zeros(n + n // 8 + 1024, dtype=torch.float32, device=device) if n > 0 else None
    _ARENA[idx] = [buf, 0, 0, task]


def zeros_f32(n: int, device) -> torch.Tensor:
    """A zeroed fp32 vector of n elements (16-byte aligned): a slice of the running backward pass's arena when there is room, else
    torch.zeros."""
    if _ARENA_ON and device.type == "cuda":
        a = _ARENA.get(device.index if device.index is not None else torch.cuda.current_device())
        if a is not None and a[3] == _graph_task():
            n4 = (n + 3) // 4 * 4
            a[2] += n4
            if a[0] is not None and a[1] + n4 <= a[0].numel():
                out = a[0][a[1] : a[1] + n]
                a[1] += n4
                return out
    return torch.zeros(n, dtype=torch.float32, device=device)


def empty(*shape, dtype, device) -> torch.Tensor:
    return _poison(torch.empty(*shape, dtype=dtype, device=device))


def empty_like(t: torch.Tensor) -> torch.Tensor:
    return _poison(torch.empty_like(t))



def split3_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 [N, K] -> fp16 [N, 3K] = [hi | hi | lo] (hi = fp16(w), lo = fp16(w - hi)): the weight side of the
    split-precision operands (``a_split`` / ``x_split`` = 3; the kernels stage activations as [hi | lo | hi])."""
    w = w.detach().float()
    hi = w.to(GEMM_DTYPE)
    lo = (w - hi.float()).to(GEMM_DTYPE)
    return torch.cat([hi, hi, lo], dim=-1).contiguous()


def pack_linear_split(w3: torch.Tensor) -> torch.Tensor:
    """split3_weight output [Npad, 3K] -> `w_regs` of grl_linear_fwd (GrlLinearArgs.w_regs, csrc/linear_split.hip): per slab of
    192 output columns and compute wave the hi and the lo MFMA A fragments of the wave's 32 columns; None for shapes the
    kernel does not take (K not in 128 / 192 / 256 / 384)."""
    Npad, K3 = w3.shape
    K = K3 // 3
    if K not in (128, 192, 256, 384) or Npad % 32:
        return None
    ns = (Npad + 191) // 192
    hi = torch.zeros(ns * 192, K, dtype=GEMM_DTYPE, device=w3.device)
    lo = torch.zeros_like(hi)
    hi[:Npad], lo[:Npad] = w3[:, :K], w3[:, 2 * K :]

    def frags(w):   # [ns * 6 tiles of 32 rows][K / 16 k-steps][64 lanes][8]: lane = 32 * (column half) + row
        return w.view(ns * 6, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(ns * 6, K // 16, 64, 8)

    blob = torch.stack([frags(hi), frags(lo)], dim=1).contiguous().view(torch.uint8).reshape(-1)
    assert blob.numel() == L.lib().grl_linear_split_blob_bytes(Npad, K)
    return blob


# ---- optional per-kernel timing with HIP events on the launching stream (used by bench.py) ----
_PROFILE = None


def profile_begin():
    """Start collecting (start, stop) event pairs around every C-ABI kernel launch."""
    global _PROFILE
    _PROFILE = {}


def profiling() -> bool:
    return _PROFILE is not None


def profile_end():
    """Stop collecting; returns {kernel: [ms, ...]} (synchronises the device)."""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in (rec or {}).items()}


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.b.record()
            _PROFILE.setdefault(self.name, []).append((self.a, self.b))
        return False


def _dev_check(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "grl_image_restoration_amd: the hot path runs only on an AMD GPU (got a CPU tensor); "
                "there is no CPU fallback"
            )


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def linear(
    a: torch.Tensor,
    w: torch.Tensor,
    bias: torch.Tensor,
    *,
    epi: int = L.EPI_PLAIN,
    out_dtype=GEMM_DTYPE,
    out: Optional[torch.Tensor] = None,
    gscale: Optional[torch.Tensor] = None,
    ln_g: Optional[torch.Tensor] = None,
    ln_b: Optional[torch.Tensor] = None,
    n_real: int = 0,
    ln_eps: float = 1e-5,
    res_scale: float = 1.0,
    resid: Optional[torch.Tensor] = None,
    add2: Optional[torch.Tensor] = None,
    add2_scale: Optional[torch.Tensor] = None,
    rows_per_image: int = 0,
    pool: Optional[Tuple[int, int, int]] = None,
    M: Optional[int] = None,
    planes: bool = False,
    a_split: int = 1,
    a_scale: float = 1.0,
    out_scale: float = 1.0,
    out_lo: Optional[torch.Tensor] = None,
    w_regs: Optional[torch.Tensor] = None,
    a_cols: int = 0,
    a_one: bool = False,
    n_store: int = 0,
    a16_out: Optional[torch.Tensor] = None,
    a_gelu: bool = False,
) -> torch.Tensor:
    """out[M, Npad] = epilogue(a[M, :Kpad] @ w[Npad, Kpad]^T + bias).  ``a``: 2-D fp32/fp16, row
    stride in elements = a.stride(0); ``pool=(df, H, W)`` averages df x df token blocks first.
    ``a_cols`` > 0: ``a`` is [M, a_cols] at its real width (multiple of 4): the kernel reads the columns up to Kpad as 0 (column
    a_cols as 1.0 with ``a_one``); ``n_store`` > 0: the result is [M, n_store] (fp32, multiple of 4) -- the training path's operands
    and results without padded copies (GrlLinearArgs, ABI 22).  ``a16_out`` [M, >= Kpad] fp16: receives the operand as the kernel contracts
    it (a_scale * a, pad / ones columns included) for the weight-gradient GEMM of the same layer.
    ``a_split=3``: split-precision operands -- ``w`` is packed by ``split3_weight`` ([hi | hi | lo], Kpad = 3 x the
    source width) and the kernel stages the fp32 ``a`` as [hi | lo | hi]."""
    _dev_check(a, w, bias, out, gscale, ln_g, ln_b, resid, add2, add2_scale, w_regs)
    if w_regs is not None:   # register image of the same weights for the weights-stationary split kernel (pack_linear_split)
        assert a_split == 3 and w_regs.dtype == torch.uint8 and w_regs.is_contiguous()
        assert w_regs.numel() == L.lib().grl_linear_split_blob_bytes(w.shape[0], w.shape[1] // 3)
    if add2 is not None:
        assert (add2.dtype == GEMM_DTYPE or (w_regs is not None and add2.dtype == torch.float32)) and add2_scale is not None and rows_per_image > 0
        assert add2_scale.dtype == torch.float32 and add2_scale.is_contiguous()
    assert a.dim() == 2 and a.stride(1) == 1 and w.dim() == 2 and w.is_contiguous() and w.dtype == GEMM_DTYPE
    assert a.dtype in (torch.float32, GEMM_DTYPE) and bias.dtype == torch.float32
    Npad, Kpad = w.shape
    assert a_split in (1, 3) and a.shape[1] >= (a_cols if a_cols > 0 else Kpad // a_split) and bias.numel() == Npad
    assert a_cols == 0 or (a.dtype == torch.float32 and pool is None and a_split == 1 and a_cols % 4 == 0 and a.stride(0) % 4 == 0)
    assert n_store == 0 or (n_store % 4 == 0 and n_store <= Npad and not planes and epi in (L.EPI_PLAIN, L.EPI_GELU, L.EPI_GELU_GRAD))
    assert not a_gelu or (a.dtype == torch.float32 and pool is None and a_split == 1)      # operand = gelu(a), taken in the loader
    assert a_split == 1 or (a.dtype == torch.float32 and Kpad % 96 == 0)
    if pool is not None:
        df, H, W = pool
        assert a.dtype == torch.float32 and a.shape[0] % (H * W) == 0
        rows = a.shape[0] // (df * df)
    else:
        df, H, W = 1, 0, 0
        rows = a.shape[0]
    M = rows if M is None else M
    if planes:
        # head-plane layout [Npad/32, M, 32] (fp16): what the attention kernel stages fastest
        if out is None:
            out = empty(Npad // 32, M, 32, dtype=PLANE_DTYPE, device=a.device)
        assert out.dtype == PLANE_DTYPE and out.is_contiguous() and out.shape == (Npad // 32, M, 32)
        ldo, plane_stride = 32, M * 32
    else:
        if out is None:
            out = empty(M, n_store if n_store > 0 else Npad, dtype=out_dtype, device=a.device)
        assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= M and out.shape[1] >= (n_store if n_store > 0 else Npad)
        assert n_store == 0 or out.dtype == torch.float32
        ldo, plane_stride = out.stride(0), 0
    args = L.GrlLinearArgs(
        a=_ptr(a), a_dtype=_KIND[a.dtype], lda=a.stride(0),
        pool_df=df, pool_H=H, pool_W=W,
        w=_ptr(w), bias=_ptr(bias), M=M, Npad=Npad, Kpad=Kpad, epi=epi,
        gscale=_ptr(gscale), ln_g=_ptr(ln_g), ln_b=_ptr(ln_b), n_real=n_real, ln_eps=ln_eps,
        res_scale=res_scale, resid=_ptr(resid), ldr=resid.stride(0) if resid is not None else 0,
        add2=_ptr(add2), add2_dtype=_KIND[add2.dtype] if add2 is not None else 0,
        ldadd2=add2.stride(0) if add2 is not None else 0,
        add2_scale=_ptr(add2_scale), rows_per_image=rows_per_image, a_split=a_split, a_scale=a_scale, out_scale=out_scale,
        out_lo=_ptr(out_lo), out=_ptr(out), out_dtype=_KIND[out.dtype], ldo=ldo, out_plane_stride=plane_stride,
        w_regs=_ptr(w_regs), a_cols=a_cols, a_one=int(a_one), n_store=n_store,
        a16_out=_ptr(a16_out), lda16=a16_out.stride(0) if a16_out is not None else 0, a_gelu=int(a_gelu),
    )
    assert a16_out is None or (a16_out.dtype == GEMM_DTYPE and a16_out.dim() == 2 and a16_out.stride(1) == 1 and a16_out.shape[0] >= M
                               and a16_out.shape[1] >= Kpad and a16_out.stride(0) % 8 == 0 and a.dtype == torch.float32)
    if out_lo is not None:   # rounding residuals of the fp16 outputs, same layout (split-precision attention operands)
        assert out_lo.dtype == torch.float16 and out_lo.shape == out.shape and out_lo.stride() == out.stride() and out.dtype == torch.float16
    if epi == L.EPI_GROUPNORM:
        assert gscale is not None and gscale.dtype == torch.float32 and gscale.numel() == Npad // 32
    if epi == L.EPI_LN_RES:
        assert resid is not None and resid.dtype == torch.float32 and ln_g.numel() == Npad and ln_b.numel() == Npad
        assert out.dtype == torch.float32
    if epi == L.EPI_GELU_GRAD:   # out = product * gelu'(resid): resid [M, >= n_store or Npad] fp32
        assert resid is not None and resid.dtype == torch.float32 and resid.stride(1) == 1 and resid.stride(0) % 4 == 0 and out.dtype == torch.float32
        assert resid.shape[0] >= M and resid.shape[1] >= (n_store if n_store > 0 else Npad)
    with _timed(f"linear {Kpad}->{Npad}" if _PROFILE is not None else "linear"):
        L.check(L.lib().grl_linear_fwd(L.stream_ptr(), C.byref(args)), "grl_linear_fwd")
    return out


def pack_mlp(fc1_w: torch.Tensor, fc1_b: torch.Tensor, fc2_w: torch.Tensor, Cpad: int, Hpad: int) -> torch.Tensor:
    """Weight chunk stream of grl_mlp_fwd (layout in include/grl_hip.h): per 32 hidden channels one LDS image
    W1 rows | W2 columns | fc1 bias, rows padded by 16 B, k-slots in the kernel's order, chunk padded to 1 KiB.
    fc1_w: (Hd, C), fc2_w: (C, Hd) -- swin_v1_block.py:29-33.  Returns a uint8 device tensor."""
    dev = fc1_w.device
    Hd, Cin = fc1_w.shape
    assert fc2_w.shape == (Cin, Hd) and Cpad % 32 == 0 and Hpad % 32 == 0 and Cpad >= Cin and Hpad >= Hd
    W1 = torch.zeros(Hpad, Cpad, dtype=torch.float32, device=dev)
    W1[:Hd, :Cin] = fc1_w.detach().float()
    W2 = torch.zeros(Cpad, Hpad, dtype=torch.float32, device=dev)
    W2[:Cin, :Hd] = fc2_w.detach().float()
    b1 = torch.zeros(Hpad, dtype=torch.float32, device=dev)
    b1[:Hd] = fc1_b.detach().float()
    nch = Hpad // 32
    # k-slot sigma = 8g + e of a 32-group  ->  channel 4g + e (e < 4) | 16 + 4g + (e - 4): the MFMA accumulator
    # fragment of one layer is the operand fragment of the next (csrc/mlp.hip)
    sig = torch.arange(32, device=dev)
    g, e = sig // 8, sig % 8
    chan = torch.where(e < 4, 4 * g + e, 16 + 4 * g + (e - 4))
    w1c = W1.view(nch, 32, Cpad // 32, 32)[..., chan].reshape(nch, 32, Cpad).to(GEMM_DTYPE)        # [chunk][32 rows][Cpad slots]
    w2c = W2.view(Cpad, nch, 32)[:, :, chan].permute(1, 0, 2).contiguous().to(GEMM_DTYPE)          # [chunk][Cpad rows][32 slots]
    total = L.lib().grl_mlp_blob_bytes(Cpad, Hpad)
    assert total > 0 and total % nch == 0
    blob = torch.zeros(nch, total // nch, dtype=torch.uint8, device=dev)
    w1row, w2row = Cpad * 2 + 16, 80
    w1b, w2b = 32 * w1row, Cpad * w2row
    blob[:, :w1b].view(nch, 32, w1row)[:, :, : Cpad * 2] = w1c.contiguous().view(torch.uint8).view(nch, 32, Cpad * 2)
    blob[:, w1b : w1b + w2b].view(nch, Cpad, w2row)[:, :, :64] = w2c.view(torch.uint8).view(nch, Cpad, 64)
    blob[:, w1b + w2b : w1b + w2b + 128] = b1.view(nch, 32).contiguous().view(torch.uint8).view(nch, 128)
    return blob.contiguous()


def _kslot_perm(dev):
    """k-slot sigma = 8g + e of a 32-group -> element 4g + e (e < 4) | 16 + 4g + (e - 4)  (csrc/mlp.hip, csrc/qkv.hip)."""
    sig = torch.arange(32, device=dev)
    g, e = sig // 8, sig % 8
    return torch.where(e < 4, 4 * g + e, 16 + 4 * g + (e - 4))


def pack_qkv(w: torch.Tensor, bias: torch.Tensor, gscale: torch.Tensor) -> torch.Tensor:
    """Weight stream of grl_qkv_fwd from the slotted QKV matrix w [nslots*32, Cpad] (any float dtype), bias
    [nslots*32] and gscale [nslots] (layout in include/grl_hip.h).  Returns a uint8 device tensor."""
    dev = w.device
    N, Cpad = w.shape
    nslots = N // 32
    assert N % 32 == 0 and Cpad % 32 == 0 and bias.numel() == N and gscale.numel() == nslots
    spc = 2 if nslots % 2 == 0 else 1
    total = L.lib().grl_qkv_blob_bytes(Cpad, nslots)
    assert total > 0
    nch = nslots // spc
    wrow = Cpad * 2 + 16
    slot_b = 32 * wrow + 128 + 16
    chan = _kslot_perm(dev)
    wp = w.detach().float().view(nslots, 32, Cpad // 32, 32)[..., chan].reshape(nslots, 32, Cpad).to(GEMM_DTYPE).contiguous()
    img = torch.zeros(nslots, slot_b, dtype=torch.uint8, device=dev)
    img[:, : 32 * wrow].view(nslots, 32, wrow)[:, :, : Cpad * 2] = wp.view(torch.uint8).view(nslots, 32, Cpad * 2)
    img[:, 32 * wrow : 32 * wrow + 128] = bias.detach().float().view(nslots, 32).contiguous().view(torch.uint8).view(nslots, 128)
    img[:, 32 * wrow + 128 : 32 * wrow + 132] = gscale.detach().float().view(nslots, 1).contiguous().view(torch.uint8).view(nslots, 4)
    blob = torch.zeros(nch, total // nch, dtype=torch.uint8, device=dev)
    blob[:, : spc * slot_b] = img.view(nch, spc * slot_b)
    return blob.contiguous()


def qkv(x: torch.Tensor, blob: torch.Tensor, nslots: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Head planes [nslots, M, 32] (fp16) of the slotted, normalised QKV projection of x [M, Cpad] (fp32)."""
    _dev_check(x, blob, out)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and blob.dtype == torch.uint8 and blob.is_contiguous()
    M, Cpad = x.shape
    if out is None:
        out = empty(nslots, M, 32, dtype=PLANE_DTYPE, device=x.device)
    assert out.dtype == PLANE_DTYPE and out.is_contiguous() and out.shape == (nslots, M, 32)
    args = L.GrlQkvArgs(x=_ptr(x), ldx=x.stride(0), blob=_ptr(blob), M=M, Cpad=Cpad, nslots=nslots, out=_ptr(out),
                        out_plane_stride=M * 32)
    with _timed("qkv"):
        L.check(L.lib().grl_qkv_fwd(L.stream_ptr(), C.byref(args)), "grl_qkv_fwd")
    return out


def pack_qkv_anchor(w: torch.Tensor, bias: torch.Tensor, gscale: torch.Tensor, wa: Optional[torch.Tensor] = None,
                    ba: Optional[torch.Tensor] = None, ga: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight stream of grl_qkv_anchor_fwd: the slotted QKV matrix w [nslots*32, Cpad] followed by the slotted anchor matrix
    wa [nanc*32, Cpad] (bias / gscale likewise), K in natural order (layout in include/grl_hip.h).  uint8 device tensor."""
    dev = w.device
    if wa is not None:
        w, bias, gscale = torch.cat([w.float(), wa.float()]), torch.cat([bias.float(), ba.float()]), torch.cat([gscale.float(), ga.float()])
    N, Cpad = w.shape
    ns = N // 32
    nanc = 0 if wa is None else wa.shape[0] // 32
    assert N % 32 == 0 and Cpad % 32 == 0 and bias.numel() == N and gscale.numel() == ns
    total = L.lib().grl_qkv_anchor_blob_bytes(Cpad, ns - nanc, nanc)
    assert total > 0
    nch = (ns + 1) // 2
    wrow = Cpad * 2 + 16
    slot_b = 32 * wrow + 128 + 16
    img = torch.zeros(2 * nch, slot_b, dtype=torch.uint8, device=dev)
    wp = w.detach().float().to(GEMM_DTYPE).contiguous()
    img[:ns, : 32 * wrow].view(ns, 32, wrow)[:, :, : Cpad * 2] = wp.view(torch.uint8).view(ns, 32, Cpad * 2)
    img[:ns, 32 * wrow : 32 * wrow + 128] = bias.detach().float().view(ns, 32).contiguous().view(torch.uint8).view(ns, 128)
    img[:ns, 32 * wrow + 128 : 32 * wrow + 132] = gscale.detach().float().view(ns, 1).contiguous().view(torch.uint8).view(ns, 4)
    blob = torch.zeros(nch, total // nch, dtype=torch.uint8, device=dev)
    blob[:, : 2 * slot_b] = img.view(nch, 2 * slot_b)
    return blob.contiguous()


def pack_qkv_anchor_lo(w: torch.Tensor, gscale: torch.Tensor) -> torch.Tensor:
    """`lo_blob` of grl_qkv_anchor_fwd's split-precision variant (layout in include/grl_hip.h) from the slotted weight matrix
    w [(nslots + nanc)*32, Cpad] (fp32, q/k/v slots then anchor slots) and the slots' gscale: for every normalised slot
    (gscale != 0) the fp16 rounding error of its weights, scaled by 2^(e+4), as e4m3.  The kernel also multiplies W_hi by 2^e
    (its split slots accumulate at that scale), so e <= 14 is lowered until 2^e max|W| stays inside fp16 and the scaled
    rounding errors inside e4m3."""
    dev = w.device
    N, Cpad = w.shape
    ns = N // 32
    assert N % 32 == 0 and gscale.numel() == ns
    w = w.detach().float()
    lo = (w - w.to(GEMM_DTYPE).float()).view(ns, 32, Cpad)[gscale.detach().float().cpu() != 0]
    nsplit = lo.shape[0]
    total = L.lib().grl_qkv_anchor_lo_blob_bytes(Cpad, nsplit)
    assert total > 0, "split-precision QKV: GRL-Base slot layout only"
    amax = float(lo.abs().max())
    wmax = float(w.view(ns, 32, Cpad)[gscale.detach().float().cpu() != 0].abs().max())
    e = 14   # as large as both ranges allow: the scaled rounding errors should sit in e4m3's normal range (3 significant bits)
    while e > -8 and (amax * 2.0 ** (e + 4) > 448.0 or wmax * 2.0 ** e > 30000.0):
        e -= 1
    lo8 = (lo * 2.0 ** (e + 4)).clamp_(-448.0, 448.0).to(torch.float8_e4m3fn)
    blob = torch.zeros(total, dtype=torch.uint8, device=dev)
    blob[:8] = torch.tensor([2.0 ** -e, 2.0 ** e], dtype=torch.float32).view(torch.uint8).to(dev)
    # row image: the 64-channel block c as [lane half h][k-step u of the block][8 channels] (k = 64 c + 16 u + 8 h + t): the 32
    # bytes a lane half feeds to one 32x32x64 fp8 MFMA are contiguous (csrc/qkv_anchor.hip, mfma64_fp8)
    img = lo8.view(torch.uint8).view(nsplit, 32, Cpad // 64, 4, 2, 8).permute(0, 1, 2, 4, 3, 5).reshape(nsplit, 32, Cpad // 16, 16)
    # ... and the 16-byte segment s of row j at position s ^ ((j >> 2) & 3): the rows carry no pad (LDS budget), the swizzle
    # keeps the kernel's 16-byte reads of 16 consecutive rows on distinct banks
    seg = torch.arange(Cpad // 16, device=dev).view(1, -1) ^ ((torch.arange(32, device=dev) >> 2) & 3).view(-1, 1)   # [row][physical] -> logical
    img = torch.gather(img, 2, seg.view(1, 32, Cpad // 16, 1).expand(nsplit, 32, Cpad // 16, 16))
    blob[16:].view(nsplit, 32, Cpad)[:] = img.reshape(nsplit, 32, Cpad)
    return blob


def qkv_anchor(x: torch.Tensor, blob: torch.Tensor, nslots: int, nanc: int, B: int, H: int, W: int, lo_blob: Optional[torch.Tensor] = None):
    """(q/k/v head planes [nslots, M, 32], anchor head planes [nanc, M/4, 32]) (fp16) of the token matrix x [B*H*W, Cpad]
    (fp32): slotted, normalised QKV projection + 2x2-pooled anchor projection in one pass (grl_qkv_anchor_fwd).  With `lo_blob`
    (pack_qkv_anchor_lo) the normalised slots are computed on split operands."""
    _dev_check(x, blob, lo_blob)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and blob.dtype == torch.uint8 and blob.is_contiguous()
    M, Cpad = x.shape
    assert M == B * H * W and H % 2 == 0 and W % 64 == 0
    out = empty(nslots, M, 32, dtype=PLANE_DTYPE, device=x.device)
    anc = empty(max(nanc, 1), M // 4, 32, dtype=PLANE_DTYPE, device=x.device)
    args = L.GrlQkvAnchorArgs(x=_ptr(x), ldx=x.stride(0), B=B, H=H, W=W, Cpad=Cpad, blob=_ptr(blob), nslots=nslots, nanc=nanc,
                              out=_ptr(out), out_plane_stride=M * 32, anc=_ptr(anc), anc_plane_stride=(M // 4) * 32, lo_blob=_ptr(lo_blob))
    with _timed("qkv_anchor"):
        L.check(L.lib().grl_qkv_anchor_fwd(L.stream_ptr(), C.byref(args)), "grl_qkv_anchor_fwd")
    return out, (anc if nanc > 0 else None)


def pack_proj(w: torch.Tensor) -> torch.Tensor:
    """Projection weight stream of grl_block_tail_fwd from the padded matrix w [Cpad, Cpad] (rows = output channels,
    columns = the slotted attention-output K): chunks of 32 rows x (2*Cpad + 16) bytes fp16, padded to 1 KiB."""
    dev = w.device
    N, K = w.shape
    assert N == K and N % 32 == 0
    total = L.lib().grl_proj_blob_bytes(N)
    assert total > 0
    nch = N // 32
    wrow = K * 2 + 16
    blob = torch.zeros(nch, total // nch, dtype=torch.uint8, device=dev)
    blob[:, : 32 * wrow].view(nch, 32, wrow)[:, :, : K * 2] = w.detach().to(GEMM_DTYPE).contiguous().view(torch.uint8).view(nch, 32, K * 2)
    return blob.contiguous()


def pack_tail_regs(proj_w: torch.Tensor, fc1_w: torch.Tensor, fc1_b: torch.Tensor, fc2_w: torch.Tensor) -> torch.Tensor:
    """`rblob` of grl_block_tail_fwd's register-resident kernel (layout in include/grl_hip.h; GRL-Base shape): the MFMA A
    fragments of proj_w [192, 192] (rows = output channels, columns = the slotted attention output), fc1_w [Hd <= 384, C <= 192]
    and fc2_w [C, Hd] (swin_v1_block.py:29-33), padded with zeros, wave by wave, followed by the padded fc1 bias."""
    dev = proj_w.device
    CP, HP = 192, 384
    Hd, Cin = fc1_w.shape
    assert proj_w.shape == (CP, CP) and fc2_w.shape == (Cin, Hd) and Cin <= CP and Hd <= HP and fc1_b.numel() == Hd
    W1 = torch.zeros(HP, CP, dtype=torch.float32, device=dev)
    W1[:Hd, :Cin] = fc1_w.detach().float()
    W2 = torch.zeros(CP, HP, dtype=torch.float32, device=dev)
    W2[:Cin, :Hd] = fc2_w.detach().float()

    def frags(w):   # [rows, K] -> [rows / 32 tiles][K / 16 k-steps][64 lanes][8]: lane = 32 * (column half) + row
        R, K = w.shape
        return w.to(GEMM_DTYPE).view(R // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(R // 32, K // 16, 64, 8)

    fp, f1, f2 = frags(proj_w.detach().float()), frags(W1), frags(W2)      # [6][12], [12][12], [6][24]
    img = torch.zeros(8, 48, 64, 8, dtype=GEMM_DTYPE, device=dev)
    for w in range(6):
        img[w, 0:12], img[w, 12:24], img[w, 24:48] = fp[w], f1[w], f2[w]
    for w in (6, 7):
        for t in range(3):
            img[w, 12 * t : 12 * t + 12] = f1[6 + 3 * (w - 6) + t]
    total = L.lib().grl_tail_regs_blob_bytes()
    blob = torch.zeros(total, dtype=torch.uint8, device=dev)
    blob[: 8 * 48 * 1024] = img.view(torch.uint8).reshape(-1)
    b1 = torch.zeros(HP, dtype=torch.float32, device=dev)
    b1[:Hd] = fc1_b.detach().float()
    blob[8 * 48 * 1024 :] = b1.view(torch.uint8)
    return blob


def block_tail(att: torch.Tensor, x: torch.Tensor, cab: torch.Tensor, gate: torch.Tensor, rows_per_image: int,
               pblob: torch.Tensor, pb: torch.Tensor, n1_g: torch.Tensor, n1_b: torch.Tensor,
               blob: torch.Tensor, b2: torch.Tensor, n2_g: torch.Tensor, n2_b: torch.Tensor, *, Hpad: int, n_real: int,
               ln_eps: float = 1e-5, res_scale: float = 1.0, out: Optional[torch.Tensor] = None, rblob: Optional[torch.Tensor] = None) -> torch.Tensor:
    """proj + norm1 + residual + gated CAB + MLP + norm2 + residual of a block in one kernel (grl_block_tail_fwd); with `rblob`
    (pack_tail_regs) the register-resident kernel where the shape qualifies."""
    _dev_check(att, x, cab, gate, pblob, pb, n1_g, n1_b, blob, b2, n2_g, n2_b, out, rblob)
    M, Cpad = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1 and att.dtype == GEMM_DTYPE and att.stride(1) == 1 and att.shape[0] == M
    assert cab.dtype == GEMM_DTYPE and cab.stride(1) == 1 and cab.shape[0] == M and gate.dtype == torch.float32 and gate.is_contiguous()
    assert gate.shape[1] == Cpad and att.shape[1] >= Cpad and cab.shape[1] >= Cpad
    if out is None:
        out = empty(M, Cpad, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.data_ptr() != x.data_ptr()
    args = L.GrlTailArgs(att=_ptr(att), ldatt=att.stride(0), x=_ptr(x), ldx=x.stride(0), cab=_ptr(cab), ldcab=cab.stride(0),
                         gate=_ptr(gate), rows_per_image=rows_per_image, pblob=_ptr(pblob), pb=_ptr(pb), n1_g=_ptr(n1_g),
                         n1_b=_ptr(n1_b), blob=_ptr(blob), M=M, Cpad=Cpad, Hpad=Hpad, b2=_ptr(b2), n2_g=_ptr(n2_g), n2_b=_ptr(n2_b),
                         n_real=n_real, ln_eps=ln_eps, res_scale=res_scale, out=_ptr(out), ldo=out.stride(0), rblob=_ptr(rblob))
    with _timed("block_tail"):
        L.check(L.lib().grl_block_tail_fwd(L.stream_ptr(), C.byref(args)), "grl_block_tail_fwd")
    return out


def mlp(x: torch.Tensor, blob: torch.Tensor, b2: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, *, Hpad: int,
        n_real: int, ln_eps: float = 1e-5, res_scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x + res_scale * LayerNorm(fc2(GELU(fc1(x)))) on the token matrix x [M, Cpad] (fp32), one kernel."""
    _dev_check(x, blob, b2, ln_g, ln_b, out)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    M, Cpad = x.shape
    assert blob.dtype == torch.uint8 and blob.is_contiguous() and b2.numel() == Cpad and ln_g.numel() == Cpad and ln_b.numel() == Cpad
    if out is None:
        out = empty(M, Cpad, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape == (M, Cpad) and out.data_ptr() != x.data_ptr()
    args = L.GrlMlpArgs(x=_ptr(x), ldx=x.stride(0), blob=_ptr(blob), M=M, Cpad=Cpad, Hpad=Hpad, b2=_ptr(b2), ln_g=_ptr(ln_g),
                        ln_b=_ptr(ln_b), n_real=n_real, ln_eps=ln_eps, res_scale=res_scale, out=_ptr(out), ldo=out.stride(0))
    with _timed("mlp"):
        L.check(L.lib().grl_mlp_fwd(L.stream_ptr(), C.byref(args)), "grl_mlp_fwd")
    return out


@dataclass
class TokenGrid:
    """An fp16 token tensor (the output grid may be fp32) viewed as windows: mirrors GrlTokenGrid.

    ``t`` is either a token-major matrix [tokens, heads*32 (+...)] (``slot`` = first 32-wide column
    group of head 0) or a stack of head planes [slots, tokens, 32] (``slot`` = plane of head 0)."""

    t: torch.Tensor
    slot: int
    Himg: int
    Wimg: int
    wh: int
    ww: int
    shy: int = 0
    shx: int = 0
    transposed: bool = False   # the grid is the transposed view of a (Wimg x Himg) row-major image (GrlTokenGrid.transposed)

    def T(self) -> "TokenGrid":
        """The same tokens seen through the transposed image: attention commutes with it (bias table transposed by the caller)."""
        return TokenGrid(self.t, self.slot, self.Wimg, self.Himg, self.ww, self.wh, self.shx, self.shy, not self.transposed)

    def c(self) -> L.GrlTokenGrid:
        t = self.t
        assert t.dtype in (torch.float16, torch.float32) and t.stride(-1) == 1
        if t.dim() == 3:  # head planes
            assert t.shape[2] == 32 and t.is_contiguous()
            return L.GrlTokenGrid(ptr=C.c_void_p(t.data_ptr() + self.slot * t.stride(0) * t.element_size()), ld=32, hstride=t.stride(0),
                                  col0=0, Himg=self.Himg, Wimg=self.Wimg, wh=self.wh, ww=self.ww, shy=self.shy, shx=self.shx,
                                  transposed=int(self.transposed))
        assert t.dim() == 2
        return L.GrlTokenGrid(ptr=_ptr(t), ld=t.stride(0), hstride=32, col0=self.slot * 32, Himg=self.Himg, Wimg=self.Wimg,
                              wh=self.wh, ww=self.ww, shy=self.shy, shx=self.shx, transposed=int(self.transposed))

    @property
    def tokens(self) -> int:
        return self.t.shape[-2] if self.t.dim() == 3 else self.t.shape[0]


def attention_rows_ok(q_win, k_win, q_shift, k_shift, masked: bool, head_dim: int) -> bool:
    """Would grl_attention_fwd serve this (query window, key window) geometry with the row-streaming kernel
    (grl_attention_rows_geometry_ok)?  Plan-time question: a geometry that is not 32-aligned may be so transposed."""
    def grid(win, sh):
        return L.GrlTokenGrid(ptr=None, ld=32, hstride=32, col0=0, Himg=win[0], Wimg=win[1], wh=win[0], ww=win[1], shy=sh[0], shx=sh[1], transposed=0)
    gq, gk = grid(q_win, q_shift), grid(k_win, k_shift)
    args = L.GrlAttnArgs(q=gq, k=gk, v=gk, o=gq, B=1, nh=1, nwy=1, nwx=1, table=None,
                         trows=(q_win[0] + k_win[0] - 1) * (q_win[1] + k_win[1] - 1), tstride=0, masked=int(masked),
                         ones_col=head_dim if head_dim < 32 else -1, head_dim=head_dim, out_dtype=L.DT_F16, k_one31=0)
    return bool(L.lib().grl_attention_rows_geometry_ok(C.byref(args)))


def transpose_table(bias: torch.Tensor, q_win, k_win) -> torch.Tensor:
    """(rows, nh) relative-position bias of a (query window, key window) pair -> the rows of the transposed pair:
    row (dy, dx) of the (q_wh + k_wh - 1) x (q_ww + k_ww - 1) offset grid becomes row (dx, dy)."""
    Dy, Dx = q_win[0] + k_win[0] - 1, q_win[1] + k_win[1] - 1
    assert bias.shape[0] == Dy * Dx
    return bias.view(Dy, Dx, -1).transpose(0, 1).reshape(Dy * Dx, -1).contiguous()


def attention(q: TokenGrid, k: TokenGrid, v: TokenGrid, o: TokenGrid, *, B: int, nh: int, table: torch.Tensor,
              masked: bool, ones_col: int, head_dim: int, k_one31: bool = False,
              lazy_floor: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
              q_lo: Optional[torch.Tensor] = None, k_lo: Optional[torch.Tensor] = None, v_lo: Optional[torch.Tensor] = None,
              o_lo: Optional[torch.Tensor] = None, lazy_ceil: Optional[torch.Tensor] = None):
    """softmax(q k^T + bias(+mask)) v over every window of every image; see grl_attention_fwd.
    ``q_lo / k_lo / v_lo``: fp16 rounding residuals of q / k / v (same tensors' layout; ``TokenGrid.t`` twins) for the
    split-precision mode, ``o_lo``: residual output twin of ``o.t``.
    ``table``: (nh, rows padded to 4) from tables.kernel_table (reversed rows); ``k_one31`` / ``lazy_floor``: the
    softmax-offset contract of include/grl_hip.h; ``lse`` (nh, tokens) fp32 optional output."""
    _dev_check(q.t, k.t, v.t, o.t, table, lazy_floor, lse)
    assert table.dtype == torch.float32 and table.is_contiguous() and table.dim() == 2 and table.shape[0] == nh
    assert q.t.dtype == PLANE_DTYPE and k.t.dtype == PLANE_DTYPE and v.t.dtype == PLANE_DTYPE
    nwy, nwx = q.Himg // q.wh, q.Wimg // q.ww
    assert q.tokens >= B * q.Himg * q.Wimg and k.tokens >= B * k.Himg * k.Wimg
    if lazy_floor is not None:
        assert lazy_floor.dtype == torch.float32 and lazy_floor.is_contiguous() and lazy_floor.numel() == nh
    if lazy_ceil is not None:
        assert lazy_ceil.dtype == torch.float32 and lazy_ceil.is_contiguous() and lazy_ceil.numel() == nh and lazy_ceil.device == table.device
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.shape[0] == nh and lse.shape[1] >= q.tokens
    def twin(g: TokenGrid, t: Optional[torch.Tensor]):
        if t is None:
            return C.c_void_p(0)
        assert t.dtype == torch.float16 and t.shape == g.t.shape and t.stride() == g.t.stride()
        return C.c_void_p(dataclasses.replace(g, t=t).c().ptr)

    args = L.GrlAttnArgs(q=q.c(), k=k.c(), v=v.c(), o=o.c(), B=B, nh=nh, nwy=nwy, nwx=nwx, table=_ptr(table),
                         trows=(q.wh + k.wh - 1) * (q.ww + k.ww - 1), tstride=table.shape[1], masked=int(masked), ones_col=ones_col,
                         head_dim=head_dim, out_dtype=_KIND[o.t.dtype], k_one31=int(k_one31), lazy_floor=_ptr(lazy_floor),
                         lse=_ptr(lse), lse_stride=lse.stride(0) if lse is not None else 0,
                         q_lo=twin(q, q_lo), k_lo=twin(k, k_lo), v_lo=twin(v, v_lo), o_lo=twin(o, o_lo), lazy_ceil=_ptr(lazy_ceil))
    with _timed(f"attention q{q.wh}x{q.ww} k{k.wh}x{k.ww}" if _PROFILE is not None else "attention"):
        L.check(L.lib().grl_attention_fwd(L.stream_ptr(), C.byref(args)), "grl_attention_fwd")
    return o.t


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, n_real: int, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev_check(x, gamma, beta, out)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    if out is None:
        out = empty_like(x)
    L.check(
        L.lib().grl_layernorm_fwd(L.stream_ptr(), _ptr(x), x.stride(0), _ptr(out), out.stride(0), _ptr(gamma),
                                  _ptr(beta), x.shape[0], n_real, x.shape[1], eps),
        "grl_layernorm_fwd",
    )
    return out


def layernorm_res(x: torch.Tensor, resid: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, n_real: int, *,
                  eps: float = 1e-5, res_scale: float = 1.0, add2: Optional[torch.Tensor] = None,
                  add2_scale: Optional[torch.Tensor] = None, rows_per_image: int = 0) -> torch.Tensor:
    """resid + res_scale * LayerNorm(x) (+ add2 * add2_scale[image]) on fp32 token matrices (grl_layernorm_res_fwd)."""
    _dev_check(x, resid, gamma, beta, add2, add2_scale)
    assert x.dim() == 2 and x.dtype == torch.float32 and resid.dtype == torch.float32 and x.stride(1) == 1 and resid.stride(1) == 1
    n_pad = gamma.numel()
    assert x.shape[1] >= n_pad and resid.shape[1] >= n_pad and beta.numel() == n_pad
    out = empty(x.shape[0], n_pad, dtype=torch.float32, device=x.device)
    args = L.GrlLnResArgs(x=_ptr(x), ldx=x.stride(0), resid=_ptr(resid), ldr=resid.stride(0), gamma=_ptr(gamma), beta=_ptr(beta),
                          add2=_ptr(add2), add2_dtype=_KIND[add2.dtype] if add2 is not None else 0,
                          ldadd2=add2.stride(0) if add2 is not None else 0, add2_scale=_ptr(add2_scale),
                          rows_per_image=rows_per_image, M=x.shape[0], n_real=n_real, n_pad=n_pad, eps=eps, res_scale=res_scale,
                          y=_ptr(out), ldy=out.stride(0))
    with _timed("layernorm_res"):
        L.check(L.lib().grl_layernorm_res_fwd(L.stream_ptr(), C.byref(args)), "grl_layernorm_res_fwd")
    return out


def pack_conv_weight(w: torch.Tensor, cin_pad: int, cout_pad: int, shuffle_r: int = 0, shuffle_cg: int = 0,
                     split: int = 1) -> torch.Tensor:
    """torch conv weight [Cout, Cin, 3, 3] -> fp16 [9, cout_pad, cin_pad] (tap = ky*3+kx, K contiguous).
    With ``shuffle_r`` the output channels are re-ordered from PixelShuffle's (c, i, j) to (i, j, c) with
    ``shuffle_cg`` (>= c, multiple of 4) slots per sub-pixel so the kernel can store whole channel groups.
    ``split=3``: [9, cout_pad, 3*cin_pad] = [hi | hi | lo] along K (split-precision operands, ``x_split=3``);
    ``split=2``: [9, cout_pad, 2*cin_pad] = [hi | hi] (``x_split=2``: only the activations are split -- for sites where the
    rounding of x matters and that of W does not)."""
    cout, cin = w.shape[:2]
    w9 = w.detach().float().permute(2, 3, 0, 1).reshape(9, cout, cin)
    out = torch.zeros(9, cout_pad, cin_pad, dtype=torch.float32, device=w.device)
    if shuffle_r > 1:
        r2 = shuffle_r * shuffle_r
        c = cout // r2
        src = w9.view(9, c, r2, cin).permute(0, 2, 1, 3)  # [9, ij, c, cin]
        out.view(9, cout_pad // shuffle_cg, shuffle_cg, cin_pad)[:, :r2, :c, :cin] = src
    else:
        out[:, :cout, :cin] = w9
    if split == 3:
        return split3_weight(out)
    if split == 2:
        hi = out.to(GEMM_DTYPE)
        return torch.cat([hi, hi], dim=-1).contiguous()
    return out.to(GEMM_DTYPE).contiguous()


def pack_conv_bias(b: torch.Tensor, cout_pad: int, shuffle_r: int = 0, shuffle_cg: int = 0) -> torch.Tensor:
    out = torch.zeros(cout_pad, dtype=torch.float32, device=b.device)
    if shuffle_r > 1:
        r2 = shuffle_r * shuffle_r
        c = b.numel() // r2
        out.view(cout_pad // shuffle_cg, shuffle_cg)[:r2, :c] = b.detach().float().view(c, r2).t()
    else:
        out[: b.numel()] = b.detach().float()
    return out


def conv3x3(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, B: int, H: int, W: int, *, act: int = 0,
            slope: float = 0.0, resid: Optional[torch.Tensor] = None, want_pool: bool = False,
            out_dtype=torch.float32, out: Optional[torch.Tensor] = None, shuffle_r: int = 0, shuffle_cg: int = 0,
            x_split: int = 1, x_scale: float = 1.0, out_scale: float = 1.0, x_cols: int = 0, n_store: int = 0):
    """3x3 conv (stride 1, pad 1) on a channels-last token matrix x[B*H*W, >=CinP]; w packed by
    pack_conv_weight.  Returns out (and the per-workgroup channel sums if want_pool).  ``x_cols`` > 0: x is [rows, x_cols] at its
    real width (fp32, multiple of 4; the channels up to CinP read as 0); ``n_store`` > 0: the result is [rows, n_store] (fp32,
    multiple of 4) -- GrlConvArgs, ABI 22."""
    _dev_check(x, w, bias, resid, out)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float32, GEMM_DTYPE)
    assert w.dtype == GEMM_DTYPE and w.is_contiguous() and w.dim() == 3 and w.shape[0] == 9
    CoutP, CinP = w.shape[1], w.shape[2]
    assert x_split in (1, 2, 3) and x.shape[0] >= B * H * W and x.shape[1] >= (x_cols if x_cols > 0 else CinP // x_split) and bias.numel() == CoutP
    assert x_split == 1 or x.dtype == torch.float32
    assert x_cols == 0 or (x.dtype == torch.float32 and x_split == 1 and x_cols % 4 == 0 and x.stride(0) % 4 == 0)
    assert n_store == 0 or (n_store % 4 == 0 and n_store <= CoutP and shuffle_r <= 1 and out_dtype == torch.float32 and not want_pool)
    if shuffle_r > 1:
        rows, cols = B * H * W * shuffle_r * shuffle_r, shuffle_cg
    else:
        rows, cols = B * H * W, (n_store if n_store > 0 else CoutP)
    if out is None:
        out = empty(rows, cols, dtype=out_dtype, device=x.device)
    pool = None
    lib = L.lib()
    if want_pool:
        nwg = lib.grl_conv3x3_num_workgroups(B, H, W)
        pool = empty(nwg, CoutP, dtype=torch.float32, device=x.device)
    # at most 192 output channels per launch; larger layers are split on the channel axis
    step = CoutP
    # Short-K layers (CAB conv2: 64 -> 192 channels, 9 tap steps in all) are epilogue/store dominated with one 104 KB-LDS
    # workgroup per CU; two launches of 96 output channels run two workgroups per CU whose store and MFMA phases
    # overlap (137 -> 101 us per 4 tiles).  Long-K layers (stage convs) lose from the split.  GRL_CONV_SPLIT overrides.
    split = int(os.environ.get("GRL_CONV_SPLIT", "96" if CinP <= 64 and x_split == 1 else "0"))
    if split and CoutP > split and CoutP % split == 0 and shuffle_r <= 1:
        step = split
    elif CoutP > 192:
        step = max(s for s in (192, 128, 96, 64, 48, 32, 16) if CoutP % s == 0 and (shuffle_r <= 1 or s % shuffle_cg == 0))
    for c0 in range(0, CoutP, step):
        ns = 0
        if n_store > 0:                     # this launch's share of the real output channels
            ns = min(step, n_store - c0)
            if ns <= 0:
                continue
            ns = 0 if ns == step else ns
        args = L.GrlConvArgs(
            x=_ptr(x), x_dtype=_KIND[x.dtype], ldx=x.stride(0),
            w=C.c_void_p(w.data_ptr() + c0 * CinP * 2), w_tap_stride=CoutP * CinP,
            bias=C.c_void_p(bias.data_ptr() + c0 * 4),
            B=B, H=H, W=W, CinP=CinP, CoutP=step, x_split=x_split, x_scale=x_scale, out_scale=out_scale, act=act, slope=slope,
            resid=C.c_void_p(resid.data_ptr() + c0 * 4) if resid is not None else C.c_void_p(0),
            ldr=resid.stride(0) if resid is not None else 0,
            pool_partial=C.c_void_p(pool.data_ptr() + c0 * 4) if pool is not None else C.c_void_p(0), pool_stride=CoutP,
            out=C.c_void_p(out.data_ptr() + (0 if shuffle_r > 1 else c0 * out.element_size())),
            out_dtype=_KIND[out.dtype], ldo=out.stride(0),
            shuffle_r=shuffle_r, shuffle_cg=shuffle_cg, shuffle_ij0=(c0 // shuffle_cg if shuffle_r > 1 else 0),
            x_cols=x_cols, n_store=ns,
        )
        with _timed(f"conv3x3 {CinP}->{CoutP} {H}x{W}" if _PROFILE is not None else "conv3x3"):
            L.check(lib.grl_conv3x3_fwd(L.stream_ptr(), C.byref(args)), "grl_conv3x3_fwd")
    return (out, pool) if want_pool else out


def pack_cab_conv2(w: torch.Tensor, bias: torch.Tensor):
    """(blob, bias[192]) of grl_cab_conv2_fwd from a Conv2d weight [Cout <= 192, Cin <= 48, 3, 3]: K = tap * 48 + cin padded to
    14 k-steps of 32, stored in MFMA fragment order (layout in include/grl_hip.h)."""
    Cout, Cin = w.shape[:2]
    assert Cout <= 192 and Cin <= 48 and tuple(w.shape[2:]) == (3, 3)
    wk = torch.zeros(192, 14 * 32, dtype=torch.float32, device=w.device)
    wk[:Cout, : 9 * 48].view(Cout, 9, 48)[:, :, :Cin] = w.detach().float().permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    blob = wk.to(GEMM_DTYPE).view(12, 16, 14, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(torch.uint8).reshape(-1)
    assert blob.numel() == L.lib().grl_cab_conv2_blob_bytes()
    b = torch.zeros(192, dtype=torch.float32, device=w.device)
    b[:Cout] = bias.detach().float()
    return blob, b


_SE_COUNTERS = {}   # (device, stream) -> zeroed int32 counters the kernel returns to zero (one launch at a time per stream)


def cab_conv2(x: torch.Tensor, blob: torch.Tensor, bias: torch.Tensor, B: int, H: int, W: int, se=None):
    """(out [B*H*W, 192] fp16, pool partial sums [B * wgs_per_image, 192]) = conv3x3(x) + bias of the CAB's second convolution
    (grl_cab_conv2_fwd: filter bank in registers); x [B*H*W, >= 56] fp16 with zero pad channels.  ``se=(w1, b1, w2, b2, C)``:
    the squeeze-excite gate [B, 192] is computed by the kernel's last workgroup per image and returned instead of the sums."""
    _dev_check(x, blob, bias)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == GEMM_DTYPE and x.shape[0] == B * H * W and x.shape[1] >= 56
    out = empty(B * H * W, 192, dtype=GEMM_DTYPE, device=x.device)
    tiles = ((H + 7) // 8) * ((W + 31) // 32)
    wgs = max(1, min(tiles, 256 // B))
    pool = empty(B * wgs, 192, dtype=torch.float32, device=x.device)
    args = L.GrlCabConv2Args(x=_ptr(x), ldx=x.stride(0), blob=_ptr(blob), bias=_ptr(bias), B=B, H=H, W=W, wgs_per_image=wgs,
                             out=_ptr(out), ldo=192, pool_partial=_ptr(pool), pool_stride=192)
    gate = None
    if se is not None:
        w1, b1, w2, b2, C_ = se
        _dev_check(w1, b1, w2, b2)
        key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
        cnt = _SE_COUNTERS.get(key)
        if cnt is None or cnt.numel() < B:
            if torch.cuda.is_current_stream_capturing():   # (the buffer would live in the capturing graph's private pool)
                raise RuntimeError("cab_conv2(se=...): run one eager forward on this stream before capturing it in a graph")
            cnt = _SE_COUNTERS[key] = torch.zeros(max(B, 64), dtype=torch.int32, device=x.device)
        cnt[:B].zero_()   # explicit: an aborted launch must not leave arrivals behind (the kernel also returns them to zero)
        gate = empty(B, 192, dtype=torch.float32, device=x.device)
        args.gate, args.se_counter = _ptr(gate), _ptr(cnt)
        args.se_w1, args.se_b1, args.se_w2, args.se_b2 = _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2)
        args.se_c, args.se_mid, args.inv_hw = C_, w1.shape[0], 1.0 / (H * W)
    with _timed("cab_conv2"):
        L.check(L.lib().grl_cab_conv2_fwd(L.stream_ptr(), C.byref(args)), "grl_cab_conv2_fwd")
    return out, (gate if se is not None else pool)


def se_scale(pool: torch.Tensor, B: int, CP: int, C_: int, HW: int, w1, b1, w2, b2) -> torch.Tensor:
    """scale[B, CP] = sigmoid(W2 relu(W1 mean + b1) + b2) from the conv kernel's partial channel sums."""
    _dev_check(pool, w1, b1, w2, b2)
    scale = empty(B, CP, dtype=torch.float32, device=pool.device)
    L.check(
        L.lib().grl_se_scale_fwd(L.stream_ptr(), _ptr(pool), B, pool.shape[0] // B, CP, C_, w1.shape[0], HW, _ptr(w1),
                                 _ptr(b1), _ptr(w2), _ptr(b2), _ptr(scale)),
        "grl_se_scale_fwd",
    )
    return scale


# ---- training path (BASELINE config 5): the contractions of the backward pass and the optimizer step --------------------
def gemm_tn(a: torch.Tensor, b: torch.Tensor, N: int, K: int, *, taps: int = 1, hw: Optional[Tuple[int, int]] = None,
            a_scale: float = 1.0, out_scale: float = 1.0, b_ones: bool = False):
    """c[taps, N, K] = out_scale * sum_m (a_scale * a[m, :N])^T b[row(m, tap), :K]  (grl_gemm_tn): the weight gradient of a
    token-wise linear (taps=1) or of a 3x3 convolution on channels-last pixel matrices (taps=9, hw=(H, W)).  N, K multiples of 4;
    nothing is read beyond column N / K of a row, so a and b may be the layer's tensors at their real widths.  ``b_ones``: returns
    (c, bias) with bias[n] = out_scale * sum_m a_scale * a[m, n] -- the bias gradient, as if b had a column of ones (ABI 22)."""
    _dev_check(a, b)
    assert a.dim() == 2 and b.dim() == 2 and a.dtype in (torch.float32, torch.float16) and b.dtype in (torch.float32, torch.float16)
    assert a.stride(1) == 1 and b.stride(1) == 1 and a.shape[0] == b.shape[0] and a.shape[1] >= N and b.shape[1] >= K
    # (an fp16 ``a`` is taken as already multiplied by a_scale: the operand copy of the data-gradient launch, linear(a16_out=...))
    M = a.shape[0]
    H, W = hw if hw is not None else (0, 0)
    det = deterministic()
    # one zeroed buffer for the taps and -- behind them -- the bias sums (a second fill per weight gradient otherwise)
    nbuf = taps * N * K + (N if b_ones else 0)
    buf = torch.zeros(nbuf, dtype=torch.int64, device=a.device) if det else zeros_f32(nbuf, a.device)
    c = buf[: taps * N * K].view(taps, N, K)
    cb = buf[taps * N * K :] if b_ones else None
    kcols = K + (1 if b_ones else 0)
    tiles = ((N + 63) // 64) * ((kcols + 63) // 64) * taps
    # ~1000 workgroups of >= 256 rows each; measured per shape with the XCD-aware tile order (tools/bench_gemm_tn.py, 512 / 768 /
    # 1024 / 1536 / 2048): the 3 x 3-tile proj gradient is fastest at 512 (42 against 51 us), the nine-tap 184 x 192 stage convolution
    # at 1536 (118 against 137), everything else at 1024
    wgs = int(os.environ.get("GRL_GEMM_TN_WGS", "0")) or (512 if tiles <= 9 else (1536 if tiles >= 81 else 1024))
    splits = max(1, min((M + 255) // 256, (wgs + tiles - 1) // tiles))
    args = L.GrlGemmTnArgs(a=_ptr(a), lda=a.stride(0), b=_ptr(b), b_dtype=_KIND[b.dtype], ldb=b.stride(0), M=M, N=N, K=K, taps=taps,
                           H=H, W=W, splits=splits, a_scale=a_scale, out_scale=out_scale, c=None if det else _ptr(c), ldc=K,
                           c_tap_stride=N * K, c_fix=_ptr(c) if det else None, b_ones=int(b_ones),
                           c_bias=None if det else _ptr(cb), c_bias_fix=_ptr(cb) if det else None, a_dtype=_KIND[a.dtype])
    with _timed("gemm_tn"):
        L.check(L.lib().grl_gemm_tn(L.stream_ptr(), C.byref(args)), "grl_gemm_tn")
    if det:
        c = (c.double() * (out_scale * 2.0 ** -30)).float()
        cb = (cb.double() * (out_scale * 2.0 ** -30)).float() if b_ones else None
    return (c, cb) if b_ones else c


def _attn_args(q: TokenGrid, k: TokenGrid, v: TokenGrid, o: TokenGrid, B, nh, table, masked, ones_col, head_dim, k_one31, lazy_floor, lse):
    nwy, nwx = q.Himg // q.wh, q.Wimg // q.ww
    return L.GrlAttnArgs(q=q.c(), k=k.c(), v=v.c(), o=o.c(), B=B, nh=nh, nwy=nwy, nwx=nwx, table=_ptr(table),
                         trows=(q.wh + k.wh - 1) * (q.ww + k.ww - 1), tstride=table.shape[1], masked=int(masked), ones_col=ones_col,
                         head_dim=head_dim, out_dtype=_KIND[o.t.dtype], k_one31=int(k_one31), lazy_floor=_ptr(lazy_floor),
                         lse=_ptr(lse), lse_stride=lse.stride(0) if lse is not None else 0)


def attention_bwd(q: TokenGrid, k: TokenGrid, v: TokenGrid, o: TokenGrid, d_o: torch.Tensor, lse: torch.Tensor, *, B: int, nh: int,
                  table: torch.Tensor, masked: bool, ones_col: int, head_dim: int, g_scale: float):
    """Gradients of grl_attention_fwd w.r.t. its operands: (d_q, d_k, d_v, d_table), fp32, laid out like q, k, v (fp32 planes of
    the same shape) and table.  ``o``: the forward output grid (fp32), ``d_o`` its gradient (same layout), ``lse`` from the forward."""
    _dev_check(q.t, k.t, v.t, o.t, d_o, lse, table)
    assert o.t.dtype == torch.float32 and d_o.dtype == torch.float32 and d_o.shape == o.t.shape and o.t.is_contiguous()
    # d_o: on o's grid; a token-major o [tokens, nh * 32] may come with a gradient that is a column block of a wider matrix (row stride)
    assert d_o.is_contiguous() or (d_o.dim() == 2 and d_o.stride(1) == 1 and d_o.stride(0) % 4 == 0 and d_o.data_ptr() % 16 == 0)
    d_o_ld = 0 if d_o.is_contiguous() else d_o.stride(0)
    assert q.t.is_contiguous() and k.t.is_contiguous() and v.t.is_contiguous() and q.slot == 0 and k.slot == 0 and v.slot == 0 and o.slot == 0
    # every token of q / k / v belongs to exactly one window of the launch: the kernels write all 32 columns of every row
    d_q = empty(q.t.shape, dtype=torch.float32, device=q.t.device)
    d_k = empty(k.t.shape, dtype=torch.float32, device=q.t.device)
    d_v = empty(v.t.shape, dtype=torch.float32, device=q.t.device)
    d_table = zeros_f32(table.numel(), table.device).view(table.shape)
    fix = torch.zeros(table.shape, dtype=torch.int64, device=table.device) if deterministic() else None
    fwd = _attn_args(q, k, v, o, B, nh, table, masked, ones_col, head_dim, False, None, lse)
    args = L.GrlAttnBwdArgs(fwd=fwd, d_o=_ptr(d_o), d_q=_ptr(d_q), d_k=_ptr(d_k), d_v=_ptr(d_v), d_table=_ptr(d_table), g_scale=g_scale,
                            d_table_fix=_ptr(fix), d_o_ld=d_o_ld)
    with _timed("attention_bwd"):
        L.check(L.lib().grl_attention_bwd(L.stream_ptr(), C.byref(args)), "grl_attention_bwd")
    if fix is not None:
        d_table = (fix.double() * (2.0 ** -32 / g_scale)).float()
    return d_q, d_k, d_v, d_table


CPB_HEADS = (1, 2, 3, 4, 6, 8)     # head counts csrc/cpb.hip is instantiated for


def cpb_table(coords: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, rows4: int) -> torch.Tensor:
    """Kernel-domain bias tables of G AffineTransforms at once (grl_cpb_table_fwd): coords [rows, 2], w1 [G, 512, 2], b1 [G, 512],
    w2 [G, nh, 512] -> [G, nh, rows4] = 16 log2(e) sigmoid(cpb_mlp(coords)), rows reversed, pad entries = source row 0."""
    _dev_check(coords, w1, b1, w2)
    G, nh, hid = w2.shape
    rows = coords.shape[0]
    assert coords.shape == (rows, 2) and w1.shape == (G, hid, 2) and b1.shape == (G, hid) and rows4 >= rows and rows4 % 4 == 0
    cs, a, b, c = (t.detach().float().contiguous() for t in (coords, w1, b1, w2))
    out = empty(G, nh, rows4, dtype=torch.float32, device=coords.device)
    args = L.GrlCpbArgs(coords=_ptr(cs), w1=_ptr(a), b1=_ptr(b), w2=_ptr(c), out=_ptr(out), G=G, rows=rows, rows4=rows4, nh=nh, hidden=hid)
    with _timed("cpb_table"):
        L.check(L.lib().grl_cpb_table_fwd(L.stream_ptr(), C.byref(args)), "grl_cpb_table_fwd")
    return out


def cpb_table_bwd(coords: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, d_out: torch.Tensor):
    """Gradients of cpb_table w.r.t. (w1, b1, w2) given d_out [G, nh, rows4] (grl_cpb_table_bwd; the hidden layer is recomputed)."""
    _dev_check(coords, w1, b1, w2, d_out)
    G, nh, hid = w2.shape
    rows, rows4 = coords.shape[0], d_out.shape[2]
    assert d_out.shape == (G, nh, rows4) and d_out.dtype == torch.float32
    cs, a, b, c, g = (t.detach().float().contiguous() for t in (coords, w1, b1, w2, d_out))
    d_w1, d_b1, d_w2 = (zeros_f32(t.numel(), t.device).view(t.shape) for t in (a, b, c))
    args = L.GrlCpbArgs(coords=_ptr(cs), w1=_ptr(a), b1=_ptr(b), w2=_ptr(c), d_out=_ptr(g), d_w1=_ptr(d_w1), d_b1=_ptr(d_b1), d_w2=_ptr(d_w2),
                        G=G, rows=rows, rows4=rows4, nh=nh, hidden=hid)
    with _timed("cpb_table_bwd"):
        L.check(L.lib().grl_cpb_table_bwd(L.stream_ptr(), C.byref(args)), "grl_cpb_table_bwd")
    return d_w1, d_b1, d_w2


STAT_REPLICAS = int(os.environ.get("GRL_STAT_REPLICAS", "32"))     # copies of small cross-workgroup sums (LayerNorm dgamma / dbeta, plane scale gradients)


def layernorm_train(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, resid: Optional[torch.Tensor] = None,
                    row_scale: Optional[torch.Tensor] = None, rows_per_image: int = 0, alpha: float = 1.0):
    """Row LayerNorm of a token matrix x [M, n] (fp32, n a multiple of 4, <= 256) with the row statistics kept for the backward
    pass (grl_layernorm_train_fwd): returns (y, mean [M], rstd [M]).  ``resid``: y = resid + alpha * row_scale[row // rows_per_image]
    * LayerNorm(x) -- a block's post-norm residual with its residual scale and DropPath keep mask in the same pass."""
    _dev_check(x, gamma, beta, resid, row_scale)
    M, n = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1 and n % 4 == 0 and n <= 256 and x.stride(0) % 4 == 0
    assert resid is None or (resid.shape == x.shape and resid.dtype == torch.float32 and resid.stride(1) == 1 and resid.stride(0) % 4 == 0)
    assert row_scale is None or (resid is not None and row_scale.dtype == torch.float32 and row_scale.is_contiguous() and rows_per_image > 0
                                 and row_scale.numel() * rows_per_image >= M)
    g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    y = empty(M, n, dtype=torch.float32, device=x.device)
    stats = empty(2, M, dtype=torch.float32, device=x.device)
    mean, rstd = stats[0], stats[1]
    args = L.GrlLnTrainArgs(x=_ptr(x), ldx=x.stride(0), gamma=_ptr(g), beta=_ptr(b), y=_ptr(y), ldy=n, mean=_ptr(mean), rstd=_ptr(rstd),
                            M=M, n=n, eps=eps, resid=_ptr(resid), ldr=resid.stride(0) if resid is not None else 0,
                            row_scale=_ptr(row_scale), rows_per_image=rows_per_image, alpha=alpha if resid is not None else 0.0)
    with _timed("layernorm_train"):
        L.check(L.lib().grl_layernorm_train_fwd(L.stream_ptr(), C.byref(args)), "grl_layernorm_train_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor,
                  row_scale: Optional[torch.Tensor] = None, rows_per_image: int = 0, alpha: float = 0.0):
    """(dx, dgamma, dbeta) of layernorm_train (grl_layernorm_bwd); ``alpha`` != 0: of its fused-residual form (dy = dL/dy of the sum)."""
    _dev_check(dy, x, mean, rstd, gamma, row_scale)
    M, n = x.shape
    assert dy.shape == x.shape and dy.dtype == torch.float32 and dy.stride(1) == 1 and dy.stride(0) % 4 == 0
    g = gamma.detach().float().contiguous()
    dx = empty(M, n, dtype=torch.float32, device=x.device)
    # dgamma | dbeta in STAT_REPLICAS copies (one fill), summed afterwards: 2048 workgroups adding into the same 2 n addresses serialise
    # at ~36 ns per atomic -- 75 us of a kernel whose rows stream in 20
    R = STAT_REPLICAS if M >= 4096 else 1
    dgb = zeros_f32(2 * R * n, x.device).view(2, R, n)
    args = L.GrlLnTrainArgs(x=_ptr(x), ldx=x.stride(0), gamma=_ptr(g), mean=_ptr(mean), rstd=_ptr(rstd), dy=_ptr(dy), lddy=dy.stride(0),
                            dx=_ptr(dx), lddx=n, dgamma=_ptr(dgb[0]), dbeta=_ptr(dgb[1]), M=M, n=n, eps=0.0,
                            row_scale=_ptr(row_scale), rows_per_image=rows_per_image, alpha=alpha, stat_replicas=R)
    with _timed("layernorm_bwd"):
        L.check(L.lib().grl_layernorm_bwd(L.stream_ptr(), C.byref(args)), "grl_layernorm_bwd")
    dgb = dgb.sum(1) if R > 1 else dgb[:, 0]
    return dx, dgb[0], dgb[1]


def _planes_args(x, scale, src, raw, one_cols, want):
    T, S_in, nh, d = x.shape
    S_out = len(src)
    i8 = lambda v: (C.c_int32 * 8)(*(list(v) + [0] * (8 - len(v))))
    return L.GrlPlanesArgs(x=_ptr(x), scale=_ptr(scale), T=T, S_in=S_in, S_out=S_out, nh=nh, d=d, src=i8(src), raw=i8([int(r) for r in raw]),
                           one_col=i8(one_cols), want_dscale=i8([int(w) for w in want]))


def head_planes_ok(x: torch.Tensor, n_out: int) -> bool:
    return x.is_cuda and x.dim() == 4 and x.shape[1] <= 8 and n_out <= 8 and x.shape[2] <= 8 and x.shape[3] <= 32 and x.shape[3] % 2 == 0


def head_planes(x: torch.Tensor, scale: torch.Tensor, src, raw, one_cols, write32: bool = True):
    """grl_head_planes_fwd: x [T, S_in, nh, d] fp32 -> (fp32 planes [S_out, nh, T, 32], fp16 copy); output slot s reads input slot
    src[s], raw[s]: copied, else L2-normalised over d and multiplied by scale[s][head]; one_cols[s] >= 0: that plane column is 1.0.
    ``write32=False``: the fp32 planes are allocated but NOT written -- for callers whose consumer reads the fp16 copy only and needs
    the fp32 tensors as autograd's handle on the operands (the attention op with ``f16=``): 40 % of the launch's bytes."""
    _dev_check(x, scale)
    assert x.dtype == torch.float32 and x.is_contiguous() and scale.dtype == torch.float32 and scale.is_contiguous()
    T, S_in, nh, d = x.shape
    S_out = len(src)
    assert scale.shape == (S_out, nh)
    out32 = empty(S_out, nh, T, 32, dtype=torch.float32, device=x.device)
    out16 = empty(S_out, nh, T, 32, dtype=PLANE_DTYPE, device=x.device)
    args = _planes_args(x, scale, src, raw, one_cols, [False] * S_out)
    args.out32, args.out16 = (_ptr(out32) if write32 else None), _ptr(out16)
    with _timed("head_planes"):
        L.check(L.lib().grl_head_planes_fwd(L.stream_ptr(), C.byref(args)), "grl_head_planes_fwd")
    return out32, out16


def head_planes_bwd(x: torch.Tensor, scale: torch.Tensor, src, raw, one_cols, grads, want_dscale):
    """grl_head_planes_bwd: ``grads[s]``: fp32 plane [nh, T, 32] or None -> (dx like x, dscale [S_out, nh])."""
    _dev_check(x, scale, *[g for g in grads if g is not None])
    S_out = len(src)
    gs = [None if g is None else g.float().contiguous() for g in grads]
    dx = empty(x.shape, dtype=torch.float32, device=x.device)
    R = STAT_REPLICAS if x.shape[0] >= 4096 else 1          # (replicas of the scale-gradient sums: see layernorm_bwd)
    dscale = zeros_f32(R * S_out * x.shape[2], x.device).view(R, S_out, x.shape[2])
    args = _planes_args(x, scale, src, raw, one_cols, want_dscale)
    args.dscale_replicas = R
    for s_, g in enumerate(gs):
        if g is not None:
            assert g.shape == (x.shape[2], x.shape[0], 32)
            args.dy[s_] = g.data_ptr()
    args.dx, args.dscale = _ptr(dx), _ptr(dscale)
    with _timed("head_planes_bwd"):
        L.check(L.lib().grl_head_planes_bwd(L.stream_ptr(), C.byref(args)), "grl_head_planes_bwd")
    return dx, (dscale.sum(0) if R > 1 else dscale[0])


def pack_conv_train(w: torch.Tensor, b: Optional[torch.Tensor], rows_pad: int, cols_pad: int, flip_t: bool = False):
    """grl_pack_conv3x3: (fp16 [9, rows_pad, cols_pad], fp32 bias [rows_pad] or None) of a conv weight [Cout, Cin, 3, 3] in ONE launch --
    what pack_conv_weight / pack_conv_bias give for the plain layout, or with ``flip_t`` the operand of the data-gradient convolution
    (taps flipped, channel roles swapped)."""
    _dev_check(w, b)
    Cout, Cin = w.shape[:2]
    wc = w.detach().float().contiguous()
    bc = None if b is None else b.detach().float().contiguous()
    ow = empty(9, rows_pad, cols_pad, dtype=GEMM_DTYPE, device=w.device)
    ob = empty(rows_pad, dtype=torch.float32, device=w.device) if (b is not None or not flip_t) else None
    L.check(L.lib().grl_pack_conv3x3(L.stream_ptr(), _ptr(wc), _ptr(bc), _ptr(ow), _ptr(ob), Cout, Cin, rows_pad, cols_pad, int(flip_t)),
            "grl_pack_conv3x3")
    return ow, ob


def pack_linear_train(w: torch.Tensor, wp: torch.Tensor, wt: Optional[torch.Tensor], b: Optional[torch.Tensor] = None,
                      bp: Optional[torch.Tensor] = None) -> None:
    """grl_pack_linear: wp [Np, Kp] (fp16) <- zero-padded w [N, K]; wt [Kp, Np] <- its transpose (optional); bp [Np] (fp32) <- the
    zero-padded bias b (optional; b None: zeros); one launch."""
    _dev_check(w, wp, wt, b, bp)
    N, K = w.shape
    Np, Kp = wp.shape
    wc = w.detach().float().contiguous()
    bc = None if b is None else b.detach().float().contiguous()
    assert wp.dtype == GEMM_DTYPE and wp.is_contiguous() and (wt is None or (wt.dtype == GEMM_DTYPE and wt.is_contiguous() and wt.shape == (Kp, Np)))
    assert bp is None or (bp.dtype == torch.float32 and bp.is_contiguous() and bp.numel() == Np and (bc is None or bc.numel() == N))
    L.check(L.lib().grl_pack_linear(L.stream_ptr(), _ptr(wc), _ptr(bc), _ptr(wp), _ptr(wt), _ptr(bp), N, K, Np, Kp), "grl_pack_linear")


def se_mlp_ok(pool: torch.Tensor, w1: torch.Tensor) -> bool:
    return pool.is_cuda and pool.dim() == 2 and pool.shape[1] <= 256 and w1.shape[0] <= 64


def se_mlp(pool: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor):
    """grl_se_mlp_fwd: (gate [B, C] = sigmoid(w2 relu(w1 pool + b1) + b2), hidden [B, Cmid]) -- the CAB's squeeze-excite MLP, one launch."""
    _dev_check(pool, w1, b1, w2, b2)
    B, C_ = pool.shape
    Cmid = w1.shape[0]
    ts = [t.detach().float().contiguous() for t in (pool, w1, b1, w2, b2)]
    assert ts[1].shape == (Cmid, C_) and ts[3].shape == (C_, Cmid) and ts[2].numel() == Cmid and ts[4].numel() == C_
    gate = empty(B, C_, dtype=torch.float32, device=pool.device)
    hidden = empty(B, Cmid, dtype=torch.float32, device=pool.device)
    args = L.GrlSeMlpArgs(pool=_ptr(ts[0]), w1=_ptr(ts[1]), b1=_ptr(ts[2]), w2=_ptr(ts[3]), b2=_ptr(ts[4]), gate=_ptr(gate), hidden=_ptr(hidden),
                          B=B, C=C_, Cmid=Cmid)
    L.check(L.lib().grl_se_mlp_fwd(L.stream_ptr(), C.byref(args)), "grl_se_mlp_fwd")
    return gate, hidden


def se_mlp_bwd(d_gate: torch.Tensor, pool: torch.Tensor, gate: torch.Tensor, hidden: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor):
    """grl_se_mlp_bwd: (d_pool, d_w1, d_b1, d_w2, d_b2) of se_mlp; one workgroup walks the batch (no atomics)."""
    _dev_check(d_gate, pool, gate, hidden, w1, w2)
    B, C_ = pool.shape
    Cmid = w1.shape[0]
    dg, pl, w1c, w2c = (t.detach().float().contiguous() for t in (d_gate, pool, w1, w2))
    dev = pool.device
    d_pool = empty(B, C_, dtype=torch.float32, device=dev)
    par = not deterministic()          # one workgroup per image + atomics into ONE zeroed buffer (else: one workgroup walks the batch)
    n1, n2 = Cmid * C_, Cmid
    flat = zeros_f32(2 * n1 + n2 + C_, dev) if par else empty(2 * n1 + n2 + C_, dtype=torch.float32, device=dev)
    d_w1, d_b1 = flat[:n1].view(Cmid, C_), flat[n1 : n1 + n2]
    d_w2, d_b2 = flat[n1 + n2 : 2 * n1 + n2].view(C_, Cmid), flat[2 * n1 + n2 :]
    args = L.GrlSeMlpArgs(pool=_ptr(pl), w1=_ptr(w1c), w2=_ptr(w2c), gate=_ptr(gate), hidden=_ptr(hidden), d_gate=_ptr(dg), d_pool=_ptr(d_pool),
                          d_w1=_ptr(d_w1), d_b1=_ptr(d_b1), d_w2=_ptr(d_w2), d_b2=_ptr(d_b2), B=B, C=C_, Cmid=Cmid,
                          b1=_ptr(d_b1), b2=_ptr(d_b2), parallel=int(par))     # (b1 / b2 are not read by the backward kernel; non-null for the argument check)
    L.check(L.lib().grl_se_mlp_bwd(L.stream_ptr(), C.byref(args)), "grl_se_mlp_bwd")
    return d_pool, d_w1, d_b1, d_w2, d_b2


def _rows_ok(t: torch.Tensor) -> bool:
    return t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def se_colsum(a: torch.Tensor, rows_per_image: int, k: float = 1.0, f: Optional[torch.Tensor] = None) -> torch.Tensor:
    """grl_se_colsum: [M / rows_per_image, C] = k * per-image column sums of a (* f): the CAB's average pool / the gate's gradient."""
    _dev_check(a, f)
    M, C_ = a.shape
    assert _rows_ok(a) and (f is None or (_rows_ok(f) and f.shape == a.shape)) and C_ % 4 == 0 and C_ <= 256 and M % rows_per_image == 0
    out = zeros_f32((M // rows_per_image) * C_, a.device).view(M // rows_per_image, C_)
    args = L.GrlSeRowsArgs(a=_ptr(a), lda=a.stride(0), f=_ptr(f), ldf=f.stride(0) if f is not None else 0, out=_ptr(out), ldo=C_, k=k,
                           M=M, C=C_, rows_per_image=rows_per_image)
    L.check(L.lib().grl_se_colsum(L.stream_ptr(), C.byref(args)), "grl_se_colsum")
    return out


def se_apply(a: torch.Tensor, g: torch.Tensor, rows_per_image: int, f: Optional[torch.Tensor] = None, h: Optional[torch.Tensor] = None,
             k: float = 0.0) -> torch.Tensor:
    """grl_se_apply: out[row] = a[row] * g[image] (+ f[row]) (+ k * h[image]) on token matrices [M, C]; g / h [images, C]."""
    _dev_check(a, g, f, h)
    M, C_ = a.shape
    assert _rows_ok(a) and (f is None or (_rows_ok(f) and f.shape == a.shape)) and C_ % 4 == 0 and C_ <= 256 and M % rows_per_image == 0
    assert g.dtype == torch.float32 and g.is_contiguous() and g.shape == (M // rows_per_image, C_)
    assert h is None or (h.dtype == torch.float32 and h.is_contiguous() and h.shape == g.shape)
    out = empty(M, C_, dtype=torch.float32, device=a.device)
    args = L.GrlSeRowsArgs(a=_ptr(a), lda=a.stride(0), f=_ptr(f), ldf=f.stride(0) if f is not None else 0, g=_ptr(g), h=_ptr(h), out=_ptr(out),
                           ldo=C_, k=k, M=M, C=C_, rows_per_image=rows_per_image)
    L.check(L.lib().grl_se_apply(L.stream_ptr(), C.byref(args)), "grl_se_apply")
    return out


def sum_tensors(ts) -> torch.Tensor:
    """grl_sum4: the sum of 2 .. 4 fp32 tensors of one shape in one launch (contiguous copies are made of inputs that are not)."""
    ts = [t.float().contiguous() for t in ts]
    _dev_check(*ts)
    assert 2 <= len(ts) <= 4 and all(t.shape == ts[0].shape for t in ts)
    n = ts[0].numel()
    if n % 4 or any(t.data_ptr() % 16 for t in ts):
        out = ts[0] + ts[1]
        for t in ts[2:]:
            out = out + t
        return out
    out = empty(ts[0].shape, dtype=torch.float32, device=ts[0].device)
    p = [_ptr(t) for t in ts] + [None] * (4 - len(ts))
    L.check(L.lib().grl_sum4(L.stream_ptr(), p[0], p[1], p[2], p[3], _ptr(out), n), "grl_sum4")
    return out
