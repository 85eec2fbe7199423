"""Composite torch path for CPU tensors (SURVEY 8(b) "errors": CPU tensors take the plain-PyTorch route; BASELINE configs[0]:
GRL-Tiny x2 SR, one 64x64 patch, CPU-only forward -- the plumbing / parity anchor of the reference).

This is NOT a fallback for the GPU path: a CUDA tensor never reaches this module (``GRL._forward_eager`` keeps failing loudly when
the HIP library is missing), and nothing here is timed or claimed as MI355X work.  It exists so that a module built on a machine
without a GPU -- checkpoint conversion, a unit test of the surrounding engine, the reference's own CPU smoke run -- produces the
reference's numbers instead of an exception.  The three contractions that are HIP kernels on the GPU (torch.ops.grl.linear /
conv3x3 / attention, autograd.py) are written here as ordinary differentiable torch expressions on the SAME operand contract
(channels-last token matrices, head planes [nh, tokens, 32], the kernel-domain bias table), so ``GRL._forward_train`` -- the one
statement of the network's data flow -- serves both devices.

Attention semantics (mixed_attn_block_efficient.py:36-58, 77-94, 128-165, 215-270; ops.py:36-157, 308-375): cyclic shift by
rolling, window / stripe partition, cosine logits (q arrives normalised and scaled by exp(min(logit_scale, ln 100)) * log2e, k
normalised) + relative-position bias + the shifted-window mask (-100), softmax, weighted sum of v -- in the log2 domain the kernels
use (exp2 instead of exp: the table already carries log2e).
"""
import warnings
from typing import Sequence

import torch
import torch.nn.functional as F

from .geometry import region1d

LOG2E = 1.4426950408889634
_warned = False


def announce():
    global _warned
    if not _warned:
        _warned = True
        warnings.warn("grl_image_restoration_amd: CPU tensor -> composite torch path (no HIP kernel runs; the MI355X path needs a CUDA tensor)",
                      stacklevel=3)


def linear(x, w, b=None):
    return F.linear(x, w, b)


def conv3x3(x, w, b, B: int, H: int, W: int):
    """x [B*H*W, Cin] channels-last tokens -> [B*H*W, Cout]; stride 1, zero pad 1."""
    y = F.conv2d(x.view(B, H, W, -1).permute(0, 3, 1, 2), w, b, padding=1)
    return y.permute(0, 2, 3, 1).reshape(B * H * W, -1)


def _windows(t, B: int, geo: Sequence[int]):
    """head planes [nh, B*Himg*Wimg, c] on grid geo = (Himg, Wimg, wh, ww, shy, shx) -> [B*nW, nh, wh*ww, c] in roll + partition order"""
    Himg, Wimg, wh, ww, shy, shx = geo
    nh, _, c = t.shape
    x = t.view(nh, B, Himg, Wimg, c)
    if shy or shx:
        x = torch.roll(x, shifts=(-shy, -shx), dims=(2, 3))      # rolled[y] = orig[(y + sh) % H]
    x = x.view(nh, B, Himg // wh, wh, Wimg // ww, ww, c).permute(1, 2, 4, 0, 3, 5, 6)
    return x.reshape(B * (Himg // wh) * (Wimg // ww), nh, wh * ww, c)


def _unwindows(o, B: int, geo: Sequence[int]):
    Himg, Wimg, wh, ww, shy, shx = geo
    nh, c = o.shape[1], o.shape[3]
    x = o.view(B, Himg // wh, Wimg // ww, nh, wh, ww, c).permute(3, 0, 1, 4, 2, 5, 6).reshape(nh, B, Himg, Wimg, c)
    if shy or shx:
        x = torch.roll(x, shifts=(shy, shx), dims=(2, 3))
    return x.reshape(nh, B * Himg * Wimg, c)


def _region_ids(geo: Sequence[int], device):
    """[nW, wh*ww] region label of every token of every window of the ROLLED image (ops.py:76-157 as closed form, geometry.region1d)"""
    Himg, Wimg, wh, ww, shy, shx = geo
    ry = torch.tensor([region1d(p, Himg, wh, shy) for p in range(Himg)], device=device)
    rx = torch.tensor([region1d(p, Wimg, ww, shx) for p in range(Wimg)], device=device)
    ids = 3 * ry.view(Himg, 1) + rx.view(1, Wimg)
    return ids.view(Himg // wh, wh, Wimg // ww, ww).permute(0, 2, 1, 3).reshape(-1, wh * ww)


def attention(q, k, v, table, qgeo: Sequence[int], kgeo: Sequence[int], B: int, nh: int, d: int, masked: bool):
    """q / k / v: fp32 head planes [nh, tokens, 32] (only the first d columns are data: the constants the HIP kernel wants in the pad
    columns are ignored); ``table``: [nh, >= rows] kernel-domain bias (bias * log2e, REVERSED: entry rows-1-i is row i of the
    reference's table).  Returns planes [nh, q_tokens, 32] with the output in the first d columns, zeros elsewhere."""
    qh, qw, kh, kw = qgeo[2], qgeo[3], kgeo[2], kgeo[3]
    dev = q.device
    qs = _windows(q[..., :d], B, qgeo)                    # [B*nW, nh, Nq, d]
    ks = _windows(k[..., :d], B, kgeo)
    vs = _windows(v[..., :d], B, kgeo)
    # relative-position row of (query (hq, wq), key (hk, wk)): (hq - hk + KH - 1) * D + (wq - wk + KW - 1)   (geometry.rel_index)
    D = qw + kw - 1
    rows = (qh + kh - 1) * D
    hq = torch.arange(qh, device=dev).view(qh, 1, 1, 1)
    wq = torch.arange(qw, device=dev).view(1, qw, 1, 1)
    hk = torch.arange(kh, device=dev).view(1, 1, kh, 1)
    wk = torch.arange(kw, device=dev).view(1, 1, 1, kw)
    idx = ((hq - hk + kh - 1) * D + (wq - wk + kw - 1)).reshape(qh * qw, kh * kw)
    bias = table[:, rows - 1 - idx]                       # [nh, Nq, Nk]
    s = qs @ ks.transpose(-1, -2) + bias.unsqueeze(0)
    if masked:
        iq, ik = _region_ids(qgeo, dev), _region_ids(kgeo, dev)            # [nW, Nq], [nW, Nk]
        m = (iq.unsqueeze(2) != ik.unsqueeze(1)).to(s.dtype) * (-100.0 * LOG2E)
        nW = m.shape[0]
        s = (s.view(B, nW, nh, s.shape[2], s.shape[3]) + m.view(1, nW, 1, m.shape[1], m.shape[2])).view_as(s)
    s = s - s.amax(dim=-1, keepdim=True)
    p = torch.exp2(s)
    o = (p @ vs) / p.sum(-1, keepdim=True)
    return F.pad(_unwindows(o, B, qgeo), (0, 32 - d))
