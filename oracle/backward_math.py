"""TEST INFRASTRUCTURE -- closed-form backward of the cosine attention in the shape a flash-style HIP kernel computes it
(SURVEY 8(f) N1 groundwork).  Pinned against autograd through ``grl_oracle.cosine_attention`` in tests/.

Forward (models/common/mixed_attn_block_efficient.py:36-58,77-94), per window b and head h:

    qh = q / max(|q|, eps)          kh = k / max(|k|, eps)
    S  = s_h * qh kh^T + bias[index] + mask          s_h = exp(min(logit_scale_h, ln 100))
    P  = softmax(S)                 O = P v

What the backward kernel needs besides (q, k, v, O, dO) is one number per query, the soft-max normaliser -- the forward
kernel already produces it as the ones-column of the PV product -- so nothing N x N is stored:

    D_i   = sum_c dO_ic O_ic                                  (rowsum(dO * O), one pass over the query's output)
    dP    = dO v^T                     dS = P * (dP - D)       (P recomputed from S and the stored normaliser)
    dv    = P^T dO
    dqh   = s_h dS kh                  dkh = s_h dS^T qh
    ds_h  = sum_ij dS_ij (qh_i . kh_j)                         -> dlogit_scale_h = ds_h * s_h * [logit_scale_h < ln 100]
    dbias[r, h] = sum over (i, j, windows) with index[i, j] == r of dS_ij      (histogram over the relative-position table)
    dq    = (dqh - qh (qh . dqh)) / max(|q|, eps)              (same for k)

The bias rows are 16 * sigmoid(cpb_mlp(table)); their chain rule (a 2 -> 512 -> heads MLP over <= 9025 rows) is dense torch
work outside the kernel.
"""
import math

import torch


def normalize_backward(x: torch.Tensor, dxh: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """Gradient of xh = x / max(|x|, eps) (F.normalize) w.r.t. x given dxh."""
    n = x.norm(dim=-1, keepdim=True).clamp_min(eps)
    xh = x / n
    return (dxh - xh * (xh * dxh).sum(-1, keepdim=True)) / n


def attention_backward(q, k, v, dO, scale_raw, bias_rows, index, mask=None, eps: float = 1e-12):
    """q (B_, nh, Nq, d), k/v (B_, nh, Nk, d), dO like the output; scale_raw (nh,) = the logit_scale parameter;
    bias_rows (rows, nh) = 16*sigmoid(cpb_mlp(table)); index (Nq, Nk) long; mask (nW, Nq, Nk) or None.

    Returns dq, dk, dv, dlogit_scale (nh,), dbias_rows (rows, nh) -- computed tile-free here, but only from quantities a
    flash-style kernel has per tile (P is rebuilt from S and the row normaliser)."""
    B_, nh, Nq, _ = q.shape
    Nk = k.shape[2]
    s = torch.clamp(scale_raw, max=math.log(100.0)).exp().view(1, nh, 1, 1)
    qh = q / q.norm(dim=-1, keepdim=True).clamp_min(eps)
    kh = k / k.norm(dim=-1, keepdim=True).clamp_min(eps)
    cos = qh @ kh.transpose(-1, -2)
    S = s * cos + bias_rows[index.reshape(-1)].view(Nq, Nk, nh).permute(2, 0, 1).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        S = (S.view(B_ // nW, nW, nh, Nq, Nk) + mask.unsqueeze(1).unsqueeze(0)).view(B_, nh, Nq, Nk)
    m = S.max(dim=-1, keepdim=True).values
    e = torch.exp(S - m)
    l = e.sum(-1, keepdim=True)              # the forward kernel's ones-column (up to the fixed / running bound)
    P = e / l
    O = P @ v
    D = (dO * O).sum(-1, keepdim=True)
    dP = dO @ v.transpose(-1, -2)
    dS = P * (dP - D)
    dv = P.transpose(-1, -2) @ dO
    dqh = s * (dS @ kh)
    dkh = s * (dS.transpose(-1, -2) @ qh)
    ds = (dS * cos).sum(dim=(0, 2, 3))
    dscale_raw = ds * s.view(-1) * (scale_raw < math.log(100.0)).to(ds.dtype)
    dbias = torch.zeros_like(bias_rows)
    dbias.index_add_(0, index.reshape(-1), dS.sum(0).permute(1, 2, 0).reshape(Nq * Nk, nh))
    return normalize_backward(q, dqh, eps), normalize_backward(k, dkh, eps), dv, dscale_raw, dbias
