"""Input-independent attention tables, computed once per (weights, geometry) on the device.

The reference recomputes ``16*sigmoid(cpb_mlp(coords_table))`` and gathers it through an
(N1 x N2) index tensor for every block on every call (mixed_attn_block_efficient.py:41-47).
At inference the table depends only on the weights, so it is built once here and handed to the
attention kernel, which performs the gather as index arithmetic on an LDS copy.
"""
import math
from typing import Sequence

import torch
import torch.nn.functional as F

LOG2E = 1.4426950408889634


def coords_table(q_or_window: Sequence[int], df: int = 1, device=None) -> torch.Tensor:
    """Log-spaced relative coordinate table between a window and its anchor window
    (closed form of models/common/ops.py:225-271 with pretrained_window_size = [0, 0]):
    integer offsets from -(a-1)-(s-a)//2 to s-1-(s-a)//2 per axis, divided by the positive
    extent, times 8, then sign(x)*log2(|x|+1)/log2(8).  Shape (rows, 2), row-major (h, w)."""
    window = list(q_or_window)
    aws = [w // df for w in window]
    pos = [w - 1 - (w - a) // 2 for w, a in zip(window, aws)]
    neg = [-(a - 1) - (w - a) // 2 for w, a in zip(window, aws)]
    ch = torch.arange(neg[0], pos[0] + 1, dtype=torch.float32, device=device)
    cw = torch.arange(neg[1], pos[1] + 1, dtype=torch.float32, device=device)
    t = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).reshape(-1, 2).contiguous()
    t[:, 0] /= pos[0]
    t[:, 1] /= pos[1]
    t = t * 8
    return torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / math.log2(8)


def clamped_scale(logit_scale: torch.Tensor) -> torch.Tensor:
    """exp(min(logit_scale, ln 100)) per head (mixed_attn_block_efficient.py:39)."""
    return torch.clamp(logit_scale.detach().float().reshape(-1), max=math.log(1.0 / 0.01)).exp()


def bias_rows(cpb0_w, cpb0_b, cpb2_w, coords: torch.Tensor) -> torch.Tensor:
    """16*sigmoid(Linear(512->nh, no bias)(ReLU(Linear(2->512)(coords)))) -> (rows, nh)."""
    h = F.relu(F.linear(coords, cpb0_w.detach().float(), cpb0_b.detach().float()))
    return 16.0 * torch.sigmoid(F.linear(h, cpb2_w.detach().float()))


def kernel_table(bias: torch.Tensor) -> torch.Tensor:
    """(rows, nh) natural-log-domain bias -> (nh, rows4) table in the kernel's exp2 domain, stored
    REVERSED along rows (entry rows-1-i = row i) and zero-padded to a multiple of 4 floats per head."""
    t = torch.flip(bias.t().contiguous() * LOG2E, dims=(1,))
    pad = (-t.shape[1]) % 4
    if pad:
        t = torch.nn.functional.pad(t, (0, pad))
    return t.contiguous()


def lazy_floor(scale: torch.Tensor) -> torch.Tensor:
    """Integer-valued lower bound (log2 domain) of every unmasked logit ``scale*cos + bias`` of a head
    (|cos| <= 1, bias >= 0): the start value of the attention kernel's running softmax offset."""
    return (-torch.ceil(scale * LOG2E) - 1.0).float().contiguous()


def lazy_ceil(scale: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """Upper bound (log2 domain) of every logit ``scale*cos + bias`` of a head: |cos| <= 1 and the largest entry of the head's
    kernel table (nh, rows4).  Masked logits only go down.  See GrlAttnArgs.lazy_ceil."""
    # (q and k are unit vectors rounded to fp16 element by element: the 0.4 % + 0.05 covers that with a wide margin)
    return (scale.reshape(-1).float() * (LOG2E * 1.004) + 0.05 + table.max(dim=1).values.clamp_min(0.0)).float().contiguous()
