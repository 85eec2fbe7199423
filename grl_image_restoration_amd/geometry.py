"""Host-side geometry of the GRL hot path (no tensors on the hot path, only integers).

The reference materialises relative-position indices and shifted-window masks as O(N1*N2)
tensors (models/common/ops.py:76-157,352-375) and re-computes them on the CPU for every new
input size (models/networks/grl.py:431-453).  The HIP kernels evaluate the same functions as
closed-form index arithmetic; this module is the single host-side statement of those closed
forms (the kernels in csrc/attention.hip mirror it) plus the per-block schedule.
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def stripe_info(stripe_size, stripe_groups, stripe_shift: bool, x_size):
    """Stripe and shift size of a block (mixed_attn_block_efficient.py:61-70)."""
    size, shift = [], []
    for s, g, d in zip(stripe_size, stripe_groups, x_size):
        if g is None:
            size.append(s)
            shift.append(s // 2 if stripe_shift else 0)
        else:
            size.append(d // g)
            shift.append(0 if g == 1 else d // (g * 2))
    return size, shift


def pad_multiple(window_size: int, stripe_size, stripe_groups, df: int) -> int:
    """Input is reflect-padded to a multiple of this (grl.py:273-276)."""
    mss = max(0 if s is None else s for s in stripe_size)
    msg = max(0 if s is None else s for s in stripe_groups) * df
    return max(window_size, mss, msg)


def region1d(p: int, n: int, s: int, sh: int) -> int:
    """Label of rolled coordinate p on an axis of length n, window s, shift sh (ops.py:76-100).

    Three bands split at n-s and n-sh; a zero shift labels the whole axis alike (the reference's
    last slice ``slice(-0, None)`` covers the full axis and overwrites the others).
    """
    if sh == 0:
        return 0
    return 0 if p < n - s else (1 if p < n - sh else 2)


def rel_index(hq: int, wq: int, hk: int, wk: int, q_win: Sequence[int], k_win: Sequence[int]) -> int:
    """Row of the flattened relative-coords table for query (hq,wq) / key (hk,wk), both as
    in-window coordinates on their own grids (closed form of ops.py:308-316,352-375):
    ``(hq-hk+KH-1) * (QW+KW-1) + (wq-wk+KW-1)``."""
    D = q_win[1] + k_win[1] - 1
    return (hq - hk + k_win[0] - 1) * D + (wq - wk + k_win[1] - 1)


def table_rows(q_win: Sequence[int], k_win: Sequence[int]) -> int:
    return (q_win[0] + k_win[0] - 1) * (q_win[1] + k_win[1] - 1)


@dataclass(frozen=True)
class BlockGeo:
    """Static geometry of one transformer block for a given padded input size."""

    window: Tuple[int, int]
    window_shift: int               # 0 or window//2 (grl.py:112)
    stripe: Tuple[int, int]         # resolved for this block's orientation
    stripe_shift: bool              # grl.py:116
    stripe_shift_size: Tuple[int, int]
    df: int
    nh_w: int
    nh_s: int

    @property
    def anchor_stripe(self):
        return (self.stripe[0] // self.df, self.stripe[1] // self.df)

    @property
    def anchor_shift_size(self):
        return (self.stripe_shift_size[0] // self.df, self.stripe_shift_size[1] // self.df)


def block_schedule(depths, num_heads_window, num_heads_stripe, window_size, stripe_size, stripe_groups,
                   stripe_shift: bool, df: int, x_size) -> List[List[BlockGeo]]:
    """grl.py:105-131: block i of a stage shifts its windows iff i is even, uses H stripes iff i is
    even (W blocks take the reversed stripe size / groups, efficient.py:466-471) and shifts its
    stripes iff i % 4 in (2, 3)."""
    window = to_2tuple(window_size)
    H, W = x_size
    out = []
    for si, depth in enumerate(depths):
        stage = []
        for i in range(depth):
            w_type = i % 2 == 1
            ss = list(stripe_size)[::-1] if w_type else list(stripe_size)
            sg = list(stripe_groups)[::-1] if w_type else list(stripe_groups)
            do_shift = (i % 4 in (2, 3)) if stripe_shift else False
            stripe, sshift = stripe_info(ss, sg, do_shift, x_size)
            if H % window[0] or W % window[1] or H % stripe[0] or W % stripe[1]:
                raise ValueError(f"input {x_size} is not divisible by window {window} / stripe {stripe}")
            if stripe[0] % df or stripe[1] % df:
                raise ValueError(f"stripe {stripe} is not divisible by the anchor down factor {df}")
            stage.append(
                BlockGeo(
                    window=tuple(window),
                    window_shift=window[0] // 2 if i % 2 == 0 else 0,
                    stripe=tuple(stripe),
                    stripe_shift=do_shift,
                    stripe_shift_size=tuple(sshift) if do_shift else (0, 0),
                    df=df,
                    nh_w=num_heads_window[si],
                    nh_s=num_heads_stripe[si],
                )
            )
        out.append(stage)
    return out


def tile_origins(dim: int, tile: int, overlap: int) -> List[int]:
    """Tile start offsets of the reference's tiled inference (engines/base.py:96-98)."""
    stride = tile - overlap
    return list(range(0, dim - tile, stride)) + [dim - tile]
