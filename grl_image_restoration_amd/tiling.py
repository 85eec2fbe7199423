"""Tiled whole-image inference, sharded over the GPUs of a node (BASELINE config 4).

Reference semantics (engines/base.py:90-116, ``forward_tile``): tiles of side ``tile`` start at
``range(0, dim - tile, tile - overlap) + [dim - tile]`` on both axes; every tile runs the full model
(its own reflect-pad and its own squeeze-excite pooling); outputs are summed into a canvas E, a
count canvas W gets ones, and the result is E / W (uniform averaging in the overlaps).

MI355X design: tiles are independent units.  The row-major tile list (the reference's loop order)
is cut into contiguous chunks, one per rank; a rank pushes its chunk through the network as
batches (B > 1 fills the 256 CUs much better than the reference's one-tile-at-a-time loop); ONE
collective -- an all-gather of the fixed-shape tile outputs over RCCL/xGMI (<= 13 MB per tile,
latency-bound, one shot) -- gives every rank all tiles, and every rank stitches them in the
reference's loop order, so the result is bitwise the serial loop's result for the same per-tile
outputs.  No other data-path collective exists.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .geometry import tile_origins


def tile_list(h: int, w: int, tile: int, overlap: int) -> Tuple[int, List[Tuple[int, int]]]:
    """(effective tile side, [(h_idx, w_idx)] in the reference's row-major loop order)."""
    tile = min(tile, h, w)
    return tile, [(hi, wi) for hi in tile_origins(h, tile, overlap) for wi in tile_origins(w, tile, overlap)]


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous chunk [lo, hi) of n tiles for ``rank`` and the padded per-rank count of the all-gather.  Balanced: the first
    n % world ranks take one tile more than the others, so no rank idles while another holds two tiles more than it needs to
    (the reference's default GoPro split, 6 tiles of 480, on 4 ranks: 2 2 1 1, not 2 2 2 0; 8 tiles of 384 on 8 ranks: one each).
    Whole tiles are the unit -- the tile list is the reference's and so is the result -- so with fewer tiles than ranks
    (6 tiles on 8 GPUs) the last ranks only take part in the collective."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0), q + (1 if r else 0)


def shard_of_tile(t: int, n: int, world: int) -> Tuple[int, int]:
    """(rank, index inside the rank's chunk) of tile t under shard_bounds."""
    q, r = divmod(n, world)
    if t < r * (q + 1):
        return divmod(t, q + 1)
    k, i = divmod(t - r * (q + 1), max(q, 1))
    return r + k, i


def stitch(outs: Sequence[torch.Tensor], origins: Sequence[Tuple[int, int]], shape, tile: int, scale: int) -> torch.Tensor:
    """E / W accumulation in the reference's order (engines/base.py:100-116)."""
    b, c, h, w = shape
    E = torch.zeros(b, c, h * scale, w * scale, dtype=outs[0].dtype, device=outs[0].device)
    Wt = torch.zeros_like(E)
    for o, (hi, wi) in zip(outs, origins):
        E[..., hi * scale : (hi + tile) * scale, wi * scale : (wi + tile) * scale].add_(o)
        Wt[..., hi * scale : (hi + tile) * scale, wi * scale : (wi + tile) * scale].add_(torch.ones_like(o))
    return E.div_(Wt)


def forward_tiled(model: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, tile: int, overlap: int, scale: int,
                  out_channels: Optional[int] = None, tile_batch: int = 8, group=None, force_collective: bool = False) -> torch.Tensor:
    """Tiled inference of ``x`` (b, c, h, w).  With an initialised process group every rank passes the
    SAME ``x`` and receives the SAME stitched output; without one it is the single-GPU batched loop.
    ``force_collective``: run the all-gather branch even at world size 1 (RCCL smoke test on a single GPU)."""
    b, c, h, w = x.shape
    oc = out_channels or c
    tile, origins = tile_list(h, w, tile, overlap)
    n = len(origins)
    distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if distributed else (0, 1)
    lo, hi, per = shard_bounds(n, rank, world)

    mine = []
    for s in range(lo, hi, tile_batch):
        chunk = origins[s : min(s + tile_batch, hi)]
        # tiles of all b images are stacked on the batch axis: (len(chunk)*b, c, tile, tile)
        patch = torch.cat([x[..., hi_ : hi_ + tile, wi_ : wi_ + tile] for hi_, wi_ in chunk], dim=0)
        out = model(patch)
        mine.extend(out.split(b, dim=0))
    ts = tile * scale
    if not distributed:
        return stitch(mine, origins, (b, oc, h, w), tile, scale)

    # one fixed-shape all-gather of this rank's tile outputs (padded to `per` tiles)
    dev = x.device
    dtype = mine[0].dtype if mine else x.dtype
    send = torch.zeros(per, b, oc, ts, ts, dtype=dtype, device=dev)
    for i, o in enumerate(mine):
        send[i].copy_(o)
    recv = torch.empty(world * per, b, oc, ts, ts, dtype=dtype, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    outs = []
    for t in range(n):
        r, i = shard_of_tile(t, n, world)
        outs.append(recv[r * per + i])
    return stitch(outs, origins, (b, oc, h, w), tile, scale)
