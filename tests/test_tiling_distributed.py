"""Tile sharding + all-gather + stitching (CPU, gloo, world_size 2 and 3): the distributed result must be
bitwise the serial reference loop (oracle/engine_oracle.forward_tile = engines/base.py:90-116)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grl_image_restoration_amd import tiling
from oracle import engine_oracle as E


class _ToyModel:
    """Deterministic, batch-invariant stand-in for the network: x2 'super-resolution' with a 3x3 conv."""

    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.w = torch.randn(3 * 4, 3, 3, 3, generator=g) * 0.2

    def __call__(self, x):
        outs = [torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(x[i : i + 1], self.w, padding=1), 2) for i in range(x.shape[0])]
        return torch.cat(outs, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, tile, overlap, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5))
    y = tiling.forward_tiled(_ToyModel(), x, tile, overlap, 2, tile_batch=2)
    ret[rank] = y
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,tile,overlap", [(2, (1, 3, 40, 56), 16, 4), (3, (2, 3, 33, 47), 16, 5), (2, (1, 3, 20, 20), 32, 8)])
def test_sharded_tiling_is_bitwise_the_serial_loop(world, shape, tile, overlap):
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5))
    model = _ToyModel()
    want = E.forward_tile(model, x, tile, overlap, 2)
    single = tiling.forward_tiled(model, x, tile, overlap, 2, tile_batch=3)
    assert torch.equal(single, want)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, tile, overlap, ret), nprocs=world, join=True)
    for r in range(world):
        assert torch.equal(ret[r], want), r


def test_tile_list_and_shards():
    tile, origins = tiling.tile_list(720, 1280, 480, 48)
    assert tile == 480 and origins == [(0, 0), (0, 432), (0, 800), (240, 0), (240, 432), (240, 800)]  # 2 x 3 tiles
    assert [tiling.shard_bounds(6, r, 4) for r in range(4)] == [(0, 2, 2), (2, 4, 2), (4, 5, 2), (5, 6, 2)]      # balanced: 2 2 1 1
    assert [tiling.shard_bounds(6, r, 8)[:2] for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 6), (6, 6)]
    assert tiling.tile_list(100, 90, 256, 32)[0] == 90
    # config 4's own tiling (BASELINE configs[3]: 384x384 tiles of a 1280x720 frame): 8 tiles, one per GPU of a node
    assert len(tiling.tile_list(720, 1280, 384, 48)[1]) == 8 and all(hi - lo == 1 for lo, hi, _ in (tiling.shard_bounds(8, r, 8) for r in range(8)))
    for n in range(0, 40):
        for world in (1, 2, 3, 4, 8):
            sizes = [tiling.shard_bounds(n, r, world)[1] - tiling.shard_bounds(n, r, world)[0] for r in range(world)]
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1 and max(sizes) == tiling.shard_bounds(n, 0, world)[2]
            for t_ in range(n):
                r, i = tiling.shard_of_tile(t_, n, world)
                lo, hi, _ = tiling.shard_bounds(n, r, world)
                assert lo + i == t_ and t_ < hi


@pytest.mark.parametrize("tile", [384, 480])
def test_gopro_frame_sharded_over_a_node_of_eight(tile):
    """World 8 (gloo) on the reference's GoPro geometry, 1280x720 at tile 384 / overlap 48 (8 tiles: one per rank) and at the
    reference's default tile 480 (6 tiles: six ranks compute, all eight take part in the all-gather): bitwise the serial loop."""
    shape = (1, 3, 720 // 4, 1280 // 4)          # (the toy model is resolution agnostic: a quarter-size frame with quarter-size tiles
    t4, ov = tile // 4, 48 // 4                  # has the same tile list as the full one -- checked below -- at 1/16 of the CPU time)
    assert [(a * 4, b * 4) for a, b in tiling.tile_list(720 // 4, 1280 // 4, t4, ov)[1]] == tiling.tile_list(720, 1280, tile, 48)[1]
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5))
    model = _ToyModel()
    want = E.forward_tile(model, x, t4, ov, 2)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(8, _free_port(), shape, t4, ov, ret), nprocs=8, join=True)
    for r in range(8):
        assert torch.equal(ret[r], want), r
