#!/usr/bin/env python3
"""BASELINE config 4: GRL-Base motion deblur on a 1280x720 frame -- whole frame (the reference's eval command uses
tile=0) and tiled (480/48 -> 6 tiles, and 384/48 -> 8 tiles) through tiling.forward_tiled on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_image_restoration_amd import GRL, baseline_config, tiling

cfg = baseline_config(4)
torch.manual_seed(0)
m = GRL(**cfg).eval().cuda()
x = torch.rand(1, 3, 720, 1280, device="cuda")

def timeit(fn, n=2):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): y = fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n, y

with torch.no_grad():
    dt, y = timeit(lambda: m(x))
    print(f"whole frame 1280x720 (padded 768x1344): {dt*1e3:.1f} ms -> {1280*720/dt/1e6:.2f} MP/s  finite={bool(torch.isfinite(y).all())} peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    for tile, ov in ((480, 48), (384, 48)):
        dt, yt = timeit(lambda: tiling.forward_tiled(m, x, tile, ov, 1, tile_batch=8))
        n = len(tiling.tile_list(720, 1280, tile, ov)[1])
        print(f"tiled {tile}/{ov} ({n} tiles, one batch): {dt*1e3:.1f} ms -> {1280*720/dt/1e6:.2f} MP/s  (tiled and whole-frame outputs differ by design: per-tile padding and SE pooling)")
