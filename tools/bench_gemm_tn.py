"""grl_gemm_tn at the weight-gradient shapes of the training step (GRL-Base, batch 8 x 64x64 LQ: M = 32768 tokens), HIP events.

Round 6 measured a templated variant with 128 x 128 / 192 x 192 output tiles per workgroup (2 x 2 / 3 x 3 MFMA tiles per wave: A and B
read once instead of 3 x / 9 x for the QKV layer) with this script -- sum over the eight shapes 471 us (64 x 64 tile), 486 (<= 128),
684 (<= 192): the larger tiles LOSE.  The 64 x 64 kernel moves 452 MB in 69 us for the QKV gradient (6.5 TB/s out of L2 / Infinity
Cache) at 8 waves per SIMD; the 193 + 144 registers of the 3 x 3 variant leave one wave per SIMD and nothing hides its global loads
(no prefetch: fp32 -> fp16 conversion on the way to LDS rules out LDS-DMA).  Not kept; the kernel is what rounds 2-5 left."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grl_image_restoration_amd import ops  # noqa: E402

M, H, W = 8 * 64 * 64, 64, 64
shapes = [("qkv 540x180", 540, 180, 1), ("proj 180x180", 180, 180, 1), ("fc1 360x180", 360, 180, 1), ("fc2 180x360", 180, 360, 1),
          ("anchor 96x180 (M/4)", 96, 180, 1), ("cab conv1 48x180 x9", 48, 180, 9), ("cab conv2 180x64 x9", 180, 64, 9), ("stage conv 184x192 x9", 184, 192, 9)]
g = torch.Generator().manual_seed(0)
tot = 0.0
for name, N, K, taps in shapes:
    m = M // 4 if "M/4" in name else M
    a = (torch.randn(m, N, generator=g) * 1e-4).cuda()          # (real widths: row strides of N and K floats, as the training path passes them)
    b = torch.randn(m, K, generator=g).cuda()
    ones = K % 32 != 0
    f = lambda: ops.gemm_tn(a, b, N, K, taps=taps, hw=(H, W) if taps == 9 else None, a_scale=4096.0, out_scale=1 / 4096.0, b_ones=ones)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * m * N * K * taps
    tot += us
    print(f"{name:26s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
print(f"sum {tot:.1f} us")
