"""A few training steps of BASELINE configs[4] (GRL-Base x4 SR, batch 8 x 64x64 LQ, L1, FusedAdamW) for rocprofv3:
    rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python tools/train_steps.py [--steps N] [--graph]
then tools/rocprof_summary.py on the .db: calls / N = kernel nodes per step by name (what a captured step replays)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grl_image_restoration_amd import GRL, FusedAdamW, GraphedTrainStep, baseline_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--graph", action="store_true", help="replay the captured step instead of eager steps (and print its time)")
a = ap.parse_args()
torch.manual_seed(0)
cfg = baseline_config(5)
m = GRL(**cfg).cuda().train()
opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
g = torch.Generator().manual_seed(100)
lq = torch.rand(a.batch, 3, 64, 64, generator=g).cuda()
gt = torch.rand(a.batch, 3, 64 * cfg["upscale"], 64 * cfg["upscale"], generator=g).cuda()
loss_fn = lambda y, t: (y - t).abs().mean()
if a.graph:
    import time
    step = GraphedTrainStep(m, opt, loss_fn, lq, gt, warmup=2)
    step(lq, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(lq, gt)
    torch.cuda.synchronize()
    print(f"graphed: {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms per step, loss {float(loss):.5f}")
else:
    for _ in range(a.steps):
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(m(lq), gt)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print(f"eager: {a.steps} steps, loss {float(loss):.5f}")
