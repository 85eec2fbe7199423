"""Scratch GPU diagnostic: times the config-3 forward (GRL-Base x4, 256x256 LQ) eagerly."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_image_restoration_amd import GRL, baseline_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = baseline_config(3)
m = GRL(**cfg).eval().cuda()
x = torch.rand(B, 3, 256, 256, device="cuda")
with torch.no_grad():
    for _ in range(2):
        y = m(x)
    torch.cuda.synchronize()
    t = time.time()
    n = 3
    for _ in range(n):
        y = m(x)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
print(f"B={B} forward {dt*1e3:.2f} ms  -> {B*256*256/dt/1e6:.3f} LQ-MP/s; out {tuple(y.shape)} finite={bool(torch.isfinite(y).all())}")
