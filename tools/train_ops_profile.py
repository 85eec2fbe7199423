"""Which torch ops (by call count and device time) surround the HIP kernels in one training step of BASELINE config 5.
Usage (GPU box): python tools/train_ops_profile.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from grl_image_restoration_amd import GRL, FusedAdamW, baseline_config

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = baseline_config(5)
torch.manual_seed(0)
dev = "cuda"
model = GRL(**cfg).to(dev).train()
opt = FusedAdamW(model.parameters(), lr=2e-4, weight_decay=1e-4)
lq = torch.rand(bsz, 3, 64, 64, device=dev)
gt = torch.rand(bsz, 3, 64 * cfg["upscale"], 64 * cfg["upscale"], device=dev)

def step():
    opt.zero_grad(set_to_none=True)
    loss = (model(lq) - gt).abs().mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print(f"eager step: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
t0 = time.perf_counter()
opt.zero_grad(set_to_none=True)
loss = (model(lq) - gt).abs().mean()
torch.cuda.synchronize(); t1 = time.perf_counter()
loss.backward()
torch.cuda.synchronize(); t2 = time.perf_counter()
opt.step()
torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms, optimizer {1e3 * (t3 - t2):.1f} ms (each synchronised)")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
