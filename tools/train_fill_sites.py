"""Who issues the fill kernels of a training step: torch.profiler events named aten::fill_ / aten::zero_ with their chain of parent ops.
Usage (GPU box): python tools/train_fill_sites.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from grl_image_restoration_amd import GRL, FusedAdamW, baseline_config

cfg = baseline_config(5)
torch.manual_seed(0)
model = GRL(**cfg).cuda().train()
opt = FusedAdamW(model.parameters(), lr=2e-4, weight_decay=1e-4)
lq = torch.rand(8, 3, 64, 64, device="cuda")
gt = torch.rand(8, 3, 256, 256, device="cuda")


def step():
    opt.zero_grad(set_to_none=True)
    (model(lq) - gt).abs().mean().backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
    torch.cuda.synchronize()
want = sys.argv[1].split(",") if len(sys.argv) > 1 else ["aten::fill_", "aten::zero_"]
rows = collections.Counter()
for e in prof.events():
    if e.name in want:
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        rows[" <- ".join(chain)] += 1
print(f"# {sum(rows.values())} events named {want} in one training step, by parent chain")
for k, c in rows.most_common(40):
    print(f"{c:6d}  {k}")
