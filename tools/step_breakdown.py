#!/usr/bin/env python3
"""Share of one forward per C-ABI entry point (HIP events around every launch, single stream so that launches do not
overlap): where a 256x256-tile batch spends its time.  Debug tool."""
import os, sys
os.environ["GRL_SPLIT_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_image_restoration_amd import GRL, baseline_config, ops

import math
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 4
cfg_i = int(args[1]) if len(args) > 1 else 3          # BASELINE config (2: Small denoise 128^2, 3: Base x4 256^2, 4: Base deblur 384^2)
side = {2: 128, 4: 384}.get(cfg_i, 256)
m = GRL(**baseline_config(cfg_i)).eval().cuda()
if "--trained" in sys.argv:   # logit scales around the clamp, the draw of bench.py's trained_scales leg
    gs = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n.endswith("logit_scale"):
                p_.copy_((math.log(100.0) + 0.3 * torch.randn(p_.shape, generator=gs)).cuda())
x = torch.rand(B, 3, side, side, device="cuda")
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    ops.profile_begin()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m(x)
    e1.record()
    prof = ops.profile_end()
tot = e0.elapsed_time(e1)
print(f"config {cfg_i}, B={B} x {side}^2, precision {m.precision}: forward {tot:.2f} ms (single stream)")
acc = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -sum(kv[1])):
    s = sum(v)
    acc += s
    print(f"  {k:12s} {len(v):5d} launches  {s:8.2f} ms  {100 * s / tot:5.1f} %   mean {1e3 * s / len(v):8.1f} us")
print(f"  (timed launches {acc:.2f} ms = {100 * acc / tot:.1f} %; rest = torch glue: pad/normalise/permute/fill/copies)")
