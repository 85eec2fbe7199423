#!/usr/bin/env python3
"""Timing probes of qkv_split_kernel (debug build of libgrl_hip.so with -DQS_DEBUG, tools/attn_asm/build_variants_generic.sh):
average s_memtime ticks (10 ns) per 32-token tile and wave, by region."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from grl_image_restoration_amd import GRL, _lib as L, baseline_config, ops
cfg = baseline_config(3); cfg.update(depths=[4], num_heads_window=[3], num_heads_stripe=[3])
torch.manual_seed(0)
m = GRL(**cfg).eval().cuda()
with torch.no_grad():
    for n, p_ in m.named_parameters():
        if n.endswith("logit_scale"): p_.fill_(math.log(100.0))
B, H, W, CP = 4, 256, 256, 192
plan = m._plan((H, W), torch.device("cuda"))
pk = plan["stages"][0]["blocks"][2]
r = torch.randn(B * H * W, CP, device="cuda"); r[:, 180:] = 0
lib = L.lib()
buf = (C.c_ulonglong * 64)()
run = lambda: ops.qkv_anchor(r, pk["qa_blob"], 18, 3, B, H, W, lo_blob=pk["qa_lo"])
for _ in range(3): run()
lib.grl_qs_debug(buf, 1)
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): run()
e1.record()
torch.cuda.synchronize()
print(f"kernel wall {e0.elapsed_time(e1) * 1e3 / N:.1f} us per launch = {e0.elapsed_time(e1) * 1e3 / N / 32:.2f} us per 32-token tile and CU")
lib.grl_qs_debug(buf, 1)
tiles = N * (B * H * W // 32) / 256 * 256   # tiles per wave index summed over the 256 workgroups
names_c = ["barrier", "pair total", "pair epi a", "pair epi b", "single total"]
names_l = ["barrier", "landed wait", "convert", "issue"]
for w in range(8):
    v = [buf[8 * w + i] / tiles for i in range(8)]
    nm = names_l if w == 7 else names_c
    print(f"wave {w}: " + "  ".join(f"{n} {x:.0f}" for n, x in zip(nm, v)) + f"   sum {sum(v[i] for i in ([0,1,2,3] if w == 7 else [0,1,4])):.0f} ticks")
