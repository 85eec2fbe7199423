#!/usr/bin/env python3
"""Timing probes of tail_regs_kernel (debug build of libgrl_hip.so with -DTR_DEBUG, tools/attn_asm/build_variants_generic.sh):
average s_memtime ticks per 32-token tile and wave, by region."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_image_restoration_amd import GRL, _lib as L, baseline_config, ops
cfg = baseline_config(3); cfg.update(depths=[4], num_heads_window=[3], num_heads_stripe=[3])
torch.manual_seed(0)
m = GRL(**cfg).eval().cuda()
B, H, W, CP, C_ = 4, 256, 256, 192, 180
M = B * H * W
plan = m._plan((H, W), torch.device("cuda"))
pk = plan["stages"][0]["blocks"][2]
blk0 = m.layers[0].blocks[2]
pk["tail_rblob"] = ops.pack_tail_regs(pk["proj_w"].float(), blk0.mlp.fc1.weight, blk0.mlp.fc1.bias, blk0.mlp.fc2.weight)
r = torch.randn(M, CP, device="cuda"); r[:, C_:] = 0
att = torch.randn(M, CP, device="cuda").to(torch.float16)
cab = torch.randn(M, CP, device="cuda").to(torch.float16); cab[:, C_:] = 0
gate = torch.ones(B, CP, device="cuda")
lib = L.lib()
buf = (C.c_ulonglong * 64)()
run = lambda: ops.block_tail(att, r, cab, gate, H * W, pk["proj_blob"], pk["proj_b"], pk["n1_g"], pk["n1_b"], pk["mlp_blob"], pk["fc2_b"], pk["n2_g"], pk["n2_b"],
                             Hpad=pk["mlp_hp"], n_real=C_, rblob=pk["tail_rblob"])
for _ in range(3): run()
lib.grl_tr_debug(buf, 1)
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / N
print(f"kernel wall {us:.1f} us per launch = {us / 32:.2f} us per 32-token tile and CU")
lib.grl_tr_debug(buf, 1)
tiles = N * (M // 32)
names = ["B0 wait", "P1 | load r0", "LN1+r1 | store r0, load r1", "fc1 tile", "-", "B4 wait", "LN2+out", "fc2 half | fc1 x2"]
for w in range(8):
    v = [buf[8 * w + i] / tiles for i in range(8)]
    print(f"wave {w:2d}: " + "  ".join(f"{n} {x:.0f}" for n, x in zip(names, v)) + f"   sum {sum(v):.0f} ticks")
