#!/usr/bin/env python3
"""Race screen: the same forward repeated must be bit-identical (all kernels are deterministic by construction: fixed
reduction orders, no atomics); a DMA / barrier ordering bug shows up as run-to-run differences.  Also compares the
two-stream schedule with the single-stream one (bit-identical: tiles are independent)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_image_restoration_amd import GRL, baseline_config

torch.manual_seed(0)
m = GRL(**baseline_config(3)).eval().cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
for B, hw in ((8, 256), (4, 256), (3, 192), (2, 64)):
    x = torch.rand(B, 3, hw, hw, device="cuda")
    with torch.no_grad():
        ref = m(x).clone()
        for i in range(n):
            y = m(x)
            if not torch.equal(y, ref):
                bad += 1
                print(f"B={B} {hw}x{hw} run {i}: differs, max |d| = {(y - ref).abs().max().item():.3e}")
        os.environ["GRL_SPLIT_STREAMS"] = "1"
        y1 = m(x)
        del os.environ["GRL_SPLIT_STREAMS"]
        if not torch.equal(y1, ref):
            bad += 1
            print(f"B={B} {hw}x{hw}: single-stream result differs from two-stream, max |d| = {(y1 - ref).abs().max().item():.3e}")
    print(f"B={B} {hw}x{hw}: {n} repeats checked")
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
