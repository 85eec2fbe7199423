#!/usr/bin/env python3
"""Runs tools/bench_kernels.py's attention launches once per ablation build (tools/attn_asm/build_variants.sh)."""
import glob, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "grl_image_restoration_amd", "libgrl_hip.so")
shutil.copy(lib, lib + ".orig")
only = sys.argv[1] if len(sys.argv) > 1 else "attn_window"
extra = sys.argv[2:]   # passed on to bench_kernels.py (e.g. --logit-scale 100)
try:
    for v in sorted(glob.glob(os.path.join(root, "tools", "attn_asm", "variants", "libgrl_*.so"))):
        shutil.copy(v, lib)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_kernels.py"), "--tiles", "4", "--iters", "20", "--only", only] + extra,
                             capture_output=True, text=True, timeout=300)
        lines = [l for l in out.stdout.splitlines() if l.split() and l.split()[0] in only.split(",")]
        print(f"{os.path.basename(v):28s}", " | ".join(f"{l.split()[0]} {l.split()[1]} us" for l in lines) or out.stderr[-300:], flush=True)
finally:
    shutil.move(lib + ".orig", lib)
