#!/usr/bin/env python3
"""Same-box A/B of library variants (tools/attn_asm/variants/libgrl_*.so): for each variant, bench.py's timed leg (checkpoint-like
scales) and the random-init leg, a few times interleaved (the pool's boxes differ by more than most effects: only same-box numbers
compare).   python tools/ab_bench.py [rounds] [extra bench.py args]"""
import glob, json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "grl_image_restoration_amd", "libgrl_hip.so")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
extra = sys.argv[2:]
variants = sorted(glob.glob(os.path.join(root, "tools", "attn_asm", "variants", "libgrl_*.so")))
shutil.copy(lib, lib + ".orig")
res = {os.path.basename(v): [] for v in variants}
try:
    for r in range(rounds):
        for v in variants:
            shutil.copy(v, lib)
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--no-train", "--no-cpu-baseline"] + extra,
                                 capture_output=True, text=True, timeout=900)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res[os.path.basename(v)].append((d["ms_per_step"], d.get("random_init_scales", {}).get("ms_per_step")))
            except Exception as e:
                res[os.path.basename(v)].append(("ERR", out.stderr[-300:]))
            print(os.path.basename(v), res[os.path.basename(v)][-1], flush=True)
finally:
    shutil.move(lib + ".orig", lib)
print(json.dumps(res))
