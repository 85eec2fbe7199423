#!/usr/bin/env python3
"""Time split of the row-streaming attention kernel (debug build with -DROWS_DEBUG, see attention_rows.hip):
   swaps the debug library in, runs one attention launch of tools/bench_kernels.py's shapes and prints the accumulators."""
import ctypes as C, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "grl_image_restoration_amd", "libgrl_hip.so")
shutil.copy(lib, lib + ".orig")
shutil.copy(os.path.join(root, "tools", "attn_asm", "libgrl_hip_dbg.so"), lib)
try:
    sys.path.insert(0, root)
    sys.argv = ["bench_kernels.py", "--tiles", "4", "--iters", "10", "--only", sys.argv[1] if len(sys.argv) > 1 else "attn_window"] + sys.argv[2:]
    from grl_image_restoration_amd import _lib as L
    import runpy
    h = L.lib()
    out = (C.c_ulonglong * 8)()
    h.grl_attn_rows_debug.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_kernels as bk  # noqa
    import torch
    # warm-up launches happen inside bench_kernels; reset, then run once more
    import io, contextlib
    bk.main()
    h.grl_attn_rows_debug(out, 1)
    bk.main()
    h.grl_attn_rows_debug(out, 1)
    v = list(out)
    nw = v[7]
    names = ["prologue", "chunk: wait DMA", "chunk: barrier", "chunk: prefetch+addr", "chunk: rows (asm)", "whole chunk loop", "trips", "waves"]
    for n, x in zip(names, v):
        print(f"{n:24s} total {x:14d}   per wave {x / max(nw, 1):12.1f}")
finally:
    shutil.move(lib + ".orig", lib)
