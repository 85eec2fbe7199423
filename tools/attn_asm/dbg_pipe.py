#!/usr/bin/env python3
"""Time split of the software-pipelined attention kernel (debug build with -DPIPE_DEBUG: tools/attn_asm/build_pipe_variants.sh z_debug):
   swaps the debug library in, runs one attention launch of tools/bench_kernels.py's shapes and prints the s_memtime accumulators."""
import ctypes as C, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "grl_image_restoration_amd", "libgrl_hip.so")
shutil.copy(lib, lib + ".orig")
shutil.copy(os.path.join(root, "tools", "attn_asm", "variants", "libgrl_z_debug.so"), lib)
try:
    sys.path.insert(0, root)
    sys.argv = ["bench_kernels.py", "--iters", "10", "--only", sys.argv[1] if len(sys.argv) > 1 else "attn_window"] + (sys.argv[2:] if "--tiles" in sys.argv else ["--tiles", "4"] + sys.argv[2:])
    from grl_image_restoration_amd import _lib as L
    h = L.lib()
    out = (C.c_ulonglong * 16)()
    h.grl_attn_pipe_debug.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_kernels as bk  # noqa
    bk.main()
    h.grl_attn_pipe_debug(out, 1)
    bk.main()
    h.grl_attn_pipe_debug(out, 1)
    v = list(out)
    nw = v[7]
    names = ["whole kernel (to stores landed)", "first pass incl. vote", "main loop", "steady: wait DMA/LDS", "steady: barrier", "steady: prefetch+addr", "steady: statement (asm)", "waves sampled",
             "prologue (Q loads, first DMA issued)", "wait + barrier of chunk 0", "prime (max fill+drain)", "main fill", "main drain", "-", "-", "-"]
    for n, x in zip(names, v):
        print(f"{n:26s} total {x:16d}   per wave {x / max(nw, 1):12.1f}")
finally:
    shutil.move(lib + ".orig", lib)
