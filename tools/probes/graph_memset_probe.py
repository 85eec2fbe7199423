"""Is a hipMemsetAsync captured into a HIP graph ordered with the kernel nodes around it on replay?  (GPU box)

grl_attention_bwd zeroes the destinations of its split launches with hipMemsetAsync on the launch stream.  Eagerly that is stream
ordered.  This probe captures  [fill NaN (kernel)] -> [hipMemsetAsync 0] (x n buffers) -> [x += 1 (kernel)]  and replays it: any
element that is not exactly 1.0 afterwards means the memset node ran before the fill or after the add."""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def memset0(t):
    rc = hip.hipMemsetAsync(t.data_ptr(), 0, t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def probe(sizes, replays, mode):
    bufs = [torch.empty(n, device="cuda") for n in sizes]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for b in bufs:
            b.fill_(0.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for b in bufs:
            b.fill_(float("nan"))
        if mode == "memset":
            for b in bufs:
                memset0(b)
        else:
            for b in bufs:
                b.mul_(0.0).nan_to_num_(0.0)     # (kernel-only control)
        for b in bufs:
            b.add_(1.0)
    bad = 0
    for r in range(replays):
        g.replay()
        if r % 8 == 7 or r == replays - 1:
            torch.cuda.synchronize()
            ok = all(bool((b == 1.0).all()) for b in bufs)
            if not ok:
                bad += 1
                if bad <= 3:
                    for i, b in enumerate(bufs):
                        n_nan = int(torch.isnan(b).sum())
                        n_bad = int((b != 1.0).sum())
                        print(f"      replay {r}: buffer {i} ({b.numel()} floats): {n_bad} wrong, {n_nan} NaN, values {b[b != 1.0][:4].tolist()}")
    return bad


if __name__ == "__main__":
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for sizes in ([3 * 2 * 4096 * 32], [3 * 2 * 1024 * 32, 3 * 2 * 1024 * 32], [3 * 8 * 4096 * 32, 3 * 8 * 1024 * 32, 3 * 8 * 1024 * 32], [1000003], [257]):
        for mode in ("memset", "kernel"):
            bad = probe(sizes, replays, mode)
            print(f"sizes {sizes} mode {mode}: {bad} bad checks of {replays // 8 + 1}", flush=True)
