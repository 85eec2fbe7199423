"""Which ingredient of graph_memset_probe.py makes a captured hipMemsetAsync misbehave: back-to-back replays?  (GPU box)"""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def memset0(t):
    assert hip.hipMemsetAsync(t.data_ptr(), 0, t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream) == 0


def describe(b, want=1.0):
    bad = (b != want) | torch.isnan(b)
    n = int(bad.sum())
    if n == 0:
        return "ok"
    idx = bad.nonzero().flatten()
    runs = int((idx[1:] != idx[:-1] + 1).sum()) + 1
    return f"{n}/{b.numel()} wrong, {runs} run(s) [{int(idx[0])}..{int(idx[-1])}], values {sorted(set(b[idx].tolist()))[:4]}"


def build(n, use_memset, with_fill=True):
    b = torch.zeros(n, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if with_fill:
            b.fill_(float("nan"))
        if use_memset:
            memset0(b)
        else:
            b.mul_(0.0).nan_to_num_(0.0)
        b.add_(1.0)
    return b, g


for n in (196608, 257):
    for use_memset in (True, False):
        for with_fill in (True, False):
            for burst in (1, 2, 8):
                b, g = build(n, use_memset, with_fill)
                res = []
                for it in range(6):
                    for _ in range(burst):
                        g.replay()
                    torch.cuda.synchronize()
                    res.append(describe(b))
                bad = [r for r in res if r != "ok"]
                print(f"n {n} {'memset' if use_memset else 'kernel'} fill={with_fill} replays between syncs {burst}: {len(bad)}/6 bad  {bad[:2]}", flush=True)
