"""Follow-up of graph_memset_probe.py: where are the wrong elements, and which node misbehaves?  (GPU box)"""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def memset(t, v=0):
    rc = hip.hipMemsetAsync(t.data_ptr(), v, t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def describe(tag, b, want):
    torch.cuda.synchronize()
    bad = (b != want) | torch.isnan(b)
    n = int(bad.sum())
    if n == 0:
        print(f"  {tag}: all {b.numel()} elements == {want}")
        return
    idx = bad.nonzero().flatten()
    runs = int((idx[1:] != idx[:-1] + 1).sum()) + 1
    print(f"  {tag}: {n} of {b.numel()} wrong in {runs} run(s), first {int(idx[0])} last {int(idx[-1])}, values {b[idx[:3]].tolist()} ... {b[idx[-2:]].tolist()}")


N = 196608
for mode in ("eager", "graph"):
    print(mode)
    for name, body, start, want in (
        ("fill NaN -> memset 0 -> add 1", lambda b: (b.fill_(float("nan")), memset(b), b.add_(1.0)), 7.0, 1.0),
        ("memset 0 -> add 1", lambda b: (memset(b), b.add_(1.0)), 7.0, 1.0),
        ("fill NaN -> memset 0", lambda b: (b.fill_(float("nan")), memset(b)), 7.0, 0.0),
        ("memset 0 only", lambda b: (memset(b),), 7.0, 0.0),
        ("memset 0x3c only (0.0115)", lambda b: (memset(b, 0x3C),), 7.0, None),
    ):
        b = torch.full((N,), start, device="cuda")
        torch.cuda.synchronize()
        if mode == "eager":
            body(b)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body(b)
            b.fill_(start)
            torch.cuda.synchronize()
            g.replay()
        if want is None:
            torch.cuda.synchronize()
            want = float(torch.tensor([0x3C3C3C3C], dtype=torch.int32).view(torch.float32))
        describe(name, b, want)
        if mode == "graph":
            g.replay()
            describe(name + " (2nd replay)", b, want)
