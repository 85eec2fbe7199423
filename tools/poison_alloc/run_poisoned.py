"""python tools/poison_alloc/run_poisoned.py <pytest arguments>   (GPU box)

Runs pytest in a process whose device allocator is tools/poison_alloc/libpoison_alloc.so (every byte not inside a live, written
tensor reads as NaN: never-written outputs, red zones behind every block, freed blocks) and with GRL_POISON=1 (the package's own
workspace tensors are NaN-filled by a launch on the stream -- that one is captured into HIP graphs too)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("GRL_POISON", "1")

import torch  # noqa: E402

so = os.path.join(HERE, "libpoison_alloc.so")
if os.environ.get("GRL_POISON_ALLOC", "1") != "0":
    alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "grl_poison_malloc", "grl_poison_free")
    torch.cuda.memory.change_current_allocator(alloc)
    print(f"[run_poisoned] device allocator: {so}", flush=True)

import pytest  # noqa: E402

rc = pytest.main(sys.argv[1:])
try:
    ctypes.CDLL(so).grl_poison_stats()
except OSError:
    pass
sys.exit(rc)
