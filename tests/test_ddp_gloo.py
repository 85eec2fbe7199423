"""Data-parallel training wrapper (grl_image_restoration_amd/ddp.py) under gloo on CPU, world size 2:
the averaged gradients of two ranks on half batches equal the single-process gradients on the whole batch -- with the
reference's DDP settings (find_unused_parameters=False) and with the bf16-on-the-wire hook (to bf16 precision)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from grl_image_restoration_amd import ddp


class _Toy(nn.Module):
    """Stand-in with the structure that matters for DDP: many small parameter tensors, every one used every step."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, 3, padding=1)
        self.blocks = nn.ModuleList([nn.Sequential(nn.Linear(8, 16), nn.GELU(), nn.Linear(16, 8), nn.LayerNorm(8)) for _ in range(4)])
        self.out = nn.Conv2d(8, 3, 3, padding=1)

    def forward(self, x):
        f = self.conv(x).permute(0, 2, 3, 1)
        for b in self.blocks:
            f = f + b(f)
        return self.out(f.permute(0, 3, 1, 2))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(3)
    return torch.rand(4, 3, 12, 12, generator=g), torch.rand(4, 3, 12, 12, generator=g)


def _worker(rank, world, port, compress, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = ddp.wrap(_Toy(), bucket_mb=1, compress_bf16=compress)
    assert model.find_unused_parameters is False
    x, y = _data()
    per = x.shape[0] // world
    loss = (model(x[rank * per : (rank + 1) * per]) - y[rank * per : (rank + 1) * per]).abs().mean()
    loss.backward()
    if rank == 0:
        ret["grads"] = {k: p.grad.clone() for k, p in model.module.named_parameters()}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("compress", [False, True])
def test_ddp_gradients_equal_single_process_full_batch(compress):
    torch.manual_seed(0)
    ref = _Toy()
    x, y = _data()
    (ref(x) - y).abs().mean().backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), compress, ret), nprocs=2, join=True)
    got = ret["grads"]
    assert set(got) == set(want)
    for k in want:
        err = (got[k] - want[k]).abs().max().item()
        tol = (2e-2 * want[k].abs().max().item() + 1e-6) if compress else 1e-6
        assert err <= tol, (k, err)


def test_wrap_needs_a_process_group():
    with pytest.raises(RuntimeError):
        ddp.wrap(_Toy())
