"""Data-parallel training of GRL over the GPUs of a node (replicas + gradient all-reduce).

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=False)`` through Lightning's DDPStrategy
(tools/trainer.py:135-142): one process per GPU, gradients all-reduced.  On MI355X that all-reduce is RCCL over xGMI
(torch.distributed backend "nccl"); GRL-Base has 77 MB of fp32 gradients in 1390 tensors.  xGMI is point-to-point (7 links per
GPU), ring collectives are per-link bound, so the gradients go out in few, large buckets (>= 25 MB: 3-4 collectives per step,
overlapped with the rest of the backward pass) and, optionally, compressed to bf16 on the wire (half the bytes; the sum is
formed in bf16, the optimizer state stays fp32).

``wrap`` is model-agnostic (it is exercised on CPU with the gloo backend in tests/test_ddp_gloo.py); every parameter of
``grl_image_restoration_amd.GRL`` receives a gradient each step, so find_unused_parameters stays False as in the reference.
"""
from typing import Optional

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


def bf16_compress_hook(process_group, bucket):
    """Gradient bucket all-reduce in bf16 (mean over ranks), result copied back into the fp32 bucket."""
    group = process_group if process_group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    comp = buf.to(torch.bfloat16).div_(world)
    fut = dist.all_reduce(comp, group=group, async_op=True).get_future()

    def done(f):
        buf.copy_(f.value()[0])
        return buf

    return fut.then(done)


def wrap(model: torch.nn.Module, device: Optional[torch.device] = None, bucket_mb: int = 32, compress_bf16: bool = False,
         process_group=None) -> DistributedDataParallel:
    """DistributedDataParallel with the reference's settings (find_unused_parameters=False) and xGMI-sized buckets."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("ddp.wrap needs an initialised process group (one process per GPU)")
    ids = [device.index] if device is not None and device.type == "cuda" else None
    ddp = DistributedDataParallel(model, device_ids=ids, find_unused_parameters=False, bucket_cap_mb=bucket_mb,
                                  gradient_as_bucket_view=True, process_group=process_group)
    if compress_bf16:
        ddp.register_comm_hook(process_group, bf16_compress_hook)
    return ddp
