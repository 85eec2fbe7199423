"""The data-parallel training step (train_graph.GraphedTrainStep with a process group: [forward, loss, backward, flat gradient
buffer] -> ONE all-reduce -> [FusedAdamW on the averaged buffer]) under gloo on CPU at world sizes 2 and 8 -- the node's real
world size: the replicas start different, are made equal by the constructor's broadcast, stay bit-identical, issue
one collective per step and follow the single-process full-batch steps of torch.optim.AdamW.

On CPU tensors nothing is captured (there is no HIP graph): the step keeps its shape and runs eagerly on the model's composite torch
path and FusedAdamW's CPU arithmetic.  What this covers is the plumbing an 8-GPU run depends on -- broadcast, flat buffer, views,
collective count, the 1 / world factor -- not kernels; the GPU tests cover the captured form with two replicas on one device."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grl_image_restoration_amd import GRL, FusedAdamW, GraphedTrainStep, make_config

STEPS = 3


def _cfg():
    return make_config("tiny", "yaml", upscale=2, img_size=16, depths=[1, 1], num_heads_window=[2, 2], num_heads_stripe=[2, 2],
                       drop_path_rate=0.0)


def _data(n):
    g = torch.Generator().manual_seed(5)
    return torch.rand(n, 3, 16, 16, generator=g), torch.rand(n, 3, 32, 32, generator=g)


def _model(seed):
    torch.manual_seed(seed)
    return GRL(**_cfg()).train()


def _worker(rank, world, port, ret, wire_bf16):
    import warnings

    warnings.filterwarnings("ignore")
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model(100 + rank)                      # every replica starts DIFFERENT, rank 0's weights are the ones that count
    if rank == 0:
        m.load_state_dict(_model(0).state_dict())
    lq, gt = _data(world)
    x, y = lq[rank : rank + 1], gt[rank : rank + 1]
    opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
    step = GraphedTrainStep(m, opt, lambda o, t: (o - t).abs().mean(), x, y, warmup=1, wire_bf16=wire_bf16)   # default group
    losses = [float(step(x, y)) for _ in range(STEPS)]
    step.finish()
    views = all(p.grad is not None and p.grad._base is not None for p in m.parameters())
    ret[rank] = ({k: p.detach().clone() for k, p in m.named_parameters()}, losses, step.collectives,
                 opt.state[next(iter(m.parameters()))]["step"], views)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,wire_bf16", [(2, False), (8, False), (8, True)])
def test_data_parallel_step_replicas_agree_and_follow_the_full_batch_step(world, wire_bf16):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, wire_bf16), nprocs=world, join=True)
    p0, l0, c0, n0, v0 = ret[0]
    for r in range(world):
        pr, lr, cr, nr, vr = ret[r]
        assert cr == 1 + STEPS and nr == 1 + STEPS and vr, (r, cr, nr, vr)          # one collective per step (warm-up + STEPS)
        for k in p0:
            assert torch.equal(p0[k], pr[k]), (r, k)                                 # same averaged gradients -> same weights, bit for bit
    # single process, full batch, torch's own optimizer
    m = _model(0)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
    lq, gt = _data(world)
    ref_losses = []
    for _ in range(1 + STEPS):
        opt.zero_grad(set_to_none=True)
        loss = (m(lq) - gt).abs().mean()
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    mean_losses = [sum(ret[r][1][i] for r in range(world)) / world for i in range(STEPS)]    # mean of the per-sample losses = the batch loss
    # Adam's first steps turn every gradient into a +-lr update whatever its size: summation-order noise of the all-reduce shows up as a
    # few per cent of ONE lr step (2e-4) in the weights; bf16 on the wire (3 significant digits per gradient) as up to a whole step
    tol_l, tol_p = (1e-5, 1e-5) if not wire_bf16 else (2e-3, 2e-4)
    for a, b in zip(mean_losses, ref_losses[1:]):
        assert abs(a - b) <= tol_l, (mean_losses, ref_losses)
    worst = max(float((p0[k] - p.detach()).abs().max()) for k, p in m.named_parameters())
    mean = sum(float((p0[k] - p.detach()).abs().sum()) for k, p in m.named_parameters()) / sum(p.numel() for p in m.parameters())
    if wire_bf16:     # a gradient whose sign the bf16 rounding flips moves by 2 lr per step: bound the worst entry by that, and the mean tightly
        assert worst <= 2 * 2e-4 * (1 + STEPS) and mean <= 2e-5, (worst, mean)
    else:
        assert worst <= tol_p, (worst, mean)


def test_data_parallel_step_names_a_parameter_without_gradient():
    """a frozen / unused parameter in the optimizer's list used to fail on ``None.reshape``; it is named now."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = _model(0)
        extra = torch.nn.Parameter(torch.zeros(3))          # in the optimizer, not in the network
        opt = FusedAdamW(list(m.parameters()) + [extra], lr=1e-4)
        lq, gt = _data(1)
        with pytest.raises(RuntimeError, match="received no gradient"):
            GraphedTrainStep(m, opt, lambda o, t: (o - t).abs().mean(), lq, gt, warmup=1, process_group=dist.group.WORLD)
    finally:
        dist.destroy_process_group()
