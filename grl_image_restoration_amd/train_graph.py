"""A whole training step -- forward, loss, backward, FusedAdamW -- as ONE captured HIP graph.

The reference's step (engines/base.py:221-236: forward, L1 loss, manual_backward, optimizer step) is ~19 000 kernel launches on
this path at batch 8 x 64x64 (1390 parameter tensors, 40 blocks x ~60 autograd nodes x forward + backward): eager, the host
issues them slower than the GPU retires them (270 ms per step of which ~90 ms are kernels).  Captured once and replayed, the
step costs what its kernels cost.  What capture needs from the rest of the package:
  * no host read inside the step: the gradient operand scale is frozen at its warm-up value (autograd.frozen_grad_scale);
  * an optimizer launch that stays correct on replay: the step count lives on the device (FusedAdamW.enable_capture,
    GrlAdamWArgs.bias_corrections_dev);
  * static input buffers: the caller's batches are copied into them before each replay.
Single process / single GPU (a DDP step with its bucketed RCCL all-reduces is left eager).

What a captured step FREEZES, and what it does not:
  * learning rate and weight decay are NOT frozen: the launch reads them from device memory and ``__call__`` refreshes them from
    ``optimizer.param_groups`` before every replay -- an LR scheduler stepping the optimizer works as in eager mode;
  * betas / eps are held by value: changing them after the capture raises (re-capture);
  * the power-of-two scale that brings the gradients into fp16 range for the backward contractions is frozen at its warm-up value
    (autograd.frozen_grad_scale, 2^10 of headroom).  That is right for losses whose top gradient does not move (L1: |dL/dy| = 1/N,
    what the reference trains GRL with, config/loss/l1.yaml).  For MSE / Charbonnier / perceptual losses the gradients shrink by
    orders of magnitude during training: pass ``recalibrate_every=N`` to re-measure the scale on an eager step every N replays and
    re-capture when it moved by more than 2^4, or train those losses eagerly.
"""
from typing import Callable

import torch

from . import autograd as AG


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer, loss_fn: Callable, lq: torch.Tensor, gt: torch.Tensor, warmup: int = 3,
                 recalibrate_every: int = 0):
        """``loss_fn(output, target) -> scalar``; ``lq`` / ``gt``: example batch (shapes are baked into the graph).  Runs ``warmup``
        eager steps (they DO update the weights), then captures one step."""
        if not hasattr(optimizer, "enable_capture"):
            raise TypeError("GraphedTrainStep needs grl_image_restoration_amd.FusedAdamW (an optimizer whose step() can be captured)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.lq, self.gt = lq.detach().clone(), gt.detach().clone()
        self.recalibrate_every = int(recalibrate_every)
        self._params = [p for g in optimizer.param_groups for p in g["params"]]
        model.train()
        self._capture(warmup)
        self.steps = 0

    def _capture(self, warmup):
        model, optimizer, lq = self.model, self.optimizer, self.lq
        # (the warm-up steps run on a side stream, the capture on the graph's own: the AccumulateGrad nodes of the parameters move
        # between streams by design here -- the warning is silenced for the capture only and restored afterwards)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        cur = torch.cuda.current_stream(lq.device)
        side = torch.cuda.Stream(lq.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):             # (warm-up on a side stream, as torch.cuda.graph's recipe asks)
            for _ in range(max(1, warmup)):
                self._eager_step()
        cur.wait_stream(side)
        torch.cuda.synchronize(lq.device)
        optimizer.enable_capture()
        optimizer.zero_grad(set_to_none=True)     # the gradients of the captured step come from the graph's own memory pool
        self.graph = torch.cuda.CUDAGraph()
        with AG.frozen_grad_scale(), torch.cuda.graph(self.graph):
            optimizer.zero_grad(set_to_none=True)
            self.loss = self.loss_fn(self.model(self.lq), self.gt)
            self.loss.backward()
            optimizer.step()
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)
        self._scale_at_capture = AG.last_grad_scale() if hasattr(AG, "last_grad_scale") else None

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(self.lq), self.gt)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, lq: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        """One optimizer step on the batch; returns the loss tensor of the graph (valid until the next call)."""
        if lq.data_ptr() != self.lq.data_ptr():
            self.lq.copy_(lq, non_blocking=True)
        if gt.data_ptr() != self.gt.data_ptr():
            self.gt.copy_(gt, non_blocking=True)
        if self.recalibrate_every and self.steps and self.steps % self.recalibrate_every == 0:
            self._recalibrate()
        self.optimizer.refresh_capture_hyper()          # lr / weight decay of the captured optimizer launch <- param_groups
        self.graph.replay()
        self.steps += 1
        # the weights changed behind the Python-side version counters: an eval forward between replays (the inference plan and the
        # fp16 weight cache key on them) must see the update (host-only, ~0.1 ms for the 1390 tensors)
        torch.autograd.graph.increment_version(self._params)
        return self.loss

    def _recalibrate(self):
        """One eager step on the current batch measures the gradient operand scale afresh; when it moved by more than 2^4 from
        the frozen one the step is captured again."""
        self.optimizer.sync_step_from_device()
        self._eager_step()
        new = AG.last_grad_scale() if hasattr(AG, "last_grad_scale") else None
        old = self._scale_at_capture
        if new and old and (new / old > 16.0 or old / new > 16.0):
            torch.cuda.synchronize(self.lq.device)
            self._capture(warmup=1)

    def finish(self):
        """Brings the host-side optimizer state (step counts) up to date, e.g. before state_dict()."""
        self.optimizer.sync_step_from_device()
        # the weights changed behind the Python-side version counters: anything that caches on them must look again
        torch.autograd.graph.increment_version([p for g in self.optimizer.param_groups for p in g["params"]])
