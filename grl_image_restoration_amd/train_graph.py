"""A whole training step -- forward, loss, backward, FusedAdamW -- as ONE captured HIP graph.

The reference's step (engines/base.py:221-236: forward, L1 loss, manual_backward, optimizer step) is ~19 000 kernel launches on
this path at batch 8 x 64x64 (1390 parameter tensors, 40 blocks x ~60 autograd nodes x forward + backward): eager, the host
issues them slower than the GPU retires them (270 ms per step of which ~90 ms are kernels).  Captured once and replayed, the
step costs what its kernels cost.  What capture needs from the rest of the package:
  * no host read inside the step: the gradient operand scale is frozen at its warm-up value (autograd.frozen_grad_scale);
  * an optimizer launch that stays correct on replay: the step count lives on the device (FusedAdamW.enable_capture,
    GrlAdamWArgs.bias_corrections_dev);
  * static input buffers: the caller's batches are copied into them before each replay.

Data-parallel replicas (the reference's real training configuration: DistributedDataParallel over 8 GPUs, tools/trainer.py:135-142)
keep the graph: the step is captured as TWO graphs with the gradient all-reduce between them --
    graph A: zero_grad, forward, loss, backward, all 1390 gradients gathered into ONE flat fp32 buffer (77 MB for GRL-Base);
    eager:   one all-reduce of that buffer over the process group (RCCL over xGMI: a single large collective is what the
             per-link-bound ring wants; optionally bf16 on the wire, ``wire_bf16=True``: half the bytes);
    graph B: FusedAdamW reading the averaged gradients straight from the flat buffer (``p.grad`` are views of it; the 1 / world
             factor rides in the kernel's grad_scale).
The collective is NOT captured: the step does not depend on the communication library's graph support, and the same code runs
over gloo (tests: two replicas on one GPU).  There is no overlap of the all-reduce with the backward pass -- 2 x 77 MB x 7/8 over
xGMI is 1-2 ms of a 200 ms step.  A ``DistributedDataParallel`` wrapper must NOT exist around the model (its reducer hooks would
fire inside the capture): pass the bare module and a process group; the parameters are broadcast from the group's rank 0 first,
as the wrapper would have done.  The returned loss is the replica's own (as under DistributedDataParallel).  While a process group is
alive the graphs are captured in "thread_local" error mode: RCCL's watchdog thread queries events from outside the capturing thread.

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
import torch.distributed as dist

from . import autograd as AG


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer, loss_fn: Callable, lq: torch.Tensor, gt: torch.Tensor, warmup: int = 3,
                 recalibrate_every: int = 0, process_group=None, wire_bf16: bool = False):
        """``loss_fn(output, target) -> scalar``; ``lq`` / ``gt``: example batch (shapes are baked into the graph).  Runs ``warmup``
        eager steps (they DO update the weights), then captures one step.
        ``process_group``: data-parallel replicas (module docstring).  None = the default group when torch.distributed is initialised
        with more than one rank (a replica that silently skipped the all-reduce would diverge), False = never, or a group."""
        if not hasattr(optimizer, "enable_capture"):
            raise TypeError("GraphedTrainStep needs grl_image_restoration_amd.FusedAdamW (an optimizer whose step() can be captured)")
        if isinstance(model, torch.nn.parallel.DistributedDataParallel):
            raise TypeError("GraphedTrainStep owns the gradient all-reduce: pass the bare module and process_group=..., not a "
                            "DistributedDataParallel wrapper (its reducer hooks cannot run inside a captured step)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.lq, self.gt = lq.detach().clone(), gt.detach().clone()
        self.recalibrate_every = int(recalibrate_every)
        self._params = [p for g in optimizer.param_groups for p in g["params"]]
        self._group, self._world, self._wire_bf16 = None, 1, bool(wire_bf16)
        if process_group is not False and dist.is_available() and dist.is_initialized():
            grp = dist.group.WORLD if process_group is None else process_group
            if process_group is not None or dist.get_world_size(grp) > 1:
                self._group, self._world = grp, dist.get_world_size(grp)
        if self._group is not None:
            if len(optimizer.param_groups) != 1:
                raise ValueError("GraphedTrainStep with a process group: one parameter group (the flat gradient buffer is one launch's table)")
            self._broadcast_replica_state()
        self.collectives = 0        # gradient all-reduces issued so far (diagnostics / tests)
        model.train()
        self._capture(warmup)
        self.steps = 0

    # ------------------------------------------------------------------ data-parallel plumbing
    def _broadcast_replica_state(self):
        """Parameters and floating-point buffers <- the group's rank 0 (what DistributedDataParallel does when it wraps a module)."""
        src = dist.get_global_rank(self._group, 0) if hasattr(dist, "get_global_rank") else 0
        with torch.no_grad():
            ts = [p for p in self.model.parameters()] + [b for b in self.model.buffers() if b.is_floating_point()]
            for dt in {t.dtype for t in ts}:
                same = [t for t in ts if t.dtype == dt]
                flat = torch.cat([t.detach().reshape(-1) for t in same])
                dist.broadcast(flat, src=src, group=self._group)
                torch._foreach_copy_([t.detach() for t in same], [v.view_as(t) for v, t in zip(flat.split([t.numel() for t in same]), same)])
        torch.autograd.graph.increment_version(list(self.model.parameters()))

    def _all_reduce(self, flat):
        dist.all_reduce(flat, group=self._group)        # SUM; the 1 / world factor rides in FusedAdamW's grad_scale
        self.collectives += 1

    def _grad_of(self, p):
        """The flat gradient buffer of the data-parallel step is split by the optimizer's parameter list: every one of them must
        have received a gradient (what DistributedDataParallel(find_unused_parameters=False), the reference's setting, requires
        as well -- tools/trainer.py:135-142).  A frozen or unused parameter is named instead of failing on None."""
        if p.grad is None:
            name = next((k for k, q in self.model.named_parameters() if q is p), "<unnamed>")
            raise RuntimeError(f"GraphedTrainStep (data parallel): parameter {name} received no gradient; pass the optimizer only "
                               "parameters that take part in the step (requires_grad=False ones excluded)")
        return p.grad

    def _flat_grad_views(self, flat):
        return [v.view_as(p) for v, p in zip(flat.split([p.numel() for p in self._params]), self._params)]

    def _capture(self, warmup):
        model, optimizer, lq = self.model, self.optimizer, self.lq
        if not lq.is_cuda:
            # CPU tensors: no HIP graph to capture.  The step keeps its SHAPE -- [zero_grad, forward, loss, backward, flat gradient
            # buffer] -> all-reduce -> [optimizer on the averaged flat buffer] -- and runs eagerly on the model's composite torch
            # path (composite.py) and FusedAdamW's CPU arithmetic: what the gloo tests of the data-parallel plumbing use
            # (tests/test_train_graph_gloo.py, world sizes 2 and 8).  Nothing of it is MI355X work.
            for _ in range(max(0, warmup)):
                self._eager_step()
            self.graph = self.graph_update = None
            self._scale_at_capture, self._graph_grads = None, None
            return
        # (the warm-up steps run on a side stream, the capture on the graph's own: the AccumulateGrad nodes of the parameters move
        # between streams by design here -- the warning is silenced for the capture only and restored afterwards)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        cur = torch.cuda.current_stream(lq.device)
        side = torch.cuda.Stream(lq.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):             # (warm-up on a side stream, as torch.cuda.graph's recipe asks)
            for _ in range(max(0, warmup)):       # (0: the caller has just run an eager step of its own -- _recalibrate)
                self._eager_step()
        cur.wait_stream(side)
        torch.cuda.synchronize(lq.device)
        optimizer.enable_capture()
        optimizer.zero_grad(set_to_none=True)     # the gradients of the captured step come from the graph's own memory pool
        self.graph, self.graph_update = torch.cuda.CUDAGraph(), None
        # With a NCCL / RCCL process group alive its watchdog thread polls the events of finished collectives (the warm-up steps'
        # all-reduces) from ANOTHER thread; under the default "global" capture mode such a query while this thread captures is an error
        # that the watchdog turns into std::terminate.  "thread_local" restricts the check to the capturing thread.
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        with AG.frozen_grad_scale(), torch.cuda.graph(self.graph, capture_error_mode=mode):
            optimizer.zero_grad(set_to_none=True)
            self.loss = self.loss_fn(self.model(self.lq), self.gt)
            self.loss.backward()
            if self._group is None:
                optimizer.step()
            else:                                 # graph A ends with the gradients gathered into one flat buffer
                self._flat = torch.cat([self._grad_of(p).reshape(-1) for p in self._params])
                self._wire = self._flat.to(torch.bfloat16) if self._wire_bf16 else self._flat
        if self._group is not None:
            # graph B: the optimizer launch reads the (all-reduced) flat buffer -- p.grad become views of it, so its pointer table,
            # and whoever looks at p.grad after a step, see the averaged gradients.  Same memory pool as graph A.
            self._raw_grads = [p.grad for p in self._params]          # (graph A's own gradient tensors stay alive: it writes them)
            for p, v in zip(self._params, self._flat_grad_views(self._flat)):
                p.grad = v
            self.graph_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_update, pool=self.graph.pool(), capture_error_mode=mode):
                if self._wire_bf16:
                    self._flat.copy_(self._wire)
                optimizer.step(grad_scale=1.0 / self._world)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)
        self._scale_at_capture = AG.last_grad_scale() if hasattr(AG, "last_grad_scale") else None
        self._graph_grads = [p.grad for p in self._params]      # what replays write / the captured optimizer launch reads

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(self.lq), self.gt)
        loss.backward()
        if self._group is None:
            self.optimizer.step()
            return loss
        flat = torch.cat([self._grad_of(p).reshape(-1) for p in self._params])
        wire = flat.to(torch.bfloat16) if self._wire_bf16 else flat
        self._all_reduce(wire)
        if self._wire_bf16:
            flat.copy_(wire)
        for p, v in zip(self._params, self._flat_grad_views(flat)):
            p.grad = v
        self.optimizer.step(grad_scale=1.0 / self._world)
        return loss

    def __call__(self, lq: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        """One optimizer step on the batch; returns the loss tensor of the graph (valid until the next call)."""
        if lq.data_ptr() != self.lq.data_ptr():
            self.lq.copy_(lq, non_blocking=True)
        if gt.data_ptr() != self.gt.data_ptr():
            self.gt.copy_(gt, non_blocking=True)
        if self.graph is None:                          # CPU tensors (see _capture)
            self.loss = self._eager_step().detach()
            self.steps += 1
            return self.loss
        if self.recalibrate_every and self.steps and self.steps % self.recalibrate_every == 0:
            # the re-calibration step IS this batch's step (it used to be followed by a replay on the same batch -- two or
            # three optimizer updates, and Adam's step count advanced as often, for one batch)
            loss = self._recalibrate()
            self.steps += 1
            torch.autograd.graph.increment_version(self._params)
            return loss
        self.optimizer.refresh_capture_hyper()          # lr / weight decay of the captured optimizer launch <- param_groups
        self.graph.replay()
        if self.graph_update is not None:
            self._all_reduce(self._wire)
            self.graph_update.replay()
        self.steps += 1
        # the weights changed behind the Python-side version counters: an eval forward between replays (the inference plan and the
        # fp16 weight cache key on them) must see the update (host-only, ~0.1 ms for the 1390 tensors)
        torch.autograd.graph.increment_version(self._params)
        return self.loss

    def _recalibrate(self):
        """One EAGER step on the current batch -- it is that batch's optimizer step, its loss is returned -- measures the gradient
        operand scale afresh; when it moved by more than 2^4 from the frozen one the step is captured again (without further
        warm-up steps: nothing else updates the weights)."""
        self.optimizer.sync_step_from_device()
        loss = self._eager_step().detach()
        new = AG.last_grad_scale() if hasattr(AG, "last_grad_scale") else None
        old = self._scale_at_capture
        again = bool(new and old and (new / old > 16.0 or old / new > 16.0))
        if self._group is not None:                  # the replicas decide together: a re-capture runs collectives
            flag = torch.tensor([float(again)], device=self.lq.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._group)
            again = bool(flag.item() > 0)
        if again:
            torch.cuda.synchronize(self.lq.device)
            self._capture(warmup=0)
        else:                                        # replays keep writing the graph's own gradient tensors: p.grad shows those again
            for p, g in zip(self._params, self._graph_grads):
                p.grad = g
        return loss

    def finish(self):
        """Brings the host-side optimizer state (step counts) up to date, e.g. before state_dict()."""
        self.optimizer.sync_step_from_device()
        # the weights changed behind the Python-side version counters: anything that caches on them must look again
        torch.autograd.graph.increment_version([p for g in self.optimizer.param_groups for p in g["params"]])
