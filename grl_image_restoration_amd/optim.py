"""Multi-tensor AdamW for the training path (BASELINE config 5).

The reference trains GRL with ``torch.optim.AdamW`` (config/optimizer/adamw.yaml: lr 2e-4, weight_decay 1e-4, torch defaults
otherwise; built in engines/base.py:451-470) over 1390 parameter tensors; a per-tensor update is host-launch bound.  ``FusedAdamW`` performs the
identical update (decoupled weight decay, bias correction as in torch.optim.AdamW, no amsgrad) for ALL tensors of a
parameter group in ONE launch of ``grl_adamw_step`` (csrc/grad.hip): a host-built list of 4096-element chunks, pointer tables
in device memory.  fp32 parameters on the GPU; CPU parameters take the same update as plain torch arithmetic (``_step_cpu``: like the
model's composite path it exists so that a module built without a GPU steps instead of raising -- never reached by GPU tensors).
"""
import ctypes as C
from typing import Iterable

import torch

from . import _lib as L

_CHUNK = 4096


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self._step_dev = None     # capture mode (enable_capture): per group, the step count as a device tensor
        self._hyper_dev = {}      # capture mode: per group, {lr, weight_decay} as a device tensor the captured launch reads
        self._frozen = {}         # capture mode: per group, the hyper-parameters a captured launch holds BY VALUE (betas, eps)
        self._pinned = {}         # per group: pinned host table of gradient pointers of the NEXT capture (one table per capture:
        self._pinned_all = []     # a replay copies from the table of its own graph; all of them are kept alive here)

    def _group_tables(self, gi, plist):
        """Device pointer / size tables of a parameter group (rebuilt when the set of tensors or their storage changes)."""
        # (the moment buffers are part of the key: load_state_dict replaces them while the parameters stay where they are)
        key = tuple((p.data_ptr(), p.numel(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in plist)
        ent = self._tables.get(gi)
        if ent is not None and ent["key"] == key:
            return ent
        dev = plist[0].device
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        chunk_t, chunk_o = [], []
        for t, p in enumerate(plist):
            for off in range(0, p.numel(), _CHUNK):
                chunk_t.append(t)
                chunk_o.append(off)
        st = [self.state[p] for p in plist]
        ent = dict(
            key=key,
            params=i64([p.data_ptr() for p in plist]),
            exp_avg=i64([s["exp_avg"].data_ptr() for s in st]),
            exp_avg_sq=i64([s["exp_avg_sq"].data_ptr() for s in st]),
            numel=i64([p.numel() for p in plist]),
            chunk_tensor=torch.tensor(chunk_t, dtype=torch.int32, device=dev),
            chunk_offset=i64(chunk_o),
            n=len(chunk_t),
        )
        self._tables[gi] = ent
        return ent

    def load_state_dict(self, state_dict):
        """Accepts this class's own state and the one of ``torch.optim.AdamW`` (what the reference's Lightning checkpoints hold
        under ``optimizer_states``: ``step`` as a 0-dim tensor per parameter)."""
        # capture mode: a captured launch holds the ADDRESSES of the moment buffers, of the pointer tables and of the step counters.
        # The loaded values therefore go INTO the existing buffers (torch's load_state_dict would replace the tensors: a replay
        # would keep updating the old moments and ignore the loaded ones -- ADVICE r4), and the tables stay.
        keep = {}
        if self._step_dev:
            keep = {p: (s["exp_avg"], s["exp_avg_sq"]) for p, s in self.state.items() if "exp_avg" in s}
        super().load_state_dict(state_dict)
        self._normalise_state()
        if keep:
            with torch.no_grad():
                for p, (m_old, v_old) in keep.items():
                    s = self.state.get(p)
                    if s is None or "exp_avg" not in s:
                        continue
                    if s["exp_avg"] is not m_old:
                        m_old.copy_(s["exp_avg"]); s["exp_avg"] = m_old
                    if s["exp_avg_sq"] is not v_old:
                        v_old.copy_(s["exp_avg_sq"]); s["exp_avg_sq"] = v_old
            for gi, group in enumerate(self.param_groups):
                steps = {int(self.state[p]["step"]) for p in group["params"] if p in self.state and "step" in self.state[p]}
                if len(steps) == 1:
                    self._step_dev[gi].fill_(steps.pop())
            self.refresh_capture_hyper()
        else:
            self._tables = {}

    def state_dict(self):
        """(In capture mode the step counts live on the device: they are read back first.)"""
        self.sync_step_from_device()
        return super().state_dict()

    def __setstate__(self, state):
        super().__setstate__(state)
        if not getattr(self, "_step_dev", None):   # (capture mode: load_state_dict keeps buffers and tables, see there)
            self._tables = {}
        self._normalise_state()

    def _normalise_state(self):
        for p, s in self.state.items():
            if "step" in s and torch.is_tensor(s["step"]):
                s["step"] = int(s["step"].item())
            for k in ("exp_avg", "exp_avg_sq"):
                if k in s and (s[k].dtype != torch.float32 or not s[k].is_contiguous() or s[k].device != p.device):
                    s[k] = s[k].to(device=p.device, dtype=torch.float32).contiguous()

    def enable_capture(self):
        """Makes ``step()`` replayable from a HIP graph (train_graph.GraphedTrainStep): the step count moves to device memory
        (incremented by a captured op; the bias corrections are computed from it on the device, in float64 like the host path,
        and handed to the kernel as GrlAdamWArgs.bias_corrections_dev) and the table of gradient pointers is copied from pinned
        host memory.  Call after at least one eager step (the moments must exist)."""
        first = not self._step_dev
        if not first:                            # already in capture mode (a second graph): the device counters are the truth and
            self.sync_step_from_device()         # keep their addresses -- an earlier graph increments them too
        else:
            self._step_dev = {}
        for gi, group in enumerate(self.param_groups):
            dev = group["params"][0].device
            if first:
                steps = {int(self.state[p]["step"]) for p in group["params"] if p in self.state and "step" in self.state[p]}
                if len(steps) != 1:
                    raise RuntimeError("FusedAdamW.enable_capture: run an eager step first (every parameter of a group needs the same step count)")
                self._step_dev[gi] = torch.full((1,), steps.pop(), dtype=torch.int64, device=dev)
                self._hyper_dev[gi] = torch.tensor([group["lr"], group["weight_decay"]], dtype=torch.float32, device=dev)
                self._frozen[gi] = (tuple(group["betas"]), group["eps"])
            # ONE pinned table of gradient pointers PER CAPTURE (allocated here: pinning host memory is not allowed while a stream is
            # capturing).  A shared table would be overwritten by the next capture, and a replay of the earlier graph would then
            # update from another graph's gradient buffers.
            host = torch.zeros(len(group["params"]), dtype=torch.int64)
            host = host.pin_memory() if dev.type == "cuda" else host
            self._pinned[gi] = host
            self._pinned_all.append(host)

    def refresh_capture_hyper(self):
        """Capture mode: brings the device copy of {lr, weight_decay} up to date with ``param_groups`` -- call before every replay
        of a captured step (GraphedTrainStep does), so that LR schedulers act on replays.  betas / eps are held by value in a
        captured launch: changing them after a capture raises here instead of being silently ignored."""
        if not self._step_dev:
            return
        for gi, group in enumerate(self.param_groups):
            if self._frozen.get(gi) != (tuple(group["betas"]), group["eps"]):
                raise RuntimeError("FusedAdamW: betas / eps changed after a step was captured; they are frozen in the graph -- re-capture")
            hd = self._hyper_dev[gi]
            want = (float(group["lr"]), float(group["weight_decay"]))
            if getattr(hd, "_host_copy", None) != want:
                hd.copy_(torch.tensor(want, dtype=torch.float32))
                hd._host_copy = want

    def sync_step_from_device(self):
        """After graph replays: the host-side ``state[p]['step']`` (what state_dict() saves) <- the device counters."""
        if self._step_dev:
            for gi, group in enumerate(self.param_groups):
                n = int(self._step_dev[gi].item())
                for p in group["params"]:
                    if p in self.state and "step" in self.state[p]:
                        self.state[p]["step"] = n

    def _step_cpu(self, group, plist, grad_scale):
        """torch.optim.AdamW's single-tensor update on CPU tensors (decoupled weight decay, bias correction), gradients multiplied by
        ``grad_scale`` first.  Exists for the same reason as the model's composite path (composite.py): a module built without a GPU
        -- the gloo tests of the data-parallel step, a unit test of a surrounding engine -- steps instead of raising.  Never reached
        by GPU parameters; nothing here is timed or claimed as MI355X work."""
        b1, b2 = group["betas"]
        lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
        for p in plist:
            s = self.state[p]
            if not s:
                s["step"] = 0
                s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            s["step"] = int(s["step"]) + 1
            g = p.grad * grad_scale if grad_scale != 1.0 else p.grad
            p.mul_(1.0 - lr * wd)
            s["exp_avg"].lerp_(g, 1.0 - b1)
            s["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            bc1, bc2s = 1.0 - b1 ** s["step"], (1.0 - b2 ** s["step"]) ** 0.5
            p.addcdiv_(s["exp_avg"], (s["exp_avg_sq"].sqrt() / bc2s).add_(eps), value=-lr / bc1)
        torch.autograd.graph.increment_version(plist)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            if all(not p.is_cuda for p in plist):       # CPU tensors: the same update as plain torch arithmetic (module docstring)
                self._step_cpu(group, plist, grad_scale)
                continue
            for p in plist:
                if (not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32
                        or not p.grad.is_contiguous()):
                    raise RuntimeError("FusedAdamW: fp32 contiguous GPU parameters only (a group is either all on the GPU or all on the CPU)")
                s = self.state[p]
                if not s:
                    s["step"] = 0
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            steps = {int(self.state[p]["step"]) for p in plist}
            if len(steps) != 1:
                raise RuntimeError("FusedAdamW: parameters of one group must share their step count")
            step = steps.pop() + 1
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing and self._step_dev is None:
                raise RuntimeError("FusedAdamW: call enable_capture() before capturing step() in a graph")
            if not capturing:              # (a captured step does not execute: the device counter advances on replay only)
                for p in plist:
                    self.state[p]["step"] = step
            b1, b2 = group["betas"]
            bc_dev = None
            if self._step_dev is not None and not capturing:
                self.refresh_capture_hyper()      # (eager steps between replays read the same device copy)
            if self._step_dev is not None:
                sd = self._step_dev[gi]
                sd.add_(1)
                t = sd.double()
                bc_dev = torch.cat([1.0 - b1 ** t, (1.0 - b2 ** t).sqrt()]).float()   # (1 - 0.999^t cancels in float32: 6e-5 off at t = 2)
            ent = self._group_tables(gi, plist)
            if capturing:
                host = self._pinned[gi][: len(plist)]
                host.copy_(torch.tensor([p.grad.data_ptr() for p in plist], dtype=torch.int64))
                grads = host.to(plist[0].device, non_blocking=True)
            else:
                grads = torch.tensor([p.grad.data_ptr() for p in plist], dtype=torch.int64, device=plist[0].device)
            args = L.GrlAdamWArgs(
                params=ent["params"].data_ptr(), grads=grads.data_ptr(), exp_avg=ent["exp_avg"].data_ptr(),
                exp_avg_sq=ent["exp_avg_sq"].data_ptr(), numel=ent["numel"].data_ptr(), weight_decay_flags=None,
                chunk_tensor=ent["chunk_tensor"].data_ptr(), chunk_offset=ent["chunk_offset"].data_ptr(), num_chunks=ent["n"],
                lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"],
                bias_correction1=1.0 - b1 ** step, bias_correction2_sqrt=(1.0 - b2 ** step) ** 0.5, grad_scale=grad_scale,
                bias_corrections_dev=bc_dev.data_ptr() if bc_dev is not None else None,
                hyper_dev=self._hyper_dev[gi].data_ptr() if (self._step_dev is not None and gi in self._hyper_dev) else None,
            )
            L.check(lib.grl_adamw_step(L.stream_ptr(), C.byref(args)), "grl_adamw_step")
            # the kernel wrote through raw pointers: tell autograd (and GRL's plan version stamp) that the tensors changed in place
            torch.autograd.graph.increment_version(plist)
        return loss
