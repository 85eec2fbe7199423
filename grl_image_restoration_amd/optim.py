"""Multi-tensor AdamW for the training path (BASELINE config 5).

The reference trains GRL with ``torch.optim.AdamW`` (config/optimizer/adamw.yaml: lr 2e-4, weight_decay 1e-4, torch defaults
otherwise; built in engines/base.py:451-470) over 1390 parameter tensors; a per-tensor update is host-launch bound.  ``FusedAdamW`` performs the
identical update (decoupled weight decay, bias correction as in torch.optim.AdamW, no amsgrad) for ALL tensors of a
parameter group in ONE launch of ``grl_adamw_step`` (csrc/grad.hip): a host-built list of 4096-element chunks, pointer tables
in device memory.  fp32 parameters on the GPU only -- there is no CPU path.
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
        super().load_state_dict(state_dict)
        self._tables = {}
        self._normalise_state()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}
        self._normalise_state()

    def _normalise_state(self):
        for p, s in self.state.items():
            if "step" in s and torch.is_tensor(s["step"]):
                s["step"] = int(s["step"].item())
            for k in ("exp_avg", "exp_avg_sq"):
                if k in s and (s[k].dtype != torch.float32 or not s[k].is_contiguous() or s[k].device != p.device):
                    s[k] = s[k].to(device=p.device, dtype=torch.float32).contiguous()

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
            for p in plist:
                if (not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32
                        or not p.grad.is_contiguous()):
                    raise RuntimeError("FusedAdamW: fp32 contiguous GPU parameters only (there is no CPU path)")
                s = self.state[p]
                if not s:
                    s["step"] = 0
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            steps = {int(self.state[p]["step"]) for p in plist}
            if len(steps) != 1:
                raise RuntimeError("FusedAdamW: parameters of one group must share their step count")
            step = steps.pop() + 1
            for p in plist:
                self.state[p]["step"] = step
            ent = self._group_tables(gi, plist)
            grads = torch.tensor([p.grad.data_ptr() for p in plist], dtype=torch.int64, device=plist[0].device)
            b1, b2 = group["betas"]
            args = L.GrlAdamWArgs(
                params=ent["params"].data_ptr(), grads=grads.data_ptr(), exp_avg=ent["exp_avg"].data_ptr(),
                exp_avg_sq=ent["exp_avg_sq"].data_ptr(), numel=ent["numel"].data_ptr(), weight_decay_flags=None,
                chunk_tensor=ent["chunk_tensor"].data_ptr(), chunk_offset=ent["chunk_offset"].data_ptr(), num_chunks=ent["n"],
                lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"],
                bias_correction1=1.0 - b1 ** step, bias_correction2_sqrt=(1.0 - b2 ** step) ** 0.5, grad_scale=grad_scale,
            )
            L.check(lib.grl_adamw_step(L.stream_ptr(), C.byref(args)), "grl_adamw_step")
            # the kernel wrote through raw pointers: tell autograd (and GRL's plan version stamp) that the tensors changed in place
            torch.autograd.graph.increment_version(plist)
        return loss
