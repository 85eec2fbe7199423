"""Which lines of the package issue the torch "glue" kernels of one training step (BASELINE config 5): every aten op that launches
a fill / copy / cat / element-wise / reduction kernel is attributed to the innermost frame inside grl_image_restoration_amd and
counted (calls, elements written).  A captured step costs the sum of its kernels, so the table is the to-do list of the training path.
Usage (GPU box): python tools/train_glue_sites.py [batch] [rows]"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from grl_image_restoration_amd import GRL, FusedAdamW, baseline_config

PKG = os.sep + "grl_image_restoration_amd" + os.sep
VIEWS = {"view", "_unsafe_view", "reshape", "t", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze", "detach", "alias",
         "as_strided", "unbind", "split", "split_with_sizes", "unflatten", "flatten", "narrow", "chunk", "empty", "empty_like", "empty_strided",
         "new_empty", "_to_copy_view", "lift_fresh", "is_contiguous", "size", "stride", "numel", "sym_size", "storage_offset", "new_empty_strided",
         "_local_scalar_dense", "item", "result_type", "can_cast", "is_same_size", "view_as", "unsafe_split", "_reshape_alias", "sym_numel",
         "sym_stride", "sym_storage_offset", "dim", "is_pinned", "record_stream"}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(lambda: [0, 0])
        self.phase = "fwd"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEWS or func.namespace == "grl":
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if PKG in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                break
        n = 0
        for o in (out if isinstance(out, (tuple, list)) else (out,)):
            if isinstance(o, torch.Tensor):
                n += o.numel()
        r = self.rows[(self.phase, name, site)]
        r[0] += 1
        r[1] += n
        return out


def main():
    bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    cfg = baseline_config(5)
    torch.manual_seed(0)
    model = GRL(**cfg).cuda().train()
    opt = FusedAdamW(model.parameters(), lr=2e-4, weight_decay=1e-4)
    lq = torch.rand(bsz, 3, 64, 64, device="cuda")
    gt = torch.rand(bsz, 3, 64 * cfg["upscale"], 64 * cfg["upscale"], device="cuda")
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        (model(lq) - gt).abs().mean().backward()
        opt.step()
    torch.cuda.synchronize()
    s = Sites()
    with s:
        opt.zero_grad(set_to_none=True)
        loss = (model(lq) - gt).abs().mean()
        s.phase = "bwd"
        loss.backward()
        s.phase = "opt"
        opt.step()
    torch.cuda.synchronize()
    rows = sorted(s.rows.items(), key=lambda kv: -kv[1][0])
    total = sum(v[0] for v in s.rows.values())
    print(f"# one training step, batch {bsz}: {total} aten calls that launch kernels (views / allocations / grl ops excluded)")
    print(f"{'phase':4s} {'op':28s} {'calls':>6s} {'Melem':>9s}  site")
    for (ph, name, site), (c, n) in rows[:nrows]:
        print(f"{ph:4s} {name:28s} {c:6d} {n / 1e6:9.2f}  {site}")
    by_op = collections.defaultdict(lambda: [0, 0])
    for (ph, name, site), (c, n) in s.rows.items():
        by_op[name][0] += c
        by_op[name][1] += n
    print("# by op")
    for name, (c, n) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"{name:28s} {c:6d} {n / 1e6:9.2f}")


if __name__ == "__main__":
    main()
