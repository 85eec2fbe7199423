"""Where does the gradient error of the training step come from?  (diagnostic, GPU box; not part of the product)

Runs the reference-gradient fixture (tests/golden_grads/train_base2x2_sr4_64.npz) through the training path several times and
prints, per variant, the worst tensors against the reference's gradients.  Variants replace one class of HIP contraction by the
plain fp32 torch expression of composite.py ON THE GPU (debug only -- the product never does that), or only its backward, so
that the error budget per site can be read off:

    hip            everything on the HIP kernels (what the test checks); repeated --runs times: run-to-run spread
    attn=torch     attention forward + backward in fp32 torch
    attn_bwd=torch HIP attention forward, fp32 torch backward (recomputed from the fp32 planes)
    linear=torch / conv=torch

usage: python tools/grad_budget.py [--runs 5] [--variants hip,attn=torch,...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from grl_image_restoration_amd import GRL, autograd as AG, composite  # noqa: E402
from oracle import grl_oracle as O  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


class _HipFwdTorchBwd(torch.autograd.Function):
    """HIP attention forward, backward by autograd through composite.attention on the same fp32 planes."""

    @staticmethod
    def forward(ctx, q, k, v, table, geo_id):
        geo = _GEOS[geo_id]
        with torch.no_grad():
            out = _ORIG_ATT(q, k, v, table, geo)
        ctx.save_for_backward(q, k, v, table)
        ctx.geo_id = geo_id
        return out

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, table = ctx.saved_tensors
        geo = _GEOS[ctx.geo_id]
        with torch.enable_grad():
            qq, kk, vv, tt = (t.detach().requires_grad_(True) for t in (q, k, v, table))
            o = composite.attention(qq, kk, vv, tt, list(geo["q"]), list(geo["k"]), geo["B"], geo["nh"], geo["d"], bool(geo["masked"]))
            gq, gk, gv, gt = torch.autograd.grad(o, (qq, kk, vv, tt), d_o)
        return gq, gk, gv, gt, None


_GEOS = []
_ORIG_ATT = AG.AttentionFn.apply
_ORIG_LIN = AG.linear
_ORIG_CONV = AG.conv3x3


def set_variant(name):
    AG.AttentionFn.apply = staticmethod(_ORIG_ATT)
    AG.linear, AG.conv3x3 = _ORIG_LIN, _ORIG_CONV
    for part in name.split("+"):
        if part == "hip":
            continue
        site, how = part.split("=")
        assert how == "torch", part
        if site == "attn":
            AG.AttentionFn.apply = staticmethod(lambda q, k, v, table, geo: composite.attention(
                q, k, v, table, list(geo["q"]), list(geo["k"]), geo["B"], geo["nh"], geo["d"], bool(geo["masked"])))
        elif site == "attn_bwd":
            def f(q, k, v, table, geo):
                _GEOS.append(geo)
                return _HipFwdTorchBwd.apply(q, k, v, table, len(_GEOS) - 1)
            AG.AttentionFn.apply = staticmethod(f)
        elif site == "linear":
            AG.linear = composite.linear
        elif site == "conv":
            AG.conv3x3 = composite.conv3x3
        else:
            raise SystemExit(f"unknown site {site}")


def one_run(meta, z, sd):
    m = GRL(**meta["cfg"]).eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x = torch.from_numpy(z["input"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(z["target"]).cuda()
    loss = (m(x) - gt).abs().mean()
    loss.backward()
    names = json.loads(str(z["grad_norm_names"]))
    grads = {k: p.grad for k, p in m.named_parameters()}
    errs = {"d/dx": rel(x.grad, torch.from_numpy(z["grad_input"]))}
    for k, n in zip(names, z["grad_norms"]):
        errs["norm::" + k] = abs(grads[k].norm().item() - n) / max(n, 1e-12)
    for k in z.files:
        if k.startswith("grad::"):
            errs[k] = rel(grads[k[6:]], torch.from_numpy(z[k]))
    return loss.item(), errs, {k: g.detach().clone() for k, g in grads.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--variants", default="hip,attn_bwd=torch,attn=torch,linear=torch,conv=torch,attn=torch+linear=torch+conv=torch")
    ap.add_argument("--top", type=int, default=8)
    a = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden_grads", "train_base2x2_sr4_64.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    shapes = {k: tuple(v.shape) for k, v in GRL(**meta["cfg"]).state_dict().items()}
    sd = O.seeded_state_dict(shapes, meta["weight_seed"])
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("GRL_")})
    for var in a.variants.split(","):
        set_variant(var)
        first = None
        for r in range(a.runs if var == "hip" else 1):
            loss, errs, grads = one_run(meta, z, sd)
            ranked = sorted(errs.items(), key=lambda t: -t[1])
            cpb = max((e for k, e in errs.items() if k.startswith("grad::") and "cpb_mlp" in k), default=0.0)
            other = max((e for k, e in errs.items() if k.startswith("grad::") and "cpb_mlp" not in k), default=0.0)
            nrm = max(e for k, e in errs.items() if k.startswith("norm::"))
            same = "" if first is None else f" bit-identical to run 0: {all(torch.equal(grads[k], first[k]) for k in grads)}"
            print(f"[{var}] run {r}: loss {loss:.6f} (ref {meta['loss']:.6f}) d/dx {errs['d/dx']:.3e} worst cpb {cpb:.3e} worst other small {other:.3e} worst norm {nrm:.3e}{same}")
            if r == 0:
                first = grads
                for k, e in ranked[: a.top]:
                    print(f"      {e:.3e}  {k}")
    set_variant("hip")




def noise():
    """Two runs of the HIP path: which tensors differ from run to run, and at which attention-backward call does the
    difference enter (inputs and outputs of every grl_attention_bwd call recorded)?"""
    from grl_image_restoration_amd import ops

    z = np.load(os.path.join(ROOT, "tests", "golden_grads", "train_base2x2_sr4_64.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    shapes = {k: tuple(v.shape) for k, v in GRL(**meta["cfg"]).state_dict().items()}
    sd = O.seeded_state_dict(shapes, meta["weight_seed"])
    orig = ops.attention_bwd
    rec = []

    def spy(q, k, v, o, d_o, lse, **kw):
        out = orig(q, k, v, o, d_o, lse, **kw)
        rec[-1].append(dict(q=q.t.clone(), k=k.t.clone(), v=v.t.clone(), o=o.t.clone(), d_o=d_o.clone(), lse=lse.clone(),
                            table=kw["table"].clone(), dq=out[0].clone(), dk=out[1].clone(), dv=out[2].clone(), dtab=out[3].clone(),
                            shape=(tuple(q.t.shape), tuple(k.t.shape))))
        return out

    ops.attention_bwd = spy
    runs = []
    for r in range(2):
        rec.append([])
        runs.append(one_run(meta, z, sd))
    ops.attention_bwd = orig
    g0, g1 = runs[0][2], runs[1][2]
    diffs = sorted(((rel(g1[k], g0[k]), k) for k in g0), reverse=True)
    print("run-to-run relative difference, noisiest parameter gradients:")
    for e, k in diffs[:10]:
        print(f"   {e:.3e}  {k}")
    print("attention-backward calls in backward order (relative run-to-run difference of inputs | outputs):")
    for i, (a, b) in enumerate(zip(rec[0], rec[1])):
        f = lambda n: f"{n} {rel(b[n], a[n]):.1e}"
        print(f"   call {i} {a['shape']}: " + " ".join(f(n) for n in ("q", "k", "v", "o", "lse", "table", "d_o")) + " | " +
              " ".join(f(n) for n in ("dq", "dk", "dv", "dtab")) + f"   |dtab| {a['dtab'].norm().item():.3e} sum|.| {a['dtab'].abs().sum().item():.3e}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "noise":
        noise()
    else:
        main()
