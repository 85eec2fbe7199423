"""TEST INFRASTRUCTURE -- freezes one training-step gradient of the REAL reference network (SURVEY 8(f) N1 groundwork).

Run in the build container (needs /root/reference):   python -m oracle.make_golden_grads

BASELINE config 5 in miniature: GRL-Base blocks (CAB on, checkpoint geometry), x4 SR, 64x64 LQ, L1 loss
(config/loss/l1.yaml), module in eval mode (DropPath is the identity; the stochastic-depth mask of train mode is not
reproducible across implementations).  Stored: input, target, loss, the gradient w.r.t. the input, the L2 norm of every
parameter gradient and the full gradient of every tensor with <= 4096 elements (norm affines, biases, logit scales,
the CPB-MLP output layers, SE layers).  A future HIP backward is checked against these without the reference tree.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from grl_image_restoration_amd.presets import make_config  # noqa: E402
from oracle import grl_oracle as O  # noqa: E402
from oracle import refshim  # noqa: E402

NAME = "train_base2x2_sr4_64"


def reference_step(cfg, sd, lq, gt):
    GRL = refshim.import_reference_grl()
    torch.manual_seed(0)
    ref = GRL(**cfg).eval()
    full = ref.state_dict()
    full.update(sd)
    ref.load_state_dict(full, strict=True)
    lq = lq.clone().requires_grad_(True)
    loss = (ref(lq) - gt).abs().mean()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    return loss.detach(), lq.grad.detach().clone(), grads


def oracle_step(cfg, sd, lq, gt):
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    lq = lq.clone().requires_grad_(True)
    loss = (O.grl_forward(lq, cfg, sd) - gt).abs().mean()
    loss.backward()
    return loss.detach(), lq.grad.detach().clone(), {k: v.grad.detach().clone() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}


def make_case():
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[2, 2], num_heads_window=[3, 3], num_heads_stripe=[3, 3])
    GRL = refshim.import_reference_grl()
    shapes = {k: tuple(v.shape) for k, v in GRL(**cfg).state_dict().items()}
    sd = O.seeded_state_dict(shapes, seed=0)
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, seed=1)
    return cfg, sd, lq, gt


def main():
    cfg, sd, lq, gt = make_case()
    loss, gin, grads = reference_step(cfg, sd, lq, gt)
    lo, gio, go = oracle_step(cfg, sd, lq, gt)
    rel = max(((grads[k] - go[k]).norm() / grads[k].norm().clamp_min(1e-20)).item() for k in grads)
    meta = dict(name=NAME, cfg=cfg, weight_seed=0, loss=float(loss), oracle_vs_reference_max_rel_grad=rel,
                oracle_vs_reference_loss=abs(float(loss) - float(lo)))
    arrays = dict(meta=json.dumps(meta), input=lq.numpy(), target=gt.numpy(), grad_input=gin.numpy(),
                  grad_norm_names=json.dumps(sorted(grads)), grad_norms=np.array([grads[k].norm().item() for k in sorted(grads)], dtype=np.float64))
    for k, g in grads.items():
        if g.numel() <= 4096:
            arrays["grad::" + k] = g.numpy()
    out = os.path.join(ROOT, "tests", "golden_grads")
    os.makedirs(out, exist_ok=True)
    np.savez_compressed(os.path.join(out, NAME + ".npz"), **arrays)
    print(f"{NAME}: loss {float(loss):.6f}, {len(grads)} parameter gradients, oracle-vs-reference max rel grad error {rel:.3e}")


if __name__ == "__main__":
    main()
