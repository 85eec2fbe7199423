"""TEST INFRASTRUCTURE -- regenerates tests/golden/*.npz by running the REAL reference network.

Run in the build container (needs /root/reference):   python -m oracle.make_golden

Each fixture freezes: the constructor kwargs, the weight seed (weights come from
grl_oracle.seeded_state_dict, so they are reproducible anywhere without the reference), the input
tensor and the output of the unmodified reference forward (fp32, CPU).  The GPU box has no
reference tree; its tests compare the HIP path and the oracle against these files.
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

# name, model, geometry, upscale, img_size (ctor), input (h, w), task
FIXTURES = [
    ("tiny_sr2_ckpt_64", "tiny", "sr_ckpt_df4", 2, 64, (64, 64), "sr"),        # BASELINE config 1
    ("tiny_sr2_yaml_64", "tiny", "yaml", 2, 64, (64, 64), "sr"),               # stripe_groups geometry
    ("small_dn_128", "small", "dn_df4", 1, 128, (128, 128), "dn"),             # BASELINE config 2
    ("base_sr4_yaml_32", "base", "yaml", 4, 32, (32, 32), "sr"),               # CAB on, 8x8 windows
    ("base_sr4_ckpt_64", "base", "sr_ckpt_df2", 4, 64, (64, 64), "sr"),        # BASELINE config 3 geometry
    ("base_deblur_ragged", "base", "deblur", 1, 96, (90, 100), "deblur"),      # reflect pad to 96x192, ws 12
]


def main():
    GRL = refshim.import_reference_grl()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, model, geom, up, size, hw, task in FIXTURES:
        cfg = make_config(model, geom, upscale=up, img_size=size)
        torch.manual_seed(0)
        ref = GRL(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        sd = O.seeded_state_dict(shapes, seed=0)
        full = ref.state_dict()
        full.update(sd)
        ref.load_state_dict(full, strict=True)
        lq, _ = O.synthetic_pair(task, hw, up, seed=1)
        lq = lq[..., : hw[0], : hw[1]].contiguous()
        with torch.no_grad():
            y = ref(lq)
            yo = O.grl_forward(lq, cfg, sd)
        err = (y - yo).abs().max().item()
        meta = dict(name=name, cfg=cfg, weight_seed=0, task=task, oracle_vs_reference_maxabs=err)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta), input=lq.numpy(), output=y.numpy())
        print(f"{name}: in {tuple(lq.shape)} out {tuple(y.shape)} oracle-vs-reference max|d| = {err:.3e}")


if __name__ == "__main__":
    main()
