"""What does `auto`'s block-by-block calibration (GRL._calibrated_plan) decide on checkpoint-like weight draws, what does the
decision cost, and what error against the float64 truth results?  (diagnostic, GPU box; the truths are tests/golden/seeds/*)"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grl_image_restoration_amd import GRL, make_config  # noqa: E402
from oracle import grl_oracle as O  # noqa: E402


def truth(tag, wseed, dseed):
    p = os.path.join(ROOT, "tests", "golden", "seeds", f"{tag}_{wseed}_{dseed}.npz")
    if not os.path.isfile(p):
        return None
    return torch.from_numpy(np.load(p)["truth"]).double()


def timed(m, x, n=5):
    with torch.no_grad():
        m(x); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def run(side, wseed, dseed, tag, batch_for_timing=4):
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=side)
    shapes = {k: tuple(v.shape) for k, v in GRL(**cfg).state_dict().items()}
    sd = O.seeded_state_dict(shapes, wseed, logit_scale_mean=math.log(100.0))
    lq, _ = O.synthetic_pair("sr", (side, side), 4, batch=1, seed=dseed)
    lq = lq[..., :side, :side].contiguous().cuda()
    want = truth(tag, wseed, dseed)
    xb = lq.expand(batch_for_timing, -1, -1, -1).contiguous()
    out = {}
    for mode, env in (("fast", {"GRL_CALIBRATE": "0"}), ("auto", {}), ("high", None)):
        for k in ("GRL_CALIBRATE",):
            os.environ.pop(k, None)
        os.environ.update(env or {})
        m = GRL(**cfg, precision="high" if mode == "high" else "auto").eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            y = m(lq).double().cpu()
        torch.cuda.synchronize(); build = time.perf_counter() - t0
        ms = timed(m, xb)
        e = (y - want).abs().max().item() if want is not None else float("nan")
        r = (y - want).pow(2).mean().sqrt().item() if want is not None else float("nan")
        out[mode] = y
        cal = getattr(m, "calibration", None)
        cal = {k: (round(v, 7) if isinstance(v, float) else v) for k, v in (cal or {}).items() if k != "split_blocks"}
        print(f"  side {side} seeds ({wseed},{dseed}) {mode:5s} [{m.precision}] first call {build:.2f} s, {ms:.2f} ms / {batch_for_timing} tiles; "
              f"vs fp64 truth max {e:.3e} rms {r:.3e}  {cal if mode == 'auto' else ''}", flush=True)
    d = (out["fast"] - out["high"])
    print(f"     fast vs high on the test input: max {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}")


if __name__ == "__main__":
    for w, d in ((0, 1), (11, 21), (12, 22), (13, 23), (14, 24), (15, 25)):
        run(64, w, d, "base_sr4")
    if "--big" in sys.argv:
        run(256, 11, 21, "base_sr4_256")
        run(256, 0, 1, "none")
