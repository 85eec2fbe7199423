"""TEST INFRASTRUCTURE -- regenerates tests/golden/*.npz by running the REAL reference network.

Run in the build container (needs /root/reference):   python -m oracle.make_golden [name ...]

Each fixture freezes: the constructor kwargs, the weight seed (weights come from
grl_oracle.seeded_state_dict, so they are reproducible anywhere without the reference), the input
tensor and the output of the unmodified reference forward (fp32, CPU).  The GPU box has no
reference tree; its tests compare the HIP path and the oracle against these files.

Large outputs (the 256x256 bench tile: 3 x 1024 x 1024 per tile) are stored as 16-bit fixed point over the
output's own [lo, hi] range (step <= 3e-5, i.e. 3 % of the 1e-3 parity bar; ``output_q``/``q_lo``/``q_step``)
and, for the second tile of a batch, as a 4x-strided subsample -- tests.util.load_golden undoes both.
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from grl_image_restoration_amd.presets import make_config  # noqa: E402
from oracle import grl_oracle as O  # noqa: E402
from oracle import refshim  # noqa: E402

LN100 = math.log(100.0)
# name, model, geometry, upscale, img_size (ctor), input (h, w), task, extras
FIXTURES = [
    ("tiny_sr2_ckpt_64", "tiny", "sr_ckpt_df4", 2, 64, (64, 64), "sr", {}),        # BASELINE config 1
    ("tiny_sr2_yaml_64", "tiny", "yaml", 2, 64, (64, 64), "sr", {}),               # stripe_groups geometry
    ("small_dn_128", "small", "dn_df4", 1, 128, (128, 128), "dn", {}),             # BASELINE config 2
    ("base_sr4_yaml_32", "base", "yaml", 4, 32, (32, 32), "sr", {}),               # CAB on, 8x8 windows
    ("base_sr4_ckpt_64", "base", "sr_ckpt_df2", 4, 64, (64, 64), "sr", {}),        # BASELINE config 3 geometry
    ("base_deblur_ragged", "base", "deblur", 1, 96, (90, 100), "deblur", {}),      # reflect pad to 96x192, ws 12
    # logit scales at the clamp (exp(min(., ln 100)), efficient.py:39): about half of the heads are clamped to 100
    ("base_sr4_ckpt_64_hiscale", "base", "sr_ckpt_df2", 4, 64, (64, 64), "sr", dict(sd=dict(logit_scale_mean=LN100))),
    ("tiny_sr2_ckpt_64_hiscale", "tiny", "sr_ckpt_df4", 2, 64, (64, 64), "sr", dict(sd=dict(logit_scale_mean=LN100))),
    # the bench shape itself: two 256x256 LQ tiles (4x4 stripes, 8x8 windows per tile, both tile groups of the 2-stream split)
    ("base_sr4_ckpt_256", "base", "sr_ckpt_df2", 4, 256, (256, 256), "sr", dict(batch=2, quantise=True, sub2=4)),
    # the bench shape with checkpoint-like logit scales (around ln 100: about half of the heads at the clamp) -- the
    # bench's trained_scales leg runs on this kernel path, so its parity is pinned at the benchmarked shape
    ("base_sr4_ckpt_256_hiscale", "base", "sr_ckpt_df2", 4, 256, (256, 256), "sr", dict(batch=1, quantise=True, sd=dict(logit_scale_mean=LN100))),
    # BASELINE config 4 at a real tile: 384x384 (2 x 4 tiles of the 1280x720 frame), window 12, stripes 48x96, anchors /4
    ("base_deblur_384", "base", "deblur", 1, 384, (384, 384), "deblur", dict(frame=(720, 1280), tile=384, overlap=48, tiles=(0, 5))),
    # round 4 (VERDICT r3 #1a): checkpoint-like logit scales for BASELINE configs 2 and 4 as well (around ln 100: about half of the
    # heads at the clamp), and the regime in between -- scales around 30, above anything random init draws and below the
    # threshold (GRL_HIQ_SCALE) at which the q / k / anchor projection switches to split operands
    # `conditioning`: the fixture also records how far the REFERENCE's own fp32 output moves when the input is multiplied by
    # 1 + 2^-20 N(0,1) (250 x below 8-bit image quantisation).  With these seeded random weights GRL-Small at the clamp is
    # a chaotic amplifier: 2.9e-3 -- its output is not defined to the 1e-3 parity bar, whatever the arithmetic (the others: ~1e-5).
    # The test reads the number and only asserts that fixture against gross errors; small_dn_128_midscale (scales around 40, still
    # above what random init draws and above the level at which `auto` moves the narrow models to split operands) is the
    # well-conditioned config-2 fixture for that regime.
    ("small_dn_128_hiscale", "small", "dn_df4", 1, 128, (128, 128), "dn", dict(sd=dict(logit_scale_mean=LN100), conditioning=True)),
    ("small_dn_128_midscale", "small", "dn_df4", 1, 128, (128, 128), "dn", dict(sd=dict(logit_scale_mean=math.log(40.0)), conditioning=True)),
    ("base_deblur_384_hiscale", "base", "deblur", 1, 384, (384, 384), "deblur",
     dict(frame=(720, 1280), tile=384, overlap=48, tiles=(5,), sd=dict(logit_scale_mean=LN100), conditioning=True)),
    ("base_sr4_ckpt_64_midscale", "base", "sr_ckpt_df2", 4, 64, (64, 64), "sr", dict(sd=dict(logit_scale_mean=math.log(30.0)), conditioning=True)),
]


def quantise(y: torch.Tensor):
    lo, hi = float(y.min()), float(y.max())
    step = (hi - lo) / 65535.0
    q = torch.round((y - lo) / step).clamp_(0, 65535).to(torch.int32).numpy().astype(np.uint16)
    return q, lo, step


def main(only=None):
    GRL = refshim.import_reference_grl()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, model, geom, up, size, hw, task, ex in FIXTURES:
        if only and name not in only:
            continue
        cfg = make_config(model, geom, upscale=up, img_size=size)
        torch.manual_seed(0)
        ref = GRL(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        sd_kw = ex.get("sd", {})
        sd = O.seeded_state_dict(shapes, seed=0, **sd_kw)
        full = ref.state_dict()
        full.update(sd)
        ref.load_state_dict(full, strict=True)
        extra_meta = {}
        if "frame" in ex:
            # tiles of a synthetic 1280x720 frame in the reference's tile order (engines/base.py:96-99)
            fh, fw = ex["frame"]
            frame, _ = O.synthetic_pair(task, (fh, fw), up, seed=1)
            frame = frame[..., :fh, :fw].contiguous()
            t, ov = ex["tile"], ex["overlap"]
            from oracle.engine_oracle import tile_origins
            origins = [(a, b) for a in tile_origins(fh, t, ov) for b in tile_origins(fw, t, ov)]
            lq = torch.cat([frame[..., a : a + t, b : b + t] for a, b in (origins[i] for i in ex["tiles"])], 0).contiguous()
            extra_meta = dict(frame=[fh, fw], tile=t, overlap=ov, tile_ids=list(ex["tiles"]), origins=[list(o) for o in origins], frame_seed=1)
        else:
            lq, _ = O.synthetic_pair(task, hw, up, batch=ex.get("batch", 1), seed=1)
            lq = lq[..., : hw[0], : hw[1]].contiguous()
        with torch.no_grad():
            y = ref(lq)
            yo = O.grl_forward(lq, cfg, sd)
        err = (y - yo).abs().max().item()
        if ex.get("conditioning"):
            with torch.no_grad():
                yp = ref(lq * (1 + 2.0**-20 * torch.randn(lq.shape, generator=torch.Generator().manual_seed(3))))
            extra_meta["conditioning_2e-20"] = (yp - y).abs().max().item()
        meta = dict(name=name, cfg=cfg, weight_seed=0, sd_kwargs=sd_kw, task=task, oracle_vs_reference_maxabs=err, **extra_meta)
        arrays = dict(input=lq.numpy())
        if ex.get("quantise"):
            q, lo, step = quantise(y[:1])
            arrays.update(output_q=q, q_lo=np.float64(lo), q_step=np.float64(step))
            s = ex.get("sub2", 1)
            if y.shape[0] > 1:
                arrays["output_b1_sub"] = y[1:, :, ::s, ::s].contiguous().numpy()
            meta.update(sub2=s, out_shape=list(y.shape),
                        out_sum=[float(y[i].double().sum()) for i in range(y.shape[0])],
                        out_abs_sum=[float(y[i].double().abs().sum()) for i in range(y.shape[0])])
        else:
            arrays["output"] = y.numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta), **arrays)
        print(f"{name}: in {tuple(lq.shape)} out {tuple(y.shape)} oracle-vs-reference max|d| = {err:.3e}", flush=True)


def main_fp64(only=None):
    """Adjudication of the checkpoint-like fixtures (VERDICT r4 #4a): the unmodified reference in FLOAT64 on the same inputs and
    weights.  tests/golden/fp64/<name>.npz stores ref_fp32 - ref_fp64 (float32, on a strided lattice for the large outputs) and its
    maximum: the GPU test then knows how far the reference's own fp32 arithmetic is from the exact network, and asserts
    |hip - fp64| <= max(1e-3, 2 x |ref_fp32 - fp64|)."""
    GRL = refshim.import_reference_grl()
    out_dir = os.path.join(ROOT, "tests", "golden", "fp64")
    os.makedirs(out_dir, exist_ok=True)
    for name, model, geom, up, size, hw, task, ex in FIXTURES:
        if "scale" not in name or (only and name not in only):
            continue
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
        cfg = make_config(model, geom, upscale=up, img_size=size)
        torch.manual_seed(0)
        ref = GRL(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        sd = O.seeded_state_dict(shapes, seed=0, **ex.get("sd", {}))
        full = ref.state_dict()
        full.update(sd)
        ref.load_state_dict(full, strict=True)
        lq = torch.from_numpy(z["input"])
        with torch.no_grad():
            y32 = ref(lq)
            y64 = ref.double()(lq.double())
        d = (y32.double() - y64)
        s_ = 4 if d.numel() > 600_000 else 1
        meta = dict(name=name, stride=s_, max_abs_ref32_minus_fp64=float(d.abs().max()), rms=float(d.pow(2).mean().sqrt()),
                    out_shape=list(d.shape))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta),
                            ref32_sub=y32[..., ::s_, ::s_].contiguous().numpy(), ref32_minus_fp64_sub=d[..., ::s_, ::s_].float().contiguous().numpy())
        print(f"{name}: max|ref_fp32 - ref_fp64| = {meta['max_abs_ref32_minus_fp64']:.3e} (rms {meta['rms']:.3e}), lattice stride {s_}", flush=True)


SEED_CASES = [   # tag, model, geometry, upscale, (h, w), task -- tests/test_gpu_model.py::test_second_seeds_at_clamp_scales_vs_pinned_oracle
    ("base_sr4", "base", "sr_ckpt_df2", 4, (64, 64), "sr"),
    ("small_dn", "small", "dn_df4", 1, (128, 128), "dn"),
    ("base_deblur", "base", "deblur", 1, (96, 192), "deblur"),
]
SEED_PAIRS = [(11, 21), (12, 22)]
# round 6: three more weight / data draws of the headline configuration, and one draw at the 256x256 bench shape
EXTRA_SEED_CASES = [
    (("base_sr4", "base", "sr_ckpt_df2", 4, (64, 64), "sr"), [(13, 23), (14, 24), (15, 25)]),
    (("base_sr4_256", "base", "sr_ckpt_df2", 4, (256, 256), "sr"), [(11, 21)]),
]


def main_seeds():
    """Second weight / data seeds at clamp scales (VERDICT r4 #4b): the truth of the GPU test is the PINNED ORACLE run in float64 (it is
    pinned to the unmodified reference within 1.2e-6 in fp32, tests/test_oracle_pinned.py); stored as float32 under tests/golden/seeds/
    so that the GPU box does not spend five minutes of host time on it.  Weights and inputs are reproducible from the seeds."""
    out_dir = os.path.join(ROOT, "tests", "golden", "seeds")
    os.makedirs(out_dir, exist_ok=True)
    from tests.util import product_shapes
    todo = [(c, SEED_PAIRS) for c in SEED_CASES] + EXTRA_SEED_CASES
    for (tag, model, geom, up, hw, task), pairs in todo:
        for wseed, dseed in pairs:
            if os.path.isfile(os.path.join(out_dir, f"{tag}_{wseed}_{dseed}.npz")) and "--force" not in sys.argv:
                continue
            cfg = make_config(model, geom, upscale=up, img_size=hw[0])
            sd = O.seeded_state_dict(product_shapes(cfg), wseed, logit_scale_mean=LN100)
            lq, _ = O.synthetic_pair(task, hw, up, batch=1, seed=dseed)
            lq = lq[..., : hw[0], : hw[1]].contiguous()
            with torch.no_grad():
                y64 = O.grl_forward(lq.double(), cfg, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
                y32 = O.grl_forward(lq, cfg, sd)
            meta = dict(tag=tag, weight_seed=wseed, data_seed=dseed, oracle_fp32_vs_fp64=float((y32.double() - y64).abs().max()))
            np.savez_compressed(os.path.join(out_dir, f"{tag}_{wseed}_{dseed}.npz"), meta=json.dumps(meta), truth=y64.float().numpy())
            print(f"{tag} seeds ({wseed}, {dseed}): out {tuple(y64.shape)}, oracle fp32 vs fp64 max|d| = {meta['oracle_fp32_vs_fp64']:.3e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--seeds":
        main_seeds()
    elif len(sys.argv) > 1 and sys.argv[1] == "--fp64":
        main_fp64(set(sys.argv[2:]) or None)
    else:
        main(set(sys.argv[1:]) or None)
