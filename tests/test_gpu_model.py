"""GPU parity of the whole network through the product module (HIP path) -- needs the MI355X.

Compared against (a) the golden fixtures = outputs of the REAL reference (tests/golden) and (b) the
CPU oracle on fresh seeded inputs.  Tolerance (floating point path): BASELINE.json asks for 1e-3
max-abs on the network output and 0.01 dB PSNR-Y; EVERY fixture is asserted at 1e-3 -- all model sizes and geometries,
logit scales at the clamp, and the bench shape itself (two 256x256 LQ tiles, 384x384 deblur tiles of a 1280x720 frame).
"""
import pytest
import torch

from oracle import engine_oracle as E
from oracle import grl_oracle as O
from tests.util import golden_names, golden_state_dict, load_fp64, load_golden

pytestmark = pytest.mark.gpu

TOL_MAXABS = 1e-3  # north_star bar, every model / geometry


def _product(cfg, seed, **sd_kwargs):
    from grl_image_restoration_amd import GRL

    m = GRL(**cfg).eval()
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed, **sd_kwargs)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0"), sd


def _golden_model(meta):
    return _product(meta["cfg"], meta["weight_seed"], **meta.get("sd_kwargs", {}))[0]


@pytest.mark.parametrize("name", [n for n in golden_names() if n != "base_sr4_ckpt_256"])
def test_hip_forward_matches_reference_golden(name):
    meta, z = load_golden(name)
    m = _golden_model(meta)
    with torch.no_grad():
        y = m(z["input"].to("cuda:0")).float().cpu()
    assert y.shape == z["output"].shape
    err = (y - z["output"]).abs().max().item()
    rms = (y - z["output"]).pow(2).mean().sqrt().item()
    print(f"{name} [{m.precision}]: max|hip - reference| = {err:.3e}  rms = {rms:.3e}")
    # Checkpoint-like fixtures carry a FLOAT64 run of the unmodified reference (oracle/make_golden.py --fp64, VERDICT r4 #4a): the
    # bar is then measured against the exact network, with the reference's own fp32 distance to it as the allowance --
    # |hip - fp64| <= max(1e-3, 2 |ref_fp32 - fp64|).  (Round 4 widened small_dn_128_hiscale to 4 x an input-perturbation number;
    # the float64 run shows the reference's fp32 arithmetic within 2.4e-4 of the truth there, i.e. the 1e-3 bar stands, and the
    # 7.2e-3 of round 4 was the fp16 rounding of the softmax weights in the split-operand attention -- fixed by its third PV term.)
    f64 = load_fp64(name)
    if f64 is not None:
        m64, z64 = f64
        s_ = m64["stride"]
        truth = z64["ref32_sub"].double() - z64["ref32_minus_fp64_sub"].double()
        e64 = (y[..., ::s_, ::s_].double() - truth).abs().max().item()
        allow = max(TOL_MAXABS, 2.0 * m64["max_abs_ref32_minus_fp64"])
        print(f"{name}: max|hip - reference_fp64| = {e64:.3e}  (reference fp32 vs fp64: {m64['max_abs_ref32_minus_fp64']:.3e}; allowance {allow:.1e})")
        assert e64 < allow, (e64, allow)
        if m64["max_abs_ref32_minus_fp64"] > 1e-4:      # the fp32 reference output itself is only defined to this much:
            # the full-resolution comparison with it stays, at the bar plus what the reference's own arithmetic contributes
            # (this branch used to return after the strided lattice)
            assert err < TOL_MAXABS + 2.0 * m64["max_abs_ref32_minus_fp64"], (err, m64["max_abs_ref32_minus_fp64"])
            return
    assert err < TOL_MAXABS, err


def test_hip_forward_at_the_bench_shape():
    """The benchmark's own shape: GRL-Base x4, checkpoint geometry, a batch of two 256x256 LQ tiles (4x4 stripes and 8x8
    windows per tile, multi-thousand-workgroup grids through xcd_remap, one tile per HIP stream of the 2-group split)
    against the reference's outputs (tile 0 complete, 16-bit fixed point; tile 1 on a 4x-strided lattice + its sums)."""
    meta, z = load_golden("base_sr4_ckpt_256")
    m = _golden_model(meta)
    assert m.stream_groups(2) == 2
    with torch.no_grad():
        y = m(z["input"].to("cuda:0")).float().cpu()
    assert list(y.shape) == meta["out_shape"]
    e0 = (y[:1] - z["output"]).abs().max().item()
    s = meta["sub2"]
    e1 = (y[1:, :, ::s, ::s] - z["output_b1_sub"]).abs().max().item()
    n = y[0].numel()
    sums = [abs(float(y[i].double().sum()) - meta["out_sum"][i]) / n for i in range(2)]
    print(f"base_sr4_ckpt_256: tile0 max|d| = {e0:.3e} (q_step {meta['q_step']:.1e}), tile1 lattice max|d| = {e1:.3e}, mean|d sum| = {sums}")
    assert e0 < TOL_MAXABS + meta["q_step"] and e1 < TOL_MAXABS and max(sums) < 1e-4
    # the single-stream schedule and a second call give the same bits
    with torch.no_grad():
        assert torch.equal(m(z["input"].to("cuda:0")).float().cpu(), y)


def test_tiled_frame_matches_reference_tiles():
    """BASELINE config 4 at full size: a synthetic 1280x720 frame, tile 384 / overlap 48 (2 x 4 tiles, engines/base.py:90-116)
    through tiling.forward_tiled; tiles 0 and 5 were run through the REAL reference (fixture base_deblur_384).  Where a tile
    is the only contributor of the stitched frame (no overlap) the stitched pixels must equal the reference tile."""
    from grl_image_restoration_amd import tiling

    meta, z = load_golden("base_deblur_384")
    m = _golden_model(meta)
    fh, fw = meta["frame"]
    t, ov = meta["tile"], meta["overlap"]
    frame, _ = O.synthetic_pair("deblur", (fh, fw), 1, seed=meta["frame_seed"])
    frame = frame[..., :fh, :fw].contiguous()
    origins = [tuple(o) for o in meta["origins"]]
    assert tiling.tile_list(fh, fw, t, ov)[1] == origins and len(origins) == 8
    for i, tid in enumerate(meta["tile_ids"]):
        a, b = origins[tid]
        assert torch.equal(frame[..., a : a + t, b : b + t], z["input"][i : i + 1])   # the fixture's tiles are tiles of this frame
    with torch.no_grad():
        got = tiling.forward_tiled(m, frame.cuda(), t, ov, 1, tile_batch=4).cpu()
    assert got.shape == (1, 3, fh, fw)
    cover = torch.zeros(fh, fw)
    for a, b in origins:
        cover[a : a + t, b : b + t] += 1
    worst = 0.0
    for i, tid in enumerate(meta["tile_ids"]):
        a, b = origins[tid]
        solo = cover[a : a + t, b : b + t] == 1
        assert solo.float().mean() > 0.3
        d = (got[0, :, a : a + t, b : b + t] - z["output"][i]).abs()[:, solo]
        worst = max(worst, d.max().item())
    print(f"1280x720 deblur frame, 8 tiles of 384: max|stitched - reference tile| over single-cover pixels = {worst:.3e}")
    assert worst < TOL_MAXABS


def test_psnr_parity_and_loaded_extension():
    """PSNR-Y of (HIP output vs GT) equals PSNR-Y of (oracle output vs GT) within 0.01 dB, and the
    process really has libgrl_hip.so mapped (no silent fallback)."""
    from grl_image_restoration_amd import make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[2, 2, 2], num_heads_window=[3] * 3, num_heads_stripe=[3] * 3)
    m, sd = _product(cfg, 3)
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=2, seed=5)
    with torch.no_grad():
        want = O.grl_forward(lq, cfg, sd)
        got = m(lq.to("cuda:0")).float().cpu()
    p_want, p_got = E.psnr_y_eval(want, gt, 4), E.psnr_y_eval(got, gt, 4)
    print("psnr-y oracle", p_want.tolist(), "hip", p_got.tolist(), "max|d|", (got - want).abs().max().item())
    assert (p_want - p_got).abs().max().item() < 0.01
    assert "libgrl_hip.so" in open("/proc/self/maps").read()


def test_batch_and_ragged_input_consistency():
    """B=2 equals two B=1 calls; a non-multiple input size is reflect-padded and cropped (grl.py:479-489,551)."""
    from grl_image_restoration_amd import make_config

    cfg = make_config("small", "sr_ckpt_df4", upscale=2, img_size=64, depths=[2, 2], num_heads_window=[2, 2], num_heads_stripe=[2, 2])
    m, sd = _product(cfg, 4)
    lq, _ = O.synthetic_pair("sr", (50, 70), 2, batch=2, seed=6)
    with torch.no_grad():
        yb = m(lq.cuda()).cpu()
        y0, y1 = m(lq[:1].cuda()).cpu(), m(lq[1:].cuda()).cpu()
        want = O.grl_forward(lq, cfg, sd)
    assert yb.shape == (2, 3, 100, 140)
    # every kernel on the path is batch-invariant (per-token / per-window / per-tile work, fixed reduction orders)
    assert torch.equal(yb[0], y0[0]) and torch.equal(yb[1], y1[0])
    assert (yb - want).abs().max().item() < TOL_MAXABS


def test_tiled_inference_matches_reference_loop():
    """BASELINE config 4 in miniature: tiled whole-image inference (engines/base.py:90-116) through the HIP
    model with batched tiles vs the serial reference loop run on the CPU oracle."""
    from grl_image_restoration_amd import make_config, tiling

    cfg = make_config("base", "deblur", upscale=1, img_size=96, depths=[2, 2], num_heads_window=[3, 3], num_heads_stripe=[3, 3])
    m, sd = _product(cfg, 7)
    lq, _ = O.synthetic_pair("deblur", (120, 200), 1, batch=1, seed=8)
    with torch.no_grad():
        want = E.forward_tile(lambda t: O.grl_forward(t, cfg, sd), lq, 96, 16, 1)
        got = tiling.forward_tiled(m, lq.cuda(), 96, 16, 1, tile_batch=4).cpu()
    assert got.shape == want.shape == (1, 3, 120, 200)
    err = (got - want).abs().max().item()
    print(f"tiled deblur 120x200 (tile 96/overlap 16, 6 tiles): max|hip - oracle| = {err:.3e}")
    assert err < TOL_MAXABS


def test_graph_replay_equals_eager_launches():
    """enable_graph(): the captured HIP graph of a forward (both tile groups on their streams) reproduces the eager
    launch sequence bit for bit, for a new input of the same shape, and the result does not alias graph memory."""
    meta, z = load_golden("base_sr4_ckpt_64")
    m = _golden_model(meta)
    g = torch.Generator().manual_seed(3)
    x1 = torch.rand(2, 3, 64, 64, generator=g).to("cuda:0")
    x2 = torch.rand(2, 3, 64, 64, generator=g).to("cuda:0")
    with torch.no_grad():
        e1, e2 = m(x1).clone(), m(x2).clone()
        m.enable_graph()
        g1 = m(x1)
        g2 = m(x2)          # replay with new data
        g1b = m(x1)
    assert torch.equal(g1, e1) and torch.equal(g2, e2) and torch.equal(g1b, e1)
    assert g1.data_ptr() != g1b.data_ptr()
    m.enable_graph(False)
    with torch.no_grad():
        assert torch.equal(m(x2), e2)


def test_repeatable_and_schedule_independent(monkeypatch):
    """Race screen: no atomics and fixed reduction orders everywhere, so repeated forwards must agree bit for bit --
    a DMA / barrier ordering bug in the streaming kernels or the attention staging shows up as run-to-run noise -- and
    the two-stream tile-group schedule must equal the single-stream one (tiles are independent)."""
    meta, z = load_golden("base_sr4_ckpt_64")
    m = _golden_model(meta)
    x = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(9)).to("cuda:0")
    with torch.no_grad():
        ref = m(x).clone()
        for _ in range(4):
            assert torch.equal(m(x), ref)
        monkeypatch.setenv("GRL_SPLIT_STREAMS", "1")
        assert torch.equal(m(x), ref)


def test_graph_owns_its_plan_across_shapes_and_weight_updates():
    """a captured graph points at the packed weights / tables of the plan it was captured with.  Shape A,
    then shape B (replaces the single cached plan), then A again must still replay correctly; load_state_dict (also through
    a PARENT module, tools/trainer.py:108-111) and in-place parameter updates must drop plans and graphs."""
    from grl_image_restoration_amd import make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[2, 2], num_heads_window=[3, 3], num_heads_stripe=[3, 3])
    m, sd = _product(cfg, 5)
    g = torch.Generator().manual_seed(4)
    xa = torch.rand(1, 3, 64, 64, generator=g).cuda()
    xb = torch.rand(1, 3, 64, 128, generator=g).cuda()
    with torch.no_grad():
        ea, eb = m(xa).clone(), m(xb).clone()
        m.enable_graph()
        assert torch.equal(m(xa), ea)
        assert torch.equal(m(xb), eb)            # second shape: the plan cache now holds B's plan only
        junk = [torch.randn(1 << 22, device="cuda") for _ in range(8)]   # recycle whatever A's plan would have freed
        assert torch.equal(m(xa), ea)            # A's graph still owns A's plan
        del junk

        class Wrapper(torch.nn.Module):          # the Lightning module owns self.model and loads through the parent
            def __init__(self, model):
                super().__init__()
                self.model = model

        wrap = Wrapper(m)
        sd2 = {"model." + k: v * 0.5 if k.endswith("conv_last.weight") else v for k, v in sd.items()}
        wrap.load_state_dict(sd2, strict=True)
        ya = m(xa).clone()
        assert not torch.equal(ya, ea)           # the stale graph / plan was dropped
        m.enable_graph(False)
        assert torch.equal(m(xa), ya)
        m.enable_graph()
        assert torch.equal(m(xa), ya)
        m.conv_last.bias.add_(0.25)              # in-place update of the Parameter itself (optimizer / EMA style; a write
        yb = m(xa)                               # through ``.data`` bypasses torch's version counter: call invalidate_plan())
        assert (yb - ya - 0.25).abs().max().item() < 1e-5


# bars of the default precision (`auto`) at checkpoint-like scales on weight sets the policy was NOT tuned on
SEED_BARS = {
    # GRL-Base SR: fp16 operands sit AT the 1e-3 bar at these scales, weight draw by weight draw (round 5: 7.7e-4 / 5.9e-4 / 1.9e-3;
    # round 6, six draws: 7.3e-4, 1.6e-3, 6.3e-4, 5.5e-4, 8.3e-4, 1.2e-3).  Since round 6 `auto` MEASURES the weights it holds
    # (GRL._calibrated_plan: a probe image through the all-split network and through the fp16-operand one) and moves the blocks whose
    # fp16 rounding costs the most to split operands until the probe passes: 0-11 of 40 blocks on these draws, 3.6e-4 .. 8.0e-4
    # against the float64 truth -- asserted at the north_star's 1e-3 like every fixture (the bar was 2.5e-3).
    "base_sr4": dict(auto_max=1e-3, auto_rms=2e-4, also_high=True),
    "base_sr4_256": dict(auto_max=1e-3, auto_rms=2e-4, also_high=False),   # the bench shape: one 256x256 tile, 3 M outputs
    "small_dn": dict(auto_max=1e-3, auto_rms=1e-3, also_high=False),       # (`auto` already resolves to `high` at these scales)
    "base_deblur": dict(auto_max=1e-3, auto_rms=1e-3, also_high=False),
}


_SEED_CASES = {
    "base_sr4": ("base", "sr_ckpt_df2", 4, (64, 64), "sr"),          # BASELINE configs[2] geometry
    "base_sr4_256": ("base", "sr_ckpt_df2", 4, (256, 256), "sr"),    # ... at the bench's tile size
    "small_dn": ("small", "dn_df4", 1, (128, 128), "dn"),            # BASELINE configs[1]
    "base_deblur": ("base", "deblur", 1, (96, 192), "deblur"),       # BASELINE configs[3] geometry (window 12, stripes 48x96, anchors /4)
}
_SEED_RUNS = ([("base_sr4", (w, w + 10)) for w in (11, 12, 13, 14, 15)] + [("base_sr4_256", (11, 21))] +
              [(t, sd) for t in ("small_dn", "base_deblur") for sd in ((11, 21), (12, 22))])


@pytest.mark.parametrize("tag,seeds", _SEED_RUNS, ids=[f"{t}-seeds{sd[0]}" for t, sd in _SEED_RUNS])
def test_second_seeds_at_clamp_scales_vs_pinned_oracle(tag, seeds):
    """VERDICT r4 #4b / r5 #3: every fixture uses weight seed 0 / data seed 1 and the precision policy was tuned on them.  Further
    (weight, data) seed pairs per BASELINE configuration -- five for the headline one, GRL-Base x4 SR, plus one of them at the
    256x256 bench shape -- with logit scales drawn around ln 100 (about half of the heads at the clamp), against the pinned CPU
    oracle (pinned to the unmodified reference within 1.2e-6 in fp32; run in float64 here).  All at the north_star's 1e-3, in the
    default mode (`auto`) -- the mode bench.py times."""
    import math

    from grl_image_restoration_amd import GRL, make_config

    model, geom, up, hw, task = _SEED_CASES[tag]
    wseed, dseed = seeds
    bars = SEED_BARS[tag]
    cfg = make_config(model, geom, upscale=up, img_size=hw[0])
    m, sd = _product(cfg, wseed, logit_scale_mean=math.log(100.0))
    lq, gt = O.synthetic_pair(task, hw, up, batch=1, seed=dseed)
    lq = lq[..., : hw[0], : hw[1]].contiguous()
    with torch.no_grad():
        # the oracle in FLOAT64 is the truth here (at these scales GRL-Small amplifies fp32 round-off to 1e-4: the test must not
        # charge the HIP path for the checker's own arithmetic).  Frozen by `python -m oracle.make_golden --seeds` under
        # tests/golden/seeds/ (five minutes of host time otherwise); recomputed when the file is missing.
        import json as _json
        import os as _os

        import numpy as _np
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "seeds", f"{tag}_{wseed}_{dseed}.npz")
        if _os.path.isfile(path):
            z = _np.load(path, allow_pickle=False)
            assert _json.loads(str(z["meta"]))["weight_seed"] == wseed
            want = torch.from_numpy(z["truth"]).double()
        else:
            want = O.grl_forward(lq.double(), cfg, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
        got = m(lq.to("cuda:0")).double().cpu()
    err = (got - want).abs().max().item()
    rms = (got - want).pow(2).mean().sqrt().item()
    cal = getattr(m, "calibration", None) or {}
    print(f"{model} {geom} {hw} seeds {seeds} [{m.precision}]: max|hip - oracle_fp64| = {err:.3e}  rms = {rms:.3e}"
          + (f"  (calibration: fp16 operands alone {cal['fast_max']:.2e} / rms {cal['fast_rms']:.2e} on the probe; "
             f"{cal['split']} of {cal['blocks']} blocks split -> {cal.get('probe_max', 0):.2e} / {cal.get('probe_rms', 0):.2e})" if "fast_max" in cal else ""))
    assert err < bars["auto_max"] and rms < bars["auto_rms"], (err, rms)
    if up > 1:
        gt = gt[..., : hw[0] * up, : hw[1] * up]
        d_psnr = (E.psnr_y_eval(want.float(), gt, up) - E.psnr_y_eval(got.float(), gt, up)).abs().max().item()
        assert d_psnr < 0.01, d_psnr
    if bars["also_high"]:
        mh = GRL(**cfg, precision="high").eval()
        mh.load_state_dict(sd, strict=True)
        with torch.no_grad():
            got_h = mh.to("cuda:0")(lq.to("cuda:0")).double().cpu()
        err_h = (got_h - want).abs().max().item()
        print(f"{model} {geom} seeds {seeds} [precision='high']: max|hip - oracle_fp64| = {err_h:.3e}")
        assert err_h < TOL_MAXABS, err_h


def test_auto_precision_is_calibrated_once_per_weight_set(monkeypatch):
    """precision='auto' on a wide SR model with checkpoint-like scales (GRL._calibrated_plan): the choice is made when the plan is
    built -- a second forward re-uses it, new weights trigger a new measurement, GRL_CALIBRATE=0 keeps the fp16-operand rules of
    round 5 -- and the numbers it decided on are available to the caller."""
    import math

    from grl_image_restoration_amd import GRL, make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64)
    m, sd = _product(cfg, 15, logit_scale_mean=math.log(100.0))          # weight draw 15: fp16 operands alone give 1.2e-3
    m = m.to("cuda:0")
    lq = O.synthetic_pair("sr", (64, 64), 4, batch=1, seed=25)[0].to("cuda:0")
    builds = []
    orig = GRL._build_plan
    monkeypatch.setattr(GRL, "_build_plan", lambda self, *a, **k: (builds.append(a[2]), orig(self, *a, **k))[1])
    with torch.no_grad():
        y1 = m(lq)
        n1 = len(builds)
        y2 = m(lq)
    cal = m.calibration
    assert m.precision.startswith("mixed(") and 1 <= cal["split"] < cal["blocks"] == 40
    assert cal["fast_max"] > cal["bar_max"] >= cal["probe_max"] and cal["probe_rms"] <= cal["bar_rms"]
    assert n1 >= 3 and len(builds) == n1 and torch.equal(y1, y2)          # (fast, all-split reference, split blocks; nothing on the 2nd call)
    m.load_state_dict(_product(cfg, 13, logit_scale_mean=math.log(100.0))[1], strict=True)    # new weights: measured again
    with torch.no_grad():
        m(lq)
    assert len(builds) > n1 and m.calibration["split"] != cal["split"] or m.calibration["fast_max"] != cal["fast_max"]
    monkeypatch.setenv("GRL_CALIBRATE", "0")
    m.invalidate_plan()
    with torch.no_grad():
        y3 = m(lq)
    assert m.precision == "fast" and m.calibration is None and bool(torch.isfinite(y3).all())
