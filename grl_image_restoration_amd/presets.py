"""Constructor presets for the GRL model family and the geometries the reference ships.

Sources (reference tree): config/model/grl/grl_{tiny,small,base}.yaml, the experiment files
config/experiment/{sr,dn,db_motion}/grl/*.yaml and the evaluation commands in
scripts/grl/grl_test.md (which override window/stripe/anchor geometry per released checkpoint).
See SURVEY.md appendix B for the file:line of every number below.
"""
import copy

_COMMON = dict(
    in_channels=3,
    img_range=1.0,
    qkv_proj_type="linear",
    anchor_proj_type="avgpool",
    anchor_one_stage=True,
    out_proj_type="linear",
    conv_type="1conv",
    init_method="n",
    stripe_shift=True,
    mlp_ratio=2,
)

MODELS = {
    # grl_tiny.yaml:5-31
    "tiny": dict(
        embed_dim=64,
        depths=[4, 4, 4, 4],
        num_heads_window=[2, 2, 2, 2],
        num_heads_stripe=[2, 2, 2, 2],
        local_connection=False,
    ),
    # grl_small.yaml:5-31
    "small": dict(
        embed_dim=128,
        depths=[4, 4, 4, 4],
        num_heads_window=[2, 2, 2, 2],
        num_heads_stripe=[2, 2, 2, 2],
        local_connection=False,
    ),
    # grl_base.yaml:5-31
    "base": dict(
        embed_dim=180,
        depths=[4, 4, 8, 8, 8, 4, 4],
        num_heads_window=[3, 3, 3, 3, 3, 3, 3],
        num_heads_stripe=[3, 3, 3, 3, 3, 3, 3],
        local_connection=True,
    ),
}

GEOMETRIES = {
    # model-YAML default (grl_base.yaml:14-22); square inputs only
    "yaml": dict(window_size=8, stripe_size=[8, None], stripe_groups=[None, 4], anchor_window_down_factor=4),
    # SR released checkpoints, Tiny/Small (sr/grl/grl_p256.yaml:34-41, grl_test.md:57-71)
    "sr_ckpt_df4": dict(window_size=32, stripe_size=[64, 64], stripe_groups=[None, None], anchor_window_down_factor=4),
    # SR released checkpoint, Base (grl_test.md:73-79)
    "sr_ckpt_df2": dict(window_size=32, stripe_size=[64, 64], stripe_groups=[None, None], anchor_window_down_factor=2),
    # denoising Tiny/Small (dn/grl/grl_p256.yaml:38-45)
    "dn_df4": dict(window_size=16, stripe_size=[64, 128], stripe_groups=[None, None], anchor_window_down_factor=4),
    # denoising Base (grl_test.md:45-50)
    "dn_df2": dict(window_size=32, stripe_size=[64, 128], stripe_groups=[None, None], anchor_window_down_factor=2),
    # motion deblurring Base (db_motion/grl_p480.yaml:33-44)
    "deblur": dict(window_size=12, stripe_size=[48, 96], stripe_groups=[None, None], anchor_window_down_factor=4),
}

_UPSAMPLER = {"tiny": "pixelshuffledirect", "small": "pixelshuffle", "base": "pixelshuffle"}


def make_config(model: str, geometry: str, upscale: int = 1, img_size=64, **overrides) -> dict:
    """kwargs for ``GRL(**cfg)`` (reference ctor: models/networks/grl.py:220-256)."""
    cfg = dict(_COMMON)
    cfg.update(copy.deepcopy(MODELS[model]))
    cfg.update(copy.deepcopy(GEOMETRIES[geometry]))
    cfg["upscale"] = upscale
    cfg["upsampler"] = _UPSAMPLER[model] if upscale > 1 else ""
    cfg["img_size"] = img_size
    cfg.update(overrides)
    return cfg


# The five BASELINE.json configurations.
def baseline_config(i: int) -> dict:
    if i == 1:  # GRL-Tiny x2 SR, 64x64 LQ
        return make_config("tiny", "sr_ckpt_df4", upscale=2, img_size=64)
    if i == 2:  # GRL-Small denoise sigma 25, 128x128
        return make_config("small", "dn_df4", upscale=1, img_size=128)
    if i == 3:  # GRL-Base x4 SR, 256x256 LQ tiles (checkpoint geometry)  <-- bench workload
        return make_config("base", "sr_ckpt_df2", upscale=4, img_size=256)
    if i == 4:  # GRL-Base motion deblur, tiled 1280x720
        return make_config("base", "deblur", upscale=1, img_size=480)
    if i == 5:  # GRL-Base x4 SR training, 64x64 LQ
        return make_config("base", "sr_ckpt_df2", upscale=4, img_size=64)
    raise ValueError(i)
