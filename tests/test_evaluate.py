"""Checkpoint loader + PSNR-Y evaluation: host logic on CPU against the oracle's engine restatement,
folder evaluation on the GPU."""
import io
import os

import pytest
import torch

from grl_image_restoration_amd import GRL, evaluate as EV, make_config
from oracle import engine_oracle as E
from oracle import grl_oracle as O


def _tiny():
    return make_config("tiny", "sr_ckpt_df4", upscale=2, img_size=64, depths=[2, 2], num_heads_window=[2, 2], num_heads_stripe=[2, 2])


def test_metric_matches_engine_restatement():
    g = torch.Generator().manual_seed(0)
    for scale in (1, 2, 4):
        out = torch.rand(2, 3, 40, 48, generator=g) * 1.2 - 0.1        # leaves [0, 1]: tensor_round clamps
        gt = torch.rand(2, 3, 40, 48, generator=g)
        a, b = EV.psnr_y(out, gt, scale), E.psnr_y_eval(out, gt, scale)
        assert torch.equal(a, b)
    x = torch.rand(1, 3, 8, 8, generator=g)
    assert torch.equal(EV.tensor_round(x), E.tensor_round(x)) and torch.equal(EV.rgb_to_y(x), E.rgb2ycbcr_y(x))


@pytest.mark.parametrize("fmt", ["lightning", "params", "plain"])
def test_load_checkpoint_formats(fmt, tmp_path):
    """tools/trainer.py:93-115: Lightning checkpoints carry 'model.' prefixes, trainer metrics and the reference's
    geometry buffers; BasicSR-style files wrap the dict in 'params'."""
    cfg = _tiny()
    src = GRL(**cfg)
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in src.state_dict().items()}, 7)
    if fmt == "lightning":
        ck = {"state_dict": {"model." + k: v for k, v in sd.items()}}
        ck["state_dict"].update({"model.table_w": torch.zeros(3), "model.index_sh_a2w": torch.zeros(3, dtype=torch.long),
                                 "model.mask_w": torch.zeros(2), "current_val_metric": torch.tensor(1.0),
                                 "best_val_metric": torch.tensor(1.0), "best_iter": torch.tensor(3)})
    elif fmt == "params":
        ck = {"params": dict(sd)}
    else:
        ck = dict(sd)
    path = tmp_path / "ck.pth"
    torch.save(ck, path)
    dst = GRL(**cfg)
    res = EV.load_checkpoint(dst, str(path))
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in dst.state_dict().items():
        assert torch.equal(v, sd[k]), k
    part = dict(sd)
    gone = next(iter(part))
    part.pop(gone)
    m2 = GRL(**cfg)
    keep = m2.state_dict()[gone].clone()
    EV.load_checkpoint(m2, part)                     # merged into the current state like tools/trainer.py:106-108
    assert torch.equal(m2.state_dict()[gone], keep)
    bad = dict(sd)
    bad["not.a.parameter"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        EV.load_checkpoint(GRL(**cfg), bad)          # strict for unknown keys, like the reference


@pytest.mark.gpu
def test_evaluate_folder_matches_oracle(tmp_path):
    """PNG folders -> PSNR-Y through the HIP module, against the CPU oracle on the same images and weights
    (0.01 dB bar of BASELINE.json)."""
    from PIL import Image
    import numpy as np

    cfg = _tiny()
    model = GRL(**cfg).eval()
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 2)
    EV.load_checkpoint(model, {"state_dict": {"model." + k: v for k, v in sd.items()}})
    lq_dir, gt_dir = tmp_path / "lq", tmp_path / "gt"
    lq_dir.mkdir(); gt_dir.mkdir()
    want = []
    for i in range(2):
        lq, gt = O.synthetic_pair("sr", (64, 64), 2, seed=10 + i)
        to8 = lambda t: (t[0].permute(1, 2, 0).clamp(0, 1) * 255).round().to(torch.uint8).numpy()
        Image.fromarray(to8(lq)).save(lq_dir / f"im{i}.png")
        Image.fromarray(to8(gt)).save(gt_dir / f"im{i}.png")
        lq8, gt8 = torch.from_numpy(to8(lq)).permute(2, 0, 1)[None].float() / 255, torch.from_numpy(to8(gt)).permute(2, 0, 1)[None].float() / 255
        with torch.no_grad():
            want.append(float(E.psnr_y_eval(O.grl_forward(lq8, cfg, sd), gt8, 2)))
    got = EV.evaluate_folder(model.to("cuda:0"), str(lq_dir), str(gt_dir), 2, verbose=False)
    assert abs(got - sum(want) / len(want)) < 0.01, (got, want)
    cli = EV.main(["--model", "tiny", "--geometry", "sr_ckpt_df4", "--scale", "2", "--lq", str(lq_dir), "--gt", str(gt_dir)])
    assert cli == cli  # runs end to end with random-init weights
