"""Checkpoint loading and PSNR-Y evaluation without Lightning / Hydra.

Restates, for the MI355X module, the three pieces of the reference's evaluation command
(``scripts/grl/grl_test.md:73-79``) that sit around ``model(x)``:

  load_checkpoint   tools/trainer.py:93-115    Lightning ``{"state_dict": {"model.<key>": ...}}``, ``{"params": ...}`` or a
                                               plain state dict; geometry buffers dropped by ``convert_checkpoint``
  psnr_y            engines/base.py:256-268    tensor_round -> shave(scale) for SR -> PSNR on the matlab-style Y channel
                    utils/utils_image.py:8-11,30-33,43-79; utils/metrics/psnr.py:44-48
  evaluate_folder   the validation loop over an LQ / GT image folder pair (whole image, or the reference's tiled
                    inference through ``tiling.forward_tiled``)

    python -m grl_image_restoration_amd.evaluate --model base --geometry sr_ckpt_df2 --scale 4 \\
        --ckpt sr_grl_base_c3x4.ckpt --lq Set5/LRbicx4 --gt Set5/GTmod12 [--tile 256 --overlap 32]
"""
import argparse
import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch

_METRIC_KEYS = ("current_val_metric", "best_val_metric", "best_iter")


def extract_state_dict(obj: Dict) -> Dict[str, torch.Tensor]:
    """The parameter dictionary inside whatever ``torch.load`` returned (tools/trainer.py:97-104): a Lightning checkpoint
    (``state_dict`` with ``model.`` prefixes and the trainer's metric entries), a BasicSR-style ``params`` dictionary,
    or already a plain state dict."""
    sd = obj
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = dict(sd["state_dict"])
        for k in _METRIC_KEYS:
            sd.pop(k, None)
    if isinstance(sd, dict) and "params" in sd:
        sd = sd["params"]
    if not isinstance(sd, dict):
        raise TypeError(f"not a checkpoint dictionary: {type(obj).__name__}")
    return sd


def load_checkpoint(model, path_or_obj, strict: bool = True):
    """Load a reference checkpoint into ``model`` (a ``grl_image_restoration_amd.GRL``).  Mirrors tools/trainer.py:93-115:
    geometry buffers (``table_*/index_*/mask_*``, ``relative_*``, ``attn_mask``) are dropped, the Lightning module prefix
    ``model.`` is removed, and -- as trainer.py:106-108 does -- the checkpoint is merged INTO the module's current state before
    the strict load: a partial checkpoint keeps the current values of the keys it lacks, an unknown key still fails.
    (The other direction: ``GRL.state_dict()`` has no ``table_/index_/mask_`` buffers, so loading it into the reference module
    needs ``strict=False`` or the reference's own buffers merged in the same way.)"""
    obj = torch.load(path_or_obj, map_location="cpu") if isinstance(path_or_obj, (str, os.PathLike)) else path_or_obj
    sd = extract_state_dict(obj)
    sd = model.convert_checkpoint(dict(sd))           # needs the un-stripped "model.table_*" names (grl.py:556-569)
    sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
    cur = dict(model.state_dict())
    cur.update(sd)
    return model.load_state_dict(cur, strict=strict)


# ---- metric (restated here: the product does not depend on the test infrastructure) -------------------------------
def tensor_round(img: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """utils/utils_image.py:30-33 (out of place)."""
    return (img.clamp(0.0, data_range) * 255.0 / data_range).round() * data_range / 255.0


def shave(img: torch.Tensor, border: int) -> torch.Tensor:
    """utils/utils_image.py:8-11."""
    return img[..., border:-border, border:-border] if border > 0 else img


def rgb_to_y(img: torch.Tensor) -> torch.Tensor:
    """Y channel of matlab ``rgb2ycbcr`` for float RGB in [0, 1] (utils/utils_image.py:43-79): (B,1,H,W) in [0,1]."""
    w = torch.tensor([65.481, 128.553, 24.966], dtype=img.dtype, device=img.device) / 255.0
    y = (img * 255.0).permute(0, 2, 3, 1) @ w + 16.0
    return (y.round() / 255.0).unsqueeze(1)


def psnr_y(restored: torch.Tensor, target: torch.Tensor, scale: int = 1) -> torch.Tensor:
    """PSNR-Y as the reference's validation step computes it (engines/base.py:256-268, utils/metrics/psnr.py:44-48):
    quantise the output to 8 bit, shave ``scale`` pixels for SR, compare the Y channels.  Grey inputs are compared as is."""
    border = scale if scale > 1 else 0
    out, tgt = shave(tensor_round(restored.float()), border), shave(target.float(), border)
    if out.shape[1] == 3:
        out, tgt = rgb_to_y(out), rgb_to_y(tgt)
    return -10.0 * (out - tgt).pow(2).mean(dim=(-3, -2, -1)).log10()


# ---- folder evaluation ---------------------------------------------------------------------------------------------
_IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")


def _read_image(path: str) -> torch.Tensor:
    import numpy as np
    from PIL import Image

    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).unsqueeze(0).contiguous()


def image_pairs(lq_dir: str, gt_dir: str) -> List[Tuple[str, str]]:
    """(lq, gt) paths matched by sorted order of the image files in the two folders (the dataset classes pair the same way,
    data/datasets/restoration_sr.py:134-141)."""
    ls = lambda d: sorted(os.path.join(d, f) for f in os.listdir(d) if f.lower().endswith(_IMG_EXT))
    lq, gt = ls(lq_dir), ls(gt_dir)
    if len(lq) != len(gt) or not lq:
        raise ValueError(f"{lq_dir}: {len(lq)} images, {gt_dir}: {len(gt)} images")
    return list(zip(lq, gt))


@torch.no_grad()
def evaluate_pairs(model, pairs: Iterable[Tuple[torch.Tensor, torch.Tensor]], scale: int, tile: int = 0, overlap: int = 32,
                   device: str = "cuda:0") -> List[float]:
    """PSNR-Y of ``model`` on (lq, gt) tensors in [0, 1], (1,3,h,w) / (1,3,h*scale,w*scale).  ``tile > 0`` uses the
    reference's tiled inference (engines/base.py:90-116) through ``tiling.forward_tiled``."""
    from . import tiling

    out = []
    for lq, gt in pairs:
        lq = lq.to(device)
        if tile and tile < min(lq.shape[-2:]):
            sr = tiling.forward_tiled(model, lq, tile, overlap, scale)
        else:
            sr = model(lq)
        gt = gt.to(sr.device)[..., : sr.shape[-2], : sr.shape[-1]]
        out.append(float(psnr_y(sr[..., : gt.shape[-2], : gt.shape[-1]], gt, scale)))
    return out


def evaluate_folder(model, lq_dir: str, gt_dir: str, scale: int, tile: int = 0, overlap: int = 32, device: str = "cuda:0",
                    verbose: bool = True) -> float:
    pairs = image_pairs(lq_dir, gt_dir)
    vals = []
    for lq_p, gt_p in pairs:
        v = evaluate_pairs(model, [(_read_image(lq_p), _read_image(gt_p))], scale, tile, overlap, device)[0]
        vals.append(v)
        if verbose:
            print(f"{os.path.basename(lq_p):32s} PSNR-Y {v:7.3f} dB")
    mean = sum(vals) / len(vals)
    if verbose:
        print(f"{'mean over ' + str(len(vals)) + ' images':32s} PSNR-Y {mean:7.3f} dB")
    return mean


def main(argv: Optional[List[str]] = None) -> float:
    from . import GRL, make_config

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", default="base", choices=["tiny", "small", "base"])
    ap.add_argument("--geometry", default="sr_ckpt_df2", help="a key of presets.GEOMETRIES")
    ap.add_argument("--scale", type=int, default=4, help="1 for denoising / deblurring")
    ap.add_argument("--ckpt", default=None, help="reference checkpoint (.ckpt / .pth); random init without it")
    ap.add_argument("--lq", required=True)
    ap.add_argument("--gt", required=True)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--overlap", type=int, default=32)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    model = GRL(**make_config(a.model, a.geometry, upscale=a.scale)).eval()
    if a.ckpt:
        load_checkpoint(model, a.ckpt)
    model = model.to(a.device)
    return evaluate_folder(model, a.lq, a.gt, a.scale, a.tile, a.overlap, a.device)


if __name__ == "__main__":
    main()
