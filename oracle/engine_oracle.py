"""TEST INFRASTRUCTURE ONLY -- restatement of the reference *engine* arithmetic around the model.

The Lightning engine (engines/base.py) cannot be imported (pytorch_lightning, hydra, torchmetrics
are absent), so the few lines that define results are restated here:

  forward_tile   engines/base.py:90-116     tiled inference with uniform overlap averaging
  tensor_round   utils/utils_image.py:30-33 clamp to [0,1], quantise to 8 bit
  shave          utils/utils_image.py:8-11  crop a border
  rgb2ycbcr_y    utils/utils_image.py:43-79 matlab-style Y channel
  psnr / psnr_y  utils/metrics/psnr.py:44-48 and engines/base.py:256-268 (round -> shave -> metric)
"""
import torch


def tile_origins(dim: int, tile: int, overlap: int):
    """engines/base.py:96-98: list(range(0, dim - tile, stride)) + [dim - tile]."""
    stride = tile - overlap
    return list(range(0, dim - tile, stride)) + [dim - tile]


def forward_tile(model, x: torch.Tensor, tile: int, overlap: int, scale: int, out_channels: int = 3):
    """engines/base.py:90-116.  ``model`` maps (b,c,t,t)->(b,c,t*s,t*s)."""
    b, _, h, w = x.shape
    tile = min(tile, h, w)
    hs, ws = tile_origins(h, tile, overlap), tile_origins(w, tile, overlap)
    E = torch.zeros(b, out_channels, h * scale, w * scale, dtype=x.dtype, device=x.device)
    Wt = torch.zeros_like(E)
    for hi in hs:
        for wi in ws:
            out = model(x[..., hi : hi + tile, wi : wi + tile])
            E[..., hi * scale : (hi + tile) * scale, wi * scale : (wi + tile) * scale].add_(out)
            Wt[..., hi * scale : (hi + tile) * scale, wi * scale : (wi + tile) * scale].add_(torch.ones_like(out))
    return E.div_(Wt)


def tensor_round(img: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    img = img.clamp(0.0, 1.0 * data_range)
    return (img * 255.0 / data_range).round() * data_range / 255.0


def shave(img: torch.Tensor, border: int) -> torch.Tensor:
    return img[..., border:-border, border:-border] if border > 0 else img


def rgb2ycbcr_y(img: torch.Tensor) -> torch.Tensor:
    """Y of matlab rgb2ycbcr for float input in [0,1]; returns (B,1,H,W) in [0,1]."""
    r = torch.tensor([65.481, 128.553, 24.966], dtype=img.dtype, device=img.device) / 255.0
    y = (img * 255.0).permute(0, 2, 3, 1) @ r + 16.0
    return (y.round() / 255.0).unsqueeze(1)


def psnr(restored: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return -10 * (restored - target).pow(2).mean([-3, -2, -1]).log10()


def psnr_y_eval(output: torch.Tensor, gt: torch.Tensor, scale: int) -> torch.Tensor:
    """What validation_step feeds the PSNR-Y metric (engines/base.py:256-268)."""
    out = shave(tensor_round(output.float()), scale if scale > 1 else 0)
    tgt = shave(gt.float(), scale if scale > 1 else 0)
    return psnr(rgb2ycbcr_y(out), rgb2ycbcr_y(tgt))
