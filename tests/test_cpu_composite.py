"""BASELINE configs[0] (GRL-Tiny x2 SR, one 64x64 LQ patch, CPU-only forward) through the PRODUCT module on a CPU tensor: the
composite torch path (grl_image_restoration_amd/composite.py) against the outputs of the unmodified reference (tests/golden,
frozen by oracle/make_golden.py).  fp32 everywhere, so the bar is round-off (1e-5), not the 1e-3 of the fp16-operand GPU path.
The oracle is used as the checker only (seeded weights); the product never imports it."""
import warnings

import pytest
import torch

from oracle import grl_oracle as O
from tests.util import load_golden


def _product(meta):
    from grl_image_restoration_amd import GRL

    m = GRL(**meta["cfg"]).eval()
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, meta["weight_seed"], **meta.get("sd_kwargs", {}))
    m.load_state_dict(sd, strict=True)
    return m


@pytest.mark.parametrize("name", ["tiny_sr2_ckpt_64", "tiny_sr2_yaml_64", "tiny_sr2_ckpt_64_hiscale", "base_sr4_yaml_32", "base_sr4_ckpt_64_hiscale"])
def test_cpu_forward_matches_reference_golden(name):
    meta, z = load_golden(name)
    m = _product(meta)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            y = m(z["input"])
    assert y.shape == z["output"].shape and not y.is_cuda
    err = (y - z["output"]).abs().max().item()
    print(f"{name}: max|composite - reference| = {err:.3e}")
    assert err < 2e-5, err


def test_cpu_path_is_differentiable_and_announced():
    meta, z = load_golden("tiny_sr2_ckpt_64")
    m = _product(meta).train()
    x = z["input"][:, :, :32, :32].clone().requires_grad_(True)
    with pytest.warns(UserWarning, match="composite torch path") if not __import__("grl_image_restoration_amd").composite._warned else warnings.catch_warnings():
        y = m(x)
    y.abs().mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    g = m.layers[0].blocks[0].attn.window_attn.attn_transform.logit_scale.grad
    assert g is not None and torch.isfinite(g).all()
