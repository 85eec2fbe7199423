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


def test_batched_plane_and_table_chains_equal_the_per_slot_chains(monkeypatch):
    """Round 5 builds the head planes of a block, and the bias tables / logit scales of all blocks, in batched chains
    (GRL._block_planes / _train_tables / _residual).  Same loss and same gradients as the per-slot chains of round 4
    (GRL_TRAIN_BATCHED_PLANES=0), here through the composite contractions on CPU at a checkpoint-like spread of logit scales, with
    DropPath on (both paths draw the same masks)."""
    from grl_image_restoration_amd import GRL, make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[2, 1], num_heads_window=[3, 3], num_heads_stripe=[3, 3],
                      drop_path_rate=0.2)

    def run(flag):
        monkeypatch.setenv("GRL_TRAIN_BATCHED_PLANES", flag)
        torch.manual_seed(0)
        m = GRL(**cfg).train()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "logit_scale" in n:
                    p.add_(torch.randn_like(p) * 0.5 + 1.0)
        g = torch.Generator().manual_seed(1)
        x, y = torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 3, 256, 256, generator=g)
        torch.manual_seed(7)                      # the DropPath draws
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = (m(x) - y).abs().mean()
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in m.named_parameters()}

    l0, g0 = run("0")
    l1, g1 = run("1")
    assert abs(l0 - l1) < 1e-6, (l0, l1)
    rel = {k: ((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-30)).item() for k in g0}
    big = {k: v for k, v in rel.items() if "cpb_mlp" not in k and "logit_scale" not in k}
    worst = max(rel, key=rel.get)
    print(f"loss {l0:.7f} / {l1:.7f}; worst relative gradient difference {rel[worst]:.2e} ({worst}); without the CPB-MLP / scales {max(big.values()):.2e}")
    # fp32 sums in a different order (measured: 4e-6 worst, 1.5e-6 without the tiny CPB-MLP / logit-scale gradients)
    assert max(big.values()) < 1e-4 and rel[worst] < 1e-3
