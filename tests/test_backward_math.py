"""Closed-form attention backward (oracle/backward_math.py, the formulas a flash-style HIP backward kernel will implement)
against autograd through the oracle's forward -- CPU only (SURVEY 8(f) N1 groundwork)."""
import math

import pytest
import torch

from oracle import backward_math as BM
from oracle import grl_oracle as O


@pytest.mark.parametrize("win,anchor_df,shift,w2a", [((8, 8), 1, 0, True), ((8, 8), 1, 4, True), ((8, 16), 2, 0, True), ((8, 16), 2, 4, False)])
def test_attention_backward_matches_autograd(win, anchor_df, shift, w2a):
    torch.manual_seed(0)
    nh, d = 3, 10
    H, W = 2 * win[0], 2 * win[1]
    table = O.coords_table(win, anchor_df)
    index = O.rel_index(win, anchor_df, w2a)                 # (N1, N2) for w2a, (N2, N1) for a2w
    Nq, Nk = index.shape
    nW = (H // win[0]) * (W // win[1])
    B_ = 2 * nW
    mask = None
    if shift:
        mode = "w" if anchor_df == 1 else ("w2a" if w2a else "a2w")
        mask = O.shift_mask((H, W), win, [shift, shift], df=anchor_df, mode=mode)
        assert mask.shape == (nW, Nq, Nk)
    p = {
        "t.logit_scale": torch.tensor([math.log(10.0), math.log(300.0), math.log(3.0)]).view(nh, 1, 1).double(),   # one head clamped
        "t.cpb_mlp.0.weight": torch.randn(512, 2).double() * 0.5,
        "t.cpb_mlp.0.bias": torch.randn(512).double() * 0.1,
        "t.cpb_mlp.2.weight": torch.randn(nh, 512).double() * 0.1,
    }
    for v_ in p.values():
        v_.requires_grad_(True)
    q = torch.randn(B_, nh, Nq, d, dtype=torch.float64, requires_grad=True)
    k = torch.randn(B_, nh, Nk, d, dtype=torch.float64, requires_grad=True)
    v = torch.randn(B_, nh, Nk, d, dtype=torch.float64, requires_grad=True)
    dO = torch.randn(B_, nh, Nq, d, dtype=torch.float64)
    m64 = mask.double() if mask is not None else None
    bias_rows = O.bias_table(p, "t.", table.double())
    bias_rows.retain_grad()
    # forward exactly as the oracle composes it, with the bias rows as an explicit leaf for the histogram check
    attn = torch.nn.functional.normalize(q, dim=-1) @ torch.nn.functional.normalize(k, dim=-1).transpose(-2, -1)
    attn = attn * O.logit_scale(p, "t.")
    attn = attn + bias_rows[index.reshape(-1)].view(Nq, Nk, nh).permute(2, 0, 1).unsqueeze(0)
    if m64 is not None:
        attn = (attn.view(B_ // nW, nW, nh, Nq, Nk) + m64.unsqueeze(1).unsqueeze(0)).view(-1, nh, Nq, Nk)
    out = torch.softmax(attn, dim=-1) @ v
    with torch.no_grad():
        ref = O.cosine_attention(q, k, v, p, "t.", table.double(), index, m64)
    assert torch.allclose(out, ref, atol=1e-12)
    out.backward(dO)
    dq, dk, dv, dscale, dbias = BM.attention_backward(q.detach(), k.detach(), v.detach(), dO, p["t.logit_scale"].detach().view(-1),
                                                      bias_rows.detach(), index, m64)
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert rel(dq, q.grad) < 1e-9 and rel(dk, k.grad) < 1e-9 and rel(dv, v.grad) < 1e-9
    assert rel(dbias, bias_rows.grad) < 1e-9
    g = p["t.logit_scale"].grad.view(-1)
    assert rel(dscale, g) < 1e-9 and g[1].item() == 0.0      # the clamped head gets no gradient (efficient.py:39)


def test_normalize_backward_matches_autograd():
    torch.manual_seed(1)
    x = torch.randn(5, 7, 12, dtype=torch.float64, requires_grad=True)
    d = torch.randn(5, 7, 12, dtype=torch.float64)
    torch.nn.functional.normalize(x, dim=-1).backward(d)
    assert torch.allclose(BM.normalize_backward(x.detach(), d), x.grad, atol=1e-12)
