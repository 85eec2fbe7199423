"""GPU parity of the training path (BASELINE config 5): backward kernels, autograd wiring, optimizer step -- needs the MI355X.

Checkers: torch fp32/fp64 autograd on the CPU for the single ops, oracle/backward_math.py (pinned against autograd through the
oracle's attention) for the attention backward, and the frozen gradients of one L1 training step of the REAL reference
(tests/golden_grads/train_base2x2_sr4_64.npz) for the whole network.  Tolerance: gradients are contracted on fp16 operands
(like the forward): 1e-2 relative (norm-wise) per tensor.
"""
import json
import copy
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import backward_math as BM
from oracle import grl_oracle as O

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,K,N", [(1000, 180, 540), (513, 64, 192), (700, 360, 180), (300, 90, 90)])
def test_linear_fn_gradients(M, K, N):
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(41)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = 0.1 * torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g) * 1e-6          # L1-loss-sized gradients: below the fp16 normal range
    xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, w, b))
    (F.linear(xr, wr, br) * dy.double()).sum().backward()
    xd, wd, bd = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    y = AG.GradScaleTop.apply(AG.linear(xd, wd, bd))
    assert _rel(y, F.linear(x.double(), w.double(), b.double())) < 2e-3
    y.backward(dy.cuda())
    # the bias gradient rides the weight-gradient contraction (ones column of the padded input): fp16-rounded dy, fp32 sums
    assert _rel(xd.grad, xr.grad) < 5e-3 and _rel(wd.grad, wr.grad) < 5e-3 and _rel(bd.grad, br.grad) < 2e-3
    assert AG.grad_scale() > 1e4                         # the pass was scaled into fp16 range


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 20, 45, 180, 180), (1, 16, 32, 180, 45), (2, 12, 20, 45, 180), (1, 17, 33, 3, 64), (1, 16, 16, 64, 3)])
def test_conv3x3_fn_gradients(B, H, W, Cin, Cout):
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(42)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g) * 1e-5
    xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, w, b))
    (F.conv2d(xr, wr, br, padding=1) * dy.double()).sum().backward()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    xd = tok(x).cuda().requires_grad_(True)
    wd, bd = w.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
    y = AG.GradScaleTop.apply(AG.conv3x3(xd, wd, bd, B, H, W))
    assert _rel(y, tok(F.conv2d(x.double(), w.double(), b.double(), padding=1))) < 2e-3
    y.backward(tok(dy).cuda())
    assert _rel(xd.grad, tok(xr.grad)) < 5e-3 and _rel(wd.grad, wr.grad) < 5e-3 and _rel(bd.grad, br.grad) < 2e-3


ATT_CASES = [
    # name, mode, (H, W), window/stripe, shift, df, nh, d
    ("win8_shift", "w", (16, 16), (8, 8), (4, 4), 1, 3, 30),
    ("win32_shift", "w", (64, 64), (32, 32), (16, 16), 1, 3, 30),
    ("win12_ragged", "w", (24, 36), (12, 12), (6, 6), 1, 3, 30),
    ("win16_d32", "w", (32, 32), (16, 16), (0, 0), 1, 2, 32),
    ("a2w_64_df2", "a2w", (64, 64), (64, 64), (32, 32), 2, 3, 30),
    ("w2a_64_df2", "w2a", (64, 64), (64, 64), (32, 32), 2, 3, 30),
    ("a2w_48x96_df4", "a2w", (48, 96), (48, 96), (24, 48), 4, 2, 16),
    ("w2a_8x16_df4", "w2a", (16, 32), (8, 16), (4, 8), 4, 3, 30),
    # 64x128 stripes with /2 anchors (denoising Base): the bias table (95 x 191 rows) no longer fits LDS twice -> the dq kernel
    # accumulates its table gradient in global memory
    ("w2a_64x128_df2_bigtable", "w2a", (64, 128), (64, 128), (32, 64), 2, 1, 30, 1),
    ("a2w_64x128_df2_bigtable", "a2w", (64, 128), (64, 128), (32, 64), 2, 1, 30, 1),
    # 3 x 32 anchors per stripe: Nq % 64 == 32 -- the last wave's second query tile is a clamped duplicate that must not reach the
    # bias-table gradient
    ("a2w_6x64_df2_odd_tiles", "a2w", (12, 64), (6, 64), (0, 0), 2, 3, 30),
]


# (cases for test_attention_backward_split_launches: streamed dimension >= 3 chunks of 128 in one or both kernels)
ATT_SPLIT_CASES = ("win32_shift", "a2w_64_df2", "w2a_64_df2", "a2w_48x96_df4", "a2w_64x128_df2_bigtable", "win12_ragged")


@pytest.mark.parametrize("case", ATT_CASES, ids=[c[0] for c in ATT_CASES])
def test_attention_fn_gradients(case):
    """AttentionFn (grl_attention_fwd + grl_attention_bwd) inside the same torch glue the model uses (normalise, scale, table)
    against the closed-form backward of oracle/backward_math.py on the reference's partitioned windows."""
    from grl_image_restoration_amd import autograd as AG, tables
    from tests.test_gpu_kernels import _windows

    name, mode, (H, W), win, shift, df, nh, d = case[:8]
    B = case[8] if len(case) > 8 else 2
    g = torch.Generator().manual_seed(43)
    awin, ashift = (win[0] // df, win[1] // df), (shift[0] // df, shift[1] // df)
    Ha, Wa = H // df, W // df
    if mode == "w":
        qg = kg = (H, W, win, shift)
    elif mode == "a2w":
        qg, kg = (Ha, Wa, awin, ashift), (H, W, win, shift)
    else:
        qg, kg = (H, W, win, shift), (Ha, Wa, awin, ashift)
    Mq, Mk = B * qg[0] * qg[1], B * kg[0] * kg[1]
    q = torch.randn(Mq, nh, d, generator=g)
    k = torch.randn(Mk, nh, d, generator=g)
    v = torch.randn(Mk, nh, d, generator=g)
    scale_raw = torch.log(torch.rand(nh, generator=g) * 12 + 4)
    rows = (qg[2][0] + kg[2][0] - 1) * (qg[2][1] + kg[2][1] - 1)
    bias = torch.rand(rows, nh, generator=g) * 16
    dO = torch.randn(Mq, nh, d, generator=g) * 1e-5
    masked = shift[0] > 0 or shift[1] > 0
    if mode == "w":
        index, mask = O.rel_index(win), (O.shift_mask((H, W), win, shift, mode="w") if masked else None)
    else:
        index, mask = O.rel_index(win, df, mode == "w2a"), (O.shift_mask((H, W), win, shift, df, mode) if masked else None)

    # reference: partition into windows exactly like the oracle's attention, closed-form backward, scatter back
    def part(t, gg):
        x = t.view(B, gg[0], gg[1], nh * t.shape[-1])
        if gg[3][0] or gg[3][1]:
            x = torch.roll(x, shifts=(-gg[3][0], -gg[3][1]), dims=(1, 2))
        return O.partition(x, gg[2]).reshape(-1, gg[2][0] * gg[2][1], nh, t.shape[-1]).permute(0, 2, 1, 3)

    def unpart(t, gg):   # (B_, nh, N, c) -> (tokens, nh, c)
        c = t.shape[-1]
        x = O.unpartition(t.permute(0, 2, 1, 3).reshape(-1, gg[2][0], gg[2][1], nh * c), gg[2], (gg[0], gg[1]))
        if gg[3][0] or gg[3][1]:
            x = torch.roll(x, shifts=(gg[3][0], gg[3][1]), dims=(1, 2))
        return x.reshape(-1, nh, c)

    dq_r, dk_r, dv_r, dsc_r, dbias_r = BM.attention_backward(part(q, qg).double(), part(k, kg).double(), part(v, kg).double(),
                                                             part(dO, qg).double(), scale_raw.double(), bias.double(), index, mask)
    dq_r, dk_r, dv_r = unpart(dq_r, qg), unpart(dk_r, kg), unpart(dv_r, kg)

    dev = "cuda"
    qd, kd, vd, sd, bd = (t.clone().to(dev).requires_grad_(True) for t in (q, k, v, scale_raw, bias))
    P = lambda t: F.pad(t, (0, 32 - t.shape[-1])).permute(1, 0, 2).contiguous()
    sc = torch.clamp(sd, max=math.log(100.0)).exp() * LOG2E
    geo = dict(q=(qg[0], qg[1], qg[2][0], qg[2][1], qg[3][0], qg[3][1]), k=(kg[0], kg[1], kg[2][0], kg[2][1], kg[3][0], kg[3][1]),
               B=B, nh=nh, d=d, masked=masked, floor=tables.lazy_floor(sc.detach() / LOG2E))
    o = AG.AttentionFn.apply(P(F.normalize(qd, dim=-1) * sc.view(1, nh, 1)), P(F.normalize(kd, dim=-1)), P(vd), tables.kernel_table(bd), geo)
    out = AG.GradScaleTop.apply(o.permute(1, 0, 2)[..., :d])
    out.backward(dO.to(dev))
    errs = dict(dq=_rel(qd.grad, dq_r), dk=_rel(kd.grad, dk_r), dv=_rel(vd.grad, dv_r), dscale=_rel(sd.grad, dsc_r), dbias=_rel(bd.grad, dbias_r))
    print(name, {k_: f"{v_:.2e}" for k_, v_ in errs.items()})
    assert max(errs.values()) < 1e-2, errs


@pytest.mark.parametrize("splits", ["2", "3"])
@pytest.mark.parametrize("case", [c for c in ATT_CASES if c[0] in ATT_SPLIT_CASES], ids=[c[0] for c in ATT_CASES if c[0] in ATT_SPLIT_CASES])
def test_attention_backward_split_launches(case, splits, monkeypatch):
    """grl_attention_bwd cuts the streamed dimension over several workgroups (atomic accumulation into a zeroed destination) when
    a launch has fewer workgroups than the chip has CUs -- the anchors -> stripe-token launches at training batch sizes.  Forced
    here (GRL_ATTN_BWD_SPLITS) on geometries with at least that many 128-row chunks; the bar is the unsplit test's."""
    monkeypatch.setenv("GRL_ATTN_BWD_SPLITS", splits)
    test_attention_fn_gradients(case)


def test_gemm_tn_matches_torch():
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(44)
    M, N, K = 5000, 192, 96
    a, b = torch.randn(M, N, generator=g) * 1e-4, torch.randn(M, K, generator=g)
    c = ops.gemm_tn(a.cuda(), b.cuda(), N, K, a_scale=4096.0, out_scale=1.0 / 4096.0)[0].cpu()
    assert _rel(c, a.double().t() @ b.double()) < 2e-3
    c16 = ops.gemm_tn(a.cuda(), b.cuda().half(), N, K, a_scale=4096.0, out_scale=1.0 / 4096.0)[0].cpu()
    assert _rel(c16, a.double().t() @ b.double()) < 2e-3


def test_fused_adamw_matches_torch():
    from grl_image_restoration_amd import FusedAdamW

    g = torch.Generator().manual_seed(45)
    shapes = [(180, 180), (540,), (1,), (64, 180, 3, 3), (3, 1, 1), (5000,), (4096,), (4097,)]
    p0 = [torch.randn(s, generator=g) for s in shapes]
    pa = [t.clone().cuda().requires_grad_(True) for t in p0]
    pb = [t.clone().cuda().requires_grad_(True) for t in p0]
    kw = dict(lr=2e-4, weight_decay=1e-4)                      # config/optimizer/adamw.yaml
    oa, ob = FusedAdamW(pa, **kw), torch.optim.AdamW(pb, **kw)
    for step in range(4):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** -step)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())
    sa, sb = oa.state[pa[0]], ob.state[pb[0]]
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-4, atol=1e-8) and torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-4, atol=1e-10)
    assert pa[0]._version > 0                                  # the raw-pointer update is visible to autograd / plan stamps


def test_fused_adamw_resumes_from_and_into_torch_adamw():
    """Mid-run restore: the reference's Lightning checkpoints carry a torch.optim.AdamW state (``step`` as a 0-dim tensor
    per parameter).  FusedAdamW must load it, must not keep pointing at the moment buffers it had before the load, and its own
    state must load into torch.optim.AdamW again; the two optimizers stay in lock-step across the hand-overs."""
    from grl_image_restoration_amd import FusedAdamW

    g = torch.Generator().manual_seed(46)
    shapes = [(96, 40), (300,), (1,), (8, 20, 3, 3), (5000,)]
    p0 = [torch.randn(s, generator=g) for s in shapes]
    pa = [t.clone().cuda().requires_grad_(True) for t in p0]
    pb = [t.clone().cuda().requires_grad_(True) for t in p0]
    kw = dict(lr=1e-3, weight_decay=1e-2)
    oa, ob = FusedAdamW(pa, **kw), torch.optim.AdamW(pb, **kw)

    def both_step(k):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (0.5 ** k)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()

    for k in range(3):
        both_step(k)
    # torch -> fused, into an optimizer that has ALREADY stepped (its pointer tables are built): the loaded moments must be used
    with torch.no_grad():
        for a, b in zip(pa, pb):
            a.copy_(b)
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))   # (a live state_dict shares its tensors with the optimizer it came from)
    assert all(isinstance(oa.state[a]["step"], int) for a in pa)
    for k in range(3, 5):
        both_step(k)
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() <= 3e-6 * max(1.0, b.abs().max().item())
    # fused -> torch
    oc = torch.optim.AdamW(pb, **kw)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))
    ob = oc
    for k in range(5, 7):
        both_step(k)
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() <= 3e-6 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("G,nh,win,df", [(3, 3, (64, 64), 2), (2, 3, (32, 32), 1), (1, 2, (16, 16), 4), (2, 6, (12, 12), 1)])
def test_cpb_table_kernels_match_torch(G, nh, win, df):
    """grl_cpb_table_fwd / _bwd (csrc/cpb.hip: the bias tables of many AffineTransforms without the [G, rows, 512] hidden layer in
    memory) against the torch expression of mixed_attn_block_efficient.py:49-58 in float64: table values (reversed rows, exp2 domain,
    pad entries = row 0) and the gradients of both layers."""
    from grl_image_restoration_amd import autograd as AG, tables

    g = torch.Generator().manual_seed(61)
    coords = tables.coords_table(win, df, device="cpu")
    rows = coords.shape[0]
    rows4 = (rows + 3) // 4 * 4
    w1 = torch.randn(G, 512, 2, generator=g) * 0.7
    b1 = torch.randn(G, 512, generator=g) * 0.5
    w2 = torch.randn(G, nh, 512, generator=g) * 0.1
    d_out = torch.randn(G, nh, rows4, generator=g) * 1e-6      # (table gradients of an L1 step are this small)
    # float64 reference
    a, b, c = (t.double().requires_grad_(True) for t in (w1, b1, w2))
    h = F.relu(torch.einsum("rk,ghk->grh", coords.double(), a) + b.unsqueeze(1))
    tab = 16.0 * LOG2E * torch.sigmoid(torch.einsum("gnh,grh->gnr", c, h))            # [G, nh, rows]
    want = torch.cat([tab.flip(2), tab[:, :, :1].expand(-1, -1, rows4 - rows)], dim=2)
    (want * d_out.double()).sum().backward()
    # kernels
    idx = torch.cat([torch.arange(rows - 1, -1, -1), torch.zeros(rows4 - rows, dtype=torch.long)]).cuda()
    ad, bd, cd = (t.clone().cuda().requires_grad_(True) for t in (w1, b1, w2))
    got = AG.cpb_tables(coords.cuda(), ad, bd, cd, idx)
    assert got.shape == want.shape and (got.double().cpu() - want.detach()).abs().max().item() < 2e-5     # values up to 23
    got.backward(d_out.cuda())
    errs = dict(w1=_rel(ad.grad, a.grad), b1=_rel(bd.grad, b.grad), w2=_rel(cd.grad, c.grad))
    print(f"cpb G={G} nh={nh} rows={rows}: {errs}")
    assert max(errs.values()) < 2e-5, errs


@pytest.mark.parametrize("M,n", [(4099, 180), (777, 64), (1024, 128), (5, 256)])
def test_layernorm_train_kernels_match_torch(M, n):
    """grl_layernorm_train_fwd / grl_layernorm_bwd (csrc/ln_train.hip) against F.layer_norm and its autograd in float64."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(62)
    x = torch.randn(M, n, generator=g) * 3 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(n, generator=g), 0.1 * torch.randn(n, generator=g)
    dy = torch.randn(M, n, generator=g) * 1e-6
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    yr = F.layer_norm(xr, (n,), gr, br, 1e-5)
    yr.backward(dy.double())
    xd, gd, bd = (t.clone().cuda().requires_grad_(True) for t in (x, gamma, beta))
    y = AG.layer_norm(xd, gd, bd, 1e-5)
    assert (y.double().cpu() - yr.detach()).abs().max().item() < 5e-6
    y.backward(dy.cuda())
    errs = dict(dx=_rel(xd.grad, xr.grad), dgamma=_rel(gd.grad, gr.grad), dbeta=_rel(bd.grad, br.grad))
    print(f"layernorm M={M} n={n}: {errs}")
    assert max(errs.values()) < 5e-6, errs


@pytest.mark.parametrize("T,nh,d,anchors", [(1000, 3, 30, False), (257, 3, 30, True), (512, 2, 32, False), (300, 2, 16, True)])
def test_head_planes_kernels_match_the_torch_chain(T, nh, d, anchors, monkeypatch):
    """grl_head_planes_fwd / _bwd (csrc/planes.hip: normalise, scale, pad constants, permute, fp16 copy in one launch; dx and the
    logit-scale gradients in another) against GRL._block_planes' torch chain (GRL_PLANES_KERNEL=0) in float64-checked fp32."""
    from grl_image_restoration_amd import GRL, make_config

    m = GRL(**make_config("tiny", "yaml", upscale=2, img_size=16, depths=[1], num_heads_window=[2], num_heads_stripe=[2]))
    # (the training path leaves the fp32 planes unwritten -- nothing reads them, the attention op takes the fp16 copies -- this test
    # compares their values too)
    monkeypatch.setenv("GRL_PLANES_WRITE32", "1")
    g = torch.Generator().manual_seed(63)
    k1, v1 = (31 if d <= 30 else -1), (d if d < 32 else -1)
    ones = torch.ones(nh, device="cuda")
    if anchors:
        x0 = torch.randn(T, 1, nh, d, generator=g).cuda()
        S, one_cols = 2, (-1, k1)
    else:
        x0 = torch.randn(T, 6, nh, d, generator=g).cuda()
        S, one_cols = 6, (-1, k1, v1, -1, k1, v1)
    s_a, s_b = (torch.rand(nh, generator=g) * 100 + 5).cuda(), (torch.rand(nh, generator=g) * 100 + 5).cuda()
    grads = [(torch.randn(nh, T, 32, generator=g) * 1e-6).cuda() for _ in range(S)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GRL_PLANES_KERNEL", mode)
        x = x0.clone().requires_grad_(True)
        sa, sb = s_a.clone().requires_grad_(True), s_b.clone().requires_grad_(True)
        scales = (sa, ones) if anchors else (sa, ones, None, sb, ones, None)
        p32, p16 = m._block_planes(x.expand(-1, 2, nh, d) if anchors else x, scales, one_cols)
        assert len(p32) == len(p16) == S and all(t.shape == (nh, T, 32) for t in p32) and all(t.dtype == torch.float16 for t in p16)
        sum((a * b).sum() for a, b in zip(p32, grads)).backward()
        res[mode] = ([t.detach().clone() for t in p32], [t.detach().clone() for t in p16], x.grad.clone(), sa.grad.clone(),
                     None if anchors else sb.grad.clone())
    for a, b in zip(res["0"][0], res["1"][0]):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())
    for a, b in zip(res["0"][1], res["1"][1]):
        assert (a.float() - b.float()).abs().max().item() <= 1e-3 * max(1.0, a.float().abs().max().item())     # (fp16 roundings may differ by one ulp)
    assert _rel(res["1"][2], res["0"][2]) < 1e-5 and _rel(res["1"][3], res["0"][3]) < 1e-4
    if not anchors:
        assert _rel(res["1"][4], res["0"][4]) < 1e-4


@pytest.mark.parametrize("Cout,Cin", [(180, 180), (45, 180), (180, 45), (64, 3), (3, 64)])
def test_pack_conv_train_matches_the_torch_packing(Cout, Cin):
    """grl_pack_conv3x3 (one launch) against ops.pack_conv_weight / pack_conv_bias, plain and in the data-gradient form."""
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(64)
    w, b = torch.randn(Cout, Cin, 3, 3, generator=g).cuda(), torch.randn(Cout, generator=g).cuda()
    CinP, CoutP = (Cin + 31) // 32 * 32, (Cout + 15) // 16 * 16
    wp, bp = ops.pack_conv_train(w, b, CoutP, CinP)
    assert torch.equal(wp, ops.pack_conv_weight(w, CinP, CoutP)) and torch.equal(bp, ops.pack_conv_bias(b, CoutP))
    gin, gout = (Cout + 31) // 32 * 32, (Cin + 15) // 16 * 16
    wt, none = ops.pack_conv_train(w, None, gout, gin, flip_t=True)
    assert none is None and torch.equal(wt, ops.pack_conv_weight(w.flip(2, 3).transpose(0, 1).contiguous(), gin, gout))


@pytest.mark.parametrize("taps", [1, 9])
def test_gemm_tn_real_widths_and_virtual_ones_column(taps):
    """grl_gemm_tn on operands at their real widths (ABI 22): N = 180, K = 180 with row strides of exactly that, the bias gradient
    from the VIRTUAL ones column -- against float64, and against the launch on the padded operands [a | 0], [b | 1 | 0]."""
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(46)
    B, H, W, N, K = 2, 24, 40, 180, 180
    M = B * H * W
    a, b = (torch.randn(M, N, generator=g) * 1e-4).cuda(), torch.randn(M, K, generator=g).cuda()
    kw = dict(taps=taps, hw=(H, W) if taps == 9 else None, a_scale=4096.0, out_scale=1.0 / 4096.0)
    c, cb = ops.gemm_tn(a, b, N, K, b_ones=True, **kw)
    assert c.shape == (taps, N, K) and c.is_contiguous() and cb.shape == (N,)
    ap = torch.cat([a, torch.zeros(M, 12).cuda()], 1)
    bp = torch.cat([b, torch.ones(M, 1).cuda(), torch.zeros(M, 11).cuda()], 1)
    cp = ops.gemm_tn(ap, bp, 192, 192, **kw)
    assert _rel(c, cp[:, :N, :K]) < 1e-5 and _rel(cb, cp[taps // 2, :N, K]) < 1e-5     # same products, another summation order
    assert _rel(cb, a.double().sum(0)) < 2e-3
    if taps == 1:
        assert _rel(c[0], a.double().t() @ b.double()) < 2e-3
    else:
        ai = a.double().view(B, H, W, N).permute(0, 3, 1, 2)
        bi = F.pad(b.double().view(B, H, W, K).permute(0, 3, 1, 2), (1, 1, 1, 1))
        for t in (0, 4, 8):
            dy, dx = t // 3, t % 3
            ref = torch.einsum("bnhw,bkhw->nk", ai, bi[:, :, dy : dy + H, dx : dx + W])
            assert _rel(c[t], ref) < 2e-3


@pytest.mark.parametrize("M,K,N", [(1000, 180, 540), (700, 360, 180), (640, 180, 90), (333, 180, 180)])
def test_linear_real_widths_equal_the_padded_path(M, K, N):
    """The linear op on operands / results at their real widths (GrlLinearArgs.a_cols / n_store, virtual ones column in the weight
    gradient) against the same op on padded copies (GRL_REAL_WIDTHS=0, the path of rounds 2-5): same fp16 products -- y and dx bit
    for bit, dW / db up to the order of the atomics.  Also with a gradient that arrives as a column slice of a wider matrix."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(47)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = (0.1 * torch.randn(N, generator=g)).cuda()
    wide = (torch.randn(M, N + 8, generator=g) * 1e-6).cuda()
    res = {}
    for flag in (True, False):
        prev, AG._REAL_WIDTHS[0] = AG._REAL_WIDTHS[0], flag
        try:
            xd, wd, bd = (t.clone().requires_grad_(True) for t in (x, w, b))
            y = AG.GradScaleTop.apply(AG.linear(xd, wd, bd))
            y.backward(wide[:, 4 : 4 + N])
            res[flag] = (y.detach(), xd.grad, wd.grad, bd.grad)
        finally:
            AG._REAL_WIDTHS[0] = prev
    assert res[True][0].is_contiguous() and res[True][0].shape == (M, N)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert _rel(res[True][2], res[False][2]) < 1e-5 and _rel(res[True][3], res[False][3]) < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 40, 180, 48), (1, 24, 32, 48, 180), (1, 17, 33, 180, 64), (2, 12, 20, 64, 180)])
def test_conv3x3_real_widths_equal_the_padded_path(B, H, W, Cin, Cout):
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(48)
    M = B * H * W
    x = torch.randn(M, Cin, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).cuda()
    b = (0.1 * torch.randn(Cout, generator=g)).cuda()
    dy = (torch.randn(M, Cout, generator=g) * 1e-5).cuda()
    res = {}
    for flag in (True, False):
        prev, AG._REAL_WIDTHS[0] = AG._REAL_WIDTHS[0], flag
        try:
            xd, wd, bd = (t.clone().requires_grad_(True) for t in (x, w, b))
            y = AG.GradScaleTop.apply(AG.conv3x3(xd, wd, bd, B, H, W))
            y.backward(dy)
            res[flag] = (y.detach(), xd.grad, wd.grad, bd.grad)
        finally:
            AG._REAL_WIDTHS[0] = prev
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert _rel(res[True][2], res[False][2]) < 1e-5 and _rel(res[True][3], res[False][3]) < 1e-5


@pytest.mark.parametrize("rows_per_image,drop", [(250, True), (250, False)])
def test_layer_norm_residual_matches_the_torch_chain(rows_per_image, drop):
    """resid + alpha * keep_mask[image] * LayerNorm(x) inside the LayerNorm launches (autograd.layer_norm_residual, csrc/ln_train.hip)
    against the torch expression it replaces (F.layer_norm + addcmul), values and all four gradients."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(49)
    M, n = 4 * rows_per_image, 180
    x, r = torch.randn(M, n, generator=g).cuda() * 3 + 1, torch.randn(M, n, generator=g).cuda()
    gm, bt = (1 + 0.1 * torch.randn(n, generator=g)).cuda(), (0.1 * torch.randn(n, generator=g)).cuda()
    mask = torch.tensor([1.0, 0.0, 1.0, 1.0]).cuda() if drop else None
    dy = torch.randn(M, n, generator=g).cuda()
    outs = []
    for fused in (True, False):
        xs, rs, gs, bs = (t.clone().requires_grad_(True) for t in (x, r, gm, bt))
        if fused:
            y = AG.layer_norm_residual(rs, xs, gs, bs, 1e-5, mask, rows_per_image, 0.4)
        else:
            t = F.layer_norm(xs, (n,), gs, bs, 1e-5)
            y = rs + 0.4 * (t if mask is None else (t.view(4, -1, n) * mask.view(4, 1, 1)).view(M, n))
        y.backward(dy)
        outs.append((y.detach(), xs.grad, rs.grad, gs.grad, bs.grad))
    for a, b in zip(*outs):
        assert _rel(a, b) < 2e-5


def test_pack_linear_train_matches_the_torch_packing():
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(50)
    for N, K, Np, Kp in [(180, 180, 192, 192), (540, 180, 576, 192), (90, 180, 96, 192), (180, 360, 192, 384), (33, 70, 64, 96)]:
        w = torch.randn(N, K, generator=g).cuda()
        wp, wt = torch.full((Np, Kp), 7.0, dtype=torch.float16).cuda(), torch.full((Kp, Np), 7.0, dtype=torch.float16).cuda()
        b, bp = torch.randn(N, generator=g).cuda(), torch.full((Np,), 7.0).cuda()
        ops.pack_linear_train(w, wp, wt, b, bp)
        ref = torch.zeros(Np, Kp, dtype=torch.float16).cuda()
        ref[:N, :K] = w.half()
        assert torch.equal(wp, ref) and torch.equal(wt, ref.t().contiguous())
        assert torch.equal(bp[:N], b) and float(bp[N:].abs().sum()) == 0.0
        ops.pack_linear_train(w, wp, None, None, bp)
        assert float(bp.abs().sum()) == 0.0


@pytest.mark.parametrize("B,C,Cmid", [(8, 180, 10), (3, 64, 4), (16, 256, 64)])
def test_se_gate_kernels_match_torch(B, C, Cmid):
    """grl_se_mlp_fwd / _bwd (csrc/se_train.hip) against the torch expression of ChannelAttention's MLP, values and all five gradients."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(51)
    pool = torch.randn(B, C, generator=g).cuda()
    w1, b1 = (torch.randn(Cmid, C, generator=g) / math.sqrt(C)).cuda(), (0.1 * torch.randn(Cmid, generator=g)).cuda()
    w2, b2 = (torch.randn(C, Cmid, generator=g) / math.sqrt(Cmid)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    dy = torch.randn(B, C, generator=g).cuda()
    outs = []
    for kernel in (True, False):
        ts = [t.clone().requires_grad_(True) for t in (pool, w1, b1, w2, b2)]
        y = AG.se_gate(*ts) if kernel else torch.sigmoid(F.linear(F.relu(F.linear(ts[0], ts[1], ts[2])), ts[3], ts[4]))
        y.backward(dy)
        outs.append([y.detach()] + [t.grad for t in ts])
    for a, b in zip(*outs):
        assert a.shape == b.shape and _rel(a, b) < 2e-5


@pytest.mark.parametrize("B,HW,C,Cmid", [(4, 1000, 180, 10), (2, 64, 64, 4), (8, 4096, 180, 10)])
def test_se_residual_matches_the_torch_chain(B, HW, C, Cmid, monkeypatch):
    """x1 + u * ChannelAttention-gate(u) as pool / MLP / apply launches (autograd.se_residual, csrc/se_train.hip) against the torch chain
    it replaces (mean, two linears with ReLU / sigmoid, addcmul) -- values and the gradients of both token matrices and all four parameters."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(52)
    M = B * HW
    x1, u = torch.randn(M, C, generator=g).cuda(), torch.randn(M, C, generator=g).cuda() + 0.3
    w1, b1 = (torch.randn(Cmid, C, generator=g) / math.sqrt(C)).cuda(), (0.1 * torch.randn(Cmid, generator=g)).cuda()
    w2, b2 = (torch.randn(C, Cmid, generator=g) / math.sqrt(Cmid)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GRL_SE_KERNEL", mode)
        ts = [t.clone().requires_grad_(True) for t in (x1, u, w1, b1, w2, b2)]
        y = AG.se_residual(*ts, HW)
        y.backward(dy)
        outs.append([y.detach()] + [t.grad for t in ts])
    for a, b in zip(*outs):
        assert a.shape == b.shape and _rel(a, b) < 3e-5


@pytest.mark.parametrize("M,K,N", [(1000, 360, 180), (513, 128, 64)])
def test_linear_with_gelu_input_matches_torch(M, K, N):
    """y = gelu(x) W^T + b with the activation in the linear kernel's loader and gelu'(x) in the epilogue of the data-gradient launch
    (autograd.linear gelu_in=True) against F.linear(F.gelu(x)) in float64."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(53)
    x = torch.randn(M, K, generator=g) * 1.5
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = 0.1 * torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g) * 1e-6
    xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(F.gelu(xr), wr, br)
    (ref * dy.double()).sum().backward()
    xd, wd, bd = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    y = AG.GradScaleTop.apply(AG.linear(xd, wd, bd, gelu_in=True))
    assert _rel(y, ref.detach()) < 2e-3
    y.backward(dy.cuda())
    assert _rel(xd.grad, xr.grad) < 5e-3 and _rel(wd.grad, wr.grad) < 5e-3 and _rel(bd.grad, br.grad) < 2e-3


def test_fan_out_adds_the_consumers_gradients_in_one_launch():
    """autograd.fan_out: n aliases forward, grl_sum4 backward -- against autograd's own pairwise accumulation (2, 3 and 4 live consumers)."""
    from grl_image_restoration_amd import autograd as AG

    g = torch.Generator().manual_seed(54)
    x0 = torch.randn(1000, 180, generator=g).cuda()
    ws = [torch.randn(1000, 180, generator=g).cuda() for _ in range(4)]
    for live in (2, 3, 4):
        x = x0.clone().requires_grad_(True)
        parts = AG.fan_out(x, 4)
        sum((p * w).sum() for p, w in zip(parts[:live], ws)).backward()
        ref = sum(ws[:live])
        assert torch.allclose(x.grad, ref, rtol=0, atol=1e-6)
