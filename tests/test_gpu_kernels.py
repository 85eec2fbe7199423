"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes).

The checker is the oracle's own index / mask / table machinery (pinned to the reference in
tests/test_oracle_pinned.py) evaluated in float64 on the *same fp16-rounded operands* the kernel
sees, so the tolerances below only cover fp16 rounding of the softmax weights / outputs:
  linear      : 2.5e-3 relative to row scale for fp16 outputs, 1e-3 for fp32 LayerNorm outputs
  attention   : 1.5e-3 x max|output| (fp16 softmax weights and fp16 output rounding, 2^-11 ulp)
  split-precision linear / conv (a_split / x_split = 3): 2e-5 against fp64 on the UNROUNDED operands
"""
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

from oracle import grl_oracle as O

pytestmark = pytest.mark.gpu

LOG2E = 1.4426950408889634


def _dev():
    return torch.device("cuda:0")


def _groupnorm_ref(x, gs):
    """(M, G, 32) fp64 -> per group x / max(|x|, 1e-12) * |gs| (gs == 0: pass through; gs < 0: column 31 := 1.0)."""
    nrm = x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    y = torch.where(gs.view(1, -1, 1) != 0, x / nrm * gs.abs().view(1, -1, 1).double(), x).clone()
    y[:, gs < 0, 31] = 1.0
    return y


# ------------------------------------------------------------------------------------------------
# linear
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N", [(1000, 192, 576), (513, 64, 384), (300, 128, 64), (777, 384, 192), (256, 192, 96)])
def test_linear_groupnorm_and_plain(M, K, N):
    from grl_image_restoration_amd import _lib as L, ops

    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.float16)
    b = 0.1 * torch.randn(N, generator=g)
    gs = torch.rand(N // 32, generator=g) * 10
    gs[::3] = 0.0  # pass-through groups
    gs[1::3] *= -1.0  # negative: |gs| scaling + 1.0 written into column 31 of the group (K planes)
    ref = a.to(torch.float16).double() @ w.double().t() + b.double()
    out = ops.linear(a.to(_dev()), w.to(_dev()), b.to(_dev()), epi=L.EPI_PLAIN, out_dtype=torch.float32)
    assert (out.cpu().double() - ref).abs().max() < 2e-3
    refn = _groupnorm_ref(ref.view(M, N // 32, 32), gs).reshape(M, N)
    outn = ops.linear(a.to(_dev()), w.to(_dev()), b.to(_dev()), epi=L.EPI_GROUPNORM, gscale=gs.to(_dev()), out_dtype=torch.float16)
    assert outn.dtype == torch.float16
    err = (outn.cpu().double() - refn).abs().max().item()
    assert err < 2.5e-3 * max(1.0, refn.abs().max().item()), err


@pytest.mark.parametrize("M,K,N,nreal", [(1000, 192, 192, 180), (300, 384, 192, 180), (513, 128, 128, 128), (300, 128, 64, 64)])
def test_linear_ln_residual_and_gelu(M, K, N, nreal):
    from grl_image_restoration_amd import _lib as L, ops

    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(torch.float16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.float16)
    w[nreal:] = 0
    b = 0.1 * torch.randn(N, generator=g)
    b[nreal:] = 0
    gam = 1 + 0.1 * torch.randn(N, generator=g)
    bet = 0.1 * torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    resid[:, nreal:] = 0
    add2 = torch.randn(M, N, generator=g).to(torch.float16)   # CAB branch (fp16) ...
    add2[:, nreal:] = 0
    rpi = 100                                                    # ... gated per image (rows_per_image)
    gate = torch.rand((M + rpi - 1) // rpi, N, generator=g)
    y = a.double() @ w.double().t() + b.double()
    ln = F.layer_norm(y[:, :nreal], (nreal,), gam[:nreal].double(), bet[:nreal].double(), 1e-5)
    ref = torch.zeros(M, N, dtype=torch.float64)
    gate_rows = gate[torch.arange(M) // rpi]
    ref[:, :nreal] = resid[:, :nreal].double() + 0.5 * ln + (add2.float() * gate_rows)[:, :nreal].double()
    d = _dev()
    out = ops.linear(a.to(d), w.to(d), b.to(d), epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=gam.to(d), ln_b=bet.to(d),
                     n_real=nreal, res_scale=0.5, resid=resid.to(d), add2=add2.to(d), add2_scale=gate.to(d), rows_per_image=rpi)
    assert (out.cpu().double() - ref).abs().max() < 1e-3
    if nreal < N:
        assert out[:, nreal:].abs().max().item() == 0.0
    outg = ops.linear(a.to(d), w.to(d), b.to(d), epi=L.EPI_GELU)
    refg = F.gelu(y)
    assert (outg.cpu().double() - refg).abs().max() < 2e-2


@pytest.mark.parametrize("M,C,Hd", [(1000, 180, 360), (4099, 180, 360), (777, 128, 256), (300, 64, 128), (50, 60, 120)])
def test_fused_mlp(M, C, Hd):
    """grl_mlp_fwd = Mlp.forward (swin_v1_block.py:37-43) + norm2 + residual (efficient.py:554) in one kernel,
    against fp64 torch on the fp16-rounded operands, and against the two-kernel path (fc1+GELU, fc2+LN)."""
    from grl_image_restoration_amd import _lib as L, ops

    CP, HP = (C + 31) // 32 * 32, (Hd + 31) // 32 * 32
    g = torch.Generator().manual_seed(11)
    x = torch.zeros(M, CP)
    x[:, :C] = torch.randn(M, C, generator=g)
    w1 = torch.randn(Hd, C, generator=g) / math.sqrt(C)
    b1 = 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(C, Hd, generator=g) / math.sqrt(Hd)
    b2 = torch.zeros(CP); b2[:C] = 0.1 * torch.randn(C, generator=g)
    gam = torch.zeros(CP); gam[:C] = 1 + 0.1 * torch.randn(C, generator=g)
    bet = torch.zeros(CP); bet[:C] = 0.1 * torch.randn(C, generator=g)
    h16 = lambda t: t.to(torch.float16).double()
    hid = F.gelu(h16(x[:, :C]) @ h16(w1).t() + b1.double())
    y = h16(hid) @ h16(w2).t() + b2[:C].double()
    ref = x[:, :C].double() + 0.5 * F.layer_norm(y, (C,), gam[:C].double(), bet[:C].double(), 1e-5)
    d = _dev()
    blob = ops.pack_mlp(w1.to(d), b1.to(d), w2.to(d), CP, HP)
    out = ops.mlp(x.to(d), blob, b2.to(d), gam.to(d), bet.to(d), Hpad=HP, n_real=C, res_scale=0.5).cpu()
    err = (out[:, :C].double() - ref).abs().max().item()
    assert err < 2e-3, err
    if C < CP:
        assert out[:, C:].abs().max().item() == 0.0
    # two-kernel path on the same operands
    W1 = torch.zeros(HP, CP); W1[:Hd, :C] = w1
    W2 = torch.zeros(CP, HP); W2[:C, :Hd] = w2
    B1 = torch.zeros(HP); B1[:Hd] = b1
    hbuf = ops.linear(x.to(d), W1.to(torch.float16).to(d), B1.to(d), epi=L.EPI_GELU)
    out2 = ops.linear(hbuf, W2.to(torch.float16).to(d), b2.to(d), epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=gam.to(d),
                      ln_b=bet.to(d), n_real=C, res_scale=0.5, resid=x.to(d)).cpu()
    assert (out - out2).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,C,nslots", [(1000, 180, 18), (4099, 180, 18), (700, 128, 12), (300, 64, 12), (333, 60, 3)])
def test_streaming_qkv(M, C, nslots):
    """grl_qkv_fwd against the weights-resident GROUPNORM linear on the same operands (bit-level agreement is not
    required: accumulation order differs) and against fp64 torch."""
    from grl_image_restoration_amd import _lib as L, ops

    CP = (C + 31) // 32 * 32
    g = torch.Generator().manual_seed(12)
    x = torch.zeros(M, CP)
    x[:, :C] = torch.randn(M, C, generator=g)
    w = torch.zeros(nslots * 32, CP)
    w[:, :C] = torch.randn(nslots * 32, C, generator=g) / math.sqrt(C)
    b = 0.1 * torch.randn(nslots * 32, generator=g)
    gs = torch.rand(nslots, generator=g) * 10
    gs[2::3] = 0.0  # pass-through slots (v)
    gs[1::3] *= -1.0  # K slots: column 31 := 1.0
    w.view(nslots, 32, CP)[1::3, 31] = 0  # (the pad row of a K slot has zero weights, as the model packs it)
    d = _dev()
    out = ops.qkv(x.to(d), ops.pack_qkv(w.to(d), b.to(d), gs.to(d)), nslots).cpu()
    assert out.dtype == torch.float16
    ref = (x.to(torch.float16).double() @ w.to(torch.float16).double().t() + b.double()).view(M, nslots, 32)
    ref = _groupnorm_ref(ref, gs).permute(1, 0, 2)
    err = (out.double() - ref).abs().max().item()
    assert err < 2.5e-3 * max(1.0, ref.abs().max().item()), err
    assert (out[1::3, :, 31] == 1.0).all()
    old = ops.linear(x.to(d), w.to(torch.float16).to(d), b.to(d), epi=L.EPI_GROUPNORM, gscale=gs.to(d), planes=True).cpu()
    assert (out.float() - old.float()).abs().max().item() <= 2.5e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,C,Hd,rpi", [(1000, 180, 360, 500), (4099, 180, 360, 4099), (1300, 128, 256, 650)])
def test_fused_block_tail(M, C, Hd, rpi, regs=False):
    """grl_block_tail_fwd (proj + norm1 + residual + gated CAB + MLP + norm2 + residual) against the separate kernels
    (LN_RES linear with add2, then the fused MLP) and against fp64 torch on fp16-rounded operands.  regs: through the
    register-resident kernel (rblob)."""
    from grl_image_restoration_amd import _lib as L, ops

    CP, HP = (C + 31) // 32 * 32, (Hd + 31) // 32 * 32
    g = torch.Generator().manual_seed(21)
    pad = lambda t, n: torch.cat([t, torch.zeros(*t.shape[:-1], n - t.shape[-1])], dim=-1)
    x = pad(torch.randn(M, C, generator=g), CP)
    att = torch.randn(M, CP, generator=g).to(torch.float16)
    cab = pad(torch.randn(M, C, generator=g), CP).to(torch.float16)
    nimg = (M + rpi - 1) // rpi
    gate = pad(torch.rand(nimg, C, generator=g), CP)
    wp = torch.zeros(CP, CP); wp[:C] = torch.randn(C, CP, generator=g) / math.sqrt(CP)
    bp = pad(0.1 * torch.randn(C, generator=g), CP)
    g1, b1n = pad(1 + 0.1 * torch.randn(C, generator=g), CP), pad(0.1 * torch.randn(C, generator=g), CP)
    w1 = torch.randn(Hd, C, generator=g) / math.sqrt(C)
    b1 = 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(C, Hd, generator=g) / math.sqrt(Hd)
    b2 = pad(0.1 * torch.randn(C, generator=g), CP)
    g2, b2n = pad(1 + 0.1 * torch.randn(C, generator=g), CP), pad(0.1 * torch.randn(C, generator=g), CP)
    d = _dev()
    blob = ops.pack_mlp(w1.to(d), b1.to(d), w2.to(d), CP, HP)
    rblob = ops.pack_tail_regs(wp.to(d), w1.to(d), b1.to(d), w2.to(d)) if regs else None
    out = ops.block_tail(att.to(d), x.to(d), cab.to(d), gate.to(d), rpi, ops.pack_proj(wp.to(d)), bp.to(d), g1.to(d), b1n.to(d),
                         blob, b2.to(d), g2.to(d), b2n.to(d), Hpad=HP, n_real=C, res_scale=0.5, rblob=rblob).cpu()
    # separate kernels
    r1 = ops.linear(att.to(d), wp.to(torch.float16).to(d), bp.to(d), epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=g1.to(d),
                    ln_b=b1n.to(d), n_real=C, res_scale=0.5, resid=x.to(d), add2=cab.to(d), add2_scale=gate.to(d), rows_per_image=rpi)
    out2 = ops.mlp(r1, blob, b2.to(d), g2.to(d), b2n.to(d), Hpad=HP, n_real=C, res_scale=0.5).cpu()
    assert (out - out2).abs().max().item() < 2e-3
    # fp64 reference
    h16 = lambda t: t.to(torch.float16).double()
    y = att.double() @ h16(wp[:C]).t() + bp[:C].double()
    gr = gate[torch.arange(M) // rpi][:, :C].double()
    r1r = x[:, :C].double() + 0.5 * F.layer_norm(y, (C,), g1[:C].double(), b1n[:C].double(), 1e-5) + cab[:, :C].double() * gr
    hid = F.gelu(h16(r1r.float()) @ h16(w1).t() + b1.double())
    y2 = h16(hid.float()) @ h16(w2).t() + b2[:C].double()
    ref = r1r + 0.5 * F.layer_norm(y2, (C,), g2[:C].double(), b2n[:C].double(), 1e-5)
    assert (out[:, :C].double() - ref).abs().max().item() < 3e-3
    if C < CP:
        assert out[:, C:].abs().max().item() == 0.0


@pytest.mark.parametrize("M,rpi", [(2048, 1024), (4096, 4096), (8192, 256), (32 * 513, 32 * 19), (128, 128)])
def test_block_tail_register_resident(M, rpi):
    """The register-resident block tail (csrc/tail_regs.hip: GrlTailArgs.rblob, Cpad 192 / Hpad 384 / M, rows_per_image
    multiples of 32): the checks of the streaming kernel's test -- against the separate kernels and fp64 torch, pad channels 0 --
    with more tiles than workgroups (513 x 32 tokens: the persistent loop, both DMA buffers) and several images per workgroup."""
    test_fused_block_tail(M, 180, 360, rpi, regs=True)


def test_block_tail_rblob_layout():
    """ops.pack_tail_regs against the fragment layout include/grl_hip.h describes."""
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(5)
    d = _dev()
    wp = torch.randn(192, 192, generator=g)
    w1, b1, w2 = torch.randn(360, 180, generator=g), torch.randn(360, generator=g), torch.randn(180, 360, generator=g)
    blob = ops.pack_tail_regs(wp.to(d), w1.to(d), b1.to(d), w2.to(d)).cpu()
    fr = blob[: 8 * 48 * 1024].view(torch.float16).view(8, 48, 64, 8)
    W1 = torch.zeros(384, 192); W1[:360, :180] = w1
    W2 = torch.zeros(192, 384); W2[:180, :360] = w2
    for (w, f, lane, mat, tile, s) in [(0, 0, 0, wp, 0, 0), (3, 7, 45, wp, 3, 7), (5, 12, 63, W1, 5, 0), (2, 30, 17, W2, 2, 6), (5, 47, 40, W2, 5, 23),
                                       (6, 0, 1, W1, 6, 0), (6, 35, 33, W1, 8, 11), (7, 13, 62, W1, 10, 1)]:
        row, c0 = 32 * tile + (lane & 31), 16 * s + 8 * (lane >> 5)
        assert torch.equal(fr[w, f, lane], mat[row, c0 : c0 + 8].to(torch.float16)), (w, f, lane)
    assert torch.equal(blob[8 * 48 * 1024 :].view(torch.float32)[:360], b1) and blob[8 * 48 * 1024 :].view(torch.float32)[360:].abs().max() == 0


@pytest.mark.parametrize("B,H,W,C,nslots,nanc", [(2, 4, 64, 180, 18, 3), (1, 6, 128, 180, 18, 3), (3, 2, 64, 128, 12, 2), (1, 8, 64, 64, 12, 2),
                                                  (2, 4, 64, 180, 18, 0)])
def test_qkv_anchor_one_pass(B, H, W, C, nslots, nanc):
    """grl_qkv_anchor_fwd (q/k/v + 2x2-pooled anchor projection in one pass, 32-token MFMA tiles) against fp64 torch on the
    fp16-rounded operands, and against the two round-2 kernels it replaces (grl_qkv_fwd, grl_linear_fwd with pooling)."""
    from grl_image_restoration_amd import _lib as L, ops

    CP = (C + 31) // 32 * 32
    M = B * H * W
    g = torch.Generator().manual_seed(21)
    x = torch.zeros(M, CP)
    x[:, :C] = torch.randn(M, C, generator=g)

    def slotted(n):
        w = torch.zeros(n * 32, CP)
        w[:, :C] = torch.randn(n * 32, C, generator=g) / math.sqrt(C)
        return w, 0.1 * torch.randn(n * 32, generator=g)

    w, b = slotted(nslots)
    gs = torch.rand(nslots, generator=g) * 10
    gs[2::3] = 0.0  # pass-through slots (v)
    gs[1::3] *= -1.0  # K slots: column 31 := 1.0
    w.view(nslots, 32, CP)[1::3, 31] = 0
    d = _dev()
    if nanc:
        wa, ba = slotted(nanc)
        wa.view(nanc, 32, CP)[:, 31] = 0
        ga = torch.full((nanc,), -1.0)
        blob = ops.pack_qkv_anchor(w.to(d), b.to(d), gs.to(d), wa.to(d), ba.to(d), ga.to(d))
    else:
        blob = ops.pack_qkv_anchor(w.to(d), b.to(d), gs.to(d))
    out, anc = ops.qkv_anchor(x.to(d), blob, nslots, nanc, B, H, W)
    out = out.cpu()
    ref = (x.to(torch.float16).double() @ w.to(torch.float16).double().t() + b.double()).view(M, nslots, 32)
    ref = _groupnorm_ref(ref, gs).permute(1, 0, 2)
    err = (out.double() - ref).abs().max().item()
    assert err < 2.5e-3 * max(1.0, ref.abs().max().item()), err
    assert (out[1::3, :, 31] == 1.0).all()
    old = ops.linear(x.to(d), w.to(torch.float16).to(d), b.to(d), epi=L.EPI_GROUPNORM, gscale=gs.to(d), planes=True).cpu()
    assert (out.float() - old.float()).abs().max().item() <= 2.5e-3 * max(1.0, ref.abs().max().item())
    if nanc:
        anc = anc.cpu()
        # the kernel pools the per-token projections (of fp16-rounded tokens); the reference pools first: same value up to rounding
        y = (x.to(torch.float16).double() @ wa.to(torch.float16).double().t()).view(B, H, W, nanc * 32)
        pooled = F.avg_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).reshape(-1, nanc, 32) + ba.double().view(1, nanc, 32)
        aref = _groupnorm_ref(pooled, ga).permute(1, 0, 2)
        assert anc.shape == (nanc, M // 4, 32)
        assert (anc.double() - aref).abs().max().item() < 2.5e-3
        assert (anc[:, :, 31] == 1.0).all()
        old_a = ops.linear(x.to(d), wa.to(torch.float16).to(d), ba.to(d), epi=L.EPI_GROUPNORM, gscale=ga.to(d), pool=(2, H, W), planes=True).cpu()
        assert (anc.float() - old_a.float()).abs().max().item() < 4e-3   # (the old kernel rounds the POOLED token to fp16)


@pytest.mark.parametrize("B,H,W", [(2, 4, 64), (1, 6, 128), (1, 66, 64)])
def test_qkv_anchor_split_precision(B, H, W):
    """grl_qkv_anchor_fwd with lo_blob (round 4): W_hi.x_hi + W_hi.x_lo + 2^-e W_lo8.x_8 for the normalised slots of the GRL-Base
    layout.  (1) The kernel computes exactly that expression (fp64 emulation of every rounding point, one fp16 ulp of output
    rounding allowed); (2) the expression is ~2^-15 away from the exact product where plain fp16 operands are 2^-11.5 away;
    (3) the kernel's planes are measurably closer to the exact ones than the unsplit kernel's."""
    from grl_image_restoration_amd import ops

    C, CP, nslots, nanc = 180, 192, 18, 3
    M = B * H * W
    g = torch.Generator().manual_seed(77)
    x = torch.zeros(M, CP)
    x[:, :C] = torch.randn(M, C, generator=g) * 1.5
    x[5, 7], x[9, 100], x[M - 1, 3] = 60.0, -400.0, 9000.0     # residual-stream outliers (9000 / 16 is clamped to the e4m3 maximum)
    w = torch.zeros((nslots + nanc) * 32, CP)
    w[:, :C] = torch.randn((nslots + nanc) * 32, C, generator=g) * 0.05
    w.view(-1, 32, CP)[:, 30:, :] = 0                             # head_dim 30
    w[40, 11] = 1.3                                               # one large weight: its rounding error sets the exponent e
    b = 0.1 * torch.randn((nslots + nanc) * 32, generator=g)
    b.view(-1, 32)[:, 30:] = 0
    gs = torch.zeros(nslots + nanc)
    for base in (0, 9):
        gs[base : base + 3] = torch.tensor([144.27, 30.0, 1.4427])   # q: logit scale x log2(e)
        gs[base + 3 : base + 6] = -1.0 if base == 0 else -100.0      # k (column 31 := 1.0)
        b.view(-1, 32)[base + 6 : base + 9, 30] = 1.0                # v: constant-1 column
    gs[nslots:] = -1.0
    d = _dev()
    blob = ops.pack_qkv_anchor(w[: nslots * 32].to(d), b[: nslots * 32].to(d), gs[:nslots].to(d), w[nslots * 32 :].to(d), b[nslots * 32 :].to(d),
                               gs[nslots:].to(d))
    lo_blob = ops.pack_qkv_anchor_lo(w.to(d), gs.to(d))
    inv_e, two_e = lo_blob[:8].cpu().view(torch.float32).tolist()
    assert inv_e * two_e == 1.0 and two_e <= 2.0 ** 14
    out, anc = ops.qkv_anchor(x.to(d), blob, nslots, nanc, B, H, W, lo_blob=lo_blob)
    out0, anc0 = ops.qkv_anchor(x.to(d), blob, nslots, nanc, B, H, W)
    torch.cuda.synchronize()
    out, anc, out0, anc0 = out.cpu(), anc.cpu(), out0.cpu(), anc0.cpu()
    assert torch.isfinite(out.float()).all() and torch.isfinite(anc.float()).all()

    f8 = lambda t: t.to(torch.float8_e4m3fn).double()
    xh = x.to(torch.float16)
    xl = (x - xh.float()).to(torch.float16).double()
    x8 = f8((xh.float() / 16.0).clamp(-448.0, 448.0))
    wh = w.to(torch.float16)
    wl8 = f8(((w - wh.float()) * (two_e * 16.0)).clamp(-448.0, 448.0))
    xh, wh = xh.double(), wh.double()
    split = (gs != 0).repeat_interleave(32)
    y_plain = xh @ wh.t() + b.double()
    y_emul = y_plain + (xl @ wh.t() + (x8 @ wl8.t()) * inv_e) * split.double()
    y_exact = x.double() @ w.double().t() + b.double()

    def planes(y):
        q = _groupnorm_ref(y[:, : nslots * 32].reshape(M, nslots, 32), gs[:nslots]).permute(1, 0, 2)
        ya = y[:, nslots * 32 :].reshape(B, H, W, nanc * 32)
        pooled = F.avg_pool2d(ya.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).reshape(-1, nanc, 32)
        return q, _groupnorm_ref(pooled, gs[nslots:]).permute(1, 0, 2)

    q_em, a_em = planes(y_emul)
    q_ex, a_ex = planes(y_exact)
    q_pl, a_pl = planes(y_plain)
    # (1) the kernel is the emulated expression (+ fp32 accumulation order, + one fp16 output rounding)
    for name, got, ref in (("planes", out, q_em), ("anchors", anc, a_em)):
        gmax = (gs[:nslots] if name == "planes" else gs[nslots:]).abs().clamp_min(1.0).view(-1, 1, 1).double()
        tol = 6e-4 * ref.abs() + 1e-6 * gmax          # one fp16 output rounding + fp32 accumulation noise relative to the slot's norm
        bad = ((got.double() - ref).abs() > tol)
        if bad.any():
            i = tuple(int(v) for v in bad.nonzero()[0])
            raise AssertionError(f"{name}: {int(bad.sum())} elements off, first {i}: got {float(got[i])} expected {float(ref[i])}")
    assert (out[3:6, :, 31] == 1.0).all() and (out[12:15, :, 31] == 1.0).all() and (anc[:, :, 31] == 1.0).all()
    # (2) the expression itself, before output rounding: relative rms error of the normalised slots against the exact product
    nz = gs[:nslots] != 0
    rel = lambda a, bb: float(((a - bb)[nz][..., :30].pow(2).mean() / bb[nz][..., :30].pow(2).mean()).sqrt())
    r_em, r_pl = rel(q_em, q_ex), rel(q_pl, q_ex)
    assert r_em < 3e-5 and r_pl > 5 * r_em, (r_em, r_pl)
    # (3) the kernel outputs: the split planes are closer to the exact ones (unit-norm k slots: output rounding is the same in both)
    ks = [3, 4, 5]
    e1 = float((out[ks].double() - q_ex[ks])[..., :30].pow(2).mean().sqrt())
    e0 = float((out0[ks].double() - q_ex[ks])[..., :30].pow(2).mean().sqrt())
    assert e1 < 0.9 * e0, (e1, e0)
    # pass-through slots are untouched by the split
    assert torch.equal(out[6:9], out0[6:9]) and torch.equal(out[15:18], out0[15:18])


def test_linear_pooled_anchor():
    """AnchorLinear: avg-pool df x df (mixed_attn_block.py:727-736) fused into the A load."""
    from grl_image_restoration_amd import _lib as L, ops

    B, H, W, CP, df, N = 2, 16, 24, 192, 4, 96
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B * H * W, CP, generator=g)
    w = (torch.randn(N, CP, generator=g) / math.sqrt(CP)).to(torch.float16)
    b = 0.1 * torch.randn(N, generator=g)
    pooled = F.avg_pool2d(x.view(B, H, W, CP).permute(0, 3, 1, 2), df, df).permute(0, 2, 3, 1).reshape(-1, CP)
    ref = pooled.to(torch.float16).double() @ w.double().t() + b.double()
    d = _dev()
    out = ops.linear(x.to(d), w.to(d), b.to(d), epi=L.EPI_PLAIN, out_dtype=torch.float32, pool=(df, H, W))
    assert out.shape[0] == B * (H // df) * (W // df)
    assert (out.cpu().double() - ref).abs().max() < 2e-3  # pooled value may round differently by 1 fp16 ulp


def test_layernorm():
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(1001, 192, generator=g) * 3 + 1
    x[:, 180:] = 0
    gam, bet = torch.randn(192, generator=g), torch.randn(192, generator=g)
    ref = F.layer_norm(x[:, :180], (180,), gam[:180], bet[:180], 1e-5)
    d = _dev()
    out = ops.layernorm(x.to(d), gam.to(d), bet.to(d), 180).cpu()
    assert (out[:, :180] - ref).abs().max() < 2e-5
    assert out[:, 180:].abs().max() == 0


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _slots(x, d, ones=False, one31=False):
    """(..., nh, d) float -> (..., nh*32) fp16 slots (zero padded; optional constant-1 column at d / at 31)."""
    pad = torch.zeros(*x.shape[:-1], 32)
    pad[..., :d] = x
    if ones and d < 32:
        pad[..., d] = 1.0
    if one31:
        pad[..., 31] = 1.0
    return pad.reshape(*x.shape[:-2], -1).to(torch.float16)


def _windows(t, B, H, W, win, shift, nh):
    """token matrix (B*H*W, nh*32) -> (B_, nh, N, 32) in the reference's roll+partition order."""
    x = t.float().view(B, H, W, nh * 32)
    if shift[0] or shift[1]:
        x = torch.roll(x, shifts=(-shift[0], -shift[1]), dims=(1, 2))
    x = O.partition(x, win).reshape(-1, win[0] * win[1], nh, 32)
    return x.permute(0, 2, 1, 3)


CASES = [
    # name, mode, (H, W), window/stripe, shift, df, nh, d
    ("win8_shift", "w", (32, 32), (8, 8), (4, 4), 1, 3, 30),
    ("win8_noshift", "w", (16, 24), (8, 8), (0, 0), 1, 2, 32),
    ("win32_shift", "w", (64, 64), (32, 32), (16, 16), 1, 3, 30),
    ("win12_ragged", "w", (24, 36), (12, 12), (6, 6), 1, 3, 30),
    ("win16_tiny", "w", (32, 32), (16, 16), (8, 8), 1, 2, 16),
    ("a2w_64_df2", "a2w", (128, 128), (64, 64), (32, 32), 2, 3, 30),
    ("w2a_64_df2", "w2a", (128, 128), (64, 64), (32, 32), 2, 3, 30),
    ("a2w_64_noshift", "a2w", (64, 128), (64, 64), (0, 0), 4, 2, 32),
    ("a2w_64_df4_tiny", "a2w", (64, 64), (64, 64), (32, 32), 4, 2, 16),
    ("w2a_64x128_df2", "w2a", (64, 128), (64, 128), (32, 64), 2, 3, 30),
    ("w2a_8x16_df4", "w2a", (16, 32), (8, 16), (4, 8), 4, 3, 30),     # 8 anchors (< one key tile)
    ("a2w_8x16_df4", "a2w", (16, 32), (8, 16), (4, 8), 4, 3, 30),
    ("w2a_16x8_df4", "w2a", (32, 16), (16, 8), (8, 4), 4, 3, 30),     # anchors 4x2 -> generic gather path
    ("a2w_48x96_df4", "a2w", (96, 96), (48, 96), (24, 48), 4, 3, 30),
    ("w2a_48x96_df4", "w2a", (96, 96), (48, 96), (24, 48), 4, 3, 30),
    ("w2a_groups_shift_one_axis", "w2a", (32, 32), (8, 32), (4, 0), 4, 3, 30),
    # head_dim 32 (GRL-Small) on 32-aligned shapes: row-streaming kernel with offset / denominator on the VALU
    ("win32_shift_d32", "w", (64, 64), (32, 32), (16, 16), 1, 2, 32),
    ("a2w_dn_d32", "a2w", (128, 128), (64, 128), (32, 64), 4, 2, 32),     # dn geometry: 16x32 anchors, 64x128 stripes
    ("w2a_dn_d32", "w2a", (128, 128), (64, 128), (32, 64), 4, 2, 32),
    ("a2w_64_df2_d32_noshift", "a2w", (64, 128), (64, 64), (0, 0), 2, 2, 32),
    # tall stripes (every other block of the dn geometry): 32-aligned only on the transposed view of the grids
    ("a2w_dn_tall_d32", "a2w", (128, 128), (128, 64), (64, 32), 4, 2, 32),
    ("w2a_dn_tall_d32", "w2a", (128, 128), (128, 64), (64, 32), 4, 2, 32),
    ("w2a_tall_d30", "w2a", (128, 64), (128, 32), (64, 16), 2, 3, 30),
]


def _attention_case(case, offset, planes, scale_hi, out_dtype=torch.float16, want_lse=False, transposed=False):
    """offset: 'lazy'   -- running offset in head-dim slot 31 (k carries 1.0 there), fast kernel for 32-aligned shapes;
               'online' -- no slot-31 contract: generic kernel, ordinary online softmax."""
    from grl_image_restoration_amd import ops, tables

    name, mode, (H, W), win, shift, df, nh, d = case
    B = 2
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    awin, ashift = (win[0] // df, win[1] // df), (shift[0] // df, shift[1] // df)
    Ha, Wa = H // df, W // df
    scale = torch.rand(nh, generator=g) * (60 if scale_hi else 12) + (40 if scale_hi else 4)   # up to the clamp (100)
    if mode == "w":
        qg = kg = (H, W, win, shift)
    elif mode == "a2w":
        qg, kg = (Ha, Wa, awin, ashift), (H, W, win, shift)
    else:
        qg, kg = (H, W, win, shift), (Ha, Wa, awin, ashift)

    def rnd(Hh, Ww):
        return torch.randn(B * Hh * Ww, nh, d, generator=g)

    qf = F.normalize(rnd(qg[0], qg[1]), dim=-1) * (scale * LOG2E).view(1, nh, 1)
    kf = F.normalize(rnd(kg[0], kg[1]), dim=-1)
    vf = rnd(kg[0], kg[1])
    ones = d < 32
    one31 = offset == "lazy" and d <= 30
    qs, ks, vs = _slots(qf, d), _slots(kf, d, one31=one31), _slots(vf, d, ones)
    rows = (qg[2][0] + kg[2][0] - 1) * (qg[2][1] + kg[2][1] - 1)
    bias = torch.rand(rows, nh, generator=g) * 16
    tab_k = tables.kernel_table(bias)                             # what the kernel receives (reversed, padded)
    tab = torch.flip(tab_k[:, : tab_k.shape[1] - (-rows) % 4], dims=(1,))  # forward order for the reference
    if transposed:   # the launch sees the transposed view of every grid and the transposed table; the reference does not change
        tab_k = tables.kernel_table(ops.transpose_table(bias, qg[2], kg[2]))
    masked = shift[0] > 0 or shift[1] > 0
    if mode == "w":
        index = O.rel_index(win)
        mask = O.shift_mask((H, W), win, shift, mode="w") if masked else None
    else:
        index = O.rel_index(win, df, mode == "w2a")
        mask = O.shift_mask((H, W), win, shift, df, mode) if masked else None
    assert int(index.max()) == rows - 1 and int(index.min()) == 0

    qw = _windows(qs, B, qg[0], qg[1], qg[2], qg[3], nh)
    kw = _windows(ks, B, kg[0], kg[1], kg[2], kg[3], nh)
    vw = _windows(vs, B, kg[0], kg[1], kg[2], kg[3], nh)
    s = qw[..., :d].double() @ kw[..., :d].double().transpose(-1, -2)   # (slot 31 belongs to the kernel's offset)
    s = s + tab.double()[:, index.reshape(-1)].view(nh, *index.shape).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        s = (s.view(B, nW, nh, *index.shape) + (mask.double() * LOG2E).unsqueeze(1).unsqueeze(0)).view(-1, nh, *index.shape)
    lse_ref = torch.log2(torch.exp2(s - s.max(dim=-1, keepdim=True).values).sum(-1)) + s.max(dim=-1).values   # (B_, nh, Nq)
    s = s - s.max(dim=-1, keepdim=True).values
    p = torch.exp2(s)
    ref = (p @ vw.double()) / p.sum(-1, keepdim=True)  # (B_, nh, Nq, 32)

    def unwin(t, c):   # (B_, nh, Nq, c) -> (tokens, nh, c) in token order
        t = t.permute(0, 2, 1, 3).reshape(-1, qg[2][0], qg[2][1], nh * c)
        t = O.unpartition(t, qg[2], (qg[0], qg[1]))
        if qg[3][0] or qg[3][1]:
            t = torch.roll(t, shifts=(qg[3][0], qg[3][1]), dims=(1, 2))
        return t.reshape(B * qg[0] * qg[1], nh, c)

    ref = unwin(ref, 32)[..., :d]
    lse_ref = unwin(lse_ref.unsqueeze(-1), 1)[..., 0]     # (tokens, nh)

    dev = _dev()
    TG = ops.TokenGrid

    def lay(t):  # token-major [tokens, nh*32] or head planes [nh, tokens, 32]
        return t.view(t.shape[0], nh, 32).permute(1, 0, 2).contiguous().to(dev) if planes else t.to(dev)

    qd, kd, vd = lay(qs), lay(ks), lay(vs)
    out = lay(torch.zeros(B * qg[0] * qg[1], nh * 32, dtype=out_dtype))
    lse = torch.zeros(nh, B * qg[0] * qg[1], dtype=torch.float32, device=dev) if want_lse else None
    grids = (TG(qd, 0, qg[0], qg[1], qg[2][0], qg[2][1], qg[3][0], qg[3][1]),
             TG(kd, 0, kg[0], kg[1], kg[2][0], kg[2][1], kg[3][0], kg[3][1]),
             TG(vd, 0, kg[0], kg[1], kg[2][0], kg[2][1], kg[3][0], kg[3][1]),
             TG(out, 0, qg[0], qg[1], qg[2][0], qg[2][1], qg[3][0], qg[3][1]))
    if transposed:
        grids = tuple(g_.T() for g_ in grids)
    ops.attention(
        *grids,
        B=B, nh=nh, table=tab_k.to(dev), masked=masked, ones_col=d if ones else -1, head_dim=d,
        k_one31=one31, lazy_floor=tables.lazy_floor(scale).to(dev) if offset == "lazy" else None, lse=lse,
        lazy_ceil=tables.lazy_ceil(scale, tab_k).to(dev) if offset == "lazy" else None,
    )
    torch.cuda.synchronize()
    got = (out.permute(1, 0, 2) if planes else out.view(-1, nh, 32)).float().cpu()[..., :d]
    assert torch.isfinite(got).all()
    err = (got.double() - ref).abs().max().item()
    print(f"{name} {offset} hi={scale_hi}: max|err| = {err:.3e} (ref max {ref.abs().max().item():.2f})")
    tol = (4e-4 if out_dtype == torch.float32 else 1.5e-3) * (3.0 if scale_hi else 1.0)   # fp32 out: fp16 weights / values remain
    assert err < tol * max(1.0, ref.abs().max().item()), err
    if want_lse:
        e2 = (lse.t().cpu().double() - lse_ref).abs().max().item()
        assert e2 < 2e-3, e2


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("offset", ["lazy", "online"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_attention_vs_oracle_indexing(case, offset, planes):
    _attention_case(case, offset, planes, scale_hi=False)


@pytest.mark.parametrize("offset", ["lazy", "online"])
@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("win32_shift", "win8_shift", "a2w_64_df2", "w2a_64_df2", "a2w_64_df4_tiny",
                                                               "w2a_64x128_df2", "a2w_48x96_df4", "w2a_8x16_df4", "win32_shift_d32", "a2w_dn_d32", "w2a_dn_d32")],
                         ids=lambda c: c[0])
def test_attention_logit_scale_at_clamp(case, offset):
    """logit scales 40 .. 100 (the clamp exp(min(., ln 100)), efficient.py:39): logits span +-144 in the log2 domain, the
    fast kernel stays selected (lazy running offset) and the weights stay inside fp16."""
    _attention_case(case, offset, True, scale_hi=True)


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("offset", ["lazy", "online"])
@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("a2w_dn_tall_d32", "w2a_dn_tall_d32", "w2a_tall_d30", "win12_ragged", "win32_shift",
                                                               "a2w_48x96_df4", "w2a_8x16_df4")], ids=lambda c: c[0])
def test_attention_transposed_view(case, offset, planes):
    """GrlTokenGrid.transposed: the same attention launched on the transposed view of its grids with the transposed bias
    table (row-streaming kernel for the tall stripes, generic kernel for the ragged shapes) equals the reference."""
    from grl_image_restoration_amd import ops
    name, mode, _, win, shift, df, nh, d = case
    if name.endswith("tall_d32") and offset == "lazy":   # the point of the flag: these are only 32-aligned when transposed
        awin, ash = (win[0] // df, win[1] // df), (shift[0] // df, shift[1] // df)
        qw, kw, qs_, ks_ = (awin, win, ash, shift) if mode == "a2w" else (win, awin, shift, ash)
        sw = lambda t: (t[1], t[0])
        assert not ops.attention_rows_ok(qw, kw, qs_, ks_, True, d) and ops.attention_rows_ok(sw(qw), sw(kw), sw(qs_), sw(ks_), True, d)
    _attention_case(case, offset, planes, scale_hi=False, transposed=True, want_lse=True, out_dtype=torch.float32 if name == "win12_ragged" else torch.float16)


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("win32_shift", "w2a_64_df2", "a2w_48x96_df4", "win12_ragged", "w2a_dn_d32")], ids=lambda c: c[0])
def test_attention_fp32_output_and_lse(case):
    """fp32 output (split-precision path) and the log2-sum-exp2 side output (what the backward kernel re-normalises with)."""
    _attention_case(case, "lazy", True, scale_hi=False, out_dtype=torch.float32, want_lse=True)


# ------------------------------------------------------------------------------------------------
# 3x3 convolution (+ activation, residual, pooled sums, pixel shuffle) and the SE gate
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,act,src_bf16", [
    (2, 20, 45, 180, 180, 0, False),   # stage conv: ragged tile edges, residual
    (1, 16, 32, 180, 45, 1, False),    # CAB conv1 + exact GELU
    (2, 24, 40, 45, 180, 0, True),     # CAB conv2 (bf16 source) + pooled sums
    (1, 17, 33, 3, 180, 0, False),     # conv_first (3 input channels -> one 32-wide chunk)
    (1, 16, 16, 64, 3, 0, True),       # conv_last (3 output channels -> one 16-wide tile)
    (1, 12, 20, 128, 64, 2, False),    # conv_before_upsample + LeakyReLU
])
def test_conv3x3(B, H, W, Cin, Cout, act, src_bf16):
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(7)
    CinP = (Cin + 31) // 32 * 32
    CoutP = (Cout + 15) // 16 * 16
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    xt = torch.zeros(B * H * W, CinP)
    xt[:, :Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin)
    if src_bf16:
        xt = xt.to(torch.float16)
    xr = xt.float()[:, :Cin].view(B, H, W, Cin).permute(0, 3, 1, 2).to(torch.float16).double()
    ref = F.conv2d(xr, w.to(torch.float16).double(), b.double(), padding=1)
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.01)
    pooled = ref.sum(dim=(2, 3))
    resid = torch.randn(B * H * W, CoutP, generator=g)
    d = _dev()
    wp, bp = ops.pack_conv_weight(w.to(d), CinP, CoutP), ops.pack_conv_bias(b.to(d), CoutP)
    out, pool = ops.conv3x3(xt.to(d), wp, bp, B, H, W, act=act, slope=0.01, resid=resid.to(d), want_pool=True)
    got = out.cpu().double().view(B, H, W, CoutP)
    want = ref.permute(0, 2, 3, 1) + resid.double().view(B, H, W, CoutP)[..., :Cout]
    assert (got[..., :Cout] - want).abs().max().item() < 2e-3
    assert (got[..., Cout:] - resid.double().view(B, H, W, CoutP)[..., Cout:]).abs().max().item() == 0 if CoutP > Cout else True
    sums = pool.cpu().double().view(B, -1, CoutP).sum(1)[:, :Cout]
    assert (sums - pooled).abs().max().item() < 2e-3 * max(1.0, pooled.abs().max().item())


@pytest.mark.parametrize("B,H,W,with_resid", [(2, 20, 45, True), (1, 64, 64, False), (3, 8, 32, True), (1, 17, 70, True)])
def test_conv3x3_stage_shape_192(B, H, W, with_resid):
    """The stage / after-body convolution of GRL-Base (180 -> 180 channels = 192 padded, fp32 in and out, + residual; grl.py:164-170,516)
    takes csrc/conv192.hip since round 5 (32x32x16 MFMAs, weights of a 16-channel chunk by LDS-DMA, one barrier per chunk): ragged
    tile edges in both directions, with and without residual, pad channels stay what the residual holds, against the fp64 convolution
    of the fp16-rounded operands."""
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(17)
    C, CP = 180, 192
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
    b = 0.1 * torch.randn(C, generator=g)
    xt = torch.zeros(B * H * W, CP)
    xt[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C)
    ref = F.conv2d(x.to(torch.float16).double(), w.to(torch.float16).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    resid = torch.randn(B * H * W, CP, generator=g) if with_resid else None
    d = _dev()
    wp, bp = ops.pack_conv_weight(w.to(d), CP, CP), ops.pack_conv_bias(b.to(d), CP)
    out = ops.conv3x3(xt.to(d), wp, bp, B, H, W, resid=resid.to(d) if with_resid else None)
    got = out.cpu().double().view(B, H, W, CP)
    want = ref + (resid.double().view(B, H, W, CP)[..., :C] if with_resid else 0.0)
    err = (got[..., :C] - want).abs().max().item()
    print(f"conv 192->192 {B}x{H}x{W} resid={with_resid}: max|err| = {err:.3e}")
    assert err < 2e-3
    pad_want = resid.double().view(B, H, W, CP)[..., C:] if with_resid else torch.zeros(B, H, W, CP - C, dtype=torch.float64)
    assert (got[..., C:] - pad_want).abs().max().item() == 0


@pytest.mark.parametrize("r,c,Cin", [(2, 64, 64), (3, 64, 64), (2, 3, 64), (4, 3, 96)])
def test_conv3x3_pixel_shuffle(r, c, Cin):
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(8)
    B, H, W = 2, 12, 20
    Cout = c * r * r
    cg = (c + 3) // 4 * 4
    CoutP = (cg * r * r + 15) // 16 * 16
    x = torch.randn(B, Cin, H, W, generator=g).to(torch.float16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    ref = F.pixel_shuffle(F.conv2d(x.double(), w.to(torch.float16).double(), b.double(), padding=1), r)
    d = _dev()
    xt = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(d)
    out = ops.conv3x3(xt, ops.pack_conv_weight(w.to(d), Cin, CoutP, r, cg), ops.pack_conv_bias(b.to(d), CoutP, r, cg),
                      B, H, W, shuffle_r=r, shuffle_cg=cg)
    got = out.cpu().double().view(B, H * r, W * r, cg)[..., :c].permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 2e-3


def test_se_gate():
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(9)
    B, nwg, CP, C, Cm, HW = 3, 7, 192, 180, 10, 1000
    pool = torch.randn(B * nwg, CP, generator=g) * 30
    w1, b1 = torch.randn(Cm, C, generator=g) / 10, torch.randn(Cm, generator=g) / 10
    w2, b2 = torch.randn(C, Cm, generator=g) / 3, torch.randn(C, generator=g) / 10
    mean = pool.view(B, nwg, CP).sum(1)[:, :C] / HW
    ref = torch.sigmoid(F.linear(F.relu(F.linear(mean, w1, b1)), w2, b2))
    d = _dev()
    got = ops.se_scale(pool.to(d), B, CP, C, HW, w1.to(d), b1.to(d), w2.to(d), b2.to(d)).cpu()
    assert (got[:, :C] - ref).abs().max().item() < 1e-5 and got[:, C:].abs().max().item() == 0


# ------------------------------------------------------------------------------------------------
# split-precision operands (precision='high' path) and the un-fused LayerNorm + residual
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,epi", [(1000, 64, 384, "plain"), (513, 128, 128, "gelu"), (700, 192, 576, "groupnorm"),
                                       (300, 256, 128, "plain"), (257, 384, 192, "plain")])
def test_linear_split_precision(M, K, N, epi):
    from grl_image_restoration_amd import _lib as L, ops

    g = torch.Generator().manual_seed(31)
    a = torch.randn(M, K, generator=g) * 3
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = 0.1 * torch.randn(N, generator=g)
    d = _dev()
    ref = a.double() @ w.double().t() + b.double()          # UNROUNDED operands
    w3 = ops.split3_weight(w.to(d))
    assert w3.shape == (N, 3 * K)
    if epi == "plain":
        out = ops.linear(a.to(d), w3, b.to(d), out_dtype=torch.float32, a_split=3).cpu().double()
    elif epi == "gelu":
        out = ops.linear(a.to(d), w3, b.to(d), epi=L.EPI_GELU, out_dtype=torch.float32, a_split=3).cpu().double()
        ref = F.gelu(ref)
    else:
        gs = torch.rand(N // 32, generator=g) * 10 + 1
        out = ops.linear(a.to(d), w3, b.to(d), epi=L.EPI_GROUPNORM, gscale=gs.to(d), planes=True, a_split=3).cpu().double()
        out = out.permute(1, 0, 2).reshape(M, N)
        ref = _groupnorm_ref(ref.view(M, N // 32, 32), gs).reshape(M, N)
        assert (out - ref).abs().max().item() < 2.5e-3 * ref.abs().max().item()   # fp16 planes
        return
    err = (out - ref).abs().max().item()
    e16 = (a.to(torch.float16).double() @ w.to(torch.float16).double().t() + b.double() - (a.double() @ w.double().t() + b.double())).abs().max().item()
    print(f"split-precision linear {M}x{K}x{N}: err {err:.2e} (fp16 operands would give {e16:.2e})")
    assert err < 2e-5 * max(1.0, ref.abs().max().item()) and err < e16 / 50


@pytest.mark.parametrize("M,K,N,epi", [(4096, 192, 576, "groupnorm"), (32 * 515, 192, 192, "plain"), (8192, 192, 384, "gelu"), (2048, 384, 192, "plain"),
                                       (32, 192, 576, "groupnorm"), (1024, 128, 384, "groupnorm"), (2080, 256, 128, "gelu"), (96, 128, 128, "plain"),
                                       (32 * 700, 192, 576, "plain16")])
def test_linear_split_weights_stationary(M, K, N, epi):
    """csrc/linear_split.hip (GrlLinearArgs.w_regs) against the unrounded fp64 product and against the generic split kernel:
    same three-term sum, so the two agree to fp32 accumulation order; planes, rounding-residual planes, slot-31 ones."""
    from grl_image_restoration_amd import _lib as L, ops

    g = torch.Generator().manual_seed(77)
    a = torch.randn(M, K, generator=g) * 3
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = 0.1 * torch.randn(N, generator=g)
    d = _dev()
    ref = a.double() @ w.double().t() + b.double()
    w3 = ops.split3_weight(w.to(d))
    wr = ops.pack_linear_split(w3)
    assert wr is not None
    ad = torch.zeros(M, K + 16, device=d)[:, :K]        # row stride != K
    ad.copy_(a)
    bd = b.to(d)
    if epi == "groupnorm":
        gs = torch.rand(N // 32, generator=g) * 10 + 1
        gs[1] = 0.0                       # pass-through group (a v slot)
        gs[2] = -gs[2]                    # a K plane: 1.0 in column 31
        lo_new = torch.zeros(N // 32, M, 32, dtype=torch.float16, device=d)
        lo_old = torch.zeros_like(lo_new)
        new = ops.linear(ad, w3, bd, epi=L.EPI_GROUPNORM, gscale=gs.to(d), planes=True, a_split=3, out_lo=lo_new, w_regs=wr)
        old = ops.linear(ad, w3, bd, epi=L.EPI_GROUPNORM, gscale=gs.to(d), planes=True, a_split=3, out_lo=lo_old)
        full_new = (new.double() + lo_new.double()).cpu().permute(1, 0, 2).reshape(M, N)
        full_old = (old.double() + lo_old.double()).cpu().permute(1, 0, 2).reshape(M, N)
        want = _groupnorm_ref(ref.view(M, N // 32, 32), gs).reshape(M, N)
        scale = want.abs().max().item()
        e_new, e_old = (full_new - want).abs().max().item(), (full_old - want).abs().max().item()
        print(f"linear_split {M}x{K}x{N} groupnorm: hi+lo err {e_new:.2e} (generic {e_old:.2e}) of {scale:.1f}")
        assert e_new < 3e-5 * scale and e_new < 2 * e_old + 1e-6 * scale
        assert (new.float() - old.float()).abs().max().item() <= 2.0 ** -10 * scale   # the fp16 planes: at most an ulp apart
        return
    kw = dict(a_split=3)
    if epi == "gelu":
        kw["epi"] = L.EPI_GELU
        ref = F.gelu(ref)
    if epi == "plain16":
        new = ops.linear(ad, w3, bd, out_dtype=torch.float16, w_regs=wr, **kw).float().cpu().double()
        old = ops.linear(ad, w3, bd, out_dtype=torch.float16, **kw).float().cpu().double()
        assert (new - ref).abs().max().item() < 1.5e-3 * ref.abs().max().item()
        assert (new - old).abs().max().item() <= 2.0 ** -9 * ref.abs().max().item()
        return
    out = torch.full((M, N + 32), 7.0, device=d)[:, :N]     # ldo != N; the columns behind stay untouched
    ops.linear(ad, w3, bd, out=out, out_dtype=torch.float32, w_regs=wr, **kw)
    old = ops.linear(ad, w3, bd, out_dtype=torch.float32, **kw).cpu().double()
    new = out.cpu().double()
    err, err_old = (new - ref).abs().max().item(), (old - ref).abs().max().item()
    print(f"linear_split {M}x{K}x{N} {epi}: err {err:.2e} (generic {err_old:.2e})")
    assert err < 2e-5 * max(1.0, ref.abs().max().item()) and err < 2 * err_old + 1e-6
    assert (out.as_strided((M, 32), (N + 32, 1), N) == 7.0).all()


@pytest.mark.parametrize("M,K,N,n_real,add2", [(2048, 192, 192, 180, "f32"), (1024, 384, 192, 180, None), (512, 256, 128, 128, "f16"),
                                                (64, 128, 128, 100, None), (32 * 300, 192, 192, 180, "f16"), (256, 192, 64, 60, "f32")])
def test_linear_split_layernorm_epilogue(M, K, N, n_real, add2):
    """out = resid + res_scale * LayerNorm(a W^T + b) (+ add2 * gate[image]) in the epilogue of the weights-stationary split
    kernel (GRL_EPI_LN_RES with w_regs) against fp64; pad channels of the output stay 0."""
    from grl_image_restoration_amd import _lib as L, ops

    g = torch.Generator().manual_seed(78)
    rpi = 32 * max(1, M // 64)                     # two images (the gate row changes inside the launch)
    nimg = (M + rpi - 1) // rpi
    a = torch.randn(M, K, generator=g) * 2
    w = torch.zeros(N, K)
    w[:n_real] = torch.randn(n_real, K, generator=g) / math.sqrt(K)
    b = torch.zeros(N)
    b[:n_real] = 0.1 * torch.randn(n_real, generator=g)
    gam, bet = torch.zeros(N), torch.zeros(N)
    gam[:n_real], bet[:n_real] = 1 + 0.2 * torch.randn(n_real, generator=g), 0.1 * torch.randn(n_real, generator=g)
    resid = torch.zeros(M, N)
    resid[:, :n_real] = torch.randn(M, n_real, generator=g) * 3
    ext = torch.zeros(M, N)
    ext[:, :n_real] = torch.randn(M, n_real, generator=g)
    gate = torch.rand(nimg, N, generator=g)
    res_scale = 0.7
    y = a.double() @ w.double().t() + b.double()
    yr = y[:, :n_real]
    mu, var = yr.mean(1, keepdim=True), yr.var(1, unbiased=False, keepdim=True)
    ref = torch.zeros(M, N, dtype=torch.float64)
    ref[:, :n_real] = resid[:, :n_real].double() + res_scale * ((yr - mu) / torch.sqrt(var + 1e-5) * gam[:n_real].double() + bet[:n_real].double())
    d = _dev()
    kw = {}
    if add2 is not None:
        e = ext.to(d).to(torch.float16 if add2 == "f16" else torch.float32)
        kw = dict(add2=e, add2_scale=gate.to(d), rows_per_image=rpi)
        img = torch.arange(M) // rpi
        ref[:, :n_real] += e.cpu().double()[:, :n_real] * gate.double()[img][:, :n_real]
    w3 = ops.split3_weight(w.to(d))
    out = ops.linear(a.to(d), w3, b.to(d), epi=L.EPI_LN_RES, out_dtype=torch.float32, a_split=3, w_regs=ops.pack_linear_split(w3),
                     ln_g=gam.to(d), ln_b=bet.to(d), n_real=n_real, res_scale=res_scale, resid=resid.to(d), **kw).cpu().double()
    err = (out - ref).abs().max().item()
    print(f"linear_split LN epilogue {M}x{K}x{N} ({n_real} real, add2 {add2}): err {err:.2e}")
    assert err < 3e-5 * ref.abs().max().item()
    assert out[:, n_real:].abs().max().item() == 0 if n_real < N else True


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [(2, 20, 45, 64, 64, 0), (1, 16, 32, 128, 128, 0), (1, 17, 33, 3, 64, 0), (1, 12, 20, 180, 45, 1)])
def test_conv3x3_split_precision(B, H, W, Cin, Cout, act):
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(32)
    CinP = (Cin + 31) // 32 * 32
    CoutP = (Cout + 15) // 16 * 16
    x = torch.randn(B, Cin, H, W, generator=g) * 2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    xt = torch.zeros(B * H * W, CinP)
    xt[:, :Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if act == 1:
        ref = F.gelu(ref)
    d = _dev()
    wp, bp = ops.pack_conv_weight(w.to(d), CinP, CoutP, split=3), ops.pack_conv_bias(b.to(d), CoutP)
    assert wp.shape == (9, CoutP, 3 * CinP)
    out = ops.conv3x3(xt.to(d), wp, bp, B, H, W, act=act, x_split=3)
    got = out.cpu().double().view(B, H, W, CoutP)[..., :Cout].permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    # x_split = 2: only the activations are split ([hi | lo] against the fp16 weights twice): exact in x, fp16-rounded in w
    wp2 = ops.pack_conv_weight(w.to(d), CinP, CoutP, split=2)
    assert wp2.shape == (9, CoutP, 2 * CinP)
    if (2 * CinP) % 32 == 0:
        ref2 = F.conv2d(x.double(), w.to(torch.float16).double(), b.double(), padding=1)
        if act == 1:
            ref2 = F.gelu(ref2)
        out2 = ops.conv3x3(xt.to(d), wp2, bp, B, H, W, act=act, x_split=2)
        got2 = out2.cpu().double().view(B, H, W, CoutP)[..., :Cout].permute(0, 3, 1, 2)
        assert (got2 - ref2).abs().max().item() < 2e-5 * max(1.0, ref2.abs().max().item())


def test_layernorm_res():
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(33)
    M, C, CP, rpi = 1001, 180, 192, 400
    x = torch.randn(M, CP, generator=g) * 3 + 1
    r = torch.randn(M, CP, generator=g)
    gam, bet = torch.randn(CP, generator=g), torch.randn(CP, generator=g)
    add2 = torch.randn(M, CP, generator=g)
    gate = torch.rand((M + rpi - 1) // rpi, CP, generator=g)
    ref = r[:, :C] + 0.5 * F.layer_norm(x[:, :C], (C,), gam[:C], bet[:C], 1e-5)
    d = _dev()
    out = ops.layernorm_res(x.to(d), r.to(d), gam.to(d), bet.to(d), C, res_scale=0.5).cpu()
    assert (out[:, :C] - ref).abs().max() < 3e-5 and out[:, C:].abs().max() == 0
    ref2 = ref + add2[:, :C] * gate[torch.arange(M) // rpi][:, :C]
    for dt in (torch.float32, torch.float16):
        out2 = ops.layernorm_res(x.to(d), r.to(d), gam.to(d), bet.to(d), C, res_scale=0.5, add2=add2.to(dt).to(d),
                                 add2_scale=gate.to(d), rows_per_image=rpi).cpu()
        assert (out2[:, :C] - ref2).abs().max() < (3e-5 if dt == torch.float32 else 3e-3)


def test_fp16_staging_saturates():
    """A residual-stream value beyond the fp16 range saturates to +-65504 in the operand staging instead of becoming inf."""
    from grl_image_restoration_amd import ops

    d = _dev()
    a = torch.zeros(64, 64)
    a[0, 0], a[1, 1] = 1.0e6, -3.0e5
    w = torch.eye(64).to(torch.float16)
    out = ops.linear(a.to(d), w.to(d), torch.zeros(64, device=d), out_dtype=torch.float32).cpu()
    assert torch.isfinite(out).all() and out[0, 0] == 65504.0 and out[1, 1] == -65504.0


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("win32_shift", "win12_ragged", "a2w_64_df4_tiny", "w2a_8x16_df4", "a2w_64_noshift")],
                         ids=lambda c: c[0])
def test_attention_split_precision_operands(case):
    """precision='high': q, k, v as fp16 hi + lo planes (3 QK^T terms, 2 PV terms) at logit scales up to the clamp, against fp64 on
    the UNROUNDED operands: the logit error of fp16 operands (scale * 2^-12) disappears; what remains is the fp16 rounding of the
    softmax weights.  Also the residual output plane (o_lo): o_hi + o_lo reproduces the fp32 result."""
    from grl_image_restoration_amd import ops, tables

    name, mode, (H, W), win, shift, df, nh, d = case
    B = 2
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000 + 7)
    awin, ashift = (win[0] // df, win[1] // df), (shift[0] // df, shift[1] // df)
    Ha, Wa = H // df, W // df
    scale = torch.rand(nh, generator=g) * 60 + 40
    if mode == "w":
        qg = kg = (H, W, win, shift)
    elif mode == "a2w":
        qg, kg = (Ha, Wa, awin, ashift), (H, W, win, shift)
    else:
        qg, kg = (H, W, win, shift), (Ha, Wa, awin, ashift)
    rnd = lambda Hh, Ww: torch.randn(B * Hh * Ww, nh, d, generator=g)
    q = F.normalize(rnd(qg[0], qg[1]), dim=-1) * (scale * LOG2E).view(1, nh, 1)
    k = F.normalize(rnd(kg[0], kg[1]), dim=-1)
    v = rnd(kg[0], kg[1])
    rows = (qg[2][0] + kg[2][0] - 1) * (qg[2][1] + kg[2][1] - 1)
    bias = torch.rand(rows, nh, generator=g) * 16
    tab_k = tables.kernel_table(bias)
    tab = torch.flip(tab_k[:, : tab_k.shape[1] - (-rows) % 4], dims=(1,))
    masked = shift[0] > 0 or shift[1] > 0
    if mode == "w":
        index, mask = O.rel_index(win), (O.shift_mask((H, W), win, shift, mode="w") if masked else None)
    else:
        index, mask = O.rel_index(win, df, mode == "w2a"), (O.shift_mask((H, W), win, shift, df, mode) if masked else None)

    def pad32(t, ones=False):
        p = torch.zeros(*t.shape[:-1], 32)
        p[..., :d] = t
        if ones and d < 32:
            p[..., d] = 1.0
        return p.reshape(t.shape[0], -1)

    q32, k32, v32 = pad32(q), pad32(k), pad32(v, True)
    qw = _windows(q32, B, qg[0], qg[1], qg[2], qg[3], nh)
    kw = _windows(k32, B, kg[0], kg[1], kg[2], kg[3], nh)
    vw = _windows(v32, B, kg[0], kg[1], kg[2], kg[3], nh)
    s = qw[..., :d].double() @ kw[..., :d].double().transpose(-1, -2) + tab.double()[:, index.reshape(-1)].view(nh, *index.shape).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        s = (s.view(B, nW, nh, *index.shape) + (mask.double() * LOG2E).unsqueeze(1).unsqueeze(0)).view(-1, nh, *index.shape)
    p = torch.exp2(s - s.max(dim=-1, keepdim=True).values)
    ref = (p @ vw.double()) / p.sum(-1, keepdim=True)
    ref = ref.permute(0, 2, 1, 3).reshape(-1, qg[2][0], qg[2][1], nh * 32)
    ref = O.unpartition(ref, qg[2], (qg[0], qg[1]))
    if qg[3][0] or qg[3][1]:
        ref = torch.roll(ref, shifts=(qg[3][0], qg[3][1]), dims=(1, 2))
    ref = ref.reshape(B * qg[0] * qg[1], nh, 32)[..., :d]

    dev = _dev()
    planes = lambda t: t.view(t.shape[0], nh, 32).permute(1, 0, 2).contiguous()

    def hl(t):
        hi = t.to(torch.float16)
        return planes(hi).to(dev), planes((t - hi.float()).to(torch.float16)).to(dev)

    (qh, ql), (kh, kl), (vh, vl) = hl(q32), hl(k32), hl(v32)
    TG = ops.TokenGrid
    gq = lambda t: TG(t, 0, qg[0], qg[1], qg[2][0], qg[2][1], qg[3][0], qg[3][1])
    gk = lambda t: TG(t, 0, kg[0], kg[1], kg[2][0], kg[2][1], kg[3][0], kg[3][1])
    out = torch.zeros(nh, B * qg[0] * qg[1], 32, dtype=torch.float32, device=dev)
    kw_ = dict(B=B, nh=nh, table=tab_k.to(dev), masked=masked, ones_col=d if d < 32 else -1, head_dim=d)
    ops.attention(gq(qh), gk(kh), gk(vh), gq(out), q_lo=ql, k_lo=kl, v_lo=vl, **kw_)
    plain = torch.zeros_like(out)
    ops.attention(gq(qh), gk(kh), gk(vh), gq(plain), **kw_)
    o16, o16lo = torch.zeros(nh, out.shape[1], 32, dtype=torch.float16, device=dev), torch.zeros(nh, out.shape[1], 32, dtype=torch.float16, device=dev)
    ops.attention(gq(qh), gk(kh), gk(vh), gq(o16), q_lo=ql, k_lo=kl, v_lo=vl, o_lo=o16lo, **kw_)
    torch.cuda.synchronize()
    got = out.permute(1, 0, 2).cpu()[..., :d].double()
    e_split = (got - ref).abs().max().item()
    e_plain = (plain.permute(1, 0, 2).cpu()[..., :d].double() - ref).abs().max().item()
    print(f"{name}: split operands {e_split:.2e}, fp16 operands {e_plain:.2e} (ref max {ref.abs().max().item():.2f})")
    assert e_split < 5e-4 * max(1.0, ref.abs().max().item()) and e_split < 0.5 * e_plain
    assert (o16.float() + o16lo.float() - out).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 16, 64), (1, 20, 50), (3, 8, 32), (1, 64, 96), (5, 9, 33)])
def test_cab_conv2_register_resident_filters(shape):
    """grl_cab_conv2_fwd (csrc/cab_conv2.hip) against torch's fp32 conv2d on the same fp16-rounded operands: output, zero pad
    channels and the partial channel sums of the SE pool; ragged tiles (H % 8, W % 32 != 0) and several images per launch."""
    from grl_image_restoration_amd import ops
    B, H, W = shape
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    Cin, Cout = 45, 180
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    bias = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(B, Cin, H, W, generator=g)
    xm = torch.zeros(B * H * W, 64, dtype=torch.float16)
    xm[:, :Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin).half()
    blob, b192 = ops.pack_cab_conv2(w.to(dev), bias.to(dev))
    out, pool = ops.cab_conv2(xm.to(dev), blob, b192, B, H, W)
    torch.cuda.synchronize()
    ref = F.conv2d(xm[:, :Cin].float().view(B, H, W, Cin).permute(0, 3, 1, 2), w.half().float(), bias, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    got = out.float().cpu()
    assert (got[:, Cout:] == 0).all()
    err = (got[:, :Cout] - ref).abs().max().item()
    print(f"cab_conv2 {shape}: max|err| = {err:.3e} (ref max {ref.abs().max().item():.2f})")
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err
    sums = pool.view(B, -1, 192).sum(1).cpu()
    ref_s = ref.view(B, H * W, Cout).sum(1)
    assert (sums[:, :Cout] - ref_s).abs().max().item() < 1e-3 * max(1.0, ref_s.abs().max().item())
    # the squeeze-excite gate from the kernel's last workgroup per image (twice: the arrival counters return to zero)
    Cm = 45
    w1, b1 = (torch.randn(Cm, Cout, generator=g) * 0.2).to(dev), (torch.randn(Cm, generator=g) * 0.1).to(dev)
    w2, b2 = (torch.randn(Cout, Cm, generator=g) * 0.2).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
    for _ in range(2):
        out2, gate = ops.cab_conv2(xm.to(dev), blob, b192, B, H, W, se=(w1, b1, w2, b2, Cout))
        torch.cuda.synchronize()
        assert torch.equal(out2, out)
        mean = ref_s.to(dev).float() / (H * W)
        gref = torch.sigmoid(torch.relu(mean @ w1.t() + b1) @ w2.t() + b2)
        assert (gate[:, :Cout] - gref).abs().max().item() < 1e-4 and (gate[:, Cout:] == 0).all()


def test_attention_pipelined_kernel_opt_in():
    """csrc/attention_pipe.hip (round 5, opt-in: GRL_ATTN_PIPE=1 -- read once per process, hence the subprocess): the software-pipelined
    kernel serves the 32-aligned head_dim <= 30 geometries; the same reference comparisons as the default kernel, at random-init and
    at clamp scales (where its first pass runs over all keys), fp16 / fp32 output and the log-sum-exp side output."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, GRL_ATTN_PIPE="1")
    sel = "(test_attention_vs_oracle_indexing or test_attention_logit_scale_at_clamp or test_attention_fp32_output_and_lse or test_attention_transposed_view) " \
          "and (win32_shift or a2w_64_df2 or w2a_64_df2 or w2a_64x128_df2 or w2a_tall_d30) and not online and not d32"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", sel], env=env, capture_output=True, text=True,
                         timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    print(tail)
    assert out.returncode == 0 and " passed" in tail and "failed" not in tail, out.stdout[-2000:]
    assert int(tail.split(" passed")[0].split()[-1]) >= 18, tail
