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


def _grad_fixture():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_grads", "train_base2x2_sr4_64.npz")
    z = np.load(path, allow_pickle=False)
    return json.loads(str(z["meta"])), z


def test_training_step_gradients_match_reference():
    """One L1 training step of GRL-Base blocks (x4 SR, 64x64 LQ, eval mode as in the fixture): loss, input gradient, the norm
    of all 156 parameter gradients and every small gradient tensor against the REAL reference (find_unused_parameters=False
    holds: every parameter receives a gradient)."""
    from grl_image_restoration_amd import GRL

    meta, z = _grad_fixture()
    cfg = meta["cfg"]
    m = GRL(**cfg).eval()
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, meta["weight_seed"])
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x = torch.from_numpy(z["input"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(z["target"]).cuda()
    loss = (m(x) - gt).abs().mean()
    loss.backward()
    assert abs(loss.item() - meta["loss"]) < 2e-4, (loss.item(), meta["loss"])
    assert _rel(x.grad, torch.from_numpy(z["grad_input"])) < 2e-2
    names = json.loads(str(z["grad_norm_names"]))
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert set(names) == set(grads) and all(g is not None for g in grads.values())
    worst = ("", 0.0)
    for k, n in zip(names, z["grad_norms"]):
        e = abs(grads[k].norm().item() - n) / max(n, 1e-12)
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < 1e-2, (k, e)
    small = [k for k in z.files if k.startswith("grad::")]
    for k in small:
        e = _rel(grads[k[6:]], torch.from_numpy(z[k]))
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < 2e-2, (k, e)
    print(f"training step: loss {loss.item():.6f} (reference {meta['loss']:.6f}); worst gradient error {worst}")


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


def test_train_mode_steps_reduce_the_loss_and_inference_sees_the_update():
    """model.train(): stochastic depth is honoured (mixed_attn_block_efficient.py:500, grl.py:299-300), a few FusedAdamW steps on
    one batch reduce the L1 loss, and the inference path afterwards runs on the UPDATED weights (plan version stamp) and agrees
    with the differentiable path."""
    from grl_image_restoration_amd import GRL, FusedAdamW, make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[2, 2], num_heads_window=[3, 3], num_heads_stripe=[3, 3])
    torch.manual_seed(0)
    m = GRL(**cfg)
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    assert m._dpr[0] == 0.0 and abs(m._dpr[-1] - 0.1) < 1e-7
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=4, seed=11)
    lq, gt = lq.cuda(), gt.cuda()
    with torch.no_grad():
        before = m(lq).clone()                                   # inference path, initial weights
    torch.manual_seed(1)
    y1 = m(lq)
    torch.manual_seed(2)
    y2 = m(lq)
    assert not torch.equal(y1, y2)                               # different stochastic-depth draws
    opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
    losses = []
    for it in range(6):
        opt.zero_grad(set_to_none=True)
        loss = (m(lq) - gt).abs().mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        opt.step()
        losses.append(loss.item())
    print("train losses:", [f"{v:.5f}" for v in losses])
    assert losses[-1] < losses[0]
    m.eval()
    with torch.no_grad():
        after = m(lq)
    assert (after - before).abs().max().item() > 1e-4             # the fast path picked the new weights up
    diff = m(lq)                                                  # grad-enabled eval = differentiable path, no drop path
    assert (after - diff.detach()).abs().max().item() < 1e-3


def _ddp_worker(rank, world, port, ret):
    """Two replicas of a small GRL on the one GPU of the box, gloo between them (RCCL refuses two ranks on one device):
    the product's DDP wrapper + autograd path + FusedAdamW end to end."""
    import os

    import torch.distributed as dist

    from grl_image_restoration_amd import GRL, FusedAdamW, ddp, make_config

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1, 1], num_heads_window=[3, 3], num_heads_stripe=[3, 3],
                      drop_path_rate=0.0)
    torch.manual_seed(0)
    m = GRL(**cfg)
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    net = ddp.wrap(m, bucket_mb=32)          # device_ids None: both replicas live on cuda:0
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=4, seed=21)
    per = lq.shape[0] // world
    x, y = lq[rank * per : (rank + 1) * per].to(dev), gt[rank * per : (rank + 1) * per].to(dev)
    opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
    loss = (net(x) - y).abs().mean()
    loss.backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    opt.step()
    after = {k: p.detach().cpu().clone() for k, p in m.named_parameters()}
    ret[rank] = (grads, after, float(loss.detach()))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_replicas_on_the_gpu():
    """DistributedDataParallel (reference settings, tools/trainer.py:135-142) over the differentiable HIP path: the all-reduced
    gradients of two half-batch replicas equal the single-process full-batch gradients, every parameter has one
    (find_unused_parameters=False holds), and the replicas stay bit-identical after the fused optimizer step."""
    import socket

    import torch.multiprocessing as mp

    from grl_image_restoration_amd import GRL, make_config

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_ddp_worker, args=(2, port, ret), nprocs=2, join=True)
    (g0, a0, l0), (g1, a1, l1) = ret[0], ret[1]
    for k in g0:
        assert torch.equal(g0[k], g1[k]) and torch.equal(a0[k], a1[k]), k       # identical all-reduced gradients / updated weights
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1, 1], num_heads_window=[3, 3], num_heads_stripe=[3, 3],
                      drop_path_rate=0.0)
    m = GRL(**cfg)
    m.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0), strict=True)
    m = m.cuda().train()
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=4, seed=21)
    loss = (m(lq.cuda()) - gt.cuda()).abs().mean()
    loss.backward()
    assert abs(float(loss.detach()) - 0.5 * (l0 + l1)) < 1e-5
    worst = max(_rel(g0[k], p.grad) for k, p in m.named_parameters())
    print(f"DDP (2 replicas) vs single process, worst relative gradient difference: {worst:.2e}")
    assert worst < 5e-3       # fp32 atomics in the weight-gradient / table reductions and per-pass gradient scales differ


@pytest.mark.parametrize("model,geom,up,hw,task", [
    ("tiny", "yaml", 2, (32, 32), "sr"),            # stripe_groups geometry, head_dim 16, pixelshuffledirect tail, no CAB
    ("small", "dn_df4", 1, (64, 128), "dn"),        # head_dim 32 (generic attention kernels), window 16, stripes 64x128 / anchors 16x32
    ("base", "deblur", 1, (48, 96), "deblur"),      # window 12 (ragged key tiles), stripes 48x96 / anchors 12x24, CAB, no upsampler
])
def test_training_gradients_other_geometries_vs_oracle_autograd(model, geom, up, hw, task):
    """Whole-network gradients on the geometries the reference ships besides the SR checkpoint one, against torch autograd through
    the CPU oracle (itself pinned to the reference's gradients, tests/test_oracle_pinned.py) on two blocks per stage."""
    from grl_image_restoration_amd import GRL, make_config

    over = dict(depths=[2, 2], num_heads_window=[2, 2] if model != "base" else [3, 3], num_heads_stripe=[2, 2] if model != "base" else [3, 3])
    cfg = make_config(model, geom, upscale=up, img_size=hw[0], **over)
    m = GRL(**cfg).eval()
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 3)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    lq, gt = O.synthetic_pair(task, hw, up, batch=2, seed=31)
    lq, gt = lq[..., : hw[0], : hw[1]].contiguous(), gt[..., : hw[0] * up, : hw[1] * up].contiguous()
    # oracle autograd
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = lq.clone().requires_grad_(True)
    lo = (O.grl_forward(xr, cfg, sdr) - gt).abs().mean()
    lo.backward()
    # HIP path
    x = lq.cuda().requires_grad_(True)
    loss = (m(x) - gt.cuda()).abs().mean()
    loss.backward()
    assert abs(loss.item() - lo.item()) < 5e-4, (loss.item(), lo.item())
    errs = {}
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        errs[k] = _rel(p.grad, sdr[k].grad)
    ex = _rel(x.grad, xr.grad)
    srt = sorted(errs.items(), key=lambda t: -t[1])
    med = srt[len(srt) // 2][1]
    print(f"{model}/{geom}: loss {loss.item():.6f} vs {lo.item():.6f}; d/dx {ex:.2e}; median {med:.2e}; worst {[(k, round(e, 4)) for k, e in srt[:4]]}")
    # fp16 operands forward and backward: 1e-2 typical; the smallest gradients (CPB-MLP biases of late blocks) up to a few percent
    assert ex < 5e-2 and med < 1e-2 and srt[0][1] < 6e-2, srt[:4]


def test_training_gradients_per_slot_plane_path(monkeypatch):
    """The per-slot chain of head planes / bias tables (round 4; still what a block whose two branches have different head counts
    takes) against the same oracle gradients as the batched chains that are the default since round 5."""
    monkeypatch.setenv("GRL_TRAIN_BATCHED_PLANES", "0")
    test_training_gradients_other_geometries_vs_oracle_autograd("base", "deblur", 1, (48, 96), "deblur")


@pytest.mark.gpu
def test_graphed_train_step_matches_eager_steps():
    """train_graph.GraphedTrainStep: forward + L1 + backward + FusedAdamW captured once as a HIP graph and replayed follows the
    eager steps as closely as two eager runs follow each other (the gradient GEMMs accumulate with atomics and Adam's first steps
    turn noise-level gradients into +-lr updates, so no two runs are bit-identical), the optimizer's step count arrives on the
    host, and the eager / inference paths pick the updated weights up afterwards."""
    from grl_image_restoration_amd import GRL, FusedAdamW, GraphedTrainStep, make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[2, 2], num_heads_window=[3, 3], num_heads_stripe=[3, 3],
                      drop_path_rate=0.0)
    models, opts = [], []
    for _ in range(3):                                   # two eager runs (the yardstick) and the graphed one
        torch.manual_seed(0)
        m = GRL(**cfg)
        m.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0), strict=True)
        models.append(m.cuda().train())
        opts.append(FusedAdamW(models[-1].parameters(), lr=2e-4, weight_decay=1e-4))
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=2, seed=12)
    lq, gt = lq.cuda(), gt.cuda()
    loss_fn = lambda y, t: (y - t).abs().mean()
    n_replays = 3

    def eager(i):
        out = []
        for _ in range(1 + n_replays):                   # 1 warm-up step + as many as the graph replays
            opts[i].zero_grad(set_to_none=True)
            loss = loss_fn(models[i](lq), gt)
            loss.backward()
            opts[i].step()
            out.append(float(loss.detach()))
        return out[1:]

    la, lb = eager(0), eager(1)
    step = GraphedTrainStep(models[2], opts[2], loss_fn, lq, gt, warmup=1)
    lg = [float(step(lq, gt).detach()) for _ in range(n_replays)]
    step.finish()

    def dist(i, j):    # mean |difference| over all parameters
        num = sum(float((p - q).abs().sum()) for p, q in zip(models[i].parameters(), models[j].parameters()))
        return num / sum(p.numel() for p in models[i].parameters())

    d_ee, d_eg = dist(0, 1), dist(0, 2)
    l_ee = max(abs(a - b) for a, b in zip(la, lb))
    l_eg = max(abs(a - b) for a, b in zip(la, lg))
    print(f"losses eager {la} | eager {lb} | graph {lg}")
    print(f"mean |dp| eager-eager {d_ee:.3e}, eager-graph {d_eg:.3e}; max |dloss| {l_ee:.3e} / {l_eg:.3e}")
    # (round 5: this line used a fixed 2e-4 relative bound on the losses and failed when two EAGER runs differed by 2.9e-4 in the third
    # loss -- the atomics of the weight-gradient GEMMs; the bound now scales with the eager-eager yardstick like the ones below)
    assert lg[-1] < lg[0] and all(abs(a - b) <= max(2e-4 * abs(a), 4 * l_ee + 5e-5) for a, b in zip(la, lg))
    assert d_eg <= 4 * d_ee + 1e-6 and l_eg <= 4 * l_ee + 5e-5      # (absolute floors: two eager runs can also happen to agree)
    p2 = next(iter(models[2].parameters()))
    assert opts[2].state[p2]["step"] == 1 + n_replays and opts[0].state[next(iter(models[0].parameters()))]["step"] == 1 + n_replays
    # one more EAGER step on the graphed model: the optimizer state and the cached fp16 weight copies are in step
    before = dist(0, 2)
    for i in (0, 2):
        opts[i].zero_grad(set_to_none=True)
        loss_fn(models[i](lq), gt).backward()
        opts[i].step()
    assert dist(0, 2) <= 2 * before + 4 * d_ee + 1e-6
    with torch.no_grad():
        y0, y2 = models[0].eval()(lq), models[2].eval()(lq)
    assert float((y0 - y2).abs().max()) <= 2e-3        # (different weights by the noise above; the inference path sees the UPDATED ones:)
    with torch.no_grad():
        torch.manual_seed(0)
        fresh = GRL(**cfg)
        fresh.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in fresh.state_dict().items()}, 0), strict=True)
        y_init = fresh.cuda().eval()(lq)
    assert float((y2 - y_init).abs().max()) > 3 * float((y0 - y2).abs().max())


@pytest.mark.gpu
def test_graphed_train_step_follows_lr_schedule_and_resume():
    """a captured optimizer launch reads lr / weight decay from device memory that GraphedTrainStep refreshes from
    param_groups before every replay (lr 0 -> the replay moves nothing; lr back -> it moves again), an eval forward between replays
    sees the replayed update without finish(), and load_state_dict in capture mode lands in the buffers the graph points at."""
    import copy

    from grl_image_restoration_amd import GRL, FusedAdamW, GraphedTrainStep, make_config

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1], num_heads_window=[3], num_heads_stripe=[3], drop_path_rate=0.0)
    torch.manual_seed(0)
    m = GRL(**cfg).cuda().train()
    opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=0.0)
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=2, seed=12)
    lq, gt = lq.cuda(), gt.cuda()
    step = GraphedTrainStep(m, opt, lambda y, t: (y - t).abs().mean(), lq, gt, warmup=1)
    snap = lambda: torch.cat([p.detach().flatten() for p in m.parameters()]).clone()
    step(lq, gt)
    w0 = snap()
    with torch.no_grad():
        y_a = m.eval()(lq).clone()
    m.train()
    opt.param_groups[0]["lr"] = 0.0                       # what an LR scheduler does between steps
    step(lq, gt)
    torch.cuda.synchronize()
    assert torch.equal(snap(), w0), "a replay at lr = 0 must not move the weights"
    opt.param_groups[0]["lr"] = 2e-4
    step(lq, gt)
    w2 = snap()
    if not bool(torch.isfinite(w2).all()):                # (diagnosis of round 5's order-dependent failure: say WHERE)
        bad = lambda f: [k for k, p in m.named_parameters() if f(p) is not None and not bool(torch.isfinite(f(p)).all())]
        bw, bg = bad(lambda p: p), bad(lambda p: p.grad)
        bm = bad(lambda p: opt.state[p]["exp_avg"])
        pytest.fail(f"non-finite weights after a replay: loss {float(step.loss)}; {len(bw)} weights, {len(bg)} gradients, {len(bm)} first moments; "
                    f"gradients (forward order) first {bg[:5]} last {bg[-5:]}; weights first {bw[:5]}")
    assert float((w2 - w0).abs().max()) > 1e-5
    with torch.no_grad():
        y_b = m.eval()(lq)                                # no finish() in between: the plan must have been rebuilt from the new weights
    m.train()
    assert float((y_b - y_a).abs().max()) > 0
    # resume in capture mode: loaded moments land in the buffers the captured launch updates
    sd = copy.deepcopy(opt.state_dict())
    p0 = next(iter(m.parameters()))
    ptr = opt.state[p0]["exp_avg"].data_ptr()
    for s in sd["state"].values():
        s["exp_avg"].zero_(); s["exp_avg_sq"].zero_()
    opt.load_state_dict(sd)
    assert opt.state[p0]["exp_avg"].data_ptr() == ptr and float(opt.state[p0]["exp_avg"].abs().max()) == 0.0
    step(lq, gt)
    torch.cuda.synchronize()
    assert float(opt.state[p0]["exp_avg"].abs().max()) > 0.0   # the replay wrote the (re-started) moments, not stale buffers
    step.finish()


# (last in the file: these tests run replicas in spawned processes on the same GPU)
def _graphed_ddp_worker(rank, world, port, ret, wire_bf16):
    """A replica of the two-graph data-parallel step (train_graph.py) on the one GPU of the box, gloo between the replicas."""
    import os

    import torch.distributed as dist

    from grl_image_restoration_amd import GRL, FusedAdamW, GraphedTrainStep, make_config

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1, 1], num_heads_window=[3, 3], num_heads_stripe=[3, 3],
                      drop_path_rate=0.0)
    torch.manual_seed(rank)                   # the replicas start DIFFERENT: the constructor's broadcast has to make them equal
    m = GRL(**cfg)
    if rank == 0:
        m.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0), strict=True)
    m = m.to(dev).train()
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=4, seed=21)
    per = lq.shape[0] // world
    x, y = lq[rank * per : (rank + 1) * per].to(dev), gt[rank * per : (rank + 1) * per].to(dev)
    opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
    step = GraphedTrainStep(m, opt, lambda o, t: (o - t).abs().mean(), x, y, warmup=1, wire_bf16=wire_bf16)    # default group
    losses = [float(step(x, y).detach()) for _ in range(3)]
    step.finish()
    grads_are_views = all(p.grad is not None and p.grad.data_ptr() >= step._flat.data_ptr() and
                          p.grad.data_ptr() < step._flat.data_ptr() + step._flat.numel() * 4 for p in m.parameters())
    ret[rank] = ({k: p.detach().cpu().clone() for k, p in m.named_parameters()}, losses, step.collectives,
                 opt.state[next(iter(m.parameters()))]["step"], grads_are_views)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire_bf16", [False, True])
def test_graphed_step_data_parallel_two_replicas(wire_bf16):
    """The captured training step under data parallelism (VERDICT r4 missing #2): graph A (forward, loss, backward, flat gradient
    buffer) -> ONE eager all-reduce -> graph B (FusedAdamW on the averaged gradients).  Two half-batch replicas stay bit-identical
    to each other and follow the single-process full-batch EAGER steps as closely as the replica test above allows."""
    import socket

    import torch.multiprocessing as mp

    from grl_image_restoration_amd import GRL, FusedAdamW, make_config

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_graphed_ddp_worker, args=(2, port, ret, wire_bf16), nprocs=2, join=True)
    (p0, l0, c0, n0, v0), (p1, l1, c1, n1, v1) = ret[0], ret[1]
    assert c0 == c1 == 1 + 3 and n0 == n1 == 1 + 3 and v0 and v1          # one collective per step (warm-up + 3 replays)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k                                # same averaged gradients -> same weights, bit for bit
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1, 1], num_heads_window=[3, 3], num_heads_stripe=[3, 3],
                      drop_path_rate=0.0)

    def single():
        m = GRL(**cfg)
        m.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0), strict=True)
        m = m.cuda().train()
        opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
        lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=4, seed=21)
        lq, gt = lq.cuda(), gt.cuda()
        out = []
        for _ in range(1 + 3):
            opt.zero_grad(set_to_none=True)
            loss = (m(lq) - gt).abs().mean()
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        return {k: p.detach().cpu() for k, p in m.named_parameters()}, out[1:]

    (pa, la), (pb, lb) = single(), single()                                # two eager runs: the yardstick (atomics, Adam's first steps)
    n = sum(v.numel() for v in pa.values())
    d_ee = sum(float((pa[k] - pb[k]).abs().sum()) for k in pa) / n
    d_eg = sum(float((pa[k] - p0[k]).abs().sum()) for k in pa) / n
    lg = [0.5 * (a + b) for a, b in zip(l0, l1)]                           # mean of the half-batch losses = the full-batch loss
    l_ee = max(abs(a - b) for a, b in zip(la, lb))
    l_eg = max(abs(a - b) for a, b in zip(la, lg))
    print(f"wire_bf16={wire_bf16}: losses eager {la} | replicas {lg}; mean |dp| eager-eager {d_ee:.3e}, eager-replicas {d_eg:.3e}; "
          f"max |dloss| {l_ee:.3e} / {l_eg:.3e}")
    slack = 4.0 if not wire_bf16 else 40.0                                 # bf16 on the wire: 3 significant digits per gradient
    assert d_eg <= slack * d_ee + 2e-6 and l_eg <= slack * l_ee + 1e-4
    assert lg[-1] < lg[0]
