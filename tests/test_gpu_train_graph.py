"""GPU tests of the captured training step (train_graph.GraphedTrainStep: forward + L1 + backward + FusedAdamW as one replayed HIP
graph) -- needs the MI355X.  A module of its own (round 6): a failure here must not hide the gradient-parity families under the
driver's `pytest -x`, nor the other way round.  Collected after test_gpu_train.py (operator-level gradients) and before the
replica / whole-network parity modules.
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


def test_split_attention_backward_replays_from_a_graph(monkeypatch):
    """Regression test of round 6's root cause for the order-dependent NaN of the captured step: grl_attention_bwd
    zeroes the destinations of its split launches; done with hipMemsetAsync, a CAPTURED launch left every fourth float of dQ / dK /
    dV un-zeroed from the second replay on (tools/probes/graph_memset_probe3.py shows the runtime doing exactly that to a bare
    memset node), so the atomics accumulated onto the previous replay's values.  The fill is a kernel now.  Each replay runs on a
    different dO and must equal the eager launch on that dO (up to the order of the fp32 atomics)."""
    import math

    from grl_image_restoration_amd import autograd as AG, ops, tables

    monkeypatch.setenv("GRL_ATTN_BWD_SPLITS", "2")
    g = torch.Generator().manual_seed(7)
    B, nh, d, H, W = 1, 3, 30, 64, 64
    tok, anc = (H, W, 64, 64, 32, 32), (H // 2, W // 2, 32, 32, 16, 16)
    Mq, Mk = B * H * W, B * (H // 2) * (W // 2)
    P = lambda t: F.pad(t, (0, 32 - t.shape[-1])).permute(1, 0, 2).contiguous().cuda()
    sc = (torch.rand(nh, generator=g) * 12 + 4) * LOG2E
    q = P(F.normalize(torch.randn(Mq, nh, d, generator=g), dim=-1) * sc.view(1, nh, 1))
    k = P(F.normalize(torch.randn(Mk, nh, d, generator=g), dim=-1))
    v = P(torch.randn(Mk, nh, d, generator=g))
    table = tables.kernel_table(torch.rand((64 + 32 - 1) ** 2, nh, generator=g) * 16).cuda()
    floor = tables.lazy_floor(sc / LOG2E).cuda()
    o, lse, q16, k16, v16 = AG.attention_op(q, k, v, table, floor, list(tok), list(anc), B, nh, d, True)
    TG = ops.TokenGrid
    d_os = [(torch.randn(nh, Mq, 32, generator=g) * 1e-6 * (r + 1)).cuda() for r in range(4)]
    for t in d_os:
        t[..., d:] = 0

    def run(d_o):
        return ops.attention_bwd(TG(q16, 0, *tok), TG(k16, 0, *anc), TG(v16, 0, *anc), TG(o, 0, *tok), d_o, lse, B=B, nh=nh, table=table,
                                 masked=True, ones_col=d, head_dim=d, g_scale=2.0 ** 20)

    eager = [[t.clone() for t in run(d_o)] for d_o in d_os]
    static = d_os[0].clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = run(static)
    for r, d_o in enumerate(d_os):
        static.copy_(d_o)
        graph.replay()
        torch.cuda.synchronize()
        for name, got, want in zip(("dq", "dk", "dv", "dtable"), outs, eager[r]):
            assert bool(torch.isfinite(got).all()), (r, name)
            e = _rel(got, want)
            assert e < 1e-5, f"replay {r}: {name} differs from the eager launch by {e:.2e}"
