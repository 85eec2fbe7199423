"""Reproducer / locator for the order-dependent non-finite weights of the captured training step (GPU box).

Dirties the driver's free memory with NaN first (a large NaN-filled tensor released with empty_cache(): what exited replica
processes leave behind), then runs the body of test_graphed_train_step_follows_lr_schedule_and_resume several times and reports,
after every replay, whether loss / gradients / weights are finite and WHICH gradients are not (in forward order)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grl_image_restoration_amd import GRL, FusedAdamW, GraphedTrainStep, make_config  # noqa: E402
from oracle import grl_oracle as O  # noqa: E402


def dirty(gb):
    n = int(gb * (1 << 30)) // 4
    t = torch.full((n,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()


def report(tag, m, step):
    torch.cuda.synchronize()
    bad_g = [k for k, p in m.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    bad_w = [k for k, p in m.named_parameters() if not bool(torch.isfinite(p).all())]
    loss = float(step.loss.detach())
    ok = not bad_g and not bad_w and loss == loss
    print(f"   {tag}: loss {loss:.6f}  non-finite grads {len(bad_g)}  non-finite weights {len(bad_w)}" + ("" if ok else f"  first grads {bad_g[:6]} last {bad_g[-3:]}"), flush=True)
    return ok


def once(i, eval_between=True):
    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1], num_heads_window=[3], num_heads_stripe=[3], drop_path_rate=0.0)
    torch.manual_seed(0)
    m = GRL(**cfg).cuda().train()
    opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=0.0)
    lq, gt = O.synthetic_pair("sr", (64, 64), 4, batch=2, seed=12)
    lq, gt = lq.cuda(), gt.cuda()
    step = GraphedTrainStep(m, opt, lambda y, t: (y - t).abs().mean(), lq, gt, warmup=1)
    ok = True
    step(lq, gt)
    ok &= report(f"run {i} replay 1", m, step)
    if eval_between:
        with torch.no_grad():
            m.eval()(lq)
        m.train()
    opt.param_groups[0]["lr"] = 0.0
    step(lq, gt)
    ok &= report(f"run {i} replay 2 (lr 0)", m, step)
    opt.param_groups[0]["lr"] = 2e-4
    step(lq, gt)
    ok &= report(f"run {i} replay 3", m, step)
    for k in range(3):
        step(lq, gt)
    ok &= report(f"run {i} replay 6", m, step)
    step.finish()
    return ok


if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    gb = float(os.environ.get("DIRTY_GB", "40"))
    print(f"GRL_POISON={os.environ.get('GRL_POISON', '0')} dirty {gb} GB, eval between replays: {os.environ.get('NO_EVAL', '0') != '1'}")
    bad = 0
    for i in range(runs):
        if gb > 0:
            dirty(gb)
        bad += not once(i, os.environ.get("NO_EVAL", "0") != "1")
    print(f"{bad} of {runs} runs produced non-finite values")
