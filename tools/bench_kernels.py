#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the bench workload's shapes (GRL-Base x4, 256x256 LQ tiles, ckpt
geometry).  Each C-ABI kernel of one transformer block is run in isolation `--iters` times and
timed with HIP events on the launching stream; prints time, algorithmic FLOPs and bytes.

    python tools/bench_kernels.py [--tiles B] [--iters N] [--only name[,name]]
Used under rocprofv3 --pmc to collect counters per kernel.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grl_image_restoration_amd import GRL, _lib as L, baseline_config, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--block", type=int, default=2, help="block index in stage 0 (2: window+stripe shift)")
    ap.add_argument("--logit-scale", type=float, default=0.0, help="set every logit scale to this value (100 = the clamp)")
    a = ap.parse_args()
    cfg = baseline_config(3)
    cfg.update(depths=[4], num_heads_window=[3], num_heads_stripe=[3])
    torch.manual_seed(0)
    m = GRL(**cfg).eval().cuda()
    if a.logit_scale > 0:
        import math
        with torch.no_grad():
            for n, p_ in m.named_parameters():
                if n.endswith("logit_scale"):
                    p_.fill_(math.log(a.logit_scale))
    B, H, W, C, CP = a.tiles, 256, 256, 180, 192
    M = B * H * W
    plan = m._plan((H, W), torch.device("cuda"))
    pk, geo = plan["stages"][0]["blocks"][a.block], plan["sched"][0][a.block]
    ceil = lambda k: None if os.environ.get("GRL_ATTN_NOCEIL") else pk.get(k)   # A/B: attention with / without the overflow test
    st = plan["stages"][0]
    r = torch.randn(M, CP, device="cuda")
    r[:, C:] = 0
    nh, df = 3, geo.df
    Ha, Wa = H // df, W // df
    TG = ops.TokenGrid
    qkv = ops.linear(r, pk["qkv_w"], pk["qkv_b"], epi=L.EPI_GROUPNORM, gscale=pk["qkv_gs"], planes=True)
    anc = ops.linear(r, pk["anc_w"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], pool=(df, H, W), planes=True)
    att = torch.zeros(M, 2 * nh * 32, dtype=torch.float16, device="cuda")
    y = torch.zeros(nh, B * Ha * Wa, 32, dtype=torch.float16, device="cuda")
    ws, sh = geo.window, geo.window_shift
    stp, ss = geo.stripe, geo.stripe_shift_size
    ast, ass = geo.anchor_stripe, geo.anchor_shift_size
    s0 = 3 * nh
    g_q = TG(qkv, s0, H, W, stp[0], stp[1], ss[0], ss[1])
    g_k = TG(qkv, s0 + nh, H, W, stp[0], stp[1], ss[0], ss[1])
    g_v = TG(qkv, s0 + 2 * nh, H, W, stp[0], stp[1], ss[0], ss[1])
    g_a = TG(anc, 0, Ha, Wa, ast[0], ast[1], ass[0], ass[1])
    g_y = TG(y, 0, Ha, Wa, ast[0], ast[1], ass[0], ass[1])
    mid = torch.zeros(M, pk["cab_mid"], dtype=torch.float16, device="cuda")
    h = torch.zeros(M, 384, dtype=torch.float16, device="cuda")
    cab = torch.zeros(M, CP, dtype=torch.float16, device="cuda")
    pool = torch.zeros(L.lib().grl_conv3x3_num_workgroups(B, H, W), CP, device="cuda")
    gate = torch.ones(B, CP, device="cuda")
    blk0 = m.layers[0].blocks[a.block]
    rblob = ops.pack_tail_regs(pk["proj_w"].float(), blk0.mlp.fc1.weight, blk0.mlp.fc1.bias, blk0.mlp.fc2.weight)   # register-resident tail (opt-in kernel)
    L_, Nw, N2 = H * W, ws[0] * ws[1], ast[0] * ast[1]
    fl_att = 2 * L_ * Nw * C * B
    fl_s = 2 * L_ * N2 * C * B

    kernels = {
        "qkv_anchor": (lambda: ops.qkv_anchor(r, pk["qa_blob"], pk["qa_slots"][0], pk["qa_slots"][1], B, H, W), (6 * L_ * C * C + L_ * C * C // 2) * B,
                       M * (CP * 4 + 18 * 32 * 2 + 3 * 32 * 2 // 4)),
        "qkv_split": (lambda: ops.qkv_anchor(r, pk["qa_blob"], pk["qa_slots"][0], pk["qa_slots"][1], B, H, W, lo_blob=pk["qa_lo"]),
                      (6 * L_ * C * C + L_ * C * C // 2) * B, M * (CP * 4 + 18 * 32 * 2 + 3 * 32 * 2 // 4)),   # needs --logit-scale > 50
        "qkv_stream": (lambda: ops.qkv(r, pk["qkv_blob"], pk["qkv_slots"], out=qkv), 6 * L_ * C * C * B, M * (CP * 4 + 18 * 32 * 2)),
        "qkv": (lambda: ops.linear(r, pk["qkv_w"], pk["qkv_b"], epi=L.EPI_GROUPNORM, gscale=pk["qkv_gs"], out=qkv, planes=True),
                6 * L_ * C * C * B, M * (CP * 4 + 576 * 2)),
        "anchor": (lambda: ops.linear(r, pk["anc_w"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], pool=(df, H, W), out=anc, planes=True),
                   L_ * C * C * B // (df * df), M * CP * 4),
        "attn_window": (lambda: ops.attention(TG(qkv, 0, H, W, ws[0], ws[1], sh, sh), TG(qkv, nh, H, W, ws[0], ws[1], sh, sh),
                                              TG(qkv, 2 * nh, H, W, ws[0], ws[1], sh, sh), TG(att, 0, H, W, ws[0], ws[1], sh, sh),
                                              B=B, nh=nh, table=pk["tab_w"], masked=sh > 0, ones_col=30, head_dim=30, k_one31=True, lazy_floor=pk["floor_w"], lazy_ceil=ceil("ceil_w")),
                        fl_att, M * 4 * 96 * 2),
        "attn_a2w": (lambda: ops.attention(g_a, g_k, g_v, g_y, B=B, nh=nh, table=pk["tab_a2w"], masked=geo.stripe_shift,
                                           ones_col=30, head_dim=30, k_one31=True, lazy_floor=pk["floor_a2w"], lazy_ceil=ceil("ceil_a2w")), fl_s, M * 2 * 96 * 2),
        "attn_w2a": (lambda: ops.attention(g_q, g_a, g_y, TG(att, nh, H, W, stp[0], stp[1], ss[0], ss[1]), B=B, nh=nh,
                                           table=pk["tab_w2a"], masked=geo.stripe_shift, ones_col=30,
                                           head_dim=30, k_one31=True, lazy_floor=pk["floor_w2a"], lazy_ceil=ceil("ceil_w2a")), fl_s, M * 2 * 96 * 2),
        "cab_conv1": (lambda: ops.conv3x3(r, pk["cab0_w"], pk["cab0_b"], B, H, W, act=1, out=mid), 2 * 9 * L_ * C * 45 * B, M * (CP * 4 + 96)),
        "cab_conv2": (lambda: ops.conv3x3(mid, pk["cab2_w"], pk["cab2_b"], B, H, W, want_pool=True, out=cab), 2 * 9 * L_ * C * 45 * B, M * (128 + CP * 2)),
        "cab_conv2_regs": (lambda: ops.cab_conv2(mid, pk["cab2_blob"], pk["cab2_bias"], B, H, W), 2 * 9 * L_ * C * 45 * B, M * (128 + CP * 2)),
        "se": (lambda: ops.se_scale(pool, B, CP, C, H * W, pk["se1_w"], pk["se1_b"], pk["se3_w"], pk["se3_b"]), 0, pool.numel() * 4),
        "proj_ln": (lambda: ops.linear(att, pk["proj_w"], pk["proj_b"], epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=pk["n1_g"],
                                       ln_b=pk["n1_b"], n_real=C, resid=r, add2=cab, add2_scale=gate, rows_per_image=H * W),
                    2 * L_ * C * C * B, M * (192 * 2 + CP * 4 + CP * 2 + CP * 4)),
        "block_tail": (lambda: ops.block_tail(att, r, cab, gate, H * W, pk["proj_blob"], pk["proj_b"], pk["n1_g"], pk["n1_b"], pk["mlp_blob"],
                                              pk["fc2_b"], pk["n2_g"], pk["n2_b"], Hpad=pk["mlp_hp"], n_real=C),
                       10 * L_ * C * C * B, M * (192 * 2 + CP * 4 + CP * 2 + CP * 4)),
        "block_tail_regs": (lambda: ops.block_tail(att, r, cab, gate, H * W, pk["proj_blob"], pk["proj_b"], pk["n1_g"], pk["n1_b"], pk["mlp_blob"],
                                                   pk["fc2_b"], pk["n2_g"], pk["n2_b"], Hpad=pk["mlp_hp"], n_real=C, rblob=rblob),
                            10 * L_ * C * C * B, M * (192 * 2 + CP * 4 + CP * 2 + CP * 4)),
        "fc1_gelu": (lambda: ops.linear(r, pk["fc1_w"], pk["fc1_b"], epi=L.EPI_GELU, out=h), 4 * L_ * C * C * B, M * (CP * 4 + 384 * 2)),
        "mlp_fused": (lambda: ops.mlp(r, pk["mlp_blob"], pk["fc2_b"], pk["n2_g"], pk["n2_b"], Hpad=pk["mlp_hp"], n_real=C),
                      8 * L_ * C * C * B, M * (CP * 4 * 2)),
        "fc2_ln": (lambda: ops.linear(h, pk["fc2_w"], pk["fc2_b"], epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=pk["n2_g"],
                                      ln_b=pk["n2_b"], n_real=C, resid=r), 4 * L_ * C * C * B, M * (384 * 2 + CP * 8)),
        "stage_conv": (lambda: ops.conv3x3(r, st["conv_w"], st["conv_b"], B, H, W, resid=r), 18 * L_ * C * C * B, M * CP * 12),
        "layernorm": (lambda: ops.layernorm(r, plan["ns_g"], plan["ns_b"], C), 0, M * CP * 8),
    }
    only = [s for s in a.only.split(",") if s]
    print(f"# tiles={B} iters={a.iters} block={a.block} (window_shift={sh}, stripe={stp}, stripe_shift={geo.stripe_shift})")
    print(f"{'kernel':14s} {'us':>9s} {'GFLOP':>9s} {'TFLOP/s':>9s} {'MB':>9s} {'GB/s':>9s}")
    for name, (fn, flops, byts) in kernels.items():
        if only and name not in only:
            continue
        if name == "qkv_split" and "qa_lo" not in pk:
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"{name:14s} {us:9.1f} {flops/1e9:9.2f} {flops/us/1e6:9.1f} {byts/1e6:9.1f} {byts/us/1e3:9.1f}")


if __name__ == "__main__":
    main()
