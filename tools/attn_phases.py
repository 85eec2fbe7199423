#!/usr/bin/env python3
"""Per-workgroup phase timing of the fast attention kernel (s_memtime stamps): prologue, wait for
K/V loads, LDS commit, MFMA loop, epilogue.  Debug tool; not part of the product path."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_image_restoration_amd import GRL, _lib as L, baseline_config, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = sys.argv[2] if len(sys.argv) > 2 else "window"
cfg = baseline_config(3); cfg.update(depths=[4], num_heads_window=[3], num_heads_stripe=[3])
m = GRL(**cfg).eval().cuda()
H = W = 256; C, CP, nh = 180, 192, 3; M = B * H * W
plan = m._plan((H, W), torch.device("cuda")); pk, geo = plan["stages"][0]["blocks"][2], plan["sched"][0][2]
r = torch.randn(M, CP, device="cuda"); r[:, C:] = 0
qkv = ops.linear(r, pk["qkv_w"], pk["qkv_b"], epi=L.EPI_GROUPNORM, gscale=pk["qkv_gs"], planes=True)
anc = ops.linear(r, pk["anc_w"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], pool=(geo.df, H, W), planes=True)
att = torch.zeros(M, 192, dtype=torch.float16, device="cuda")
Ha, Wa = H // geo.df, W // geo.df
y = torch.zeros(nh, B * Ha * Wa, 32, dtype=torch.bfloat16, device="cuda")
TG = ops.TokenGrid
ws, sh = geo.window, geo.window_shift
stp, ss, ast, ass = geo.stripe, geo.stripe_shift_size, geo.anchor_stripe, geo.anchor_shift_size
def run():
    if which == "window":
        ops.attention(TG(qkv, 0, H, W, ws[0], ws[1], sh, sh), TG(qkv, nh, H, W, ws[0], ws[1], sh, sh), TG(qkv, 2 * nh, H, W, ws[0], ws[1], sh, sh),
                      TG(att, 0, H, W, ws[0], ws[1], sh, sh), B=B, nh=nh, table=pk["tab_w"], masked=True, fixed_max=True, ones_col=30, head_dim=30)
    else:
        ops.attention(TG(anc, 0, Ha, Wa, ast[0], ast[1], ass[0], ass[1]), TG(qkv, 4 * nh, H, W, stp[0], stp[1], ss[0], ss[1]),
                      TG(qkv, 5 * nh, H, W, stp[0], stp[1], ss[0], ss[1]), TG(y, 0, Ha, Wa, ast[0], ast[1], ass[0], ass[1]), B=B, nh=nh,
                      table=pk["tab_a2w"], masked=True, fixed_max=True, ones_col=30, head_dim=30)
run(); torch.cuda.synchronize()
buf = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.grl_debug_attention_timestamps.argtypes = [ctypes.c_void_p]
lib.grl_debug_attention_timestamps(ctypes.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
lib.grl_debug_attention_timestamps(ctypes.c_void_p(0))
t = buf.view(-1, 8).cpu()
t = t[t[:, 6] > 0]
n = t.shape[0]
start = t[:, 0] - t[:, 0].min()
print(f"{which} B={B}: {n} workgroups; s_memtime ticks (100 MHz?) span {int((t[:,0]+t[:,6]).max()-t[:,0].min())}")
names = ["prologue", "wait_loads", "commit", "mfma_loop", "epilogue", "total"]
for i, nm in enumerate(names):
    c = t[:, i + 1].float()
    print(f"  {nm:10s} mean {c.mean():10.1f}  min {c.min():8.0f}  max {c.max():8.0f}")
print("  XCC ids:", torch.bincount(t[:, 7] & 15).tolist())
# concurrency: how many WGs overlap the midpoint of the launch
mid = (start.max() + t[:, 6].max()) // 2
print("  resident at midpoint:", int(((start <= mid) & (start + t[:, 6] >= mid)).sum()))
