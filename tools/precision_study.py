"""CPU emulation of the rounding points of the HIP path (which bf16/fp16 roundings dominate the output error).
Usage: python tools/precision_study.py [golden fixture name].  Test infrastructure (imports oracle/)."""
import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import grl_oracle as O
from tests.util import load_golden, product_shapes
torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else "base_sr4_ckpt_64"
meta, z = load_golden(name)
cfg = meta["cfg"]; sd = O.seeded_state_dict(product_shapes(cfg), 0)
x = z["input"]; ref = z["output"]
FL, FC = F.linear, F.conv2d
MODE = {}
def rnd(t, key):
    dt = MODE.get(key)
    return t if dt is None else t.to(dt).float()
def lin(inp, w, b=None):
    if w.shape[-1] in (2, 512) and w.shape[0] in (512, 1, 2, 3, 4, 6):  # cpb mlp: exact
        return FL(inp, w, b)
    return FL(rnd(inp, "lin_in"), rnd(w, "lin_w"), b)
def conv(inp, w, b=None, **kw):
    if w.shape[-1] == 1: return FC(inp, w, b, **kw)
    return FC(rnd(inp, "conv_in"), rnd(w, "conv_w"), b, **kw)
F.linear, F.conv2d = lin, conv
orig_attn = O.cosine_attention
def attn(q, k, v, p, prefix, table, index, mask):
    B_, nh, Nq, _ = q.shape; Nk = k.shape[2]
    qn = rnd(F.normalize(q, dim=-1) * O.logit_scale(p, prefix).unsqueeze(0), "qk")
    kn = rnd(F.normalize(k, dim=-1), "qk")
    a = qn @ kn.transpose(-2, -1)
    bt = O.bias_table(p, prefix, table)
    a = a + bt[index.reshape(-1)].view(Nq, Nk, nh).permute(2, 0, 1).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        a = (a.view(B_ // nW, nW, nh, Nq, Nk) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nh, Nq, Nk)
    a = a - a.max(-1, keepdim=True).values
    pnum = rnd(torch.exp(a), "p")
    o = (pnum @ rnd(v, "v")) / pnum.sum(-1, keepdim=True)
    return rnd(o, "attn_out")
O.cosine_attention = attn
orig_gelu = F.gelu
def run(**mode):
    MODE.clear(); MODE.update(mode)
    with torch.no_grad():
        y = O.grl_forward(x, cfg, sd)
    return (y - ref).abs().max().item(), (y - ref).pow(2).mean().sqrt().item()
bf, hf = torch.bfloat16, torch.float16
print(name)
print("exact           ", run())
allk = ["lin_in", "lin_w", "conv_in", "conv_w", "qk", "p", "v", "attn_out"]
print("all bf16        ", run(**{k: bf for k in allk}))
print("all fp16        ", run(**{k: hf for k in allk}))
for k in allk:
    print(f"only {k:9s} bf16", run(**{k: bf}))
print("bf16 but qk fp16", run(**{**{k: bf for k in allk}, "qk": hf}))
print("bf16 but weights fp16", run(**{**{k: bf for k in allk}, "lin_w": hf, "conv_w": hf}))
print("bf16 but acts fp16 (lin_in conv_in attn_out)", run(**{**{k: bf for k in allk}, "lin_in": hf, "conv_in": hf, "attn_out": hf}))
