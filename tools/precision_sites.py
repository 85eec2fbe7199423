"""Per-site CPU emulation of the 16-bit rounding points of the HIP path: which layer's operand rounding dominates the
output error of a fixture.  Usage: python tools/precision_sites.py [fixture] .  Test infrastructure (imports oracle/)."""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import grl_oracle as O
from tests.util import load_golden, product_shapes
torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_sr2_ckpt_64"
meta, z = load_golden(name)
cfg = meta["cfg"]; sd = O.seeded_state_dict(product_shapes(cfg), meta.get("weight_seed", 0), **meta.get("sd_kwargs", {}))
x = z["input"]; ref = z["output"]
names = {id(v): k for k, v in sd.items()}
def site_of(w):
    k = names.get(id(w), "?")
    for s in ("conv_first", "conv_after_body", "conv_before_upsample", "upsample", "conv_last", "qkv", "anchor", "proj", "fc1", "fc2", "cab.0", "cab.2", "cpb_mlp", "attention"):
        if s in k: return s
    if k.startswith("layers") and k.endswith("conv.weight"): return "stage_conv"
    return "?"
FL, FC = F.linear, F.conv2d
MODE = {}
def rnd(t, key):
    dt = MODE.get(key, MODE.get("*"))
    return t if dt is None else t.to(dt).float()
def lin(inp, w, b=None):
    s = site_of(w)
    if s in ("cpb_mlp", "?"): return FL(inp, w, b)
    return FL(rnd(inp, s + ".in"), rnd(w, s + ".w"), b)
def conv(inp, w, b=None, **kw):
    s = site_of(w)
    if w.shape[-1] == 1 or s == "?": return FC(inp, w, b, **kw)
    return FC(rnd(inp, s + ".in"), rnd(w, s + ".w"), b, **kw)
F.linear, F.conv2d = lin, conv
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
def run(**mode):
    MODE.clear(); MODE.update(mode)
    with torch.no_grad():
        y = O.grl_forward(x, cfg, sd)
    return "max %.2e rms %.2e" % ((y - ref).abs().max().item(), (y - ref).pow(2).mean().sqrt().item())
hf = torch.float16
print(name)
if len(sys.argv) > 2 and sys.argv[2] == "combo":
    # python tools/precision_sites.py <fixture> combo "site.in,site.w;..."  -- everything fp16 EXCEPT the listed operands, one run per
    # semicolon-separated list: which half of a split (activations `.in`, weights `.w`) a site actually needs
    for combo in sys.argv[3].split(";"):
        mode = {"*": hf}
        mode.update({k: torch.float32 for k in combo.split(",") if k})
        print(("fp16 except " + combo)[:100].ljust(102), run(**mode), flush=True)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "only":
    # python tools/precision_sites.py <fixture> only "site.in;site.w,other.w;..."  -- everything exact EXCEPT the listed operands
    for combo in sys.argv[3].split(";"):
        print(("only " + combo + " fp16")[:100].ljust(102), run(**{k: hf for k in combo.split(",") if k}), flush=True)
    sys.exit(0)
print("all fp16".ljust(40), run(**{"*": hf}))
if len(sys.argv) > 2 and sys.argv[2] == "attn":
    for s in ("qk", "p", "v", "attn_out"):
        print(("only " + s + " fp16").ljust(40), run(**{s: hf}))
    sys.exit(0)
sites = ["conv_first", "stage_conv", "conv_after_body", "conv_before_upsample", "upsample", "conv_last", "qkv", "anchor", "proj", "fc1", "fc2", "cab.0", "cab.2"]
for s in sites:
    print(("only " + s + " fp16 (in+w)").ljust(40), run(**{s + ".in": hf, s + ".w": hf}))
for s in ("qk", "p", "v", "attn_out"):
    print(("only " + s + " fp16").ljust(40), run(**{s: hf}))
