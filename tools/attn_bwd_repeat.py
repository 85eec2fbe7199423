"""Run-to-run behaviour of grl_attention_bwd (diagnostic, GPU box): the same launch repeated under GRL_ATTN_BWD_SPLITS = 1 / auto /
2 / 4, outputs compared with the unsplit run (relative, norm-wise) and with each other (bitwise)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grl_image_restoration_amd import autograd as AG, ops, tables  # noqa: E402

LOG2E = 1.4426950408889634


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def case(mode, H, W, win, shift, df, nh, d, B, g_scale, seed=3, do_mag=1e-6):
    g = torch.Generator().manual_seed(seed)
    awin, ashift = (win[0] // df, win[1] // df), (shift[0] // df, shift[1] // df)
    Ha, Wa = H // df, W // df
    tok, anc = (H, W, win[0], win[1], shift[0], shift[1]), (Ha, Wa, awin[0], awin[1], ashift[0], ashift[1])
    qg, kg = {"w": (tok, tok), "a2w": (anc, tok), "w2a": (tok, anc)}[mode]
    Mq, Mk = B * qg[0] * qg[1], B * kg[0] * kg[1]
    P = lambda t: F.pad(t, (0, 32 - t.shape[-1])).permute(1, 0, 2).contiguous().cuda()
    sc = (torch.rand(nh, generator=g) * 12 + 4) * LOG2E
    q = P(F.normalize(torch.randn(Mq, nh, d, generator=g), dim=-1) * sc.view(1, nh, 1))
    k = P(F.normalize(torch.randn(Mk, nh, d, generator=g), dim=-1))
    v = P(torch.randn(Mk, nh, d, generator=g))
    rows = (qg[2] + kg[2] - 1) * (qg[3] + kg[3] - 1)
    table = tables.kernel_table(torch.rand(rows, nh, generator=g) * 16).cuda()
    floor = tables.lazy_floor(sc / LOG2E).cuda()
    d_o = (torch.randn(nh, Mq, 32, generator=g) * do_mag).cuda()
    d_o[..., d:] = 0
    masked = shift[0] > 0 or shift[1] > 0
    o, lse, q16, k16, v16 = AG.attention_op(q, k, v, table, floor, list(qg), list(kg), B, nh, d, masked)
    TG = ops.TokenGrid

    def run():
        return ops.attention_bwd(TG(q16, 0, *qg), TG(k16, 0, *kg), TG(v16, 0, *kg), TG(o, 0, *qg), d_o, lse, B=B, nh=nh, table=table,
                                 masked=masked, ones_col=d, head_dim=d, g_scale=g_scale)

    return run


def main():
    names = ("dq", "dk", "dv", "dtab")
    cases = [("w2a 64x64 df2 B1", ("w2a", 64, 64, (64, 64), (32, 32), 2, 3, 30, 1)),
             ("a2w 64x64 df2 B1", ("a2w", 64, 64, (64, 64), (32, 32), 2, 3, 30, 1)),
             ("w2a 64x64 df2 B1 noshift", ("w2a", 64, 64, (64, 64), (0, 0), 2, 3, 30, 1)),
             ("win32 B1", ("w", 64, 64, (32, 32), (16, 16), 1, 3, 30, 1))]
    for title, c in cases:
        for gs in (2.0 ** 20, 2.0 ** 24):
            run = case(*c, g_scale=gs)
            os.environ["GRL_ATTN_BWD_SPLITS"] = "1"
            base = [t.clone() for t in run()]
            again = run()
            print(f"{title} g_scale 2^{int(math.log2(gs))}: unsplit repeat bitwise {[bool(torch.equal(a, b)) for a, b in zip(base, again)]}")
            for sp in ("", "2", "4"):
                if sp:
                    os.environ["GRL_ATTN_BWD_SPLITS"] = sp
                else:
                    os.environ.pop("GRL_ATTN_BWD_SPLITS", None)
                outs = [[t.clone() for t in run()] for _ in range(4)]
                errs = [[rel(o[i], base[i]) for i in range(4)] for o in outs]
                same = [bool(torch.equal(outs[0][i], outs[1][i])) for i in range(4)]
                print(f"   splits {sp or 'auto':>4}: vs unsplit " + "  ".join(f"{n} {max(e[i] for e in errs):.2e}" for i, n in enumerate(names)) + f"  | bitwise run0==run1 {same}")
    os.environ.pop("GRL_ATTN_BWD_SPLITS", None)


if __name__ == "__main__":
    main()
