"""CPU tests of the host-side logic: closed-form geometry vs the oracle, the boundary contract
(constructor kwargs, state_dict names, loud failure on CPU), and that libgrl_hip.so loads and
exports every symbol include/grl_hip.h declares (no kernels are launched here)."""
import ctypes
import os
import re

import pytest
import torch

from grl_image_restoration_amd import GRL, _lib, geometry, make_config, tables
from oracle import grl_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("win,df", [((8, 8), 1), ((32, 32), 1), ((64, 64), 2), ((8, 64), 4), ((64, 8), 4), ((48, 96), 4), ((12, 12), 1)])
def test_closed_form_index_and_tables(win, df):
    a = (win[0] // df, win[1] // df)
    for w2a in (True, False):
        idx = O.rel_index(win, df, w2a)
        qw, kw = (win, a) if w2a else (a, win)
        assert idx.shape == (qw[0] * qw[1], kw[0] * kw[1])
        for nq in (0, 1, qw[1], qw[0] * qw[1] - 1):
            for nk in (0, kw[1] - 1, kw[0] * kw[1] - 1):
                got = geometry.rel_index(nq // qw[1], nq % qw[1], nk // kw[1], nk % kw[1], qw, kw)
                assert got == int(idx[nq, nk])
        assert geometry.table_rows(qw, kw) == O.coords_table(win, df).numel() // 2
    assert torch.equal(tables.coords_table(win, df), O.coords_table(win, df).reshape(-1, 2))


@pytest.mark.parametrize("res,win,shift", [((32, 32), (8, 8), (4, 4)), ((16, 64), (8, 32), (4, 16)), ((32, 32), (8, 32), (4, 0)),
                                           ((24, 24), (12, 12), (6, 6))])
def test_closed_form_regions(res, win, shift):
    lab = O._region_labels(res, win, shift)  # (nW, N)
    nwx = res[1] // win[1]
    for w in range(lab.shape[0]):
        wy, wx = divmod(w, nwx)
        for n in range(0, win[0] * win[1], 5):
            hy, hx = divmod(n, win[1])
            rid = 3 * geometry.region1d(wy * win[0] + hy, res[0], win[0], shift[0]) + geometry.region1d(
                wx * win[1] + hx, res[1], win[1], shift[1])
            # labels only matter up to equality: compare the induced partition against token 0 of the window
            rid0 = 3 * geometry.region1d(wy * win[0], res[0], win[0], shift[0]) + geometry.region1d(wx * win[1], res[1], win[1], shift[1])
            assert (rid == rid0) == bool(lab[w, n] == lab[w, 0])


def test_schedule_matches_oracle():
    for model, geom, size in [("base", "yaml", (64, 64)), ("base", "sr_ckpt_df2", (128, 64)), ("small", "dn_df4", (128, 256))]:
        cfg = make_config(model, geom, img_size=64)
        mine = geometry.block_schedule(cfg["depths"], cfg["num_heads_window"], cfg["num_heads_stripe"], cfg["window_size"],
                                       cfg["stripe_size"], cfg["stripe_groups"], cfg["stripe_shift"],
                                       cfg["anchor_window_down_factor"], size)
        theirs = O.block_schedule(cfg, size)
        for sm, so in zip(mine, theirs):
            for bm, bo in zip(sm, so):
                assert list(bm.window) == list(bo["window"]) and bm.window_shift == bo["window_shift"]
                assert list(bm.stripe) == list(bo["stripe"]) and bm.stripe_shift == bo["stripe_shift"]
                if bm.stripe_shift:
                    assert list(bm.stripe_shift_size) == list(bo["stripe_shift_size"])
        assert geometry.pad_multiple(cfg["window_size"], cfg["stripe_size"], cfg["stripe_groups"],
                                     cfg["anchor_window_down_factor"]) == O.pad_size(cfg)
    with pytest.raises(ValueError):
        cfg = make_config("base", "sr_ckpt_df2")
        geometry.block_schedule(cfg["depths"], cfg["num_heads_window"], cfg["num_heads_stripe"], 32, [64, 64], [None, None], True, 2, (96, 64))


def test_boundary_contract_on_cpu():
    cfg = make_config("tiny", "sr_ckpt_df4", upscale=2, img_size=64)
    # extra YAML keys are swallowed like the reference does (grl.py:255)
    m = GRL(**cfg, name="grl_tiny", double_window=False, stripe_square=False, separable_conv_act=False, use_buffer=True)
    keys = set(m.state_dict().keys())
    for k in ["conv_first.weight", "norm_start.bias", "layers.0.blocks.0.attn.qkv.body.weight",
              "layers.0.blocks.0.attn.anchor.body.0.reduction.bias",
              "layers.0.blocks.1.attn.window_attn.attn_transform.logit_scale",
              "layers.0.blocks.1.attn.stripe_attn.attn_transform2.cpb_mlp.2.weight", "layers.3.conv.bias",
              "layers.0.blocks.0.mlp.fc2.weight", "conv_after_body.weight", "upsample.up.0.weight"]:
        assert k in keys, k
    assert not any(k.startswith(("table_", "index_", "mask_")) for k in keys)
    # a reference-style state_dict that still carries the geometry buffers loads strictly
    sd = m.state_dict()
    sd["table_w"] = torch.zeros(1)
    sd["index_sh_a2w"] = torch.zeros(1)
    sd["mask_sv_w2a"] = torch.zeros(1)
    m.load_state_dict(sd, strict=True)
    ck = {"model." + k: v for k, v in m.state_dict().items()}
    ck["model.table_w"] = torch.zeros(1)
    assert "model.table_w" not in m.convert_checkpoint(ck)
    os.environ["GRL_NO_CPU_COMPOSITE"] = "1"
    try:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.eval()(torch.zeros(1, 3, 64, 64))
    finally:
        del os.environ["GRL_NO_CPU_COMPOSITE"]
    with pytest.raises(NotImplementedError):
        GRL(**{**cfg, "qkv_proj_type": "separable_conv"})
    base = GRL(**make_config("base", "sr_ckpt_df2", upscale=4))
    assert sum(p.numel() for p in base.parameters()) == 20201299  # paper: 20.20 M (figs/task4.png)


def test_library_exports_every_declared_symbol():
    assert os.path.isfile(_lib.LIB_PATH), "run __graft_entry__.build() first"
    header = open(os.path.join(ROOT, "include", "grl_hip.h")).read()
    declared = set(re.findall(r"\b(grl_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), (declared, _lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym
    lib = _lib.lib()
    assert lib.grl_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.grl_build_info()


def test_ctypes_layout_matches_the_c_header(tmp_path):
    """Compile include/grl_hip.h with gcc and compare sizeof/offsetof of every struct field with the
    ctypes mirrors in _lib.py (a silent mismatch would scramble kernel arguments)."""
    import subprocess

    structs = {"GrlLinearArgs": _lib.GrlLinearArgs, "GrlTokenGrid": _lib.GrlTokenGrid, "GrlAttnArgs": _lib.GrlAttnArgs,
               "GrlConvArgs": _lib.GrlConvArgs, "GrlMlpArgs": _lib.GrlMlpArgs, "GrlQkvArgs": _lib.GrlQkvArgs, "GrlQkvAnchorArgs": _lib.GrlQkvAnchorArgs, "GrlCabConv2Args": _lib.GrlCabConv2Args,
               "GrlTailArgs": _lib.GrlTailArgs,
               "GrlLnResArgs": _lib.GrlLnResArgs, "GrlGemmTnArgs": _lib.GrlGemmTnArgs, "GrlAttnBwdArgs": _lib.GrlAttnBwdArgs,
               "GrlAdamWArgs": _lib.GrlAdamWArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "grl_hip.h"', 'int main(void) {']
    for name, st in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in st._fields_:
            lines.append(f'printf("{name}.{f[0]} %zu\\n", offsetof({name}, {f[0]}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, st in structs.items():
        assert int(out[name]) == ctypes.sizeof(st), name
        for f in st._fields_:
            assert int(out[f"{name}.{f[0]}"]) == getattr(st, f[0]).offset, (name, f[0])


def test_softmax_offset_tables():
    """Kernel table = reversed bias rows in the log2 domain; lazy floor = an integer below every logit, exact in fp16."""
    scale = torch.tensor([10.0, 100.0])
    bias = torch.rand(50, 2) * 16
    t0 = tables.kernel_table(bias)
    assert t0.shape == (2, 52) and torch.equal(t0[:, 50:], torch.zeros(2, 2))
    assert torch.allclose(torch.flip(t0[:, :50], dims=(1,)), bias.t() * tables.LOG2E)
    fl = tables.lazy_floor(scale)
    assert torch.equal(fl, torch.round(fl)) and (fl <= -scale * tables.LOG2E).all() and fl.abs().max() < 2048


def test_lazy_ceil_bounds_every_logit():
    """GrlAttnArgs.lazy_ceil: the per-head bound the row-streaming attention kernel uses to stop testing for overflow must
    dominate every logit scale*log2e*cos + bias the kernel can form from fp16-rounded unit vectors (element-wise rounding of q
    and k, the scale folded into q before rounding, as the QKV epilogue does)."""
    g = torch.Generator().manual_seed(3)
    nh, d, n = 4, 30, 4096
    scale = torch.tensor([4.0, 10.0, 37.5, 100.0])
    bias = torch.rand(200, nh, generator=g) * 16
    tab = tables.kernel_table(bias)
    ceil = tables.lazy_ceil(scale, tab)
    assert ceil.shape == (nh,) and ceil.dtype == torch.float32
    k = torch.nn.functional.normalize(torch.randn(n, nh, d, generator=g), dim=-1)
    q = k.clone()                                  # the worst case: q parallel to k (cos = 1 before rounding)
    q16 = (q * (scale * tables.LOG2E).view(1, nh, 1)).half().float()
    k16 = k.half().float()
    logits = (q16 * k16).sum(-1) + tab.max(dim=1).values.view(1, nh)
    assert (logits.max(dim=0).values <= ceil).all()
    assert (ceil - logits.max(dim=0).values < 1.0 + 0.01 * scale * tables.LOG2E).all()   # and it is not sloppy either


def test_cab_conv2_blob_layout():
    """ops.pack_cab_conv2 writes the MFMA-fragment layout include/grl_hip.h documents for grl_cab_conv2_fwd: lane (g4, r) of
    (group G, k-step s) holds W[16 G + r][k = 32 s + 8 g4 ..], k = (ky * 3 + kx) * 48 + cin; zero beyond Cin / Cout / 432."""
    from grl_image_restoration_amd import ops

    g = torch.Generator().manual_seed(5)
    Cout, Cin = 180, 45
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    b = torch.randn(Cout, generator=g)
    blob, b192 = ops.pack_cab_conv2(w, b)
    assert blob.dtype == torch.uint8 and blob.numel() == 12 * 14 * 64 * 16
    assert torch.equal(b192[:Cout], b) and torch.equal(b192[Cout:], torch.zeros(192 - Cout))
    frag = blob.view(torch.float16).view(12, 14, 4, 16, 8)      # [G][s][g4][r][8]
    wk = frag.permute(0, 3, 1, 2, 4).reshape(192, 14 * 32).float()   # [16 G + r][32 s + 8 g4 + j]
    want = torch.zeros(192, 14 * 32)
    want[:Cout, : 9 * 48].view(Cout, 9, 48)[:, :, :Cin] = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).half().float()
    assert torch.equal(wk, want)


def test_ctypes_structs_refuse_unknown_fields():
    """ctypes would silently ignore a misspelt keyword and leave the C field zero."""
    with pytest.raises(TypeError):
        _lib.GrlConvArgs(Cin_pad=3)
    a = _lib.GrlConvArgs(CinP=64, CoutP=192, w_tap_stride=5)
    assert a.CinP == 64 and a.CoutP == 192 and a.w_tap_stride == 5


def test_mlp_blob_layout():
    """ops.pack_mlp writes the chunk images include/grl_hip.h documents (W1 rows | W2 columns | b1, padded rows,
    k-slot order 8g+e <-> 4g+e / 16+4g+e-4); checked by un-permuting the blob on the host."""
    from grl_image_restoration_amd import ops

    C_, Hd, CP, HP = 60, 120, 64, 128
    g = torch.Generator().manual_seed(5)
    w1, b1, w2 = torch.randn(Hd, C_, generator=g), torch.randn(Hd, generator=g), torch.randn(C_, Hd, generator=g)
    blob = ops.pack_mlp(w1, b1, w2, CP, HP)
    nch = HP // 32
    assert blob.dtype == torch.uint8 and blob.shape[0] == nch and blob.numel() == _lib.lib().grl_mlp_blob_bytes(CP, HP)
    assert blob.shape[1] % 1024 == 0
    slot_chan = [4 * (s // 8) + s % 8 if s % 8 < 4 else 16 + 4 * (s // 8) + s % 8 - 4 for s in range(32)]
    w1row, w2row = CP * 2 + 16, 80
    for c in range(nch):
        img = blob[c]
        W1 = img[: 32 * w1row].view(32, w1row)[:, : CP * 2].contiguous().view(torch.float16).view(32, CP // 32, 32)
        W2 = img[32 * w1row : 32 * w1row + CP * w2row].view(CP, w2row)[:, :64].contiguous().view(torch.float16)
        B1 = img[32 * w1row + CP * w2row :][:128].contiguous().view(torch.float32)
        ref1 = torch.zeros(32, CP)
        rows = min(32, max(0, Hd - 32 * c))
        ref1[:rows, :C_] = w1[32 * c : 32 * c + rows]
        ref2 = torch.zeros(CP, 32)
        ref2[:C_, :rows] = w2[:, 32 * c : 32 * c + rows]
        refb = torch.zeros(32)
        refb[:rows] = b1[32 * c : 32 * c + rows]
        for s_, ch in enumerate(slot_chan):
            assert torch.equal(W1[:, :, s_], ref1.view(32, CP // 32, 32)[:, :, ch].to(torch.float16))
            assert torch.equal(W2[:, s_], ref2[:, ch].to(torch.float16))
        assert torch.equal(B1, refb)


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under the package (nor the C sources) may import, open or name it,
    and nothing may read /root/reference at run time."""
    import re

    pkg = os.path.join(ROOT, "grl_image_restoration_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "oracle/" in txt or "/root/reference" in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    for f in ("bench.py", "__graft_entry__.py"):
        txt = open(os.path.join(ROOT, f)).read()
        assert "/root/reference" not in txt and "refshim" not in txt


def test_training_contractions_are_registered_torch_ops():
    """torch.ops.grl.{linear, conv3x3, attention}: registered custom ops (autograd + fake kernels), so shape inference works
    without a device and without touching the extension."""
    import grl_image_restoration_amd.autograd  # noqa: F401  (registers the ops)

    x = torch.empty(100, 180, device="meta")
    y = torch.ops.grl.linear(x, torch.empty(540, 180, device="meta"), torch.empty(540, device="meta"))
    assert y.shape == (100, 540) and y.dtype == torch.float32
    c = torch.ops.grl.conv3x3(torch.empty(2 * 8 * 8, 180, device="meta"), torch.empty(45, 180, 3, 3, device="meta"), torch.empty(45, device="meta"), 2, 8, 8)
    assert c.shape == (128, 45)
    q = torch.empty(3, 256, 32, device="meta")
    o, lse, q16, k16, v16 = torch.ops.grl.attention(q, q, q, torch.empty(3, 228, device="meta"), torch.empty(3, device="meta"), [16, 16, 8, 8, 4, 4],
                                                    [16, 16, 8, 8, 4, 4], 1, 3, 30, True)
    assert o.shape == (3, 256, 32) and lse.shape == (3, 256)
    assert all(t.shape == q.shape and t.dtype == torch.float16 for t in (q16, k16, v16))   # the operand planes the kernel ran on (kept for backward)


def test_training_operand_caches():
    """autograd.py's operand preparation (host logic, runs on the CPU): one-launch padding with the ones column that carries
    the bias gradient, and the per-parameter padded fp16 weights -- kept only for registered module parameters, refreshed when the
    optimizer bumps the version counter, dropped when the addresses are registered again (a new model at a recycled address)."""
    from grl_image_restoration_amd import autograd as AG

    x = torch.randn(5, 180)
    xp = AG._padded(x, 192, ones=True)
    assert xp.shape == (5, 192) and xp.is_contiguous() and torch.equal(xp[:, :180], x)
    assert torch.all(xp[:, 180] == 1) and torch.all(xp[:, 181:] == 0)
    assert AG._padded(x, 192)[:, 180:].abs().max() == 0 and AG._padded(x, 180) is x
    # dy^T [x | 1 | 0]: column 180 of the weight-gradient product is the bias gradient
    dy = torch.randn(5, 7)
    full = dy.t() @ xp
    assert torch.allclose(full[:, 180], dy.sum(0), atol=1e-6) and torch.allclose(full[:, :180], dy.t() @ x, atol=1e-5)

    lin = torch.nn.Linear(180, 100)
    w = lin.weight
    a = AG._padded_weight(w, 128, 192)
    assert a.shape == (128, 192) and a.dtype == torch.float16 and torch.equal(a[:100, :180], w.detach().half()) and a[100:].abs().max() == 0
    assert AG._padded_weight(w, 128, 192) is not a              # not registered: rebuilt every call (temporaries can alias)
    AG.register_parameters(lin)
    a = AG._padded_weight(w, 128, 192)
    assert AG._padded_weight(w, 128, 192) is a                  # forward -> backward of the same step: the same copy
    at = AG._padded_weight(w, 128, 192, transposed=True)
    assert at.shape == (192, 128) and torch.equal(at, a.t()) and AG._padded_weight(w, 128, 192, transposed=True) is at
    with torch.no_grad():
        w.add_(1.0)                                             # an optimizer step bumps the version
    b = AG._padded_weight(w, 128, 192)
    assert torch.equal(b[:100, :180], w.detach().half())
    assert torch.equal(AG._padded_weight(w, 128, 192, transposed=True), b.t())
    AG.register_parameters(lin)                                 # re-registration forgets the copies made for the old owner
    assert w.data_ptr() not in AG._WEIGHTS
    del lin


def test_diagonal_ring_reduction_index_math():
    """The register-level diagonal sum of csrc/attention_bwd.hip (diag_ring), restated lane by lane in numpy: Horner from key
    row 31 down to 0 on a 64-lane ring that rotates one lane towards lane 0 per row.  Lane p must end with the sum of the tile's
    entries on the diagonal key - query = -p (p < 32) or 64 - p (p > 32) -- the table entry  ebase + key - query  the kernel
    then adds it to with one atomic per lane."""
    import numpy as np

    rng = np.random.default_rng(0)
    tile = rng.standard_normal((32, 32))          # [key row][query]
    z = np.zeros(64)
    for key in range(31, -1, -1):
        z = np.roll(z, -1)                        # wave_rol:1 -- lane i takes lane (i + 1) mod 64
        z[:32] += tile[key]                       # the row sits in lanes 0-31 over zeros (v_permlane32_swap against 0)
    for p in range(64):
        if p == 32:
            assert z[p] == 0.0
            continue
        d = -p if p < 32 else 64 - p              # key - query
        want = sum(tile[k, k - d] for k in range(32) if 0 <= k - d < 32)
        assert abs(z[p] - want) < 1e-12, (p, d)


@pytest.mark.parametrize("W,ww,shx", [(64, 32, 16), (256, 32, 16), (64, 64, 32), (128, 64, 32), (32, 32, 16), (256, 128, 64), (128, 64, 0)])
def test_banded_shift_mask_assumption(W, ww, shx):
    """The fast attention kernels (forward and backward) apply the shifted-window mask per 16-column band: with windows a whole
    multiple of 32 wide and shifts a multiple of 16, the 16 consecutive keys (queries) of each half of a 32-aligned tile row
    share one region label.  Checked against the closed-form region function for the shipped geometries."""
    assert ww % 32 == 0 and shx % 16 == 0
    for x0 in range(0, W, 16):
        labels = {geometry.region1d(x, W, ww, shx) for x in range(x0, x0 + 16)}
        assert len(labels) == 1, (x0, labels)


def test_integration_stub_matches_the_header():
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would copy.  Its structures must have the
    layout of include/grl_hip.h (checked against gcc elsewhere in this file via _lib.py): a short GrlAttnArgs would make the
    kernel read its trailing pointers from whatever follows."""
    import ctypes as C
    import re
    import types

    from grl_image_restoration_amd import _lib

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "class GrlAttnArgs" in b]
    assert len(stub) == 1
    # executed with the library load stubbed out: only the structure definitions matter here
    fake_c = types.SimpleNamespace(**{k: getattr(C, k) for k in dir(C) if not k.startswith("_")})
    class _NoLib:
        def __getattr__(self, name):
            return types.SimpleNamespace(argtypes=None, restype=None)
    fake_c.CDLL = lambda *a, **k: _NoLib()
    code = stub[0].replace("import ctypes as C, torch", "torch = None")
    ns = {"C": fake_c, "__builtins__": __builtins__}
    exec(code, ns)
    for name in ("GrlTokenGrid", "GrlAttnArgs"):
        mine, theirs = getattr(_lib, name), ns[name]
        assert C.sizeof(theirs) == C.sizeof(mine), name
        assert [(f[0], getattr(theirs, f[0]).offset) for f in theirs._fields_] == [(f[0], getattr(mine, f[0]).offset) for f in mine._fields_], name


def test_forget_parameters_follows_replaced_tensors():
    """GRL.invalidate_plan after parameters were REPLACED (load_state_dict(assign=True), m.weight = nn.Parameter(...)): the fp16
    weight copies cached under the old addresses are dropped and the module is registered under the new ones."""
    import torch

    from grl_image_restoration_amd import autograd as AG

    m = torch.nn.Linear(8, 8)
    AG.register_parameters(m)
    old = set(p.data_ptr() for p in m.parameters())
    assert AG._REGISTERED[m] == old
    for ptr in old:
        AG._WEIGHTS[ptr] = ("stale",)
    keep_alive = list(m.parameters())          # the old storages stay allocated: the new tensors get new addresses
    m.weight = torch.nn.Parameter(torch.zeros(8, 8))
    m.bias = torch.nn.Parameter(torch.zeros(8))
    new = set(p.data_ptr() for p in m.parameters())
    assert not (new & old)
    AG.forget_parameters(m)
    assert not any(ptr in AG._WEIGHTS for ptr in old | new)
    assert AG._REGISTERED[m] == new and all(AG._is_registered(ptr) for ptr in new)
    assert not any(AG._is_registered(ptr) for ptr in old)
    del keep_alive
    # a module that was never registered stays unregistered
    m2 = torch.nn.Linear(4, 4)
    AG.forget_parameters(m2)
    assert m2 not in AG._REGISTERED


def test_forget_parameters_keeps_the_buffers_a_captured_step_points_at():
    """A captured training step rewrites the padded fp16 weight copies through their ADDRESSES on every replay.  invalidate_plan ->
    forget_parameters (an eval forward between replays gets there through the version stamp) must therefore only mark the copies of
    parameters that stayed in place stale -- freeing them let a replay write into whatever tensor the allocator handed the memory to
    next (round 5)."""
    import torch

    from grl_image_restoration_amd import autograd as AG

    lin = torch.nn.Linear(180, 100)
    AG.register_parameters(lin)
    w = lin.weight
    a = AG._padded_weight(w, 128, 192)
    at = AG._padded_weight(w, 128, 192, transposed=True)
    with torch.no_grad():
        w.data.mul_(2.0)                                      # a change the version counter does not see
    AG.forget_parameters(lin)
    assert w.data_ptr() in AG._WEIGHTS and AG._is_registered(w.data_ptr())
    b = AG._padded_weight(w, 128, 192)
    assert b is a and b.data_ptr() == a.data_ptr()            # the same buffer, refreshed in place
    assert torch.equal(b[:100, :180], w.detach().half())
    assert torch.equal(AG._padded_weight(w, 128, 192, transposed=True), b.t())
    del lin, at


def test_transposed_view_is_the_same_attention_on_the_oracle():
    """GrlTokenGrid.transposed + ops.transpose_table (round 4): launching a (query window, key window) pair on the transposed view
    of its grids with the transposed bias table is the SAME attention.  Checked on the oracle's own index arithmetic
    (models/common/ops.py:352-375 as restated in oracle/grl_oracle.py): for every (query, key) pair the row of the transposed
    table that the transposed geometry addresses holds the entry the original geometry addresses -- and the shifted-window masks
    of the transposed problem are the original ones with the token order transposed."""
    import torch

    from grl_image_restoration_amd import ops
    from oracle import grl_oracle as O

    for win, df, w2a in (((8, 4), 2, True), ((8, 4), 2, False), ((12, 24), 4, True), ((6, 6), 1, True), ((16, 32), 4, False)):
        awin = (win[0] // df, win[1] // df)
        q_win, k_win = (win, awin) if w2a else (awin, win)
        idx = O.rel_index(win, df, w2a)                                   # [Nq, Nk] -> row of the (Dy*Dx)-row table
        winT, q_winT, k_winT = (win[1], win[0]), (q_win[1], q_win[0]), (k_win[1], k_win[0])
        idxT = O.rel_index(winT, df, w2a)                                 # the transposed problem's rows
        rows = (q_win[0] + k_win[0] - 1) * (q_win[1] + k_win[1] - 1)
        assert int(idx.max()) == rows - 1 == int(idxT.max())
        bias = torch.arange(rows, dtype=torch.float32).view(rows, 1) * 1.0 + 0.25     # a table whose entries identify their row
        biasT = ops.transpose_table(bias, q_win, k_win)
        # token (h, w) of a window is token (w, h) of the transposed window
        def perm(wn):
            h, w = torch.meshgrid(torch.arange(wn[0]), torch.arange(wn[1]), indexing="ij")
            return (w * wn[0] + h).reshape(-1)                            # position in the transposed window's row-major order
        pq, pk = perm(q_win), perm(k_win)
        got = biasT[idxT.reshape(-1), 0].view(idxT.shape)[pq][:, pk]      # entry the transposed launch adds for the ORIGINAL (q, k) pair
        want = bias[idx.reshape(-1), 0].view(idx.shape)
        assert torch.equal(got, want), (win, df, w2a)

    # region masks: transposing image, window and shift transposes the token order inside each window and the order of the windows
    res, win, shift = (16, 24), (8, 12), (4, 6)
    m = O.shift_mask(res, win, shift, mode="w")                           # [nW, N, N]
    mT = O.shift_mask((res[1], res[0]), (win[1], win[0]), (shift[1], shift[0]), mode="w")
    nwy, nwx = res[0] // win[0], res[1] // win[1]
    h, w = torch.meshgrid(torch.arange(win[0]), torch.arange(win[1]), indexing="ij")
    p = (w * win[0] + h).reshape(-1)
    for wy in range(nwy):
        for wx in range(nwx):
            assert torch.equal(mT[wx * nwy + wy][p][:, p], m[wy * nwx + wx])


def test_split_site_spec_and_linear_split_blob_layout():
    """model._split_sites ('name' = both operands split, 'name:x' = activations only) and ops.pack_linear_split: the register image
    of include/grl_hip.h (GrlLinearArgs.w_regs) -- per slab of 192 columns and compute wave the hi, then the lo A fragments, lane l
    element e of k-step s = W[32 (6 slab + wave) + (l & 31)][16 s + 8 (l >> 5) + e], columns beyond Npad zero."""
    import torch

    from grl_image_restoration_amd import _lib, ops
    from grl_image_restoration_amd.model import _split_sites

    assert _split_sites("stage_conv:x,after,last,cab0") == {"stage_conv": 2, "after": 3, "last": 3, "cab0": 3}
    assert _split_sites("") == {}
    g = torch.Generator().manual_seed(5)
    for N, K in ((576, 192), (96, 192), (128, 128), (192, 384)):
        w = torch.randn(N, K, generator=g)
        w3 = ops.split3_weight(w)
        blob = ops.pack_linear_split(w3)
        ns = (N + 191) // 192
        assert blob.numel() == _lib.lib().grl_linear_split_blob_bytes(N, K) == ns * 6 * 2 * (K // 16) * 1024
        img = blob.view(torch.float16).view(ns, 6, 2, K // 16, 64, 8)
        hi, lo = w3[:, :K], w3[:, 2 * K :]
        for (sl, wv, s, l, e) in ((0, 0, 0, 0, 0), (ns - 1, 2, K // 16 - 1, 63, 7), (0, 5, 3, 37, 2), (ns - 1, 0, 1, 31, 5)):
            col, k = 32 * (6 * sl + wv) + (l & 31), 16 * s + 8 * (l >> 5) + e
            want_hi = hi[col, k] if col < N else torch.tensor(0.0, dtype=torch.float16)
            want_lo = lo[col, k] if col < N else torch.tensor(0.0, dtype=torch.float16)
            assert img[sl, wv, 0, s, l, e] == want_hi and img[sl, wv, 1, s, l, e] == want_lo
    assert ops.pack_linear_split(ops.split3_weight(torch.randn(64, 96, generator=g))) is None      # K = 96: not a shape the kernel takes


def test_training_glue_shortcuts_equal_the_plain_torch_chains():
    """Round-4 launch-count shortcuts of the training path, on the CPU: GRL._attn_table's single gather + scale equals
    tables.kernel_table(16 * sigmoid(CPB-MLP)) on every entry a (query, key) pair can address (the pad entries repeat row 0 instead
    of being zero), with the same gradient for the CPB-MLP; GRL._to_planes's single cat equals the pad + permute it replaced, with
    the constants the attention kernel wants (k slot 31, v column d) in place."""
    import math

    import torch
    import torch.nn.functional as F

    from grl_image_restoration_amd import GRL, make_config, tables

    cfg = make_config("base", "sr_ckpt_df2", upscale=4, img_size=64, depths=[1], num_heads_window=[3], num_heads_stripe=[3])
    torch.manual_seed(3)
    m = GRL(**cfg)
    blk = m.layers[0].blocks[0]
    for tr, win, df in ((blk.attn.window_attn.attn_transform, (32, 32), 1), (blk.attn.stripe_attn.attn_transform1, (64, 64), 2)):
        coords = tables.coords_table(win, df, device="cpu")
        rows = coords.shape[0]
        got = m._attn_table(tr, win, df, torch.device("cpu"))
        h = F.relu(F.linear(coords, tr.cpb_mlp[0].weight, tr.cpb_mlp[0].bias))
        want = tables.kernel_table(16.0 * torch.sigmoid(F.linear(h, tr.cpb_mlp[2].weight)))
        assert got.shape == want.shape and got.is_contiguous()
        assert torch.allclose(got[:, :rows], want[:, :rows], rtol=2e-7, atol=0)
        assert torch.equal(got[:, rows:], got[:, rows - 1 : rows].expand(-1, got.shape[1] - rows))      # pad = row 0 (last after the reversal)
        gw = torch.randn(got.shape, generator=torch.Generator().manual_seed(4))
        gw[:, rows:] = 0                                                                             # nothing reads the pad entries
        g1 = torch.autograd.grad((got * gw).sum(), tr.cpb_mlp[2].weight, retain_graph=True)[0]
        g2 = torch.autograd.grad((want * gw).sum(), tr.cpb_mlp[2].weight)[0]
        # (two fp32 evaluation orders of the same expression -- the K = 2 layer as two multiply-adds + bmm against F.linear: both
        # are 1.4e-5 from the float64 gradient on entries up to 35, and 4e-6 from each other)
        assert torch.allclose(g1, g2, rtol=1e-5, atol=2e-5)

    t = torch.randn(96, 3, 30, generator=torch.Generator().manual_seed(6), requires_grad=True)
    plain = F.pad(t, (0, 2)).permute(1, 0, 2).contiguous()
    assert torch.equal(m._to_planes(t), plain)
    k = m._to_planes(t, 31)
    assert torch.equal(k[..., :31], plain[..., :31]) and bool((k[..., 31] == 1).all())
    v = m._to_planes(t, 30)
    assert torch.equal(v[..., :30], plain[..., :30]) and bool((v[..., 30] == 1).all()) and bool((v[..., 31] == 0).all())
    (g,) = torch.autograd.grad(k.sum(), t)
    assert bool((g == 1).all())                                                                        # the constant block takes no gradient
    t32 = torch.randn(64, 2, 32)
    assert torch.equal(m._to_planes(t32, -1), t32.permute(1, 0, 2).contiguous())
    assert math.isclose(float((m._to_planes(t, 31).sum() - plain.sum()).detach()), 96 * 3, rel_tol=1e-5)


def test_fused_adamw_capture_mode_bookkeeping():
    """FusedAdamW.enable_capture (host logic, CPU tensors: no launch): the step counts move to device counters, state_dict() and a
    second enable_capture read them back, load_state_dict pushes loaded counts into the SAME counter tensors (a captured graph
    holds their addresses)."""
    import torch

    from grl_image_restoration_amd import FusedAdamW

    ps = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
    opt = FusedAdamW(ps, lr=1e-3)
    with pytest.raises(RuntimeError):
        opt.enable_capture()                                  # no eager step yet: no state
    for p in ps:
        opt.state[p] = dict(step=7, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
    opt.enable_capture()
    ctr = opt._step_dev[0]
    assert int(ctr) == 7 and opt._pinned[0].numel() == 2
    ctr.add_(3)                                               # three replays of a captured step
    sd = opt.state_dict()
    assert all(int(s["step"]) == 10 for s in sd["state"].values())
    opt.enable_capture()                                      # a second graph shares the counters of the first
    assert opt._step_dev[0] is ctr and int(ctr) == 10
    keep = opt._step_dev[0]
    for s in sd["state"].values():
        s["step"] = 4
    opt.load_state_dict(sd)
    assert opt._step_dev[0] is keep and int(keep) == 4
    assert all(opt.state[p]["step"] == 4 for p in ps)


def test_fused_adamw_capture_mode_keeps_addresses_and_follows_schedulers():
    """ADVICE r4 (all three mediums), host logic on CPU tensors: in capture mode (a) load_state_dict copies the loaded moments INTO the
    buffers a captured launch points at and keeps the pointer tables, (b) every capture gets its own pinned gradient-pointer table,
    (c) lr / weight decay live in a device tensor that refresh_capture_hyper() keeps equal to param_groups, and betas / eps -- held by
    value in a captured launch -- refuse to change silently."""
    import torch

    from grl_image_restoration_amd import FusedAdamW

    ps = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
    opt = FusedAdamW(ps, lr=1e-3, weight_decay=1e-2)
    for p in ps:
        opt.state[p] = dict(step=2, exp_avg=torch.full_like(p, 0.5), exp_avg_sq=torch.full_like(p, 0.25))
    opt.enable_capture()
    first_table = opt._pinned[0]
    m_ptr = [opt.state[p]["exp_avg"].data_ptr() for p in ps]
    v_ptr = [opt.state[p]["exp_avg_sq"].data_ptr() for p in ps]
    opt._tables[0] = dict(key="sentinel")                     # stands for the device pointer tables of a captured launch
    import copy
    sd = copy.deepcopy(opt.state_dict())                      # (a checkpoint read back from disk: state_dict() itself aliases the live state)
    for s in sd["state"].values():
        s["exp_avg"] = s["exp_avg"] * 0 + 3.0
        s["exp_avg_sq"] = s["exp_avg_sq"] * 0 + 9.0
        s["step"] = 11
    opt.load_state_dict(sd)
    assert [opt.state[p]["exp_avg"].data_ptr() for p in ps] == m_ptr and [opt.state[p]["exp_avg_sq"].data_ptr() for p in ps] == v_ptr
    assert all(float(opt.state[p]["exp_avg"].mean()) == 3.0 and float(opt.state[p]["exp_avg_sq"].mean()) == 9.0 for p in ps)
    assert opt._tables[0]["key"] == "sentinel" and int(opt._step_dev[0]) == 11
    opt.enable_capture()                                      # a second capture: its own pinned table, the first one stays alive
    assert opt._pinned[0] is not first_table and any(t is first_table for t in opt._pinned_all)
    # schedulers: lr / weight decay follow param_groups through the device copy
    assert torch.allclose(opt._hyper_dev[0], torch.tensor([1e-3, 1e-2]))
    opt.param_groups[0]["lr"] = 5e-4
    opt.param_groups[0]["weight_decay"] = 0.0
    opt.refresh_capture_hyper()
    assert torch.allclose(opt._hyper_dev[0], torch.tensor([5e-4, 0.0]))
    opt.param_groups[0]["betas"] = (0.8, 0.999)
    with pytest.raises(RuntimeError):
        opt.refresh_capture_hyper()


def test_calibration_predicts_the_blocks_to_split():
    """GRL._calibrated_plan's variance bookkeeping (model.predicted_split_count): blocks are sorted by what their fp16 rounding
    costs, the cheapest stay on fp16 operands while the summed variance fits the bar."""
    from grl_image_restoration_amd.model import predicted_split_count as psc

    bar = 1.3e-4
    assert psc([], 0.0, bar) == 0
    assert psc([1e-10] * 40, 0.0, bar) == 0                                 # 40 x (1e-5)^2 = (6.3e-5)^2: everything stays fast
    assert psc([(3e-4) ** 2] + [1e-10] * 39, 0.0, bar) == 1                  # one dominant block (weight draw 15 of the tests)
    assert psc([(1e-4) ** 2] * 4, 0.0, bar) == 3                            # one 1e-4 fits under 0.9 * 1.3e-4 = 1.17e-4, two (1.41e-4) do not
    assert psc([(1e-4) ** 2] * 4, (1.2e-4) ** 2, bar) == 4                  # the convolutions around the blocks already exceed the margin
    v = sorted(((i + 1) * 1e-5) ** 2 for i in range(10))[::-1]
    k = psc(v, (2e-5) ** 2, bar)
    assert (2e-5) ** 2 + sum(v[k:]) <= (0.9 * bar) ** 2 < (2e-5) ** 2 + sum(v[k - 1:])   # minimal


def test_poison_and_deterministic_switches_are_off_by_default():
    from grl_image_restoration_amd import _lib, ops

    assert not ops._POISON and not ops.deterministic() and not _lib._DIRTY_LDS
    prev = ops.set_poison(True)
    try:
        t = ops.empty(3, 5, dtype=torch.float32, device="cpu")
        assert bool(torch.isnan(t).all())
        i = ops.empty(4, dtype=torch.int32, device="cpu")
        assert bool((i == 0x7F7F7F7F).all())
    finally:
        ops.set_poison(prev)
    assert bool(torch.isfinite(ops.empty(2, dtype=torch.float32, device="cpu").fill_(0)).all())
