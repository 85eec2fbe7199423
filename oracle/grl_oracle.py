"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32) restatement of the GRL forward pass.

This file is the *checker* for the HIP hot path: tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py are the only callers.  The product package never imports it.

It restates, op by op, what the reference network computes, so it can travel to the GPU box
(where /root/reference does not exist).  Pinning: tests/test_oracle_pinned.py runs the real,
unmodified reference (through oracle/refshim.py) on seeded inputs/weights in the build
container and requires this restatement to agree to float round-off; the outputs are also
frozen as fixtures under tests/golden/ (oracle/make_golden.py) so the pin can be re-checked
anywhere.  The reference ships no tests/golden vectors of its own for this path.

All functions are functional: parameters come from a ``state_dict`` using the reference's key
names, so a reference checkpoint drives the oracle unchanged.

Reference file:line for every restated function is given in its docstring; paths are relative
to the reference root (models/...).
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# geometry helpers
# ----------------------------------------------------------------------------------------------
def stripe_info(stripe_size, stripe_groups, stripe_shift, x_size):
    """models/common/mixed_attn_block_efficient.py:61-70 (_get_stripe_info)."""
    size, shift = [], []
    for s, g, d in zip(stripe_size, stripe_groups, x_size):
        if g is None:
            size.append(s)
            shift.append(s // 2 if stripe_shift else 0)
        else:
            size.append(d // g)
            shift.append(0 if g == 1 else d // (g * 2))
    return size, shift


def pad_size(cfg) -> int:
    """models/networks/grl.py:273-276."""
    mss = max(0 if s is None else s for s in cfg["stripe_size"])
    msg = max(0 if s is None else s for s in cfg["stripe_groups"]) * cfg["anchor_window_down_factor"]
    return max(cfg["window_size"], mss, msg)


def coords_table(window: Sequence[int], df: int = 1) -> Tensor:
    """models/common/ops.py:225-271 (get_relative_coords_table_all, pretrained size = [0,0]).

    Log-spaced relative coordinates between a window of size ``window`` and its anchor window
    ``window // df``.  Returns (1, Wh+AWh-1, Ww+AWw-1, 2) float32.
    """
    aws = [w // df for w in window]
    pos = [w - 1 - (w - a) // 2 for w, a in zip(window, aws)]
    neg = [-(a - 1) - (w - a) // 2 for w, a in zip(window, aws)]
    ch = torch.arange(neg[0], pos[0] + 1, dtype=torch.float32)
    cw = torch.arange(neg[1], pos[1] + 1, dtype=torch.float32)
    t = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).unsqueeze(0).contiguous()
    t[..., 0] /= pos[0]
    t[..., 1] /= pos[1]
    t *= 8
    t = torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / math.log2(8)
    return t


def rel_index(window: Sequence[int], df: int = 1, window_to_anchor: bool = True) -> Tensor:
    """models/common/ops.py:352-375 (get_relative_position_index_simple) + :308-316.

    (N1, N2) int64 for window->anchor, (N2, N1) for anchor->window; values index the flattened
    coords table above.
    """
    aws = [w // df for w in window]
    hw, ww = torch.meshgrid(torch.arange(window[0]), torch.arange(window[1]), indexing="ij")
    ha, wa = torch.meshgrid(torch.arange(aws[0]), torch.arange(aws[1]), indexing="ij")
    hw, ww, ha, wa = hw.reshape(-1), ww.reshape(-1), ha.reshape(-1), wa.reshape(-1)
    D = aws[1] + window[1] - 1
    if window_to_anchor:
        return (hw[:, None] - ha[None, :] + aws[0] - 1) * D + (ww[:, None] - wa[None, :] + aws[1] - 1)
    return (ha[:, None] - hw[None, :] + window[0] - 1) * D + (wa[:, None] - ww[None, :] + window[1] - 1)


def _region_labels(res: Sequence[int], window: Sequence[int], shift: Sequence[int]) -> Tensor:
    """models/common/ops.py:76-100 (_fill_window): 3x3 region label image, window-partitioned.

    Returns (nW, Wh*Ww) float labels.  Slices follow Python semantics including the
    ``slice(-0, None)`` == whole-axis corner case when a shift is 0.
    """
    img = torch.zeros(res[0], res[1])
    hs = (slice(0, -window[0]), slice(-window[0], -shift[0]), slice(-shift[0], None))
    ws_ = (slice(0, -window[1]), slice(-window[1], -shift[1]), slice(-shift[1], None))
    c = 0
    for h in hs:
        for w in ws_:
            img[h, w] = c
            c += 1
    nh, nw = res[0] // window[0], res[1] // window[1]
    img = img.view(nh, window[0], nw, window[1]).permute(0, 2, 1, 3)
    return img.reshape(nh * nw, window[0] * window[1])


def shift_mask(res, window, shift, df: int = 1, mode: str = "w") -> Tensor:
    """models/common/ops.py:112-157 (calculate_mask / calculate_mask_all).

    mode 'w'  : (nW, N, N) window self-attention
    mode 'w2a': (nW, N1, N2) queries = window tokens, keys = anchors
    mode 'a2w': (nW, N2, N1) queries = anchors, keys = window tokens
    Values 0 / -100.
    """
    lw = _region_labels(res, window, shift)
    if mode == "w":
        d = lw.unsqueeze(1) - lw.unsqueeze(2)
    else:
        la = _region_labels([s // df for s in res], [s // df for s in window], [s // df for s in shift])
        d = lw.unsqueeze(2) - la.unsqueeze(1) if mode == "w2a" else la.unsqueeze(2) - lw.unsqueeze(1)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def partition(x: Tensor, window: Sequence[int]) -> Tensor:
    """models/common/ops.py:36-53 (window_partition): (B,H,W,C) -> (nW*B, Wh, Ww, C)."""
    B, H, W, C = x.shape
    x = x.view(B, H // window[0], window[0], W // window[1], window[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, window[0], window[1], C)


def unpartition(w: Tensor, window: Sequence[int], size: Sequence[int]) -> Tensor:
    """models/common/ops.py:56-73 (window_reverse)."""
    H, W = size
    B = w.shape[0] // ((H // window[0]) * (W // window[1]))
    x = w.view(B, H // window[0], W // window[1], window[0], window[1], -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def bias_table(p: Dict[str, Tensor], prefix: str, table: Tensor) -> Tensor:
    """16*sigmoid(cpb_mlp(table)) -> (rows, heads).

    models/common/mixed_attn_block_efficient.py:41-47 with CPB_MLP = Linear(2,512)+ReLU+
    Linear(512,heads,no bias) (mixed_attn_block.py:24-31).  The reference applies the sigmoid
    after the gather; gather and elementwise sigmoid commute.
    """
    w0 = p[prefix + "cpb_mlp.0.weight"]
    h = F.relu(F.linear(table.reshape(-1, 2).to(w0.dtype), w0, p[prefix + "cpb_mlp.0.bias"]))   # (the oracle also runs in float64: tests' "truth")
    return 16.0 * torch.sigmoid(F.linear(h, p[prefix + "cpb_mlp.2.weight"]))


def logit_scale(p: Dict[str, Tensor], prefix: str) -> Tensor:
    """exp(clamp(logit_scale, max=ln 100)), shape (heads,1,1) (efficient.py:39)."""
    return torch.clamp(p[prefix + "logit_scale"], max=math.log(1.0 / 0.01)).exp()


def cosine_attention(q, k, v, p, prefix, table, index, mask):
    """models/common/mixed_attn_block_efficient.py:77-94 + :36-58.

    q (B_, nh, Nq, d), k/v (B_, nh, Nk, d); index (Nq, Nk); mask (nW, Nq, Nk) or None.
    Returns (B_, nh, Nq, d).
    """
    B_, nh, Nq, _ = q.shape
    Nk = k.shape[2]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    attn = attn * logit_scale(p, prefix)
    bt = bias_table(p, prefix, table)  # rows, nh
    bias = bt[index.reshape(-1)].view(Nq, Nk, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, nh, Nq, Nk) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nh, Nq, Nk)
    attn = torch.softmax(attn, dim=-1)
    return attn @ v


def window_attention(qkv: Tensor, x_size, window, shift: int, nh: int, p, prefix: str) -> Tensor:
    """models/common/mixed_attn_block_efficient.py:128-165 (WindowAttention.forward).

    qkv (B, L, 3*Cb) with channel order [q|k|v][head][d]; returns (B, L, Cb).
    """
    H, W = x_size
    B, L, C3 = qkv.shape
    x = qkv.view(B, H, W, C3)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    x = partition(x, window).reshape(-1, window[0] * window[1], C3)
    B_, N, _ = x.shape
    x = x.reshape(B_, N, 3, nh, -1).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    table = coords_table(window)
    index = rel_index(window)
    mask = shift_mask(x_size, window, [shift, shift], mode="w") if shift > 0 else None
    o = cosine_attention(q, k, v, p, prefix + "attn_transform.", table, index, mask)
    o = o.transpose(1, 2).reshape(B_, N, C3 // 3)
    o = unpartition(o.view(-1, window[0], window[1], C3 // 3), window, x_size)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B, L, C3 // 3)


def anchor_stripe_attention(
    qkv: Tensor, anchor: Tensor, x_size, stripe, shift, do_shift: bool, df: int, nh: int, p, prefix: str
) -> Tensor:
    """models/common/mixed_attn_block_efficient.py:215-270 (AnchorStripeAttention.forward).

    qkv (B, L, 3*Cb); anchor (B, H/df, W/df, Cb); ``stripe``/``shift`` are the already
    resolved sizes for this block's orientation.  Returns (B, L, Cb).
    """
    H, W = x_size
    B, L, C3 = qkv.shape
    Cb = C3 // 3
    x = qkv.view(B, H, W, C3)
    astripe = [s // df for s in stripe]
    ashift = [s // df for s in shift]
    if do_shift:
        x = torch.roll(x, shifts=(-shift[0], -shift[1]), dims=(1, 2))
        anchor = torch.roll(anchor, shifts=(-ashift[0], -ashift[1]), dims=(1, 2))
    x = partition(x, stripe).reshape(-1, stripe[0] * stripe[1], C3)
    a = partition(anchor, astripe).reshape(-1, astripe[0] * astripe[1], Cb)
    B_, N1, _ = x.shape
    N2 = a.shape[1]
    x = x.reshape(B_, N1, 3, nh, -1).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    a = a.reshape(B_, N2, nh, -1).permute(0, 2, 1, 3)
    table = coords_table(stripe, df)
    idx_a2w = rel_index(stripe, df, False)
    idx_w2a = rel_index(stripe, df, True)
    m_a2w = shift_mask(x_size, stripe, shift, df, "a2w") if do_shift else None
    m_w2a = shift_mask(x_size, stripe, shift, df, "w2a") if do_shift else None
    y = cosine_attention(a, k, v, p, prefix + "attn_transform1.", table, idx_a2w, m_a2w)
    o = cosine_attention(q, a, y, p, prefix + "attn_transform2.", table, idx_w2a, m_w2a)
    o = o.transpose(1, 2).reshape(B_, N1, Cb)
    o = unpartition(o.view(B_, stripe[0], stripe[1], Cb), stripe, x_size)
    if do_shift:
        o = torch.roll(o, shifts=(shift[0], shift[1]), dims=(1, 2))
    return o.reshape(B, L, Cb)


def anchor_projection(x: Tensor, x_size, df: int, p, prefix: str) -> Tensor:
    """models/common/mixed_attn_block.py:714-736 (AnchorLinear, avgpool) via :739-785.

    x (B, L, C) -> (B, H/df, W/df, C/2).
    """
    B, L, C = x.shape
    H, W = x_size
    y = F.avg_pool2d(x.transpose(1, 2).reshape(B, C, H, W), df, df)
    y = y.flatten(2).transpose(1, 2)
    y = F.linear(y, p[prefix + "body.0.reduction.weight"], p[prefix + "body.0.reduction.bias"])
    return y.view(B, H // df, W // df, -1)


def mixed_attention(x: Tensor, x_size, blk: dict, p, prefix: str) -> Tensor:
    """models/common/mixed_attn_block_efficient.py:351-381 (MixedAttention.forward)."""
    B, L, C = x.shape
    qkv = F.linear(x, p[prefix + "qkv.body.weight"], p[prefix + "qkv.body.bias"])
    qkv_w, qkv_s = torch.split(qkv, C * 3 // 2, dim=-1)
    anchor = anchor_projection(x, x_size, blk["df"], p, prefix + "anchor.")
    xw = window_attention(
        qkv_w.contiguous(), x_size, blk["window"], blk["window_shift"], blk["nh_w"], p, prefix + "window_attn."
    )
    xs = anchor_stripe_attention(
        qkv_s.contiguous(),
        anchor,
        x_size,
        blk["stripe"],
        blk["stripe_shift_size"],
        blk["stripe_shift"],
        blk["df"],
        blk["nh_s"],
        p,
        prefix + "stripe_attn.",
    )
    y = torch.cat([xw, xs], dim=-1)
    return F.linear(y, p[prefix + "proj.weight"], p[prefix + "proj.bias"])


def cab(x: Tensor, x_size, p, prefix: str) -> Tensor:
    """models/common/mixed_attn_block.py:948-983 (CAB + ChannelAttention).

    conv3x3 C->C/4, exact GELU, conv3x3 C/4->C, squeeze-excite (global mean, 1x1, ReLU, 1x1,
    sigmoid, scale).  x (B, L, C) -> (B, L, C).
    """
    B, L, C = x.shape
    y = x.transpose(1, 2).reshape(B, C, *x_size)
    y = F.conv2d(y, p[prefix + "cab.0.weight"], p[prefix + "cab.0.bias"], padding=1)
    y = F.gelu(y)
    y = F.conv2d(y, p[prefix + "cab.2.weight"], p[prefix + "cab.2.bias"], padding=1)
    s = y.mean(dim=(2, 3), keepdim=True)
    s = F.relu(F.conv2d(s, p[prefix + "cab.3.attention.1.weight"], p[prefix + "cab.3.attention.1.bias"]))
    s = torch.sigmoid(F.conv2d(s, p[prefix + "cab.3.attention.3.weight"], p[prefix + "cab.3.attention.3.bias"]))
    y = y * s
    return y.flatten(2).transpose(1, 2)


def mlp(x: Tensor, p, prefix: str) -> Tensor:
    """models/common/swin_v1_block.py:15-43 (Mlp): fc1 -> exact GELU -> fc2."""
    y = F.gelu(F.linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"]))
    return F.linear(y, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])


def layer_norm(x: Tensor, p, prefix: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), p[prefix + "weight"], p[prefix + "bias"], 1e-5)


def block_forward(x: Tensor, x_size, blk: dict, p, prefix: str, local_connection: bool, res_scale=1.0) -> Tensor:
    """models/common/mixed_attn_block_efficient.py:539-556 (post-norm block, eval mode)."""
    a = layer_norm(mixed_attention(x, x_size, blk, p, prefix + "attn."), p, prefix + "norm1.")
    if local_connection:
        x = x + res_scale * a + cab(x, x_size, p, prefix + "conv.")
    else:
        x = x + res_scale * a
    x = x + res_scale * layer_norm(mlp(x, p, prefix + "mlp."), p, prefix + "norm2.")
    return x


def conv3x3_blc(x: Tensor, x_size, w: Tensor, b: Tensor) -> Tensor:
    B, L, C = x.shape
    y = F.conv2d(x.transpose(1, 2).reshape(B, C, *x_size), w, b, padding=1)
    return y.flatten(2).transpose(1, 2)


# ----------------------------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------------------------
def block_schedule(cfg: dict, x_size) -> List[List[dict]]:
    """Per-stage, per-block static geometry.

    models/networks/grl.py:105-131 (window shift iff i even; stripe type H/W alternating;
    stripe shift iff i%4 in {2,3}) and mixed_attn_block_efficient.py:466-471 (W blocks use the
    reversed stripe size/groups), :229-233 (sizes resolved against x_size).
    """
    ws = cfg["window_size"]
    window = list(ws) if isinstance(ws, (list, tuple)) else [ws, ws]
    out = []
    for si, depth in enumerate(cfg["depths"]):
        stage = []
        for i in range(depth):
            typ = "H" if i % 2 == 0 else "W"
            ss = cfg["stripe_size"][::-1] if typ == "W" else cfg["stripe_size"]
            sg = cfg["stripe_groups"][::-1] if typ == "W" else cfg["stripe_groups"]
            do_shift = (i % 4 in (2, 3)) if cfg["stripe_shift"] else False
            stripe, sshift = stripe_info(ss, sg, do_shift, x_size)
            stage.append(
                dict(
                    window=window,
                    window_shift=window[0] // 2 if i % 2 == 0 else 0,
                    stripe=stripe,
                    stripe_shift=do_shift,
                    stripe_shift_size=sshift,
                    df=cfg["anchor_window_down_factor"],
                    nh_w=cfg["num_heads_window"][si],
                    nh_s=cfg["num_heads_stripe"][si],
                )
            )
        out.append(stage)
    return out


def pad_input(x: Tensor, ps: int) -> Tensor:
    """models/networks/grl.py:479-489 (check_image_size): reflect pad, constant on failure."""
    _, _, h, w = x.shape
    ph, pw = (ps - h % ps) % ps, (ps - w % ps) % ps
    try:
        return F.pad(x, (0, pw, 0, ph), "reflect")
    except BaseException:
        return F.pad(x, (0, pw, 0, ph), "constant")


def forward_features(x: Tensor, cfg: dict, p, capture: Optional[dict] = None) -> Tensor:
    """models/networks/grl.py:491-504 + TransformerStage.forward :164-170."""
    x_size = (x.shape[2], x.shape[3])
    t = x.flatten(2).transpose(1, 2)
    t = layer_norm(t, p, "norm_start.")
    sched = block_schedule(cfg, x_size)
    for si, stage in enumerate(sched):
        r = t
        for bi, blk in enumerate(stage):
            r = block_forward(r, x_size, blk, p, f"layers.{si}.blocks.{bi}.", cfg["local_connection"],
                              0.1 if cfg.get("init_method", "n") == "r" else 1.0)
            if capture is not None:
                capture[f"layers.{si}.blocks.{bi}"] = r
        t = conv3x3_blc(r, x_size, p[f"layers.{si}.conv.weight"], p[f"layers.{si}.conv.bias"]) + t
    t = layer_norm(t, p, "norm_end.")
    return t.transpose(1, 2).reshape(x.shape[0], -1, *x_size)


def grl_forward(x: Tensor, cfg: dict, p: Dict[str, Tensor], capture: Optional[dict] = None) -> Tensor:
    """models/networks/grl.py:506-551 (GRL.forward), eval mode, fp32."""
    H, W = x.shape[2:]
    s = cfg["upscale"]
    x = pad_input(x, pad_size(cfg))
    if cfg.get("in_channels", 3) == 3:
        mean = torch.tensor((0.4488, 0.4371, 0.4040), dtype=x.dtype).view(1, 3, 1, 1)
    else:
        mean = torch.zeros(1, 1, 1, 1, dtype=x.dtype)
    rng = cfg.get("img_range", 1.0)
    x = (x - mean) * rng
    up = cfg.get("upsampler", "")

    def conv(name, t):
        return F.conv2d(t, p[name + ".weight"], p[name + ".bias"], padding=1)

    f = conv("conv_first", x)
    body = conv("conv_after_body", forward_features(f, cfg, p, capture)) + f
    if up == "pixelshuffle":
        y = F.leaky_relu(conv("conv_before_upsample.0", body), 0.01)
        if (s & (s - 1)) == 0:
            for i in range(int(math.log2(s))):
                y = F.pixel_shuffle(conv(f"upsample.up.{2 * i}", y), 2)
        elif s == 3:
            y = F.pixel_shuffle(conv("upsample.up.0", y), 3)
        else:
            raise ValueError(s)
        y = conv("conv_last", y)
    elif up == "pixelshuffledirect":
        y = F.pixel_shuffle(conv("upsample.up.0", body), s)
    elif up == "nearest+conv":
        y = F.leaky_relu(conv("conv_before_upsample.0", body), 0.01)
        y = F.leaky_relu(conv("conv_up1", F.interpolate(y, scale_factor=2, mode="nearest")), 0.2)
        y = F.leaky_relu(conv("conv_up2", F.interpolate(y, scale_factor=2, mode="nearest")), 0.2)
        y = conv("conv_last", F.leaky_relu(conv("conv_hr", y), 0.2))
    else:
        r = conv("conv_last", body)
        in_ch = cfg.get("in_channels", 3)
        out_ch = cfg.get("out_channels") or in_ch
        y = x + r if in_ch == out_ch else r
    y = y / rng + mean
    return y[:, :, : H * s, : W * s]


# ----------------------------------------------------------------------------------------------
# seeded weights / inputs shared by every parity test
# ----------------------------------------------------------------------------------------------
def perturb_state_dict(sd: Dict[str, Tensor], seed: int = 0) -> Dict[str, Tensor]:
    """Move parameters away from their init so that every code path carries signal.

    At init LN is identity-affine, all biases are 0 and logit_scale is ln 10 for every head;
    a kernel that dropped a bias or swapped two heads would still pass.  Buffers are untouched.
    """
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.startswith(("table_", "index_", "mask_")) or not v.is_floating_point():
            out[k] = v.clone()
            continue
        r = torch.randn(v.shape, generator=g, dtype=v.dtype)
        if k.endswith("logit_scale"):
            out[k] = v + 0.3 * r
        elif ".norm" in k or k.startswith("norm_"):
            out[k] = v + 0.1 * r
        elif k.endswith(".bias"):
            out[k] = v + 0.02 * r
        else:
            out[k] = v + 0.1 * v.std().clamp_min(1e-3) * r if v.numel() > 1 else v + 0.02 * r
    return out


def synthetic_pair(task: str, hw: Tuple[int, int], scale: int = 1, batch: int = 1, seed: int = 1):
    """Seeded synthetic (LQ, GT) pair in [0,1]: GT = 5x5-box-blurred uniform noise;
    SR LQ = avg_pool(GT, s); dn LQ = GT + 25/255*randn; deblur LQ = 9-tap horizontal box of GT."""
    g = torch.Generator().manual_seed(seed)
    H, W = hw[0] * scale, hw[1] * scale
    gt = torch.rand(batch, 3, H + 4, W + 4, generator=g)
    gt = F.avg_pool2d(gt, 5, 1)
    if task == "sr":
        lq = F.avg_pool2d(gt, scale) if scale > 1 else gt.clone()
    elif task == "dn":
        lq = gt + (25.0 / 255.0) * torch.randn(gt.shape, generator=g)
    elif task == "deblur":
        k = torch.ones(3, 1, 1, 9) / 9.0
        lq = F.conv2d(F.pad(gt, (4, 4, 0, 0), "replicate"), k, groups=3)
    else:
        raise ValueError(task)
    return lq.contiguous(), gt.contiguous()


def seeded_state_dict(shapes: Dict[str, Sequence[int]], seed: int = 0, logit_scale_mean: float = math.log(10.0)) -> Dict[str, Tensor]:
    """Deterministic, init-order-independent parameters for a GRL of the given key->shape map.

    Every key gets its own generator (seed mixed with crc32 of the key) so the reference module
    (build container) and the product module (GPU box) obtain bit-identical weights without
    either depending on the other's construction order.  Distributions mimic a trained net
    closely enough to exercise every path: Linear/conv weights ~ U(-1/sqrt(fan_in), ..),
    LayerNorm gamma ~ 1 + 0.1 N, biases ~ 0.02 N, logit_scale ~ ``logit_scale_mean`` (ln 10 = the
    init value) + 0.3 N; ``logit_scale_mean = ln 100`` puts about half of the heads above the clamp
    of mixed_attn_block_efficient.py:39 (what a trained Swin-V2-style checkpoint looks like).
    Buffer keys (table_/index_/mask_) are skipped.
    """
    import zlib

    out = {}
    for k in sorted(shapes):
        if k.startswith(("table_", "index_", "mask_")):
            continue
        shp = tuple(shapes[k])
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(k.encode())) % (2**31 - 1))
        if k.endswith("logit_scale"):
            v = logit_scale_mean + 0.3 * torch.randn(shp, generator=g)
        elif (".norm" in k or k.startswith("norm_")) and k.endswith("weight"):
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            v = 0.02 * torch.randn(shp, generator=g)
        elif "cpb_mlp.0.weight" in k:
            v = 0.5 * torch.randn(shp, generator=g)
        elif "cpb_mlp.2.weight" in k:
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            b = 1.0 / math.sqrt(max(fan_in, 1))
            v = (torch.rand(shp, generator=g) * 2 - 1) * b
        out[k] = v.float()
    return out
