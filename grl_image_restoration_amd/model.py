"""``GRL`` -- drop-in replacement for the reference network at its own boundary.

Boundary: the reference engine does ``self.model = hydra.utils.instantiate(cfg.model)``
(engines/base.py:44) and then only calls ``model(x)``, ``parameters()``, ``state_dict()`` /
``load_state_dict(strict=True)`` and ``convert_checkpoint`` (tools/trainer.py:93-115).  This class
accepts the constructor arguments of ``models.networks.grl.GRL`` (grl.py:220-256, unknown YAML keys
swallowed like the reference does), owns parameters with the *same names and shapes*, and runs
the forward pass (grl.py:506-551) on MI355X through libgrl_hip.so.

There is no CPU / eager fallback for the hot path: calling the model on CPU tensors raises.
"""
import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import autograd as AG
from . import ops, tables
from .geometry import BlockGeo, block_schedule, pad_multiple, table_rows, to_2tuple

LOG2E = tables.LOG2E


def _pad32(n: int) -> int:
    return (n + 31) // 32 * 32


# ------------------------------------------------------------------------------------------------
# parameter containers (names mirror the reference so state_dicts are interchangeable)
# ------------------------------------------------------------------------------------------------
class _Affine(nn.Module):  # mixed_attn_block_efficient.py:23-34
    def __init__(self, nh):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((nh, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, nh, bias=False))


class _Body(nn.Module):
    def __init__(self, body):
        super().__init__()
        self.body = body


class _AnchorLinear(nn.Module):  # mixed_attn_block.py:714-725
    def __init__(self, cin, cout):
        super().__init__()
        self.reduction = nn.Linear(cin, cout, bias=True)


class _WindowAttn(nn.Module):
    def __init__(self, nh):
        super().__init__()
        self.attn_transform = _Affine(nh)


class _StripeAttn(nn.Module):
    def __init__(self, nh):
        super().__init__()
        self.attn_transform1 = _Affine(nh)
        self.attn_transform2 = _Affine(nh)


class _MixedAttention(nn.Module):  # mixed_attn_block_efficient.py:317-349
    def __init__(self, dim, nh_w, nh_s):
        super().__init__()
        self.qkv = _Body(nn.Linear(dim, dim * 3, bias=True))
        self.anchor = _Body(nn.ModuleList([_AnchorLinear(dim, dim // 2)]))
        self.window_attn = _WindowAttn(nh_w)
        self.stripe_attn = _StripeAttn(nh_s)
        self.proj = nn.Linear(dim, dim)


class _ChannelAttention(nn.Module):  # mixed_attn_block.py:948-963
    def __init__(self, c, reduction):
        super().__init__()
        self.attention = nn.Sequential(
            nn.AdaptiveAvgPool2d(1), nn.Conv2d(c, c // reduction, 1, padding=0), nn.ReLU(inplace=True),
            nn.Conv2d(c // reduction, c, 1, padding=0), nn.Sigmoid(),
        )


class _CAB(nn.Module):  # mixed_attn_block.py:970-979
    def __init__(self, c, compress_ratio=4, reduction=18):
        super().__init__()
        self.cab = nn.Sequential(
            nn.Conv2d(c, c // compress_ratio, 3, 1, 1), nn.GELU(), nn.Conv2d(c // compress_ratio, c, 3, 1, 1),
            _ChannelAttention(c, reduction),
        )


class _Mlp(nn.Module):  # swin_v1_block.py:15-35
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):  # mixed_attn_block_efficient.py:406-508
    def __init__(self, dim, nh_w, nh_s, mlp_ratio, local_connection):
        super().__init__()
        self.attn = _MixedAttention(dim, nh_w, nh_s)
        self.norm1 = nn.LayerNorm(dim)
        if local_connection:
            self.conv = _CAB(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.LayerNorm(dim)


class _Stage(nn.Module):  # grl.py:31-137
    def __init__(self, dim, depth, nh_w, nh_s, mlp_ratio, local_connection):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(dim, nh_w, nh_s, mlp_ratio, local_connection) for _ in range(depth)])
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1)


class _Up(nn.Module):  # upsample.py:6-50
    def __init__(self, mods):
        super().__init__()
        self.up = nn.Sequential(*mods)


_BUFFER_PREFIXES = ("table_", "index_", "mask_")


def _split_sites(spec: str) -> dict:
    """'stage_conv:x,after,last,cab0' -> {site: x_split}: 3 = activations and weights split (three MFMA terms), ':x' = 2 = only the
    activations (two terms; for sites where the rounding of x matters and that of W does not, tools/precision_sites.py combo)."""
    out = {}
    for item in spec.split(","):
        if item:
            name, _, mode = item.partition(":")
            out[name] = 2 if mode == "x" else 3
    return out


def predicted_split_count(var_desc, base_var: float, bar_rms: float, margin: float = 0.9) -> int:
    """How many blocks -- taken from the front of ``var_desc``, the per-block error variances in DESCENDING order -- must move to split
    operands so that the variance left (``base_var``: what remains with every block split, plus the variances of the blocks that stay on
    fp16 operands; independent rounding errors add in variance) stays within ``(margin * bar_rms)^2``.  The calibration verifies the
    prediction on the probe and raises the count until it passes (GRL._calibrated_plan)."""
    k, acc = len(var_desc), base_var
    for j in range(len(var_desc) - 1, -1, -1):        # blocks that may stay on fp16 operands, cheapest first
        if acc + var_desc[j] > (margin * bar_rms) ** 2:
            break
        acc += var_desc[j]
        k = j
    return k


class GRL(nn.Module):
    """MI355X-native GRL.  Constructor signature of models/networks/grl.py:220-256."""

    def __init__(
        self,
        img_size=64,
        in_channels=3,
        out_channels=None,
        embed_dim=96,
        upscale=2,
        img_range=1.0,
        upsampler="",
        depths=[6, 6, 6, 6, 6, 6],
        num_heads_window=[3, 3, 3, 3, 3, 3],
        num_heads_stripe=[3, 3, 3, 3, 3, 3],
        window_size=8,
        stripe_size=[8, 8],
        stripe_groups=[None, None],
        stripe_shift=False,
        mlp_ratio=4.0,
        qkv_bias=True,
        qkv_proj_type="linear",
        anchor_proj_type="avgpool",
        anchor_one_stage=True,
        anchor_window_down_factor=1,
        out_proj_type="linear",
        local_connection=False,
        drop_rate=0.0,
        attn_drop_rate=0.0,
        drop_path_rate=0.1,
        norm_layer=nn.LayerNorm,
        pretrained_window_size=[0, 0],
        pretrained_stripe_size=[0, 0],
        conv_type="1conv",
        init_method="n",
        fairscale_checkpoint=False,
        offload_to_cpu=False,
        euclidean_dist=False,
        precision="auto",  # not a reference option: "fast" | "high" | "auto" (operand precision of the HIP path, see below)
        **kwargs,  # name, double_window, stripe_square, separable_conv_act, use_buffer, ... (swallowed, grl.py:255)
    ):
        super().__init__()
        unsupported = []
        if qkv_proj_type != "linear":
            unsupported.append(f"qkv_proj_type={qkv_proj_type!r}")
        if anchor_proj_type != "avgpool" or not anchor_one_stage:
            unsupported.append(f"anchor_proj_type={anchor_proj_type!r}/anchor_one_stage={anchor_one_stage}")
        if out_proj_type != "linear":
            unsupported.append(f"out_proj_type={out_proj_type!r}")
        if conv_type != "1conv":
            unsupported.append(f"conv_type={conv_type!r}")
        if euclidean_dist:
            unsupported.append("euclidean_dist=True")
        if not qkv_bias:
            unsupported.append("qkv_bias=False")
        if list(pretrained_window_size) != [0, 0] or list(pretrained_stripe_size) != [0, 0]:
            unsupported.append("pretrained_*_size != [0, 0]")
        if init_method not in ("n", "r", "l", "w") and init_method.find("t") < 0:
            unsupported.append(f"init_method={init_method!r}")
        if unsupported:
            raise NotImplementedError(
                "grl_image_restoration_amd.GRL implements the configurations the reference ships "
                "(config/model/grl/*.yaml); not supported: " + ", ".join(unsupported)
            )
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.embed_dim, self.upscale, self.upsampler, self.img_range = embed_dim, upscale, upsampler, img_range
        self.depths = list(depths)
        self.num_heads_window, self.num_heads_stripe = list(num_heads_window), list(num_heads_stripe)
        self.window_size = to_2tuple(window_size)
        self.stripe_size, self.stripe_groups = list(stripe_size), list(stripe_groups)
        self.stripe_shift = stripe_shift
        self.mlp_ratio = mlp_ratio
        self.df = anchor_window_down_factor
        self.local_connection = local_connection
        self.res_scale = 0.1 if init_method == "r" else 1.0
        self.input_resolution = to_2tuple(img_size)
        self.pad_size = pad_multiple(self.window_size[0], self.stripe_size, self.stripe_groups, self.df)
        # Operand precision of the linear / conv contractions (attention always runs on fp16 operands):
        #   fast : fp16 operands (2^-11 relative rounding), fused block-tail / register-resident / streaming kernels; per-site
        #          exceptions on split operands: conv_first always, the convolutions named in `split_sites`, and the q / k /
        #          anchor projection of blocks whose logit scale exceeds `hiq_scale`
        #   high : split operands a = hi + lo, w = hi + lo (3 MFMA terms, ~22 mantissa bits) everywhere, fp32 intermediates
        #   auto : (resolved per weight set, _resolve_precision: the narrow models switch to `high` at checkpoint-like logit scales)
        #          measured on the fixtures (max |err| against the reference, bar 1e-3; tests/test_gpu_model.py):
        #          GRL-Base SR                 fast                                    2.1e-4 (clamp-scale checkpoints 6.5e-4)
        #          GRL-Base deblur (no upsampler: y = x + conv_last(body), no smoothing tail)
        #                                      fast 1.16e-3 -> + stage/after/last convs split 1.0e-3 -> + CAB conv1 split 8.2e-4
        #                                      (with the q/k projection split instead: 7.8e-4, but 4.0 instead of 5.6 MP/s)
        #          GRL-Small denoise           fast 1.14e-3 -> + stage/after/last convs split 7.2e-4
        #          GRL-Tiny                    high (fast + splits 6e-4 .. 8e-4, but 8e-3 at clamp scales)
        precision = os.environ.get("GRL_PRECISION", precision)
        if precision not in ("auto", "fast", "high"):
            raise ValueError(f"precision={precision!r}: expected 'auto', 'fast' or 'high'")
        narrow = embed_dim < 160 or not upsampler
        self._precision_arg, self._narrow = precision, narrow
        # (`auto` is resolved again whenever the packed weights are rebuilt, see _resolve_precision: it depends on the logit scales)
        self.precision = precision if precision != "auto" else ("high" if embed_dim < 100 else "fast")
        # fast mode: comma list of conv sites kept on split operands (see _plan); logit scale above which a block's q / k / anchor
        # planes come from the split-operand projection (0: always)
        # (a site may be given as `name:x` = only its activations split, two MFMA terms.  Tried as the default for the stage conv --
        # the per-operand emulation said its weights' rounding does not matter: deblur 384 8.3e-4 with both split, 8.0e-4 with x only,
        # 1.04e-3 with W only -- but measured on the GPU the fixtures moved from 8.2e-4 to 8.8e-4 (deblur) and from 7.2e-4 to 9.0e-4
        # (Small) for 4-5 % of a step: not kept, the margin to the 1e-3 bar is worth more.)
        self.split_sites = ("stage_conv,after,last,cab0" if embed_dim >= 160 else "stage_conv,after,last") if narrow else ""
        self.hiq_scale = float(os.environ.get("GRL_HIQ_SCALE", "50"))
        if embed_dim % 2 or any((embed_dim // 2) % h for h in self.num_heads_window + self.num_heads_stripe):
            raise ValueError("embed_dim/2 must be divisible by the number of heads")
        if max((embed_dim // 2) // h for h in self.num_heads_window + self.num_heads_stripe) > 32:
            raise NotImplementedError("head_dim > 32 is not supported by the gfx950 attention kernel")
        if in_channels == 3:
            mean = torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
        else:
            mean = torch.zeros(1, 1, 1, 1)
        self.register_buffer("_mean", mean, persistent=False)

        num_out_feats = 64
        self.conv_first = nn.Conv2d(in_channels, embed_dim, 3, 1, 1)
        self.norm_start = nn.LayerNorm(embed_dim)
        self.layers = nn.ModuleList(
            [
                _Stage(embed_dim, depths[i], num_heads_window[i], num_heads_stripe[i], mlp_ratio, local_connection)
                for i in range(len(depths))
            ]
        )
        self.norm_end = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        if upsampler == "pixelshuffle":
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_out_feats, 3, 1, 1), nn.LeakyReLU(inplace=True))
            m = []
            if (upscale & (upscale - 1)) == 0:
                for _ in range(int(math.log(upscale, 2))):
                    m += [nn.Conv2d(num_out_feats, 4 * num_out_feats, 3, 1, 1), nn.PixelShuffle(2)]
            elif upscale == 3:
                m += [nn.Conv2d(num_out_feats, 9 * num_out_feats, 3, 1, 1), nn.PixelShuffle(3)]
            else:
                raise ValueError(f"scale {upscale} is not supported. Supported scales: 2^n and 3.")
            self.upsample = _Up(m)
            self.conv_last = nn.Conv2d(num_out_feats, out_channels, 3, 1, 1)
        elif upsampler == "pixelshuffledirect":
            self.upsample = _Up([nn.Conv2d(embed_dim, (upscale**2) * out_channels, 3, 1, 1), nn.PixelShuffle(upscale)])
        elif upsampler == "nearest+conv":
            assert upscale == 4, "only support x4 now."
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_out_feats, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.conv_up1 = nn.Conv2d(num_out_feats, num_out_feats, 3, 1, 1)
            self.conv_up2 = nn.Conv2d(num_out_feats, num_out_feats, 3, 1, 1)
            self.conv_hr = nn.Conv2d(num_out_feats, num_out_feats, 3, 1, 1)
            self.conv_last = nn.Conv2d(num_out_feats, out_channels, 3, 1, 1)
        else:
            self.conv_last = nn.Conv2d(embed_dim, out_channels, 3, 1, 1)

        self.apply(self._init_weights)  # grl.py:381,455-469
        self._init_method_rescale(init_method)
        # stochastic depth decay rule (grl.py:299-300): block j of the whole network drops its branches with probability dpr[j]
        self.drop_path_rate = float(drop_path_rate)
        self._dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self._plan_cache: Dict = {}
        self._register_load_state_dict_pre_hook(self._drop_reference_buffers)
        # fires on the recursive path too (a parent module's load_state_dict, tools/trainer.py:108-111)
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.invalidate_plan())

    # ---- init / checkpoint contract ------------------------------------------------------------
    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _init_method_rescale(self, init_method):
        """TransformerStage._init_weights (grl.py:139-162) for the 'l' / 'w' / 't*' variants."""
        if not (init_method in ("l", "w") or init_method.find("t") >= 0):
            return
        for layer in self.layers:
            for n, m in layer.named_modules():
                if init_method == "w":
                    if isinstance(m, (nn.Linear, nn.Conv2d)) and n.find("cpb_mlp") < 0:
                        m.weight.data *= 0.1
                elif init_method == "l":
                    if isinstance(m, nn.LayerNorm):
                        nn.init.constant_(m.bias, 0)
                        nn.init.constant_(m.weight, 0)
                else:
                    scale = 0.1 ** (len(init_method) - 1) * int(init_method[-1])
                    if isinstance(m, nn.Linear) and n.find("cpb_mlp") < 0:
                        nn.init.trunc_normal_(m.weight, std=scale)
                    elif isinstance(m, nn.Conv2d):
                        m.weight.data *= 0.1

    @staticmethod
    def _drop_reference_buffers(state_dict, prefix, *args):
        """The reference keeps 13 table/index/mask buffers in its state_dict (grl.py:309-310) and never
        trusts them from disk (convert_checkpoint, grl.py:556-569).  This implementation needs none of
        them, so they are dropped on load to keep ``load_state_dict(strict=True)`` working."""
        for k in list(state_dict.keys()):
            if k[len(prefix):].startswith(_BUFFER_PREFIXES):
                state_dict.pop(k)

    def convert_checkpoint(self, state_dict):
        """grl.py:556-569: drop buffers that depend on the geometry from a Lightning checkpoint."""
        for k in list(state_dict.keys()):
            if (
                k.find("relative_coords_table") >= 0 or k.find("relative_position_index") >= 0
                or k.find("attn_mask") >= 0 or k.find("model.table_") >= 0 or k.find("model.index_") >= 0
                or k.find("model.mask_") >= 0
            ):
                state_dict.pop(k)
        return state_dict

    def invalidate_plan(self):
        """Drops the packed weights / tables (rebuilt lazily), every captured graph that points at them, the cached fp16
        training weights and the cached parameter list.  Called by the load_state_dict post-hook; in-place parameter updates
        through autograd-visible ops (optimizer steps, ``p.mul_()``, ``p.copy_()``) are caught by the version stamp in ``_plan``.
        NOT caught -- call this method yourself afterwards: updates through ``.data`` (``p.data.mul_()``, ``m.weight.data *= s``:
        torch does not bump the version counter for them) and parameters REPLACED by new tensors
        (``load_state_dict(assign=True)``, ``m.weight = nn.Parameter(...)``)."""
        self._plan_cache = {}
        self._plist = None
        if getattr(self, "_graphs", None):
            self._graphs = {}
        AG.forget_parameters(self)

    def _param_stamp(self):
        """Changes whenever a parameter is modified in place or replaced (torch bumps ``_version`` on every in-place op)."""
        plist = getattr(self, "_plist", None)
        if plist is None:
            plist = self._plist = list(self.parameters())
        return sum(p._version for p in plist)

    def train(self, mode: bool = True):
        if mode:
            self.invalidate_plan()
        return super().train(mode)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    # ---- weight packing ------------------------------------------------------------------------
    def _pack_block(self, blk: _Block, geo: BlockGeo, dev, hi: Optional[bool] = None, cab_split: Optional[bool] = None,
                    allow_hiq: bool = True) -> dict:
        """Packed weights / tables of one block.  ``hi``: this block runs on split operands (None: as the model's precision says;
        `auto` may choose it block by block, see _calibrated_plan); ``cab_split``: the CAB convolutions of a split block on split
        operands too (None: _high_cab_fp16 decides)."""
        C = self.embed_dim
        CP = _pad32(C)
        nh_w, nh_s = geo.nh_w, geo.nh_s
        d_w, d_s = C // 2 // nh_w, C // 2 // nh_s
        a = blk.attn
        f32 = dict(dtype=torch.float32, device=dev)

        def aff(m: _Affine):
            return tables.clamped_scale(m.logit_scale).to(dev)

        sc_w, sc_1, sc_2 = aff(a.window_attn.attn_transform), aff(a.stripe_attn.attn_transform1), aff(a.stripe_attn.attn_transform2)
        # K planes carry 1.0 in the spare head-dim slot 31 (negative gscale): partner of the attention kernel's running
        # softmax offset, which lives in slot 31 of its Q fragments (include/grl_hip.h)
        one_w, one_s = d_w <= 30, d_s <= 30
        if hi is None:
            hi = self.precision == "high"

        # --- QKV: one 32-wide slot per (branch, q|k|v, head); v slots carry a constant-1 column ---
        W = a.qkv.body.weight.detach().float()
        b = a.qkv.body.bias.detach().float()
        G = 3 * nh_w + 3 * nh_s
        Wp = torch.zeros(G * 32, CP, **f32)
        bp = torch.zeros(G * 32, **f32)
        gs = torch.zeros(G, **f32)
        for br, (nh, d, base_o, base_g) in enumerate(((nh_w, d_w, 0, 0), (nh_s, d_s, 3 * C // 2, 3 * nh_w))):
            for which in range(3):
                for h in range(nh):
                    g = base_g + which * nh + h
                    o0 = base_o + which * (C // 2) + h * d
                    Wp[g * 32 : g * 32 + d, :C] = W[o0 : o0 + d]
                    bp[g * 32 : g * 32 + d] = b[o0 : o0 + d]
                    if which == 2 and d < 32:
                        bp[g * 32 + d] = 1.0
                    if which == 0:
                        gs[g] = (sc_w[h] if br == 0 else sc_2[h]) * LOG2E
                    elif which == 1:
                        one = one_w if br == 0 else one_s
                        gs[g] = (1.0 if br == 0 else sc_1[h] * LOG2E) * (-1.0 if one else 1.0)
        G16 = ops.GEMM_DTYPE
        pk = dict(hi=hi, hi_c=False, qkv_b=bp, qkv_gs=gs, one_w=one_w, one_s=one_s, floor_w=tables.lazy_floor(sc_w),
                  floor_a2w=tables.lazy_floor(sc_1), floor_w2a=tables.lazy_floor(sc_2))
        pk["qkv_w"] = ops.split3_weight(Wp) if hi else Wp.to(G16)
        # fast mode, logit scales beyond GRL_HIQ_SCALE (trained checkpoints sit at the clamp, 100): the q / k / anchor planes come from
        # the split-operand projection -- at scale 100 the fp16 rounding of x and W in this one GEMM is the largest single
        # contribution to the output error (tools/precision_sites.py: rms 8.9e-5 of 1.6e-4), amplified by the scale itself
        hiq = (not hi) and allow_hiq and float(max(sc_w.max(), sc_1.max(), sc_2.max())) > self.hiq_scale
        if hiq:
            pk.update(hiq=True, qkv_w3=ops.split3_weight(Wp))
            pk["qkv_w3r"] = ops.pack_linear_split(pk["qkv_w3"])
        if not hi and CP in (64, 128, 192):  # one-pass streaming QKV kernel (csrc/qkv.hip)
            pk.update(qkv_blob=ops.pack_qkv(Wp, bp, gs), qkv_slots=G)

        # --- anchor projection (avg-pool fused in the kernel) ---
        Wa = a.anchor.body[0].reduction.weight.detach().float()
        ba = a.anchor.body[0].reduction.bias.detach().float()
        Wap = torch.zeros(nh_s * 32, CP, **f32)
        bap = torch.zeros(nh_s * 32, **f32)
        for h in range(nh_s):
            Wap[h * 32 : h * 32 + d_s, :C] = Wa[h * d_s : (h + 1) * d_s]
            bap[h * 32 : h * 32 + d_s] = ba[h * d_s : (h + 1) * d_s]
        pk.update(anc_w=ops.split3_weight(Wap) if hi else Wap.to(G16), anc_b=bap,
                  anc_gs=torch.full((nh_s,), -1.0 if one_s else 1.0, **f32))
        if hiq:
            pk["anc_w3"] = ops.split3_weight(Wap)
        if not hi and CP in (64, 128, 192):   # q/k/v + 2x2-pooled anchors in one pass over x (csrc/qkv_anchor.hip)
            pk.update(qa_blob=ops.pack_qkv_anchor(Wp, bp, gs, Wap, bap, pk["anc_gs"]), qa_slots=(G, nh_s))
            if hiq and CP == 192 and (nh_w, nh_s) == (3, 3):   # the same pass on split operands (qkv_split_kernel, round 4)
                pk["qa_lo"] = ops.pack_qkv_anchor_lo(torch.cat([Wp, Wap]), torch.cat([gs, pk["anc_gs"]]))

        # --- output projection over the slotted attention output + norm1 ---
        Wo = a.proj.weight.detach().float()
        KA = (nh_w + nh_s) * 32
        Wop = torch.zeros(CP, KA, **f32)
        for h in range(nh_w):
            Wop[:C, h * 32 : h * 32 + d_w] = Wo[:, h * d_w : (h + 1) * d_w]
        for h in range(nh_s):
            Wop[:C, (nh_w + h) * 32 : (nh_w + h) * 32 + d_s] = Wo[:, C // 2 + h * d_s : C // 2 + (h + 1) * d_s]

        def padv(v, n=CP):
            out = torch.zeros(n, **f32)
            out[: v.numel()] = v.detach().float()
            return out

        pk.update(proj_w=ops.split3_weight(Wop) if hi else Wop.to(G16), proj_b=padv(a.proj.bias), n1_g=padv(blk.norm1.weight),
                  n1_b=padv(blk.norm1.bias))

        # --- MLP + norm2 ---
        Hd = blk.mlp.fc1.weight.shape[0]
        HP = _pad32(Hd)
        W1 = torch.zeros(HP, CP, **f32)
        W1[:Hd, :C] = blk.mlp.fc1.weight.detach().float()
        W2 = torch.zeros(CP, HP, **f32)
        W2[:C, :Hd] = blk.mlp.fc2.weight.detach().float()
        pk.update(fc1_w=ops.split3_weight(W1) if hi else W1.to(G16), fc1_b=padv(blk.mlp.fc1.bias, HP),
                  fc2_w=ops.split3_weight(W2) if hi else W2.to(G16),
                  fc2_b=padv(blk.mlp.fc2.bias), n2_g=padv(blk.norm2.weight), n2_b=padv(blk.norm2.bias))
        if hi:   # register images of the split weights for the weights-stationary kernel (csrc/linear_split.hip; None: generic kernel)
            for k_ in ("qkv", "anc", "proj", "fc1", "fc2"):
                pk[k_ + "_wr"] = ops.pack_linear_split(pk[k_ + "_w"])
        if not hi and CP in (64, 128, 192) and KA == CP and self.local_connection:   # + proj/norm1/CAB in front: one kernel per block tail
            pk["proj_blob"] = ops.pack_proj(Wop)
            # weights stationary in registers (csrc/tail_regs.hip, round 4): 255 against 290 us per 4 tiles; GRL_TAIL_REGS=0: streaming kernel
            if CP == 192 and HP == 384 and C > 160 and os.environ.get("GRL_TAIL_REGS", "1") != "0":
                pk["tail_rblob"] = ops.pack_tail_regs(Wop, blk.mlp.fc1.weight.to(dev), blk.mlp.fc1.bias.to(dev), blk.mlp.fc2.weight.to(dev))
        if not hi and CP in (64, 128, 192):  # fused fc1 -> GELU -> fc2 -> norm2 -> residual kernel (csrc/mlp.hip)
            pk.update(mlp_blob=ops.pack_mlp(blk.mlp.fc1.weight.to(dev), blk.mlp.fc1.bias.to(dev), blk.mlp.fc2.weight.to(dev), CP, HP),
                      mlp_hp=HP)

        # --- relative-position bias tables in the kernel's exp2 domain ---
        # A (query window, key window) pair that is not 32-aligned as it stands but is so with the image axes swapped -- the
        # 128x64 stripes / 32x16 anchors of every other block of the dn geometry -- is launched on the transposed view of its
        # grids (GrlTokenGrid.transposed) with the transposed table: row-streaming kernel instead of the generic one.
        def table(m: _Affine, win, df, q_win, k_win, q_sh, k_sh, masked, d):
            coords = tables.coords_table(win, df, device=dev)
            bias = tables.bias_rows(m.cpb_mlp[0].weight.to(dev), m.cpb_mlp[0].bias.to(dev), m.cpb_mlp[2].weight.to(dev), coords)
            sw = lambda t: (t[1], t[0])
            tr = (not hi and os.environ.get("GRL_ATTN_TRANSPOSE", "1") != "0"
                  and not ops.attention_rows_ok(q_win, k_win, q_sh, k_sh, masked, d)
                  and ops.attention_rows_ok(sw(q_win), sw(k_win), sw(q_sh), sw(k_sh), masked, d))
            if tr:
                bias = ops.transpose_table(bias, q_win, k_win)
            return tables.kernel_table(bias), tr

        wsh = (geo.window_shift, geo.window_shift)
        pk["tab_w"], pk["tr_w"] = table(a.window_attn.attn_transform, geo.window, 1, geo.window, geo.window, wsh, wsh, geo.window_shift > 0, d_w)
        pk["tab_a2w"], pk["tr_a2w"] = table(a.stripe_attn.attn_transform1, geo.stripe, geo.df, geo.anchor_stripe, geo.stripe,
                                            geo.anchor_shift_size, geo.stripe_shift_size, geo.stripe_shift, d_s)
        pk["tab_w2a"], pk["tr_w2a"] = table(a.stripe_attn.attn_transform2, geo.stripe, geo.df, geo.stripe, geo.anchor_stripe,
                                            geo.stripe_shift_size, geo.anchor_shift_size, geo.stripe_shift, d_s)
        pk.update(ceil_w=tables.lazy_ceil(sc_w, pk["tab_w"]), ceil_a2w=tables.lazy_ceil(sc_1, pk["tab_a2w"]),
                  ceil_w2a=tables.lazy_ceil(sc_2, pk["tab_w2a"]))
        assert pk["tab_w"].shape[1] == (table_rows(geo.window, geo.window) + 3) // 4 * 4
        assert pk["tab_a2w"].shape[1] == (table_rows(geo.anchor_stripe, geo.stripe) + 3) // 4 * 4

        # --- CAB: conv3x3 C->C/4 (GELU), conv3x3 C/4->C, squeeze-excite gate (mixed_attn_block.py:948-983) ---
        if self.local_connection:
            c0, c2 = blk.conv.cab[0], blk.conv.cab[2]
            se = blk.conv.cab[3].attention
            Cm = c0.weight.shape[0]
            CmO, CmI = (Cm + 15) // 16 * 16, _pad32(Cm)  # conv1 writes CmO channels of a zeroed CmI-wide matrix
            # (the CAB convs of an auto-resolved `high` Base-width model stay on fp16 operands)
            hi_c = hi and (cab_split if cab_split is not None else not self._high_cab_fp16())
            pk["hi_c"] = hi_c
            if hi_c:
                CmO = CmI   # fp32 mid tensor written by the plain store path: every channel of its row comes from the conv
            sp = 3 if hi_c else 1
            sites = _split_sites(os.environ.get("GRL_SPLIT_SITES", self.split_sites))
            # the CAB's first conv on split operands: everything-split `high`; the fast path's per-site choice (see _plan); fp16 in the
            # auto-resolved `high` of a Base-width model (_high_cab_fp16: every other site is split there, so this one can afford it)
            pk["cab0_split"] = 3 if hi_c else (1 if hi else sites.get("cab0", 1))
            pk.update(
                cab0_w=ops.pack_conv_weight(c0.weight.to(dev), CP, CmO, split=pk["cab0_split"]), cab0_b=ops.pack_conv_bias(c0.bias.to(dev), CmO),
                cab2_w=ops.pack_conv_weight(c2.weight.to(dev), CmI, CP, split=sp), cab2_b=ops.pack_conv_bias(c2.bias.to(dev), CP),
                cab_mid=CmI,
                se1_w=se[1].weight.detach().float().reshape(se[1].weight.shape[0], C).to(dev).clone(),   # copies: a plan never aliases
                se1_b=se[1].bias.detach().float().to(dev).clone(),                                        # the live parameters
                se3_w=se[3].weight.detach().float().reshape(C, -1).to(dev).clone(),
                se3_b=se[3].bias.detach().float().to(dev).clone(),
            )
            if not hi_c and CP == 192 and Cm <= 48 and CmI >= 56 and os.environ.get("GRL_CAB_CONV2", "1") != "0":
                pk["cab2_blob"], pk["cab2_bias"] = ops.pack_cab_conv2(c2.weight.to(dev), c2.bias.to(dev))   # csrc/cab_conv2.hip
        return pk

    def _high_cab_fp16(self) -> bool:
        """In a `high` that `auto` chose for a Base-width model (deblur / denoise at checkpoint-like scales) the two CAB convolutions
        stay on fp16 operands with the fast path's kernels: emulated per operand on the clamp-scale deblur fixture
        (tools/precision_sites.py only ...) they are the least sensitive sites of the net -- conv1 3.4e-4 / 2.5e-4 (weights /
        input alone), conv2 1.9e-4 / 2.3e-4, against 1e-3 for the stage conv's weights alone -- and on split operands they were 41
        of a 188 ms forward.  An explicit precision='high' (and GRL-Tiny) keeps every contraction split.  GRL_HIGH_CAB=split|fp16."""
        mode = os.environ.get("GRL_HIGH_CAB", "")
        if mode in ("split", "fp16"):
            return mode == "fp16"
        return self._precision_arg == "auto" and self.embed_dim >= 160

    def _resolve_precision(self) -> str:
        """precision='auto' for the weights the module holds NOW (called when a plan is built, i.e. after every weight change).
        GRL-Tiny: high.  The narrow / same-resolution models (GRL-Small, anything without the smoothing upsampler tail: denoise,
        deblur) hold the 1e-3 bar on fp16 operands only at random-init logit scales (7.2e-4 / 8.2e-4); with checkpoint-like
        scales the round-4 clamp-scale fixtures measure 2.5e-3 (Base deblur) and worse (Small), against 5.9e-4 in the `high` chosen here
        (5.0e-5 with every contraction split, _high_cab_fp16) -- the
        cosine logits are multiplied by up to 100 and these nets have no tail that averages the error out.  So above
        GRL_NARROW_HIGH_SCALE (25: random init draws 5 .. 20) they run on split operands throughout.  GRL-Base SR stays fast at
        every scale (blocks above GRL_HIQ_SCALE take the split-operand q / k / anchor projection: 8.1e-4 at the clamp)."""
        if self._precision_arg != "auto":
            return self._precision_arg
        if self.embed_dim < 100:
            return "high"
        if self._narrow:
            smax = 0.0
            for layer in self.layers:
                for blk in layer.blocks:
                    a = blk.attn
                    for t in (a.window_attn.attn_transform, a.stripe_attn.attn_transform1, a.stripe_attn.attn_transform2):
                        smax = max(smax, float(tables.clamped_scale(t.logit_scale).max()))
            if smax > float(os.environ.get("GRL_NARROW_HIGH_SCALE", "25")):
                # round 6: not `high` throughout any more -- the blocks are chosen by measurement (_calibrated_plan), everything split
                # only if the probe asks for it; GRL_CALIBRATE=0 restores the blanket rule
                if os.environ.get("GRL_CALIBRATE", "1") == "0":
                    return "high"
                self._calibrate_narrow = True
        return "fast"

    def _plan(self, x_size, dev):
        key = (tuple(x_size), str(dev), self._param_stamp())
        plan = self._plan_cache.get(key)
        if plan is not None:
            return plan
        self._calibrate_narrow = False
        self.precision = self._resolve_precision()
        self.calibration = None
        if (self._precision_arg == "auto" and self.precision == "fast" and (not self._narrow or self._calibrate_narrow)
                and os.environ.get("GRL_CALIBRATE", "1") != "0"):
            plan = self._calibrated_plan(x_size, dev, force=self._calibrate_narrow)
        else:
            plan = self._build_plan(x_size, dev, self.precision)
        self._plan_cache = {key: plan}  # one geometry at a time keeps memory bounded (captured graphs hold their own plan)
        return plan

    def _build_plan(self, x_size, dev, precision: str, cab_split: Optional[bool] = None, allow_hiq: bool = True):
        """Packed weights / tables of the whole network for one input size, every block in ``precision`` ('fast' | 'high').
        ``allow_hiq=False``: no block takes the split-operand q / k / anchor projection (plain fp16 operands everywhere)."""
        hi = precision == "high"
        sp = 3 if hi else 1
        C, CP = self.embed_dim, _pad32(self.embed_dim)
        f32 = dict(dtype=torch.float32, device=dev)
        sched = block_schedule(self.depths, self.num_heads_window, self.num_heads_stripe, self.window_size,
                               self.stripe_size, self.stripe_groups, self.stripe_shift, self.df, x_size)

        def padv(v):
            out = torch.zeros(CP, **f32)
            out[:C] = v.detach().float()
            return out

        # fast mode: convolutions named in GRL_SPLIT_SITES (stage_conv, after, last) still run on split operands -- per-site
        # precision for the models whose fast-mode error sits at the 1e-3 limit (tools/precision_sites.py)
        sites = _split_sites(os.environ.get("GRL_SPLIT_SITES", self.split_sites))
        xs = {k: (3 if hi else sites.get(k, 1)) for k in ("stage_conv", "after", "last")}

        def pconv(conv, cin_pad, cout_pad, r=0, cg=0, site=None):
            return (ops.pack_conv_weight(conv.weight.to(dev), cin_pad, cout_pad, r, cg, split=xs.get(site, sp)),
                    ops.pack_conv_bias(conv.bias.to(dev), cout_pad, r, cg))

        with torch.no_grad():
            stages = []
            for si, stage in enumerate(self.layers):
                blocks = [self._pack_block(blk, sched[si][bi], dev, hi, cab_split, allow_hiq) for bi, blk in enumerate(stage.blocks)]
                cw, cb = pconv(stage.conv, CP, CP, site="stage_conv")
                stages.append(dict(blocks=blocks, conv_w=cw, conv_b=cb))
            plan = dict(
                sched=sched, stages=stages, split=sp, xs=xs,
                ns_g=padv(self.norm_start.weight), ns_b=padv(self.norm_start.bias),
                ne_g=padv(self.norm_end.weight), ne_b=padv(self.norm_end.bias),
                # conv_first always runs on split operands: K = 27, the cost is nil, and its operand rounding alone is 4e-4 of the
                # 1e-3 budget of a clamp-scale checkpoint (tools/precision_sites.py base_sr4_ckpt_256_hiscale)
                first=(ops.pack_conv_weight(self.conv_first.weight.to(dev), _pad32(self.in_channels), CP, split=3),
                       ops.pack_conv_bias(self.conv_first.bias.to(dev), CP)),
                after=pconv(self.conv_after_body, CP, CP, site="after"),
            )
            out_p = (self.out_channels + 15) // 16 * 16
            if self.upsampler == "pixelshuffle":
                plan["cbu"] = pconv(self.conv_before_upsample[0], CP, 64)
                r = 3 if self.upscale == 3 else 2
                plan["ups"] = [pconv(m, 64, (64 * r * r + 15) // 16 * 16, r, 64) for m in self.upsample.up if isinstance(m, nn.Conv2d)]
                plan["ups_r"] = r
                plan["last"] = pconv(self.conv_last, 64, out_p)
            elif self.upsampler == "pixelshuffledirect":
                r = self.upscale
                cg = (self.out_channels + 3) // 4 * 4
                plan["upd"] = pconv(self.upsample.up[0], CP, (cg * r * r + 15) // 16 * 16, r, cg)
                plan["upd_cg"] = cg
            elif self.upsampler == "nearest+conv":
                plan["cbu"] = pconv(self.conv_before_upsample[0], CP, 64)
                plan["up1"], plan["up2"] = pconv(self.conv_up1, 64, 64), pconv(self.conv_up2, 64, 64)
                plan["hr"], plan["last"] = pconv(self.conv_hr, 64, 64), pconv(self.conv_last, 64, out_p)
            else:
                plan["last"] = pconv(self.conv_last, CP, out_p, site="last")
        return plan

    # ---- precision `auto` for the wide SR models at checkpoint-like logit scales: chosen block by block, by measurement ----------
    def _probe_input(self, H: int, W: int, dev):
        """A fixed smooth probe image in [0, 1]: box-blurred uniform noise at the output resolution, down-sampled by the model's
        scale (the statistics of a low-quality SR input: SURVEY 8(d)'s synthetic recipe, own seed).  How far fp16 operands move the
        output depends on the input as well as on the weights -- on a high-contrast probe (coarse random blobs + fine noise) a
        clamp-scale random-weight network is 50x more sensitive than on smooth ones -- so the probe has to look like what the
        network restores."""
        g = torch.Generator().manual_seed(20240607)
        s = max(int(self.upscale), 1) if self.upsampler else 1
        hr = F.avg_pool2d(torch.rand(1, self.in_channels, H * s + 4, W * s + 4, generator=g), 5, 1)
        x = F.avg_pool2d(hr, s) if s > 1 else hr
        if not self.upsampler:                # same-resolution tasks: the input may be a NOISY image (denoising, sigma 25 / 255:
            x = x + (25.0 / 255.0) * torch.randn(x.shape, generator=g)    # data/datasets/restoration_dn.py:126-144) -- the harder case
        return x.contiguous().to(dev)

    def _calibrated_plan(self, x_size, dev, force: bool = False):
        """GRL-Base SR on fp16 operands sits AT the 1e-3 parity bar when the logit scales are checkpoint-like (clamped at 100), weight
        set by weight set: 7.7e-4 / 5.9e-4 / 1.9e-3 on three draws (round 5, float64 reference), with no single site to blame (q.k
        rounding 38 % of the variance, fc1 17 %, CAB conv2 11 %, fc2 9 %).  So `auto` MEASURES the weights it holds: a fixed probe
        image runs through the all-split network (the reference here: 5e-6 from the float64 truth) and through the fp16-operand
        one; if the difference exceeds the calibration bars, blocks move to split operands -- the ones whose fp16 rounding costs
        the most first (error of the network with ONLY that block on fp16 operands) -- until the probe passes.  One-time cost per
        weight set and input size: 3 plan builds + ~(blocks + 8) probe forwards.  Bars: rms <= GRL_CAL_RMS (1.3e-4: the maximum
        over the 3 M outputs of a 256x256 tile sits 5.5-6.6 rms above zero) and max <= GRL_CAL_MAX (8.5e-4) on the probe."""
        fast = self._build_plan(x_size, dev, "fast")
        blocks = [(si, bi) for si, st in enumerate(fast["stages"]) for bi in range(len(st["blocks"]))]
        info = dict(blocks=len(blocks), split=0)
        self.calibration = info
        if not force and not any(fast["stages"][si]["blocks"][bi].get("hiq") for si, bi in blocks):
            return fast                       # random-init-like scales: fp16 operands hold 2e-4 (fixtures); nothing to measure
        bar_rms = float(os.environ.get("GRL_CAL_RMS", "1.3e-4"))
        bar_max = float(os.environ.get("GRL_CAL_MAX", "8.5e-4"))
        H, W = x_size
        # the probe is a crop when the image is large and the block geometry does not depend on the image size
        ph, pw = min(H, 256 // self.pad_size * self.pad_size or self.pad_size), min(W, 256 // self.pad_size * self.pad_size or self.pad_size)
        if (ph, pw) != (H, W):
            small = block_schedule(self.depths, self.num_heads_window, self.num_heads_stripe, self.window_size, self.stripe_size,
                                   self.stripe_groups, self.stripe_shift, self.df, (ph, pw))
            if small != fast["sched"]:
                ph, pw = H, W
        x = self._probe_input(ph, pw, dev)
        with torch.no_grad():
            ref_plan = self._build_plan(x_size, dev, "high", cab_split=True)
            y_ref = self._forward_eager(x, ref_plan).double()
            del ref_plan
            hi_plan = self._build_plan(x_size, dev, "high")       # its BLOCKS are what a split block runs (CAB as _high_cab_fp16 says)

            def mixed(split_set):
                plan = dict(fast)
                plan["stages"] = [dict(st, blocks=[(hi_plan if (si, bi) in split_set else fast)["stages"][si]["blocks"][bi]
                                                   for bi in range(len(st["blocks"]))]) for si, st in enumerate(fast["stages"])]
                return plan

            def err(plan):
                d = self._forward_eager(x, plan).double() - y_ref
                return float(d.abs().max()), float(d.pow(2).mean().sqrt())

            ok = lambda e: e[0] <= bar_max and e[1] <= bar_rms
            n_hiq = sum(bool(fast["stages"][si]["blocks"][bi].get("hiq")) for si, bi in blocks)
            info.update(bar_max=bar_max, bar_rms=bar_rms, probe=(ph, pw), qkv_split_blocks=n_hiq)
            if n_hiq:
                # cheapest first: plain fp16 operands in EVERY projection (the split q / k / anchor projection of blocks above
                # hiq_scale costs 206 against 150 us per 4 tiles and block) -- kept only where the measurement asks for it
                plain = self._build_plan(x_size, dev, "fast", allow_hiq=False)
                e_plain = err(plain)
                info.update(plain_max=e_plain[0], plain_rms=e_plain[1])
                if ok(e_plain):
                    info.update(probe_max=e_plain[0], probe_rms=e_plain[1], qkv_split_blocks=0)
                    return plain
                del plain
            e_fast = err(fast)
            info.update(fast_max=e_fast[0], fast_rms=e_fast[1])
            if ok(e_fast):
                info.update(probe_max=e_fast[0], probe_rms=e_fast[1])
                return fast
            every = frozenset(blocks)
            e_all = err(mixed(every))
            info.update(all_split_max=e_all[0], all_split_rms=e_all[1])
            if not ok(e_all):                 # the fp16 convolutions around the blocks alone exceed the bars: everything split
                self.precision = "high"
                info.update(split=len(blocks), probe_max=0.0, probe_rms=0.0, everything=True)
                return self._build_plan(x_size, dev, "high", cab_split=True)
            # cost of each block's fp16 operands: the network with ONLY that block fast
            var = {b: max(err(mixed(every - {b}))[1] ** 2 - e_all[1] ** 2, 0.0) for b in blocks}
            order = sorted(blocks, key=lambda b: -var[b])
            # predicted number of blocks (variances add), then verified by measurement and raised until the probe passes
            k = predicted_split_count([var[b] for b in order], e_all[1] ** 2, bar_rms)
            while True:
                e = err(mixed(frozenset(order[:k])))
                if ok(e) or k >= len(order):
                    break
                k = min(len(order), k + max(1, len(order) // 16))
            info.update(split=k, probe_max=e[0], probe_rms=e[1], split_blocks=sorted(order[:k]))
            self.precision = f"mixed({k}/{len(order)} blocks split)"
            plan = mixed(frozenset(order[:k]))
        return plan

    # ---- forward -------------------------------------------------------------------------------
    def check_image_size(self, x):
        """grl.py:479-489."""
        _, _, h, w = x.size()
        ph = (self.pad_size - h % self.pad_size) % self.pad_size
        pw = (self.pad_size - w % self.pad_size) % self.pad_size
        try:
            x = F.pad(x, (0, pw, 0, ph), "reflect")
        except BaseException:
            x = F.pad(x, (0, pw, 0, ph), "constant")
        return x

    def _cab(self, r, pk, B, H, W, CP):
        """CAB branch (mixed_attn_block.py:948-983): returns the un-gated conv output and the per-image squeeze-excite
        gate; the gate is applied inside the proj+norm1 epilogue.  fast: fp16 intermediates; high: fp32 + split operands."""
        hi = pk["hi_c"]
        sp, dt = (3, torch.float32) if hi else (1, ops.GEMM_DTYPE)
        mid = ops.empty(B * H * W, pk["cab_mid"], dtype=dt, device=r.device)  # fast: pad channels zero-filled by the conv store
        ops.conv3x3(r, pk["cab0_w"], pk["cab0_b"], B, H, W, act=1, out=mid, x_split=pk["cab0_split"])
        # (GRL_SE_FOLD=1: conv2 + pool + squeeze-excite gate in one launch, the gate by the last workgroup of each image.  Measured
        # SLOWER in the two-stream bench, 88.1 against 82.6 ms/step: the serial tail of one workgroup per image holds the whole
        # launch, while the separate 10-us se_kernel hides behind the other tile group's kernels.  Kept as an option, off.)
        if "cab2_blob" in pk and os.environ.get("GRL_SE_FOLD", "0") == "1":
            return ops.cab_conv2(mid, pk["cab2_blob"], pk["cab2_bias"], B, H, W,
                                 se=(pk["se1_w"], pk["se1_b"], pk["se3_w"], pk["se3_b"], self.embed_dim))
        elif "cab2_blob" in pk:
            raw, pool = ops.cab_conv2(mid, pk["cab2_blob"], pk["cab2_bias"], B, H, W)
        else:
            raw, pool = ops.conv3x3(mid, pk["cab2_w"], pk["cab2_b"], B, H, W, want_pool=True, out_dtype=dt, x_split=sp)
        gate = ops.se_scale(pool, B, CP, self.embed_dim, H * W, pk["se1_w"], pk["se1_b"], pk["se3_w"], pk["se3_b"])
        return raw, gate

    def _attention(self, qkv, anc, att, pk, geo: BlockGeo, B, H, W, lse=None, qkv_lo=None, anc_lo=None):
        """The three attention launches of a block on head planes: window (efficient.py:128-165), anchors -> stripe tokens
        and stripe tokens -> anchors (:215-270).  ``att``: [M, (nh_w+nh_s)*32] output (fp16 or fp32).  ``qkv_lo`` / ``anc_lo``:
        rounding-residual twins of the planes (precision 'high': split-precision attention operands)."""
        C = self.embed_dim
        nh_w, nh_s, df = geo.nh_w, geo.nh_s, geo.df
        d_w, d_s = C // 2 // nh_w, C // 2 // nh_s
        Ha, Wa = H // df, W // df
        y = ops.empty(nh_s, B * Ha * Wa, 32, dtype=ops.PLANE_DTYPE, device=att.device)
        split = qkv_lo is not None
        y_lo = ops.empty_like(y) if split else None

        ws, sh = geo.window, geo.window_shift
        TG = ops.TokenGrid
        ls = lse if lse is not None else (None, None, None)
        tr = lambda key, *gs: tuple(g.T() for g in gs) if pk.get(key) else gs    # transposed view where the plan chose it
        ops.attention(
            *tr("tr_w", TG(qkv, 0, H, W, ws[0], ws[1], sh, sh), TG(qkv, nh_w, H, W, ws[0], ws[1], sh, sh),
                TG(qkv, 2 * nh_w, H, W, ws[0], ws[1], sh, sh), TG(att, 0, H, W, ws[0], ws[1], sh, sh)),
            B=B, nh=nh_w, table=pk["tab_w"], masked=sh > 0,
            ones_col=d_w if d_w < 32 else -1, head_dim=d_w, k_one31=pk["one_w"], lazy_floor=pk["floor_w"], lse=ls[0], lazy_ceil=pk.get("ceil_w"),
            q_lo=qkv_lo, k_lo=qkv_lo, v_lo=qkv_lo,
        )
        s0 = 3 * nh_w
        st, ss = geo.stripe, geo.stripe_shift_size
        ast, ass = geo.anchor_stripe, geo.anchor_shift_size
        g_q = TG(qkv, s0, H, W, st[0], st[1], ss[0], ss[1])
        g_k = TG(qkv, s0 + nh_s, H, W, st[0], st[1], ss[0], ss[1])
        g_v = TG(qkv, s0 + 2 * nh_s, H, W, st[0], st[1], ss[0], ss[1])
        g_a = TG(anc, 0, Ha, Wa, ast[0], ast[1], ass[0], ass[1])
        g_y = TG(y, 0, Ha, Wa, ast[0], ast[1], ass[0], ass[1])
        oc = d_s if d_s < 32 else -1
        ops.attention(*tr("tr_a2w", g_a, g_k, g_v, g_y), B=B, nh=nh_s, table=pk["tab_a2w"], masked=geo.stripe_shift,
                      ones_col=oc, head_dim=d_s, k_one31=pk["one_s"], lazy_floor=pk["floor_a2w"], lse=ls[1], lazy_ceil=pk.get("ceil_a2w"),
                      q_lo=anc_lo, k_lo=qkv_lo, v_lo=qkv_lo, o_lo=y_lo)
        ops.attention(*tr("tr_w2a", g_q, g_a, g_y, TG(att, nh_w, H, W, st[0], st[1], ss[0], ss[1])), B=B, nh=nh_s, table=pk["tab_w2a"],
                      masked=geo.stripe_shift, ones_col=oc, head_dim=d_s, k_one31=pk["one_s"],
                      lazy_floor=pk["floor_w2a"], lse=ls[2], q_lo=qkv_lo, k_lo=anc_lo, v_lo=y_lo, lazy_ceil=pk.get("ceil_w2a"))
        return y

    def _block(self, r, pk, geo: BlockGeo, B, H, W):
        C, CP = self.embed_dim, r.shape[1]
        M = B * H * W
        nh_w, nh_s, df = geo.nh_w, geo.nh_s, geo.df
        dev = r.device
        if pk["hi"]:
            return self._block_high(r, pk, geo, B, H, W)
        # q/k/v, anchors and the anchor-side values live as head planes [slot][token][32]: a key tile of 32
        # consecutive tokens is 2 KB contiguous for the attention kernel's staging loads
        one_pass = "qa_blob" in pk and df == 2 and H % 2 == 0 and W % 64 == 0 and os.environ.get("GRL_QKV_ANCHOR", "1") != "0"
        if pk.get("hiq") and one_pass and "qa_lo" in pk and os.environ.get("GRL_QKV_SPLIT", "1") != "0":
            qkv, anc = ops.qkv_anchor(r, pk["qa_blob"], pk["qa_slots"][0], pk["qa_slots"][1], B, H, W, lo_blob=pk["qa_lo"])
        elif pk.get("hiq"):
            qkv = ops.linear(r, pk["qkv_w3"], pk["qkv_b"], epi=L.EPI_GROUPNORM, gscale=pk["qkv_gs"], planes=True, a_split=3, w_regs=pk.get("qkv_w3r"))
            anc = ops.linear(r, pk["anc_w3"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], pool=(df, H, W), planes=True, a_split=3)
        elif one_pass:
            qkv, anc = ops.qkv_anchor(r, pk["qa_blob"], pk["qa_slots"][0], pk["qa_slots"][1], B, H, W)
        else:
            if "qkv_blob" in pk and os.environ.get("GRL_STREAM_QKV", "1") != "0":
                qkv = ops.qkv(r, pk["qkv_blob"], pk["qkv_slots"])
            else:
                qkv = ops.linear(r, pk["qkv_w"], pk["qkv_b"], epi=L.EPI_GROUPNORM, gscale=pk["qkv_gs"], planes=True)
            anc = ops.linear(r, pk["anc_w"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], pool=(df, H, W), planes=True)
        att = ops.empty(M, (nh_w + nh_s) * 32, dtype=ops.GEMM_DTYPE, device=dev)  # operand of the proj GEMM
        self._attention(qkv, anc, att, pk, geo, B, H, W)
        cab, gate = self._cab(r, pk, B, H, W, CP) if self.local_connection else (None, None)
        if "proj_blob" in pk and "mlp_blob" in pk and H * W >= 128 and os.environ.get("GRL_FUSED_TAIL", "1") != "0":
            return ops.block_tail(att, r, cab, gate, H * W, pk["proj_blob"], pk["proj_b"], pk["n1_g"], pk["n1_b"], pk["mlp_blob"],
                                  pk["fc2_b"], pk["n2_g"], pk["n2_b"], Hpad=pk["mlp_hp"], n_real=C, res_scale=self.res_scale,
                                  rblob=pk.get("tail_rblob"))
        # x = x + res_scale * norm1(proj(attn)) + cab(x)   (efficient.py:543-548)
        r1 = ops.linear(att, pk["proj_w"], pk["proj_b"], epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=pk["n1_g"],
                        ln_b=pk["n1_b"], n_real=C, res_scale=self.res_scale, resid=r, add2=cab, add2_scale=gate,
                        rows_per_image=H * W)
        # x = x + res_scale * norm2(mlp(x))                 (efficient.py:554)
        if "mlp_blob" in pk and os.environ.get("GRL_FUSED_MLP", "1") != "0":
            return ops.mlp(r1, pk["mlp_blob"], pk["fc2_b"], pk["n2_g"], pk["n2_b"], Hpad=pk["mlp_hp"], n_real=C,
                           res_scale=self.res_scale)
        h = ops.linear(r1, pk["fc1_w"], pk["fc1_b"], epi=L.EPI_GELU)
        return ops.linear(h, pk["fc2_w"], pk["fc2_b"], epi=L.EPI_LN_RES, out_dtype=torch.float32, ln_g=pk["n2_g"],
                          ln_b=pk["n2_b"], n_real=C, res_scale=self.res_scale, resid=r1)

    def _block_high(self, r, pk, geo: BlockGeo, B, H, W):
        """precision='high': every linear / conv contraction on split operands (activation hi+lo staged in-kernel from fp32,
        weights packed hi|hi|lo), fp32 intermediates, the row norms as separate launches; attention on hi + lo operand planes
        (3 QK^T terms, 2 PV terms in the generic kernel) with an fp32 output."""
        C, CP = self.embed_dim, r.shape[1]
        M = B * H * W
        f32 = torch.float32
        G = pk["qkv_w"].shape[0] // 32
        Ma = M // (geo.df * geo.df)
        qkv_lo = ops.empty(G, M, 32, dtype=ops.PLANE_DTYPE, device=r.device)
        anc_lo = ops.empty(geo.nh_s, Ma, 32, dtype=ops.PLANE_DTYPE, device=r.device)
        qkv = ops.linear(r, pk["qkv_w"], pk["qkv_b"], epi=L.EPI_GROUPNORM, gscale=pk["qkv_gs"], planes=True, a_split=3, out_lo=qkv_lo, w_regs=pk.get("qkv_wr"))
        if pk.get("anc_wr") is not None and Ma % 32 == 0:
            # AnchorLinear's avg-pool (mixed_attn_block.py:727-736) as a reduction of its own, then the weights-stationary kernel on
            # the M / df^2 pooled rows (the generic kernel with the pool fused into its A load: 440 us of a 384x384 x4 deblur block)
            pooled = r.view(B, H // geo.df, geo.df, W // geo.df, geo.df, CP).mean(dim=(2, 4)).view(Ma, CP)
            anc = ops.linear(pooled, pk["anc_w"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], planes=True, a_split=3,
                             out_lo=anc_lo, w_regs=pk["anc_wr"])
        else:
            anc = ops.linear(r, pk["anc_w"], pk["anc_b"], epi=L.EPI_GROUPNORM, gscale=pk["anc_gs"], pool=(geo.df, H, W), planes=True,
                             a_split=3, out_lo=anc_lo)
        att = ops.empty(M, (geo.nh_w + geo.nh_s) * 32, dtype=f32, device=r.device)
        # attention on split operands too: q, k, v (and the anchor-side values) as fp16 hi + lo planes -> generic kernel
        self._attention(qkv, anc, att, pk, geo, B, H, W, qkv_lo=qkv_lo, anc_lo=anc_lo)
        cab, gate = self._cab(r, pk, B, H, W, CP) if self.local_connection else (None, None)
        # norm + residual (+ gated CAB branch) in the epilogue of the weights-stationary kernel where it takes the shape (a row of
        # <= 192 channels is one slab): the fp32 products p1 / p2 never reach memory
        fuse = (CP <= 192 and M % 32 == 0 and (H * W) % 32 == 0 and pk.get("proj_wr") is not None and pk.get("fc2_wr") is not None
                and os.environ.get("GRL_HIGH_FUSE_LN", "1") != "0")
        if fuse:
            r1 = ops.linear(att, pk["proj_w"], pk["proj_b"], epi=L.EPI_LN_RES, out_dtype=f32, a_split=3, w_regs=pk["proj_wr"],
                            ln_g=pk["n1_g"], ln_b=pk["n1_b"], n_real=C, res_scale=self.res_scale, resid=r, add2=cab, add2_scale=gate,
                            rows_per_image=H * W)
        else:
            p1 = ops.linear(att, pk["proj_w"], pk["proj_b"], out_dtype=f32, a_split=3, w_regs=pk.get("proj_wr"))
            r1 = ops.layernorm_res(p1, r, pk["n1_g"], pk["n1_b"], C, res_scale=self.res_scale, add2=cab, add2_scale=gate,
                                   rows_per_image=H * W)
        h = ops.linear(r1, pk["fc1_w"], pk["fc1_b"], epi=L.EPI_GELU, out_dtype=f32, a_split=3, w_regs=pk.get("fc1_wr"))
        if fuse:
            return ops.linear(h, pk["fc2_w"], pk["fc2_b"], epi=L.EPI_LN_RES, out_dtype=f32, a_split=3, w_regs=pk["fc2_wr"],
                              ln_g=pk["n2_g"], ln_b=pk["n2_b"], n_real=C, res_scale=self.res_scale, resid=r1)
        p2 = ops.linear(h, pk["fc2_w"], pk["fc2_b"], out_dtype=f32, a_split=3, w_regs=pk.get("fc2_wr"))
        return ops.layernorm_res(p2, r1, pk["n2_g"], pk["n2_b"], C, res_scale=self.res_scale)

    def forward_features(self, f, plan, B, H, W):
        """grl.py:491-504 on the token matrix f [B*H*W, CP] (fp32) -> [B*H*W, CP]."""
        C = self.embed_dim
        t = ops.layernorm(f, plan["ns_g"], plan["ns_b"], C)
        n = self.stream_groups(B)
        if n > 1:
            return self._features_streams(t, plan, B, H, W, n)
        check = os.environ.get("GRL_CHECK_RANGE", "0") == "1"   # debug: largest residual-stream magnitude per block (fp16 operand
        for si, st in enumerate(plan["stages"]):                # staging saturates at 65504; this reports how close a checkpoint gets)
            r = t
            for bi, pk in enumerate(st["blocks"]):
                r = self._block(r, pk, plan["sched"][si][bi], B, H, W)
                if check:
                    print(f"GRL_CHECK_RANGE layers.{si}.blocks.{bi}: max|x| = {r.abs().max().item():.4g}  (fp16 operand limit 65504)")
            # TransformerStage.forward (grl.py:164-170): conv3x3 + residual
            t = ops.conv3x3(r, st["conv_w"], st["conv_b"], B, H, W, resid=t, x_split=plan["xs"]["stage_conv"])
        return ops.layernorm(t, plan["ne_g"], plan["ne_b"], C)

    @staticmethod
    def stream_groups(B: int) -> int:
        """Number of tile groups / HIP streams a batch of B tiles is processed in (GRL_SPLIT_STREAMS, default 2;
        measured on MI355X: 2 groups +6 % tiles/s over one stream, 4 groups are host-launch bound)."""
        n = int(os.environ.get("GRL_SPLIT_STREAMS", "2"))
        if os.environ.get("GRL_CHECK_RANGE", "0") == "1":
            return 1
        return n if n > 1 and B >= n and B % n == 0 else 1

    def _features_streams(self, t, plan, B, H, W, n):
        """The tile batch is cut into n groups that advance block by block on n HIP streams: tiles are independent,
        and the HBM-bound linear kernels of one group overlap the MFMA-bound attention of another."""
        dev = t.device
        main = torch.cuda.current_stream(dev)
        pool = getattr(self, "_streams", None)
        if pool is None or len(pool) < n or pool[0].device != dev:
            pool = self._streams = [torch.cuda.Stream(dev) for _ in range(n)]
        Bg = B // n
        Mg = t.shape[0] // n
        parts = [t[g * Mg : (g + 1) * Mg] for g in range(n)]
        for g in range(n):
            pool[g].wait_stream(main)
        for si, st in enumerate(plan["stages"]):
            r = list(parts)
            for bi, pk in enumerate(st["blocks"]):
                for g in range(n):
                    with torch.cuda.stream(pool[g]):
                        r[g] = self._block(r[g], pk, plan["sched"][si][bi], Bg, H, W)
            for g in range(n):
                with torch.cuda.stream(pool[g]):
                    parts[g] = ops.conv3x3(r[g], st["conv_w"], st["conv_b"], Bg, H, W, resid=parts[g], x_split=plan["xs"]["stage_conv"])
        out = ops.empty_like(t)
        for g in range(n):
            with torch.cuda.stream(pool[g]):
                ops.layernorm(parts[g], plan["ne_g"], plan["ne_b"], self.embed_dim, out=out[g * Mg : (g + 1) * Mg])
            main.wait_stream(pool[g])
        return out

    # ---- training path (BASELINE config 5; reference: engines/base.py:221-236 = autograd through grl.py / efficient.py) --------
    @staticmethod
    def _drop_path(x, rows_per_image: int, p: float, training: bool):
        """timm DropPath (scale_by_keep) on a token matrix: one Bernoulli draw per image (mixed_attn_block_efficient.py:500)."""
        if p == 0.0 or not training:
            return x
        keep = 1.0 - p
        m = x.new_empty(x.shape[0] // rows_per_image, 1, 1).bernoulli_(keep) / keep
        return (x.view(-1, rows_per_image, x.shape[1]) * m).view_as(x)

    def _to_planes(self, t, one_col: int = -1, extra: int = 32):
        """[tokens, nh, d] -> fp32 head planes [nh, tokens, 32] in ONE launch: a cat with a cached constant block for the pad
        columns (F.pad is a fill plus a strided copy) -- zeros, and 1.0 in plane column ``one_col`` where the attention kernel
        wants a constant (k: slot 31, the partner of the running softmax offset; v: column d, the softmax denominator), so that
        the attention op needs no index fills (``prepared`` operands)."""
        M, nh, d = t.shape
        if d == extra:
            return t.permute(1, 0, 2).contiguous()
        cache = self.__dict__.setdefault("_coords_cache", {})
        key = ("padblk", nh, M, extra - d, one_col, str(t.device))
        blk = cache.get(key)
        if blk is None:
            blk = torch.zeros(nh, M, extra - d, dtype=torch.float32, device=t.device)
            if one_col >= d:
                blk[..., one_col - d] = 1.0
            cache[key] = blk
        return torch.cat([t.permute(1, 0, 2), blk], dim=2)

    def _block_planes(self, x, scales, one_cols, sc=None):
        """All head planes of a block's projection in ONE chain: ``x`` [tokens, S, nh, d] (S slots: q / k / v of the window branch and of
        the stripe branch; or the anchors, used twice) -> fp32 planes [S, nh, tokens, 32] plus their fp16 copy (the kernels' operands).
        ``scales[j]``: None = slot j is taken as it is (values), a tensor [nh] = L2-normalise over d and multiply (q: the clamped
        logit scale * log2e, k: ones); ``one_cols[j]``: plane column of slot j that holds 1.0 (-1: none).  The per-slot chain of
        round 4 (normalize, scale, permute + cat, fp16 copy -- and in the backward a strided [tokens, nh, d] -> [nh] reduction per
        logit scale) was ~35 launches forward and ~80 backward per block; here the scale rides on the per-token inverse norm
        (a tensor 1/d the size, so its gradient is a last-dim reduction plus a small column sum): 6 + ~12 launches."""
        T, S, nh, d = x.shape
        dev = x.device
        cache = self.__dict__.setdefault("_coords_cache", {})
        if ops.head_planes_ok(x, S) and os.environ.get("GRL_PLANES_KERNEL", "1") != "0" and not ops.deterministic():
            # round 6: one launch forward (normalise, scale, pad constants, permute, fp16 copy), one backward (csrc/planes.hip).
            # An expanded input (the anchors, used as scaled queries and as keys) is passed once: both slots read input slot 0.
            expanded = x.stride(1) == 0
            xin = x[:, :1] if expanded else x
            ones = cache.get(("ones_nh", nh, str(dev)))
            if ones is None:
                ones = cache[("ones_nh", nh, str(dev))] = torch.ones(nh, dtype=torch.float32, device=dev)
            if sc is None:                                                                   # (else: prebuilt for all blocks, _train_tables)
                sc = torch.stack([ones if s is None else s for s in scales])                 # [S, nh] (differentiable in the q scales)
            # (the fp32 planes are autograd's handle on the operands only -- every consumer takes the fp16 copies, f16= of the attention
            # op -- so the kernel does not write them: GRL_PLANES_WRITE32=1 restores the values)
            outs = AG.HeadPlanesFn.apply(xin, sc, tuple(0 if expanded else j for j in range(S)), tuple(s is None for s in scales),
                                         tuple(int(c) for c in one_cols), os.environ.get("GRL_PLANES_WRITE32", "0") == "1")
            return outs[:S], outs[S:]
        key = ("planes_const", T, S, nh, d, tuple(s is None for s in scales), tuple(one_cols), str(dev))
        const = cache.get(key)
        if const is None:
            blk = torch.zeros(S, nh, T, 32 - d, dtype=torch.float32, device=dev)
            for j, c in enumerate(one_cols):
                if c >= d:
                    blk[j, :, :, c - d] = 1.0
            raw = torch.tensor([s is None for s in scales], device=dev).view(1, S, 1, 1)
            const = cache[key] = (blk, raw, torch.ones(nh, dtype=torch.float32, device=dev))
        blk, raw, ones = const
        sc = torch.stack([ones if s is None else s for s in scales]).view(1, S, nh, 1)
        nrm = torch.linalg.vector_norm(x, dim=-1, keepdim=True)                      # F.normalize: v / max(|v|, 1e-12)
        inv = torch.where(raw, ones.view(1, 1, nh, 1), sc / nrm.clamp_min(1e-12))
        y = x * inv
        planes = torch.cat([y.permute(1, 2, 0, 3), blk], dim=3) if d < 32 else y.permute(1, 2, 0, 3).contiguous()
        return planes.unbind(0), planes.detach().to(ops.PLANE_DTYPE).unbind(0)

    def _train_tables(self, sched, dev):
        """The relative-position bias tables and clamped logit scales of EVERY block in a few batched chains (training path).
        The tables depend on the CPB-MLP weights only, not on activations, so nothing forces them to be built block by block:
        per geometry class (same coordinate table and head count) the 2 -> 512 -> nh MLPs of all blocks run as one broadcast
        layer and one bmm -- per block they were 3 x (7 launches forward, ~12 backward) with a K = 2 GEMM that the BLAS library
        takes 45-50 us for (16 ms of a 182 ms step).  Returns {(stage, block): ((table_w, table_a2w, table_w2a), scales [3, nh],
        floors [3, nh])}; only for blocks whose two branches have the same head count (the batched plane path)."""
        cache = self.__dict__.setdefault("_coords_cache", {})
        groups, blocks = {}, []
        for si, stage in enumerate(self.layers):
            for bi, blk in enumerate(stage.blocks):
                geo, a = sched[si][bi], blk.attn
                if geo.nh_w != geo.nh_s:
                    continue
                ts = (a.window_attn.attn_transform, a.stripe_attn.attn_transform1, a.stripe_attn.attn_transform2)
                blocks.append(((si, bi), ts))
                for slot, (m, win, df) in enumerate(zip(ts, (geo.window, geo.stripe, geo.stripe), (1, geo.df, geo.df))):
                    key = (tuple(win), df, tuple(m.cpb_mlp[2].weight.shape))
                    groups.setdefault(key, []).append(((si, bi, slot), m))
        tabs = {}
        for (win, df, _), items in groups.items():
            ck = (win, df, str(dev))
            coords = cache.get(ck)
            if coords is None:
                coords = cache[ck] = tables.coords_table(win, df, device=dev)
            rows = coords.shape[0]
            idx = cache.get(("revidx", rows, str(dev)))
            if idx is None:
                idx = cache[("revidx", rows, str(dev))] = torch.cat([torch.arange(rows - 1, -1, -1, device=dev),
                                                                       torch.zeros((-rows) % 4, dtype=torch.long, device=dev)])
            W1 = torch.stack([m.cpb_mlp[0].weight for _, m in items])            # [G, 512, 2]
            b1 = torch.stack([m.cpb_mlp[0].bias for _, m in items])              # [G, 512]
            W2 = torch.stack([m.cpb_mlp[2].weight for _, m in items])            # [G, nh, 512]
            # round 6: one launch forward, one backward, the [G, rows, 512] hidden layer (1.5 GB for the stripe transforms of GRL-Base)
            # never in memory (autograd.cpb_tables -> csrc/cpb.hip; CPU tensors: the torch expression)
            t = AG.cpb_tables(coords, W1, b1, W2, idx)                            # [G, nh, rows4], see _attn_table
            for (key, _), tt in zip(items, t.unbind(0)):
                tabs[key] = tt
        out = {}
        if blocks:
            ls = torch.stack([m.logit_scale.reshape(-1) for _, ts in blocks for m in ts]).view(len(blocks), 3, -1)
            scales = torch.clamp(ls, max=math.log(1.0 / 0.01)).exp() * LOG2E      # efficient.py:39, exp2 domain
            floors = -1.0 - torch.ceil(scales.detach())                           # tables.lazy_floor
            # the [slots, nh] scale matrices of the two plane launches of every block (q k v q k v | anchors as q, as k) from one cat
            # each and one unbind (per block: a stack forward, a stack backward)
            one = torch.ones(len(blocks), 1, scales.shape[2], dtype=scales.dtype, device=scales.device)
            sc6 = torch.cat([scales[:, 0:1], one, one, scales[:, 2:3], one, one], dim=1).unbind(0)
            sc2 = torch.cat([scales[:, 1:2], one], dim=1).unbind(0)
            for i, ((key, _), sc, fl) in enumerate(zip(blocks, scales.unbind(0), floors.unbind(0))):
                out[key] = (tuple(tabs[key + (slot,)] for slot in range(3)), sc, fl, sc6[i], sc2[i])
        return out

    def _attn_table(self, m: _Affine, win, df, dev):
        key = (tuple(win), df, str(dev))
        cache = self.__dict__.setdefault("_coords_cache", {})      # constant per geometry: a dozen tiny launches per call otherwise
        coords = cache.get(key)
        if coords is None:
            coords = cache[key] = tables.coords_table(win, df, device=dev)
        # tables.kernel_table(16 * sigmoid(cpb_mlp(coords))) -- transpose, exp2 domain, reversed rows, padded to 4; the pad
        # entries repeat row 0 instead of being zero: no valid (query, key) pair addresses them
        rows = coords.shape[0]
        idx = cache.get(("revidx", rows, str(dev)))
        if idx is None:
            idx = cache[("revidx", rows, str(dev))] = torch.cat([torch.arange(rows - 1, -1, -1, device=dev),
                                                                   torch.zeros((-rows) % 4, dtype=torch.long, device=dev)])
        return AG.cpb_tables(coords, m.cpb_mlp[0].weight.unsqueeze(0), m.cpb_mlp[0].bias.unsqueeze(0), m.cpb_mlp[2].weight.unsqueeze(0), idx)[0]

    @staticmethod
    def _scale(m: _Affine):
        """exp(min(logit_scale, ln 100)) * log2e per head (efficient.py:39), differentiable below the clamp."""
        return torch.clamp(m.logit_scale.reshape(-1), max=math.log(1.0 / 0.01)).exp() * LOG2E

    def _block_train(self, r, blk: _Block, geo: BlockGeo, B, H, W, dp: float, pre=None):
        """EfficientMixAttnTransformerBlock.forward (efficient.py:539-556) on the token matrix r [B*H*W, C] with autograd."""
        C = self.embed_dim
        M = B * H * W
        nh_w, nh_s, df = geo.nh_w, geo.nh_s, geo.df
        d_w, d_s = C // 2 // nh_w, C // 2 // nh_s
        Ha, Wa = H // df, W // df
        a = blk.attn
        dev = r.device
        # (r has four consumers -- QKV projection, anchor pooling, the CAB, the residual: their gradients are added in one launch)
        r_q, r_p, r_c, r = AG.fan_out(r, 4)
        qkv = AG.linear(r_q, a.qkv.body.weight, a.qkv.body.bias)                               # QKVProjection (mixed_attn_block.py:669-676)
        pooled = r_p.view(B, Ha, df, Wa, df, C).mean(dim=(2, 4)).reshape(B * Ha * Wa, C)        # AnchorLinear avg-pool (:727-736)
        anc = AG.linear(pooled, a.anchor.body[0].reduction.weight, a.anchor.body[0].reduction.bias).view(-1, nh_s, d_s)
        same = (nh_w, d_w) == (nh_s, d_s) and os.environ.get("GRL_TRAIN_BATCHED_PLANES", "1") != "0"
        if same:
            att = self._attention_train_batched(qkv, anc, a, geo, B, H, W, pre)
            return self._block_train_tail(r, att, blk, B, H, W, dp, r_c)
        if (nh_w, d_w) == (nh_s, d_s):   # one view, one unbind: the backward is a single stack instead of two slice-backwards (zeros + copy) and an add
            qw, kw, vw, qs, ks, vs = qkv.view(M, 6, nh_w, d_w).unbind(1)
        else:
            qw, kw, vw = qkv[:, : 3 * C // 2].reshape(M, 3, nh_w, d_w).unbind(1)
            qs, ks, vs = qkv[:, 3 * C // 2 :].reshape(M, 3, nh_s, d_s).unbind(1)
        P = self._to_planes
        k1_w, k1_s = (31 if d_w <= 30 else -1), (31 if d_s <= 30 else -1)     # plane columns that hold a constant 1.0 (see _to_planes)
        v1_w, v1_s = (d_w if d_w < 32 else -1), (d_s if d_s < 32 else -1)

        def floor(sc):   # tables.lazy_floor from the already scaled value: sc = clamped scale * log2e
            return -1.0 - torch.ceil(sc.detach())

        ws, sh = geo.window, geo.window_shift
        st, ss = geo.stripe, geo.stripe_shift_size
        ast, ass = geo.anchor_stripe, geo.anchor_shift_size
        g_tok_w = (H, W, ws[0], ws[1], sh, sh)
        g_tok_s = (H, W, st[0], st[1], ss[0], ss[1])
        g_anc = (Ha, Wa, ast[0], ast[1], ass[0], ass[1])
        # window attention (efficient.py:128-165)
        tw = a.window_attn.attn_transform
        sw = self._scale(tw)
        ow = AG.AttentionFn.apply(P(F.normalize(qw, dim=-1) * sw.view(1, nh_w, 1)), P(F.normalize(kw, dim=-1), k1_w), P(vw, v1_w),
                                  self._attn_table(tw, geo.window, 1, dev),
                                  dict(q=g_tok_w, k=g_tok_w, B=B, nh=nh_w, d=d_w, masked=sh > 0, floor=floor(sw), prepared=True))
        # anchored stripe attention (efficient.py:215-270): anchors -> stripe tokens, then stripe tokens -> anchors
        t1, t2 = a.stripe_attn.attn_transform1, a.stripe_attn.attn_transform2
        an = F.normalize(anc, dim=-1)
        s1, s2 = self._scale(t1), self._scale(t2)
        y = AG.AttentionFn.apply(P(an * s1.view(1, nh_s, 1)), P(F.normalize(ks, dim=-1), k1_s), P(vs, v1_s),
                                 self._attn_table(t1, geo.stripe, df, dev),
                                 dict(q=g_anc, k=g_tok_s, B=B, nh=nh_s, d=d_s, masked=geo.stripe_shift, floor=floor(s1), prepared=True))
        cache = self.__dict__.setdefault("_coords_cache", {})
        dmask = cache.get(("dmask", d_s, str(dev)))
        if dmask is None:
            dmask = cache[("dmask", d_s, str(dev))] = (torch.arange(32, device=dev) < d_s).float()
        onev = cache.get(("onev", d_s, str(dev)))
        if onev is None:
            onev = cache[("onev", d_s, str(dev))] = (torch.arange(32, device=dev) == v1_s).float()
        yv = torch.addcmul(onev, y, dmask)                              # real head dims only, and the constant 1.0 in column d again
        os_ = AG.AttentionFn.apply(P(F.normalize(qs, dim=-1) * s2.view(1, nh_s, 1)), P(an, k1_s), yv,
                                   self._attn_table(t2, geo.stripe, df, dev),
                                   dict(q=g_tok_s, k=g_anc, B=B, nh=nh_s, d=d_s, masked=geo.stripe_shift, floor=floor(s2), prepared=True))
        if d_w == d_s:                    # one cat of the planes, one slice (its backward: one zeros + copy instead of two)
            att = torch.cat([ow, os_], dim=0).permute(1, 0, 2)[..., :d_w].reshape(M, C)
        else:
            att = torch.cat([ow.permute(1, 0, 2)[..., :d_w].reshape(M, C // 2), os_.permute(1, 0, 2)[..., :d_s].reshape(M, C // 2)], dim=1)
        return self._block_train_tail(r, att, blk, B, H, W, dp)

    def _attention_train_batched(self, qkv, anc, a, geo: BlockGeo, B, H, W, pre=None):
        """The three attention calls of a block (as in _block_train) with all head planes built by two _block_planes chains."""
        C = self.embed_dim
        M = B * H * W
        nh, df = geo.nh_w, geo.df
        d = C // 2 // nh
        Ha, Wa = H // df, W // df
        dev = qkv.device
        k1, v1 = (31 if d <= 30 else -1), (d if d < 32 else -1)
        tw, t1, t2 = a.window_attn.attn_transform, a.stripe_attn.attn_transform1, a.stripe_attn.attn_transform2
        # the three clamped logit scales (efficient.py:39) and their lazy-offset floors in one chain each instead of three
        if pre is None:
            scales = torch.clamp(torch.stack([tw.logit_scale.reshape(-1), t1.logit_scale.reshape(-1), t2.logit_scale.reshape(-1)]),
                                 max=math.log(1.0 / 0.01)).exp() * LOG2E
            floors = -1.0 - torch.ceil(scales.detach())                      # tables.lazy_floor from the already scaled values
            tabs = (self._attn_table(tw, geo.window, 1, dev), self._attn_table(t1, geo.stripe, df, dev),
                    self._attn_table(t2, geo.stripe, df, dev))
        sc6 = sc2 = None
        if pre is not None:                                                  # (built for all blocks at once: _train_tables)
            tabs, scales, floors, sc6, sc2 = pre
        sw, s1, s2 = scales.unbind(0)
        fw, f1, f2 = floors.unbind(0)
        cache = self.__dict__.setdefault("_coords_cache", {})
        ones = cache.get(("ones_nh", nh, str(dev)))
        if ones is None:
            ones = cache[("ones_nh", nh, str(dev))] = torch.ones(nh, dtype=torch.float32, device=dev)
        # slots of the projection: q k v (window branch), q k v (stripe branch); the anchors serve as queries (scaled) and as keys
        (qw, kw, vw, qs, ks, vs), (qw16, kw16, vw16, qs16, ks16, vs16) = self._block_planes(
            qkv.view(M, 6, nh, d), (sw, ones, None, s2, ones, None), (-1, k1, v1, -1, k1, v1), sc=sc6)
        (aq, ak), (aq16, ak16) = self._block_planes(anc.view(-1, 1, nh, d).expand(-1, 2, nh, d), (s1, ones), (-1, k1), sc=sc2)

        ws, sh = geo.window, geo.window_shift
        st, ss = geo.stripe, geo.stripe_shift_size
        ast, ass = geo.anchor_stripe, geo.anchor_shift_size
        g_tok_w = (H, W, ws[0], ws[1], sh, sh)
        g_tok_s = (H, W, st[0], st[1], ss[0], ss[1])
        g_anc = (Ha, Wa, ast[0], ast[1], ass[0], ass[1])
        # (round 6: the two branch outputs as token matrices [M, nh * 32]: one cat along the channels gives the projection's input --
        # _block_train_tail places the weight columns accordingly -- and the cat's backward hands each attention backward its column
        # block of the gradient in place; before: cat of the planes, permute, slice, copy, and zeros + copy + two copies back)
        tm = os.environ.get("GRL_TRAIN_TOKEN_MAJOR", "1") != "0"
        ow = AG.AttentionFn.apply(qw, kw, vw, tabs[0],
                                  dict(q=g_tok_w, k=g_tok_w, B=B, nh=nh, d=d, masked=sh > 0, floor=fw, prepared=True,
                                       f16=(qw16, kw16, vw16), token_major=tm))
        y = AG.AttentionFn.apply(aq, ks, vs, tabs[1],
                                 dict(q=g_anc, k=g_tok_s, B=B, nh=nh, d=d, masked=geo.stripe_shift, floor=f1, prepared=True,
                                      f16=(aq16, ks16, vs16)))
        dmask = cache.get(("dmask", d, str(dev)))
        if dmask is None:
            dmask = cache[("dmask", d, str(dev))] = (torch.arange(32, device=dev) < d).float()
        onev = cache.get(("onev", d, str(dev)))
        if onev is None:
            onev = cache[("onev", d, str(dev))] = (torch.arange(32, device=dev) == v1).float()
        if y.is_cuda and d < 31:
            # the kernel's output already IS the prepared value operand: column d = the softmax denominator over itself (1.0, exact
            # once rounded to fp16), column 31 = 0.  Only the gradient of the pad columns has to go (round 6: an addcmul forward and
            # three multiplies backward before).
            yv = AG.PadGradMask.apply(y, dmask)
        else:
            yv = torch.addcmul(onev, y, dmask)                          # real head dims only, and the constant 1.0 in column d again
        os_ = AG.AttentionFn.apply(qs, ak, yv, tabs[2],
                                   dict(q=g_tok_s, k=g_anc, B=B, nh=nh, d=d, masked=geo.stripe_shift, floor=f2, prepared=True,
                                        f16=(qs16, ak16, None), token_major=tm))
        if tm:
            return torch.cat([ow, os_], dim=1)                          # [M, 2 * nh * 32]
        return torch.cat([ow, os_], dim=0).permute(1, 0, 2)[..., :d].reshape(M, C)

    def _block_train_tail(self, r, att, blk: _Block, B, H, W, dp: float, r_conv=None):
        """proj + norm1 + residual, CAB, MLP + norm2 + residual of a block (efficient.py:543-556) on token matrices."""
        C = self.embed_dim
        M = B * H * W
        a = blk.attn
        if att.shape[1] != C:
            # att = [M, heads * 32]: head h's d channels at columns 32 h .. 32 h + d - 1 (the attention kernels' own layout); the weight
            # columns go where their channels are, zero elsewhere.  Column d of a head holds the softmax denominator over itself = 1.0
            # (v's ones column through the PV product): the bias gradient's ones column.
            nht = att.shape[1] // 32
            dh = C // nht
            cache = self.__dict__.setdefault("_coords_cache", {})
            zkey = ("wzero", C, nht, 32 - dh, str(att.device))
            zb = cache.get(zkey)
            if zb is None:
                zb = cache[zkey] = torch.zeros(C, nht, 32 - dh, dtype=torch.float32, device=att.device)
            wpad = torch.cat([a.proj.weight.view(C, nht, dh), zb], dim=2).view(C, nht * 32)
            x1 = AG.linear(att, wpad, a.proj.bias, one_col=dh if dh < 32 else -1)
        else:
            x1 = AG.linear(att, a.proj.weight, a.proj.bias)
        x1 = self._norm_residual(r, x1, blk.norm1, H * W, dp)
        if self.local_connection:   # CAB + ChannelAttention (mixed_attn_block.py:948-983)
            c0, c2, se = blk.conv.cab[0], blk.conv.cab[2], blk.conv.cab[3].attention
            u = AG.conv3x3(F.gelu(AG.conv3x3(r if r_conv is None else r_conv, c0.weight, c0.bias, B, H, W)), c2.weight, c2.bias, B, H, W)
            # x1 + u * gate(u): pool, squeeze-excite MLP and the gated residual as three launches each way (autograd.se_residual)
            x1 = AG.se_residual(x1, u, se[1].weight.flatten(1), se[1].bias, se[3].weight.flatten(1), se[3].bias, H * W)
        # Mlp (swin_v1_block.py:37-43): the GELU between fc1 and fc2 is taken by fc2's loader, its adjoint by the epilogue of fc2's
        # data-gradient launch (autograd.linear gelu_in; GRL_GELU_FUSED=0: the torch activation)
        h1 = AG.linear(x1, blk.mlp.fc1.weight, blk.mlp.fc1.bias)
        if os.environ.get("GRL_GELU_FUSED", "1") != "0":
            m = AG.linear(h1, blk.mlp.fc2.weight, blk.mlp.fc2.bias, gelu_in=True)
        else:
            m = AG.linear(F.gelu(h1), blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        return self._norm_residual(x1, m, blk.norm2, H * W, dp)

    def _norm_residual(self, r, t, norm, rows_per_image: int, p: float):
        """r + res_scale * DropPath(norm(t)) (efficient.py:543-556; timm DropPath, scale_by_keep: one Bernoulli draw per image) inside the
        LayerNorm launches (autograd.layer_norm_residual; round 6 -- _residual's addcmul was one more pass forward and two backward)."""
        if p == 0.0 or not self.training:
            return AG.layer_norm_residual(r, t, norm.weight, norm.bias, 1e-5, None, rows_per_image, self.res_scale)
        keep = 1.0 - p
        m = t.new_empty(t.shape[0] // rows_per_image).bernoulli_(keep)
        return AG.layer_norm_residual(r, t, norm.weight, norm.bias, 1e-5, m, rows_per_image, self.res_scale / keep)

    def _residual(self, r, t, rows_per_image: int, p: float):
        """r + res_scale * DropPath(t) (efficient.py:543-556, timm DropPath with scale_by_keep: one Bernoulli draw per image) in ONE
        launch: the residual scale and the keep mask ride in an addcmul / add-with-alpha instead of a multiply each."""
        if p == 0.0 or not self.training:
            return torch.add(r, t, alpha=self.res_scale)
        keep = 1.0 - p
        m = t.new_empty(t.shape[0] // rows_per_image, 1, 1).bernoulli_(keep)
        return torch.addcmul(r.view(-1, rows_per_image, r.shape[1]), t.view(-1, rows_per_image, t.shape[1]), m,
                             value=self.res_scale / keep).view_as(r)

    def _forward_train(self, x):
        """GRL.forward (grl.py:506-551) as a differentiable graph over the HIP kernels (autograd.py)."""
        H0, W0 = x.shape[2:]
        first = self.conv_first.weight
        if x.is_cuda and getattr(self, "_ag_registered", None) != (first.data_ptr(), first.device):   # (re)register after .to() / load
            AG.register_parameters(self)
            self._ag_registered = (first.data_ptr(), first.device)
        x = self.check_image_size(x.float())
        mean = self._mean.to(x.device, x.dtype)
        x = (x - mean) * self.img_range
        B, Cin, H, W = x.shape
        C, s, oc = self.embed_dim, self.upscale, self.out_channels
        sched = block_schedule(self.depths, self.num_heads_window, self.num_heads_stripe, self.window_size, self.stripe_size,
                               self.stripe_groups, self.stripe_shift, self.df, (H, W))

        def conv(t, m, b=B, h=H, w=W):
            return AG.conv3x3(t, m.weight, m.bias, b, h, w)

        def shuffle(t, b, h, w, r):   # PixelShuffle(r) on a token matrix [b*h*w, c*r*r] -> [b*h*r*w*r, c]
            c = t.shape[1] // (r * r)
            return t.view(b, h, w, c, r, r).permute(0, 1, 4, 2, 5, 3).reshape(b * h * r * w * r, c)

        def image(t, h, w):
            return t.view(B, h, w, -1).permute(0, 3, 1, 2)

        f = conv(x.permute(0, 2, 3, 1).reshape(B * H * W, Cin), self.conv_first)
        z = AG.layer_norm(f, self.norm_start.weight, self.norm_start.bias, 1e-5)
        pre = self._train_tables(sched, x.device) if os.environ.get("GRL_TRAIN_BATCHED_PLANES", "1") != "0" else {}
        j = 0
        for si, stage in enumerate(self.layers):
            r = z
            for bi, blk in enumerate(stage.blocks):
                r = self._block_train(r, blk, sched[si][bi], B, H, W, self._dpr[j], pre.get((si, bi)))
                j += 1
            z = conv(r, stage.conv) + z
        z = AG.layer_norm(z, self.norm_end.weight, self.norm_end.bias, 1e-5)
        body = conv(z, self.conv_after_body) + f
        if self.upsampler == "pixelshuffle":
            y = F.leaky_relu(conv(body, self.conv_before_upsample[0]), 0.01)
            h, w = H, W
            r = 3 if self.upscale == 3 else 2
            for m in self.upsample.up:
                if isinstance(m, nn.Conv2d):
                    y = shuffle(conv(y, m, B, h, w), B, h, w, r)
                    h, w = h * r, w * r
            y = image(conv(y, self.conv_last, B, h, w), h, w)
        elif self.upsampler == "pixelshuffledirect":
            y = image(shuffle(conv(body, self.upsample.up[0]), B, H, W, s), H * s, W * s)
        elif self.upsampler == "nearest+conv":
            def up2(t, h, w):
                return t.view(B, h, 1, w, 1, -1).expand(B, h, 2, w, 2, t.shape[1]).reshape(B * 4 * h * w, -1)

            y = F.leaky_relu(conv(body, self.conv_before_upsample[0]), 0.01)
            y = F.leaky_relu(conv(up2(y, H, W), self.conv_up1, B, 2 * H, 2 * W), 0.2)
            y = F.leaky_relu(conv(up2(y, 2 * H, 2 * W), self.conv_up2, B, 4 * H, 4 * W), 0.2)
            y = F.leaky_relu(conv(y, self.conv_hr, B, 4 * H, 4 * W), 0.2)
            y = image(conv(y, self.conv_last, B, 4 * H, 4 * W), 4 * H, 4 * W)
        else:
            y = image(conv(body, self.conv_last), H, W)
            if self.in_channels == self.out_channels:
                y = x + y
        y = y / self.img_range + mean
        y = y[:, :, : H0 * s, : W0 * s].contiguous()
        return AG.GradScaleTop.apply(y) if y.is_cuda else y      # (the gradient operand scale belongs to the fp16 HIP contractions)

    @staticmethod
    def _tokens(x, cpad):
        """(B, C, H, W) -> channels-last token matrix [B*H*W, cpad] (zero padded)."""
        B, C, H, W = x.shape
        t = torch.zeros(B * H * W, cpad, dtype=torch.float32, device=x.device)
        t[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C)
        return t

    @staticmethod
    def _image(t, B, H, W, C):
        return t.view(B, H, W, -1)[..., :C].permute(0, 3, 1, 2)

    def enable_graph(self, flag: bool = True):
        """Replay the whole forward as one captured HIP graph per input shape.  A forward is ~500
        kernel launches; at one 256x256 tile the eager path is bound by the host issuing them, the graph is not.
        Every cached graph owns the plan (packed weights / tables) it was captured with; ``load_state_dict`` and in-place
        parameter updates drop the graphs (post-hook / version stamp).
        """
        self._use_graph = bool(flag)
        self._graphs = {}
        return self

    def forward(self, x):
        """grl.py:506-551.  Eager launch sequence, or a captured HIP graph (``enable_graph`` / GRL_GRAPH=1)."""
        if getattr(self, "_use_graph", None) is None:
            self._use_graph, self._graphs = os.environ.get("GRL_GRAPH", "0") == "1", {}
        if not (self._use_graph and x.is_cuda) or ops.profiling() or torch.is_grad_enabled():
            return self._forward_eager(x)
        key = (tuple(x.shape), x.dtype, str(x.device))
        stamp = self._param_stamp()
        ent = self._graphs.get(key)
        if ent is not None and ent[4] != stamp:            # parameters changed in place since the capture
            self.invalidate_plan()
            ent = None
        if ent is None:
            with torch.no_grad():
                self._forward_eager(x)                     # builds the plan, warms the allocator and kernel attributes
                torch.cuda.synchronize(x.device)
                static_in = x.clone()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self._forward_eager(static_in)
            # the captured kernels point at the packed weights / tables of THIS shape's plan: the entry keeps them alive
            # (the plan cache itself holds one geometry at a time)
            ent = self._graphs[key] = (static_in, graph, static_out, next(iter(self._plan_cache.values())), stamp)
        static_in, graph, static_out = ent[:3]
        static_in.copy_(x)
        graph.replay()
        return static_out.clone()                          # the engine clamps its output in place (utils_image.py:31)

    def _forward_eager(self, x, plan=None):
        if not x.is_cuda:
            # CPU tensor: the composite torch path (composite.py; SURVEY 8(b) "errors", BASELINE configs[0]).  Not a fallback of the
            # GPU path -- a CUDA tensor never gets here and still fails loudly below when the HIP library is missing.
            if os.environ.get("GRL_NO_CPU_COMPOSITE", "0") == "1":
                raise RuntimeError("grl_image_restoration_amd.GRL: got a CPU tensor and GRL_NO_CPU_COMPOSITE=1 forbids the composite torch path "
                                   "(no CPU fallback)")
            if any(p.is_cuda for p in self.parameters()):
                raise RuntimeError("grl_image_restoration_amd.GRL: CPU input but the parameters live on the GPU")
            from . import composite
            composite.announce()
            return self._forward_train(x)
        L.lib()  # fail loudly if the extension is missing
        if plan is None and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_train(x)      # autograd path: every contraction, forward and backward, in libgrl_hip.so
        H0, W0 = x.shape[2:]
        x = self.check_image_size(x.float())
        mean = self._mean.to(x.device, x.dtype)
        x = (x - mean) * self.img_range
        B, _, H, W = x.shape
        s, oc = self.upscale, self.out_channels
        if plan is None:
            plan = self._plan((H, W), x.device)
        sp = plan["split"]

        def conv(*a, **kw):
            return ops.conv3x3(*a, x_split=sp, **kw)

        # fast: 16-bit intermediates of the tail feed fp16-operand convolutions; high: fp32 + split operands
        bf = torch.float32 if sp == 3 else ops.GEMM_DTYPE

        f = ops.conv3x3(self._tokens(x, plan["first"][0].shape[2] // 3), *plan["first"], B, H, W, x_split=3)   # conv_first
        body = ops.conv3x3(self.forward_features(f, plan, B, H, W), *plan["after"], B, H, W, resid=f, x_split=plan["xs"]["after"])  # conv_after_body + f
        if self.upsampler == "pixelshuffle":
            y = conv(body, *plan["cbu"], B, H, W, act=2, slope=0.01, out_dtype=bf)
            h, w, r = H, W, plan["ups_r"]
            for wt, bs in plan["ups"]:
                y = conv(y, wt, bs, B, h, w, out_dtype=bf, shuffle_r=r, shuffle_cg=64)         # conv + PixelShuffle
                h, w = h * r, w * r
            y = self._image(conv(y, *plan["last"], B, h, w), B, h, w, oc)
        elif self.upsampler == "pixelshuffledirect":
            y = conv(body, *plan["upd"], B, H, W, shuffle_r=s, shuffle_cg=plan["upd_cg"])
            y = self._image(y, B, H * s, W * s, oc)
        elif self.upsampler == "nearest+conv":
            y = conv(body, *plan["cbu"], B, H, W, act=2, slope=0.01, out_dtype=bf)

            def up2(t, h, w):  # nearest x2 on a token matrix
                return t.view(B, h, 1, w, 1, -1).expand(B, h, 2, w, 2, t.shape[1]).reshape(B * 4 * h * w, -1)

            y = conv(up2(y, H, W), *plan["up1"], B, 2 * H, 2 * W, act=2, slope=0.2, out_dtype=bf)
            y = conv(up2(y, 2 * H, 2 * W), *plan["up2"], B, 4 * H, 4 * W, act=2, slope=0.2, out_dtype=bf)
            y = conv(y, *plan["hr"], B, 4 * H, 4 * W, act=2, slope=0.2, out_dtype=bf)
            y = self._image(conv(y, *plan["last"], B, 4 * H, 4 * W), B, 4 * H, 4 * W, oc)
        else:
            y = self._image(ops.conv3x3(body, *plan["last"], B, H, W, x_split=plan["xs"]["last"]), B, H, W, oc)
            if self.in_channels == self.out_channels:
                y = x + y
        y = y / self.img_range + mean
        return y[:, :, : H0 * s, : W0 * s].contiguous()
