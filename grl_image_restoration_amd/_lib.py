"""ctypes binding of libgrl_hip.so (the C ABI declared in include/grl_hip.h).

There is deliberately NO fallback: if the shared library is missing or stale, or a kernel
returns an error, a RuntimeError is raised.  Build it with ``python __graft_entry__.py`` (or
``make -C grl_image_restoration_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgrl_hip.so")
ABI_VERSION = 22
DT_F32, DT_BF16, DT_F16 = 0, 1, 2

EPI_PLAIN, EPI_GELU, EPI_GROUPNORM, EPI_LN_RES, EPI_GELU_GRAD = 0, 1, 2, 3, 4

# every symbol include/grl_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "grl_linear_fwd",
    "grl_linear_split_blob_bytes",
    "grl_mlp_fwd",
    "grl_mlp_blob_bytes",
    "grl_block_tail_fwd",
    "grl_proj_blob_bytes",
    "grl_tail_regs_blob_bytes",
    "grl_qkv_fwd",
    "grl_qkv_blob_bytes",
    "grl_qkv_anchor_fwd",
    "grl_qkv_anchor_lo_blob_bytes",
    "grl_qkv_anchor_blob_bytes",
    "grl_cab_conv2_fwd",
    "grl_cab_conv2_blob_bytes",
    "grl_attention_fwd",
    "grl_attention_rows_geometry_ok",
    "grl_layernorm_fwd",
    "grl_layernorm_res_fwd",
    "grl_conv3x3_fwd",
    "grl_conv3x3_num_workgroups",
    "grl_se_scale_fwd",
    "grl_gemm_tn",
    "grl_attention_bwd",
    "grl_adamw_step",
    "grl_layernorm_train_fwd",
    "grl_layernorm_bwd",
    "grl_pack_conv3x3",
    "grl_pack_linear",
    "grl_sum4",
    "grl_se_mlp_fwd",
    "grl_se_mlp_bwd",
    "grl_se_colsum",
    "grl_se_apply",
    "grl_head_planes_fwd",
    "grl_head_planes_bwd",
    "grl_cpb_table_fwd",
    "grl_cpb_table_bwd",
    "grl_debug_dirty_lds",
    "grl_abi_version",
    "grl_build_info",
]


class _Strict(C.Structure):
    """ctypes silently turns unknown keyword arguments into plain attributes (leaving the C field
    zero); refuse them instead."""

    def __init__(self, **kw):
        bad = set(kw) - {f[0] for f in self._fields_}
        if bad:
            raise TypeError(f"{type(self).__name__}: unknown field(s) {sorted(bad)}")
        super().__init__(**kw)


class GrlLinearArgs(_Strict):
    _fields_ = [
        ("a", C.c_void_p),
        ("a_dtype", C.c_int32),
        ("lda", C.c_int64),
        ("pool_df", C.c_int32),
        ("pool_H", C.c_int32),
        ("pool_W", C.c_int32),
        ("w", C.c_void_p),
        ("bias", C.c_void_p),
        ("M", C.c_int32),
        ("Npad", C.c_int32),
        ("Kpad", C.c_int32),
        ("epi", C.c_int32),
        ("gscale", C.c_void_p),
        ("ln_g", C.c_void_p),
        ("ln_b", C.c_void_p),
        ("n_real", C.c_int32),
        ("ln_eps", C.c_float),
        ("res_scale", C.c_float),
        ("resid", C.c_void_p),
        ("ldr", C.c_int64),
        ("add2", C.c_void_p),
        ("add2_dtype", C.c_int32),
        ("ldadd2", C.c_int64),
        ("add2_scale", C.c_void_p),
        ("rows_per_image", C.c_int32),
        ("a_split", C.c_int32),
        ("a_scale", C.c_float),
        ("out_scale", C.c_float),
        ("out", C.c_void_p),
        ("out_dtype", C.c_int32),
        ("ldo", C.c_int64),
        ("out_plane_stride", C.c_int64),
        ("out_lo", C.c_void_p),
        ("w_regs", C.c_void_p),
        ("a_cols", C.c_int32),
        ("a_one", C.c_int32),
        ("n_store", C.c_int32),
        ("a_gelu", C.c_int32),
        ("a16_out", C.c_void_p),
        ("lda16", C.c_int64),
    ]


class GrlMlpArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("blob", C.c_void_p),
        ("M", C.c_int32),
        ("Cpad", C.c_int32),
        ("Hpad", C.c_int32),
        ("b2", C.c_void_p),
        ("ln_g", C.c_void_p),
        ("ln_b", C.c_void_p),
        ("n_real", C.c_int32),
        ("ln_eps", C.c_float),
        ("res_scale", C.c_float),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
    ]


class GrlTailArgs(_Strict):
    _fields_ = [
        ("att", C.c_void_p),
        ("ldatt", C.c_int64),
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("cab", C.c_void_p),
        ("ldcab", C.c_int64),
        ("gate", C.c_void_p),
        ("rows_per_image", C.c_int32),
        ("pblob", C.c_void_p),
        ("pb", C.c_void_p),
        ("n1_g", C.c_void_p),
        ("n1_b", C.c_void_p),
        ("blob", C.c_void_p),
        ("M", C.c_int32),
        ("Cpad", C.c_int32),
        ("Hpad", C.c_int32),
        ("b2", C.c_void_p),
        ("n2_g", C.c_void_p),
        ("n2_b", C.c_void_p),
        ("n_real", C.c_int32),
        ("ln_eps", C.c_float),
        ("res_scale", C.c_float),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("rblob", C.c_void_p),
    ]


class GrlQkvAnchorArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("Cpad", C.c_int32),
        ("blob", C.c_void_p),
        ("nslots", C.c_int32),
        ("nanc", C.c_int32),
        ("out", C.c_void_p),
        ("out_plane_stride", C.c_int64),
        ("anc", C.c_void_p),
        ("anc_plane_stride", C.c_int64),
        ("lo_blob", C.c_void_p),
    ]


class GrlCabConv2Args(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("blob", C.c_void_p),
        ("bias", C.c_void_p),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("wgs_per_image", C.c_int32),
        ("out", C.c_void_p),
        ("ldo", C.c_int64),
        ("pool_partial", C.c_void_p),
        ("pool_stride", C.c_int64),
        ("gate", C.c_void_p),
        ("se_counter", C.c_void_p),
        ("se_w1", C.c_void_p),
        ("se_b1", C.c_void_p),
        ("se_w2", C.c_void_p),
        ("se_b2", C.c_void_p),
        ("se_c", C.c_int32),
        ("se_mid", C.c_int32),
        ("inv_hw", C.c_float),
    ]


class GrlQkvArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("blob", C.c_void_p),
        ("M", C.c_int32),
        ("Cpad", C.c_int32),
        ("nslots", C.c_int32),
        ("out", C.c_void_p),
        ("out_plane_stride", C.c_int64),
    ]


class GrlTokenGrid(_Strict):
    _fields_ = [
        ("ptr", C.c_void_p),
        ("ld", C.c_int64),
        ("hstride", C.c_int64),
        ("col0", C.c_int32),
        ("Himg", C.c_int32),
        ("Wimg", C.c_int32),
        ("wh", C.c_int32),
        ("ww", C.c_int32),
        ("shy", C.c_int32),
        ("shx", C.c_int32),
        ("transposed", C.c_int32),
    ]


class GrlAttnArgs(_Strict):
    _fields_ = [
        ("q", GrlTokenGrid),
        ("k", GrlTokenGrid),
        ("v", GrlTokenGrid),
        ("o", GrlTokenGrid),
        ("B", C.c_int32),
        ("nh", C.c_int32),
        ("nwy", C.c_int32),
        ("nwx", C.c_int32),
        ("table", C.c_void_p),
        ("trows", C.c_int32),
        ("tstride", C.c_int32),
        ("masked", C.c_int32),
        ("ones_col", C.c_int32),
        ("head_dim", C.c_int32),
        ("out_dtype", C.c_int32),
        ("k_one31", C.c_int32),
        ("lazy_floor", C.c_void_p),
        ("lse", C.c_void_p),
        ("lse_stride", C.c_int64),
        ("q_lo", C.c_void_p),
        ("k_lo", C.c_void_p),
        ("v_lo", C.c_void_p),
        ("o_lo", C.c_void_p),
        ("lazy_ceil", C.c_void_p),
    ]


class GrlConvArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("x_dtype", C.c_int32),
        ("ldx", C.c_int64),
        ("w", C.c_void_p),
        ("w_tap_stride", C.c_int64),
        ("bias", C.c_void_p),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("CinP", C.c_int32),
        ("CoutP", C.c_int32),
        ("x_split", C.c_int32),
        ("x_scale", C.c_float),
        ("out_scale", C.c_float),
        ("act", C.c_int32),
        ("slope", C.c_float),
        ("resid", C.c_void_p),
        ("ldr", C.c_int64),
        ("pool_partial", C.c_void_p),
        ("pool_stride", C.c_int64),
        ("out", C.c_void_p),
        ("out_dtype", C.c_int32),
        ("ldo", C.c_int64),
        ("shuffle_r", C.c_int32),
        ("shuffle_cg", C.c_int32),
        ("shuffle_ij0", C.c_int32),
        ("x_cols", C.c_int32),
        ("n_store", C.c_int32),
    ]


class GrlLnResArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("resid", C.c_void_p),
        ("ldr", C.c_int64),
        ("gamma", C.c_void_p),
        ("beta", C.c_void_p),
        ("add2", C.c_void_p),
        ("add2_dtype", C.c_int32),
        ("ldadd2", C.c_int64),
        ("add2_scale", C.c_void_p),
        ("rows_per_image", C.c_int32),
        ("M", C.c_int32),
        ("n_real", C.c_int32),
        ("n_pad", C.c_int32),
        ("eps", C.c_float),
        ("res_scale", C.c_float),
        ("y", C.c_void_p),
        ("ldy", C.c_int64),
    ]


class GrlGemmTnArgs(_Strict):
    _fields_ = [
        ("a", C.c_void_p),
        ("lda", C.c_int64),
        ("b", C.c_void_p),
        ("b_dtype", C.c_int32),
        ("ldb", C.c_int64),
        ("M", C.c_int32),
        ("N", C.c_int32),
        ("K", C.c_int32),
        ("taps", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("splits", C.c_int32),
        ("a_scale", C.c_float),
        ("out_scale", C.c_float),
        ("c", C.c_void_p),
        ("ldc", C.c_int64),
        ("c_tap_stride", C.c_int64),
        ("c_fix", C.c_void_p),
        ("b_ones", C.c_int32),
        ("reserved0", C.c_int32),
        ("c_bias", C.c_void_p),
        ("c_bias_fix", C.c_void_p),
        ("a_dtype", C.c_int32),
        ("reserved1", C.c_int32),
    ]


class GrlAttnBwdArgs(_Strict):
    _fields_ = [
        ("fwd", GrlAttnArgs),
        ("d_o", C.c_void_p),
        ("d_q", C.c_void_p),
        ("d_k", C.c_void_p),
        ("d_v", C.c_void_p),
        ("d_table", C.c_void_p),
        ("g_scale", C.c_float),
        ("d_table_fix", C.c_void_p),
        ("d_o_ld", C.c_int64),
    ]


class GrlAdamWArgs(_Strict):
    _fields_ = [
        ("params", C.c_void_p),
        ("grads", C.c_void_p),
        ("exp_avg", C.c_void_p),
        ("exp_avg_sq", C.c_void_p),
        ("numel", C.c_void_p),
        ("weight_decay_flags", C.c_void_p),
        ("chunk_tensor", C.c_void_p),
        ("chunk_offset", C.c_void_p),
        ("num_chunks", C.c_int32),
        ("lr", C.c_float),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("eps", C.c_float),
        ("weight_decay", C.c_float),
        ("bias_correction1", C.c_float),
        ("bias_correction2_sqrt", C.c_float),
        ("grad_scale", C.c_float),
        ("bias_corrections_dev", C.c_void_p),
        ("hyper_dev", C.c_void_p),
    ]


class GrlLnTrainArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gamma", C.c_void_p),
        ("beta", C.c_void_p),
        ("y", C.c_void_p), ("ldy", C.c_int64),
        ("mean", C.c_void_p),
        ("rstd", C.c_void_p),
        ("dy", C.c_void_p), ("lddy", C.c_int64),
        ("dx", C.c_void_p), ("lddx", C.c_int64),
        ("dgamma", C.c_void_p),
        ("dbeta", C.c_void_p),
        ("M", C.c_int32), ("n", C.c_int32),
        ("eps", C.c_float),
        ("resid", C.c_void_p), ("ldr", C.c_int64),
        ("row_scale", C.c_void_p),
        ("rows_per_image", C.c_int32),
        ("alpha", C.c_float),
        ("stat_replicas", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class GrlPlanesArgs(_Strict):
    _fields_ = [
        ("x", C.c_void_p),
        ("scale", C.c_void_p),
        ("out32", C.c_void_p),
        ("out16", C.c_void_p),
        ("dy", C.c_void_p * 8),
        ("dx", C.c_void_p),
        ("dscale", C.c_void_p),
        ("T", C.c_int32), ("S_in", C.c_int32), ("S_out", C.c_int32), ("nh", C.c_int32), ("d", C.c_int32),
        ("src", C.c_int32 * 8), ("raw", C.c_int32 * 8), ("one_col", C.c_int32 * 8), ("want_dscale", C.c_int32 * 8),
        ("dscale_replicas", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class GrlSeMlpArgs(_Strict):
    _fields_ = [
        ("pool", C.c_void_p),
        ("w1", C.c_void_p),
        ("b1", C.c_void_p),
        ("w2", C.c_void_p),
        ("b2", C.c_void_p),
        ("gate", C.c_void_p),
        ("hidden", C.c_void_p),
        ("d_gate", C.c_void_p),
        ("d_pool", C.c_void_p),
        ("d_w1", C.c_void_p),
        ("d_b1", C.c_void_p),
        ("d_w2", C.c_void_p),
        ("d_b2", C.c_void_p),
        ("B", C.c_int32), ("C", C.c_int32), ("Cmid", C.c_int32),
        ("parallel", C.c_int32),
    ]


class GrlSeRowsArgs(_Strict):
    _fields_ = [
        ("a", C.c_void_p), ("lda", C.c_int64),
        ("f", C.c_void_p), ("ldf", C.c_int64),
        ("g", C.c_void_p),
        ("h", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("k", C.c_float),
        ("M", C.c_int32), ("C", C.c_int32), ("rows_per_image", C.c_int32),
    ]


class GrlCpbArgs(_Strict):
    _fields_ = [
        ("coords", C.c_void_p),
        ("w1", C.c_void_p),
        ("b1", C.c_void_p),
        ("w2", C.c_void_p),
        ("out", C.c_void_p),
        ("d_out", C.c_void_p),
        ("d_w1", C.c_void_p),
        ("d_b1", C.c_void_p),
        ("d_w2", C.c_void_p),
        ("G", C.c_int32),
        ("rows", C.c_int32),
        ("rows4", C.c_int32),
        ("nh", C.c_int32),
        ("hidden", C.c_int32),
    ]


_lib = None


def lib():
    """Loads the library once; raises if it is missing or built against another ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the GRL HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` from the repo root. "
            "There is no CPU/PyTorch fallback for the hot path."
        )
    L = C.CDLL(LIB_PATH)
    L.grl_abi_version.restype = C.c_int
    if L.grl_abi_version() != ABI_VERSION:
        raise RuntimeError(f"stale {LIB_PATH}: ABI {L.grl_abi_version()} != {ABI_VERSION}; rebuild")
    L.grl_build_info.restype = C.c_char_p
    L.grl_linear_fwd.argtypes = [C.c_void_p, C.POINTER(GrlLinearArgs)]
    L.grl_linear_fwd.restype = C.c_int
    L.grl_linear_split_blob_bytes.argtypes = [C.c_int32, C.c_int32]
    L.grl_linear_split_blob_bytes.restype = C.c_int64
    L.grl_mlp_fwd.argtypes = [C.c_void_p, C.POINTER(GrlMlpArgs)]
    L.grl_mlp_fwd.restype = C.c_int
    L.grl_mlp_blob_bytes.argtypes = [C.c_int32, C.c_int32]
    L.grl_mlp_blob_bytes.restype = C.c_int64
    L.grl_block_tail_fwd.argtypes = [C.c_void_p, C.POINTER(GrlTailArgs)]
    L.grl_block_tail_fwd.restype = C.c_int
    L.grl_proj_blob_bytes.argtypes = [C.c_int32]
    L.grl_proj_blob_bytes.restype = C.c_int64
    L.grl_tail_regs_blob_bytes.argtypes = []
    L.grl_tail_regs_blob_bytes.restype = C.c_int64
    L.grl_qkv_fwd.argtypes = [C.c_void_p, C.POINTER(GrlQkvArgs)]
    L.grl_qkv_fwd.restype = C.c_int
    L.grl_qkv_blob_bytes.argtypes = [C.c_int32, C.c_int32]
    L.grl_qkv_blob_bytes.restype = C.c_int64
    L.grl_qkv_anchor_fwd.argtypes = [C.c_void_p, C.POINTER(GrlQkvAnchorArgs)]
    L.grl_qkv_anchor_fwd.restype = C.c_int
    L.grl_qkv_anchor_blob_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.grl_qkv_anchor_blob_bytes.restype = C.c_int64
    L.grl_qkv_anchor_lo_blob_bytes.argtypes = [C.c_int32, C.c_int32]
    L.grl_qkv_anchor_lo_blob_bytes.restype = C.c_int64
    L.grl_cab_conv2_fwd.argtypes = [C.c_void_p, C.POINTER(GrlCabConv2Args)]
    L.grl_cab_conv2_fwd.restype = C.c_int
    L.grl_cab_conv2_blob_bytes.argtypes = []
    L.grl_cab_conv2_blob_bytes.restype = C.c_int64
    L.grl_attention_fwd.argtypes = [C.c_void_p, C.POINTER(GrlAttnArgs)]
    L.grl_attention_fwd.restype = C.c_int
    L.grl_attention_rows_geometry_ok.argtypes = [C.POINTER(GrlAttnArgs)]
    L.grl_attention_rows_geometry_ok.restype = C.c_int
    L.grl_layernorm_fwd.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_int32, C.c_int32, C.c_int32, C.c_float,
    ]
    L.grl_layernorm_fwd.restype = C.c_int
    L.grl_layernorm_res_fwd.argtypes = [C.c_void_p, C.POINTER(GrlLnResArgs)]
    L.grl_layernorm_res_fwd.restype = C.c_int
    L.grl_conv3x3_fwd.argtypes = [C.c_void_p, C.POINTER(GrlConvArgs)]
    L.grl_conv3x3_fwd.restype = C.c_int
    L.grl_conv3x3_num_workgroups.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.grl_conv3x3_num_workgroups.restype = C.c_int
    L.grl_se_scale_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.grl_se_scale_fwd.restype = C.c_int
    L.grl_gemm_tn.argtypes = [C.c_void_p, C.POINTER(GrlGemmTnArgs)]
    L.grl_gemm_tn.restype = C.c_int
    L.grl_attention_bwd.argtypes = [C.c_void_p, C.POINTER(GrlAttnBwdArgs)]
    L.grl_attention_bwd.restype = C.c_int
    L.grl_adamw_step.argtypes = [C.c_void_p, C.POINTER(GrlAdamWArgs)]
    L.grl_adamw_step.restype = C.c_int
    L.grl_layernorm_train_fwd.argtypes = [C.c_void_p, C.POINTER(GrlLnTrainArgs)]
    L.grl_layernorm_train_fwd.restype = C.c_int
    L.grl_layernorm_bwd.argtypes = [C.c_void_p, C.POINTER(GrlLnTrainArgs)]
    L.grl_layernorm_bwd.restype = C.c_int
    L.grl_pack_conv3x3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.grl_pack_conv3x3.restype = C.c_int
    L.grl_pack_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.grl_pack_linear.restype = C.c_int
    L.grl_sum4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.grl_sum4.restype = C.c_int
    L.grl_se_mlp_fwd.argtypes = [C.c_void_p, C.POINTER(GrlSeMlpArgs)]
    L.grl_se_mlp_fwd.restype = C.c_int
    L.grl_se_mlp_bwd.argtypes = [C.c_void_p, C.POINTER(GrlSeMlpArgs)]
    L.grl_se_mlp_bwd.restype = C.c_int
    L.grl_se_colsum.argtypes = [C.c_void_p, C.POINTER(GrlSeRowsArgs)]
    L.grl_se_colsum.restype = C.c_int
    L.grl_se_apply.argtypes = [C.c_void_p, C.POINTER(GrlSeRowsArgs)]
    L.grl_se_apply.restype = C.c_int
    L.grl_head_planes_fwd.argtypes = [C.c_void_p, C.POINTER(GrlPlanesArgs)]
    L.grl_head_planes_fwd.restype = C.c_int
    L.grl_head_planes_bwd.argtypes = [C.c_void_p, C.POINTER(GrlPlanesArgs)]
    L.grl_head_planes_bwd.restype = C.c_int
    L.grl_cpb_table_fwd.argtypes = [C.c_void_p, C.POINTER(GrlCpbArgs)]
    L.grl_cpb_table_fwd.restype = C.c_int
    L.grl_cpb_table_bwd.argtypes = [C.c_void_p, C.POINTER(GrlCpbArgs)]
    L.grl_cpb_table_bwd.restype = C.c_int
    L.grl_debug_dirty_lds.argtypes = [C.c_void_p]
    L.grl_debug_dirty_lds.restype = C.c_int
    _lib = L
    return L


def check(code: int, what: str):
    if code != 0:
        kind = {-1: "bad argument", -2: "unsupported shape"}.get(code, f"hipError {code}")
        raise RuntimeError(f"libgrl_hip: {what} failed: {kind}")


_DIRTY_LDS = os.environ.get("GRL_DIRTY_LDS", "0") == "1"


def stream_ptr():
    """The current torch HIP stream as the `stream` argument of a C-ABI launch (every wrapper evaluates this right before its
    call).  GRL_DIRTY_LDS=1 (debug): first fills the LDS of every CU with NaN bytes on that stream, so that the kernel about to be
    launched cannot profit from what its predecessor left there (grl_debug_dirty_lds)."""
    import torch

    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if _DIRTY_LDS:
        check(lib().grl_debug_dirty_lds(s), "grl_debug_dirty_lds")
    return s
