// Token-wise linear layers of the GRL block as one MFMA kernel family (gfx950).
//
//   out[m, :] = epilogue( A[m, :] . W^T + bias )          M = B*H*W tokens (huge), N,K <= 576
//
// Replaces, per SURVEY 8(a): P1 QKVProjection (mixed_attn_block.py:661-676), P2 AnchorLinear
// (:714-736, avg-pool fused into the A load), M1 proj + B1 norm1/residual
// (mixed_attn_block_efficient.py:379,543-548), F1 Mlp (swin_v1_block.py:37-43) + norm2/residual
// (efficient.py:554).
//
// Design (MI355X): these GEMMs are HBM-bound (108 FLOP/B for QKV), so the kernel is built
// around one pass over the activations:
//   * a wave owns 32 token rows; its whole A slab (32 x Kpad, fp16 operands) lives in VGPRs and is
//     re-used for every N chunk, so A is read from HBM exactly once;
//   * weights (fp16, zero-padded to [Npad][Kpad]) stream through LDS in chunks of NT*16 rows,
//     padded by 16 B per row so the ds_read_b128 fragment reads are bank-conflict free;
//   * the product is computed transposed (D^T = W . A^T, mfma_f32_16x16x32_f16) so every
//     lane ends up with 4 consecutive output channels of one token: row reductions for
//     LayerNorm / per-head L2 normalisation need only two cross-lane steps and stores are
//     8-16 B wide;
//   * epilogues fuse bias, exact GELU, cosine-attention q/k normalisation + logit scale,
//     LayerNorm + residual (+ the CAB branch) so no intermediate goes back to HBM.
#include "linear_impl.h"

// The kernel instantiations are spread over three translation units by K (linear.hip: K <= 256, linear_k576.hip: K = 384 / 576,
// linear_k1152.hip: K = 768 / 1152) so that they compile in parallel: one file took 8 minutes, then 2, now under one.
int grl_linear_launch_k576(const GrlLinearArgs& p, hipStream_t st);
int grl_linear_launch_k1152(const GrlLinearArgs& p, hipStream_t st);
// csrc/linear_split.hip: weights-stationary kernel of the split-precision layers (GrlLinearArgs.w_regs)
int grl_linear_split_launch(const GrlLinearArgs& p, hipStream_t st);

extern "C" int grl_linear_fwd(void* stream, const GrlLinearArgs* args) {
    const GrlLinearArgs& p = *args;
    if (p.M <= 0) return 0;
    if (p.Kpad % 32 != 0 || p.Npad % 32 != 0 || p.lda % (p.a_cols > 0 ? 4 : 8) != 0 || p.ldo % 4 != 0) return GRL_ERR_BAD_ARG;
    if (p.a_cols < 0 || p.n_store < 0) return GRL_ERR_BAD_ARG;
    if (p.a16_out != nullptr && (p.a_dtype != GRL_DT_F32 || p.a_split == 3 || p.pool_df > 1 || (p.lda16 % 8) || p.lda16 < p.Kpad || p.w_regs != nullptr))
        return GRL_ERR_BAD_ARG;
    if (p.a_cols > 0 && ((p.a_cols % 4) || p.a_cols > p.Kpad || p.lda < p.a_cols || p.a_dtype != GRL_DT_F32 || p.pool_df > 1 || p.a_split == 3 ||
                         (p.a_one && p.a_cols >= p.Kpad)))
        return GRL_ERR_BAD_ARG;
    if (p.n_store > 0 && ((p.n_store % 4) || p.n_store > p.Npad || p.ldo < p.n_store || p.out_dtype != GRL_DT_F32 || p.out_plane_stride > 0 ||
                          (p.epi != GRL_EPI_PLAIN && p.epi != GRL_EPI_GELU && p.epi != GRL_EPI_GELU_GRAD)))
        return GRL_ERR_BAD_ARG;
    if (p.epi == GRL_EPI_GELU_GRAD && (p.resid == nullptr || (p.ldr % 4) || p.out_dtype != GRL_DT_F32 || p.out_plane_stride > 0 ||
                                       (p.n_store == 0 && p.ldr < p.Npad) || (p.n_store > 0 && p.ldr < p.n_store)))
        return GRL_ERR_BAD_ARG;
    if (p.a_gelu && (p.a_dtype != GRL_DT_F32 || p.pool_df > 1 || p.a_split == 3)) return GRL_ERR_BAD_ARG;
    if (p.out_dtype != GRL_DT_F32 && p.out_plane_stride <= 0 && (p.ldo % 8) != 0) return GRL_ERR_BAD_ARG;  // 16-B stores
    if (p.epi == GRL_EPI_LN_RES && (p.Npad > 192 || p.n_real > p.Npad || p.resid == nullptr)) return GRL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (p.a_split == 3 && ((p.Kpad / 32) % 3 != 0 || p.a_dtype != GRL_DT_F32)) return GRL_ERR_BAD_ARG;
    if (p.a_split != 0 && p.a_split != 1 && p.a_split != 3) return GRL_ERR_BAD_ARG;
    if (p.out_lo != nullptr && p.out_dtype != GRL_DT_F16) return GRL_ERR_BAD_ARG;
    if (p.add2 != nullptr && (p.add2_scale == nullptr || p.rows_per_image <= 0)) return GRL_ERR_BAD_ARG;
    if (p.a_split == 3 && p.w_regs != nullptr && !getenv("GRL_LINEAR_SPLIT_GENERIC")) {   // weights-stationary split kernel (csrc/linear_split.hip)
        const int rc = grl_linear_split_launch(p, st);
        if (rc != GRL_ERR_UNSUPPORTED) return rc;
    }
    if (p.add2 != nullptr && p.add2_dtype != GRL_DT_F16) return GRL_ERR_UNSUPPORTED;   // (the generic LN_RES epilogue reads a 16-bit extra branch)
    switch (p.Kpad / 32) {
        case 2: return launch_split<2>(p, st);
        case 3: return launch_split<3>(p, st);
        case 4: return launch_split<4>(p, st);
        case 6: return launch_split<6>(p, st);
        case 8: return launch_split<8>(p, st);
        case 12: case 18: return grl_linear_launch_k576(p, st);    // 18, 24, 36: split-precision operands of the 192-, 256-, 384-wide layers
        case 24: case 36: return grl_linear_launch_k1152(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}
