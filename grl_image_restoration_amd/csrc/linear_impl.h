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
#pragma once
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

// Workgroup shapes: MT m-tiles (16 token rows each) per wave, WV waves per workgroup.
//   MT = 2, WV = 8  : 256 rows / workgroup, weight fragments shared by two m-tiles, <= 256 VGPRs
//   MT = 1, WV = 16 : 256 rows / workgroup, <= 128 VGPRs -> 4 waves per SIMD hide the load / epilogue latency
//                     of these HBM-bound layers better (measured), at twice the LDS fragment reads per MFMA

// hi / lo halves of a split-precision operand: hi = fp16(x), lo = fp16(x - hi)
__device__ __forceinline__ f16 split_part(float x, bool lo) {
    const f16 h = to_f16(x);
    return lo ? (f16)(x - (float)h) : h;
}

template <int KSTEPS, int MT>
__device__ __forceinline__ void load_a_slab(const GrlLinearArgs& p, int row0, int lane, gemm_x8 (&a)[MT][KSTEPS]) {
    const int r = lane & 15, kg = lane >> 4;
    // a_split == 3: virtual K = [hi | lo | hi] over a source of KS3 k-steps (KSTEPS = 3 * KS3)
    const bool split = p.a_split == 3;
    const float asc = p.a_scale != 0.0f ? p.a_scale : 1.0f;   // backward pass: gradients pre-scaled into fp16 range
    constexpr int KS3 = KSTEPS >= 3 ? KSTEPS / 3 : 1;   // (never used as a divisor of 0: instantiations with KSTEPS < 3 are never run with a_split == 3)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = row0 + 16 * mt + r;
        const bool valid = m < p.M;
        if (!valid) m = p.M - 1;
        if (p.pool_df > 1) {
            // A row = mean over a df x df block of token rows (AnchorLinear avg-pool, fp32 source)
            const int df = p.pool_df;
            const int Wa = p.pool_W / df, Ha = p.pool_H / df;
            const int xa = m % Wa, ya = (m / Wa) % Ha, b = m / (Wa * Ha);
            const float inv = 1.0f / (float)(df * df);
            const float* base = (const float*)p.a + ((int64_t)(b * p.pool_H + ya * df) * p.pool_W + xa * df) * p.lda;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int ss = split ? s % KS3 : s;
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int dy = 0; dy < df; ++dy)
                    for (int dx = 0; dx < df; ++dx) {
                        const float4* q = (const float4*)(base + ((int64_t)dy * p.pool_W + dx) * p.lda + 32 * ss + 8 * kg);
                        float4 v0 = q[0], v1 = q[1];
                        acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w;
                        acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
                    }
#pragma unroll
                for (int e = 0; e < 8; ++e) a[mt][s][e] = split_part(valid ? acc[e] * inv : 0.0f, split && s / KS3 == 1);
            }
        } else if (p.a_dtype == GRL_DT_F16) {
            const gemm_t* base = (const gemm_t*)p.a + (int64_t)m * p.lda;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                gemm_x8 v = *(const gemm_x8*)(base + 32 * s + 8 * kg);
                if (!valid) v = gemm_x8{0, 0, 0, 0, 0, 0, 0, 0};
                a[mt][s] = v;
            }
        } else {
            const float* base = (const float*)p.a + (int64_t)m * p.lda;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int ss = split ? s % KS3 : s;
                const bool lo = split && s / KS3 == 1;
                const float4* q = (const float4*)(base + 32 * ss + 8 * kg);
                float4 v0, v1;
                const int c0 = 32 * ss + 8 * kg;
                const bool in0 = p.a_cols <= 0 || c0 < p.a_cols, in1 = p.a_cols <= 0 || c0 + 4 < p.a_cols;
                v0 = in0 ? q[0] : float4{0, 0, 0, 0};
                v1 = in1 ? q[1] : float4{0, 0, 0, 0};
                if (p.a_gelu) {       // the operand is gelu(A) (fc2 of the Mlp on fc1's pre-activation); gelu(0) = 0 keeps the pad columns
                    const f32x2v g0 = gelu_erf2(f32x2v{v0.x, v0.y}), g1 = gelu_erf2(f32x2v{v0.z, v0.w});
                    const f32x2v g2 = gelu_erf2(f32x2v{v1.x, v1.y}), g3 = gelu_erf2(f32x2v{v1.z, v1.w});
                    v0 = float4{g0.x, g0.y, g1.x, g1.y}; v1 = float4{g2.x, g2.y, g3.x, g3.y};
                }
                if (p.a_cols > 0 && p.a_one) {   // operand at its real width: 1.0 in column a_cols (the other pad columns are 0)
                    if (c0 == p.a_cols) v0.x = 1.0f;
                    if (c0 + 4 == p.a_cols) v1.x = 1.0f;
                }
                if (!valid) { v0 = float4{0, 0, 0, 0}; v1 = v0; }
                gemm_x8 v;
                v[0] = split_part(v0.x * asc, lo); v[1] = split_part(v0.y * asc, lo); v[2] = split_part(v0.z * asc, lo); v[3] = split_part(v0.w * asc, lo);
                v[4] = split_part(v1.x * asc, lo); v[5] = split_part(v1.y * asc, lo); v[6] = split_part(v1.z * asc, lo); v[7] = split_part(v1.w * asc, lo);
                a[mt][s] = v;
            }
        }
    }
}

// Epilogue of one 16-row m-tile over NCH chunks of NT n-tiles held in registers.
template <int NT, int NCH, int MT, int EPI, bool ADD2, int mt>
__device__ __forceinline__ void epilogue(const GrlLinearArgs& p, f32x4 (&acc)[NCH][MT][NT], int m, bool valid, int n0, int g4) {
    if constexpr (EPI == GRL_EPI_GROUPNORM) {
        // per 32-channel group (= one attention head slot): x / max(|x|,1e-12) * |gscale[g]|;
        // gscale == 0 marks a pass-through group (v).  F.normalize eps: efficient.py:85.
#pragma unroll
        for (int g = 0; g < NT / 2; ++g) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += acc[0][mt][2 * g][e] * acc[0][mt][2 * g][e];
                ss += acc[0][mt][2 * g + 1][e] * acc[0][mt][2 * g + 1][e];
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float gs = p.gscale[(n0 >> 5) + g];
            const float f = gs != 0.0f ? fabsf(gs) / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[0][mt][2 * g][e] *= f; acc[0][mt][2 * g + 1][e] *= f; }
            if (gs < 0.0f && g4 == 3) acc[0][mt][2 * g + 1][3] = 1.0f;   // column 31 of a K plane (grl_hip.h)
        }
    } else if constexpr (EPI == GRL_EPI_GELU) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][mt][nt][e] = gelu_erf(acc[0][mt][nt][e]);
    } else if constexpr (EPI == GRL_EPI_GELU_GRAD) {
        // the product times gelu'(h) of the same element: h = resid [M, ldr] (fc1's pre-activation), columns < n_store only
        const float* hrow = p.resid + (valid ? (int64_t)m : (int64_t)p.M - 1) * p.ldr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + 16 * nt + 4 * g4;
            if (p.n_store > 0 && col >= p.n_store) continue;
            const float4 h4 = *(const float4*)(hrow + col);
            acc[0][mt][nt][0] *= gelu_grad(h4.x); acc[0][mt][nt][1] *= gelu_grad(h4.y);
            acc[0][mt][nt][2] *= gelu_grad(h4.z); acc[0][mt][nt][3] *= gelu_grad(h4.w);
        }
    } else if constexpr (EPI == GRL_EPI_LN_RES) {
        // LayerNorm over the n_real real channels (eps 1e-5), then residual (+ gated extra branch).
        // Branch-free and batched: out-of-range rows are clamped (only the store is predicated) and the
        // residual / extra-branch loads of a whole chunk are issued before they are consumed.
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) s1 += (16 * (c * NT + nt) + 4 * g4 + e) < p.n_real ? acc[c][mt][nt][e] : 0.f;
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 / (float)p.n_real;
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc[c][mt][nt][e] - mean;
                    s2 += (16 * (c * NT + nt) + 4 * g4 + e) < p.n_real ? d * d : 0.f;
                }
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 / (float)p.n_real + p.ln_eps);
        const int64_t mc = valid ? m : (int64_t)p.M - 1;
        const float* rrow = p.resid + mc * p.ldr;
        const gemm_t* arow = ADD2 ? (const gemm_t*)p.add2 + mc * p.ldadd2 : nullptr;
        const float* grow = ADD2 ? p.add2_scale + (mc / p.rows_per_image) * p.Npad : nullptr;
        constexpr int SB = 4;  // n-tiles per load batch: 4 x (16 + 16 + 8) B per lane in flight, ~40 VGPRs
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int nb = 0; nb < NT; nb += SB) {
                // keep the load batches apart: hoisting all of them would need > 256 VGPRs
                __builtin_amdgcn_sched_barrier(0);
                float4 res[SB], gate[SB];
                gemm_x4 ext[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (nb + j < NT) {
                        const int col = 16 * (c * NT + nb + j) + 4 * g4;
                        res[j] = *(const float4*)(rrow + col);
                        if constexpr (ADD2) {
                            ext[j] = *(const gemm_x4*)(arow + col);
                            gate[j] = *(const float4*)(grow + col);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (nb + j < NT) {
                        const int nt = nb + j;
                        const int col = 16 * (c * NT + nt) + 4 * g4;
                        const float4 g = *(const float4*)(p.ln_g + col);
                        const float4 bb = *(const float4*)(p.ln_b + col);
                        float y[4];
                        y[0] = res[j].x + p.res_scale * ((acc[c][mt][nt][0] - mean) * rstd * g.x + bb.x);
                        y[1] = res[j].y + p.res_scale * ((acc[c][mt][nt][1] - mean) * rstd * g.y + bb.y);
                        y[2] = res[j].z + p.res_scale * ((acc[c][mt][nt][2] - mean) * rstd * g.z + bb.z);
                        y[3] = res[j].w + p.res_scale * ((acc[c][mt][nt][3] - mean) * rstd * g.w + bb.w);
                        if constexpr (ADD2) {  // CAB branch times its squeeze-excite gate (per image)
                            y[0] += (float)ext[j][0] * gate[j].x;
                            y[1] += (float)ext[j][1] * gate[j].y;
                            y[2] += (float)ext[j][2] * gate[j].z;
                            y[3] += (float)ext[j][3] * gate[j].w;
                        }
                        float4 o4;
                        o4.x = (col + 0) < p.n_real ? y[0] : 0.f;  // keep pad channels 0
                        o4.y = (col + 1) < p.n_real ? y[1] : 0.f;
                        o4.z = (col + 2) < p.n_real ? y[2] : 0.f;
                        o4.w = (col + 3) < p.n_real ? y[3] : 0.f;
                        if (valid) *(float4*)((float*)p.out + (int64_t)m * p.ldo + n0 + col) = o4;  // LN output is fp32
                    }
                }
            }
        }
        return;
    }
    if (p.out_dtype != GRL_DT_F32) {
        // 16-bit outputs: a lane owns 4 channels (8 B) of tile nt and of tile nt+1.  The lane pairs
        // (g4, g4^1) -- 16 lanes apart -- swap one of the two so that every lane holds 8 CONSECUTIVE
        // channels and issues one 16-B store per tile pair: 64 B contiguous per token instead of two
        // instructions of 32-B pieces (8-B-per-lane stores were the bottleneck of these epilogues).
        // All lanes take part in the exchange; only the store is predicated.
        const bool odd = g4 & 1;
        // pass 1 (out_lo given): the rounding residual v - fp16(v) of every value, same layout: the low half of a split-precision
        // operand for a consumer that contracts hi + lo (the attention kernel in precision "high")
        for (int pass = 0; pass < (p.out_lo != nullptr ? 2 : 1); ++pass) {
            f16* obase = (f16*)(pass ? p.out_lo : p.out);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int nt = 0; nt < NT; nt += 2) {
                    float v[8] = {acc[c][mt][nt][0], acc[c][mt][nt][1], acc[c][mt][nt][2], acc[c][mt][nt][3],
                                  acc[c][mt][nt + 1][0], acc[c][mt][nt + 1][1], acc[c][mt][nt + 1][2], acc[c][mt][nt + 1][3]};
                    if (pass) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] -= (float)to_f16(v[e]);
                    }
                    uint2 lo, hi;  // this lane's 4 channels of tile nt / tile nt+1
                    lo.x = pack_f16(v[0], v[1]);
                    lo.y = pack_f16(v[2], v[3]);
                    hi.x = pack_f16(v[4], v[5]);
                    hi.y = pack_f16(v[6], v[7]);
                    const uint2 send = odd ? lo : hi;       // even lanes keep tile nt, odd lanes keep tile nt+1
                    uint2 recv;
                    recv.x = __shfl_xor(send.x, 16, 64);
                    recv.y = __shfl_xor(send.y, 16, 64);
                    const uint4 w4 = odd ? uint4{recv.x, recv.y, hi.x, hi.y} : uint4{lo.x, lo.y, recv.x, recv.y};
                    // even lane g4: channels 4*g4 .. 4*g4+7 of tile nt; odd lane: channels 4*(g4-1) .. of tile nt+1
                    const int col = n0 + 16 * (c * NT + nt + (odd ? 1 : 0)) + 4 * (g4 & ~1);
                    const int64_t off = p.out_plane_stride > 0 ? (int64_t)(col >> 5) * p.out_plane_stride + (int64_t)m * 32 + (col & 31)
                                                               : (int64_t)m * p.ldo + col;
                    if (valid) *(uint4*)(obase + off) = w4;
                }
        }
        return;
    }
    if (!valid) return;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + 16 * (c * NT + nt) + 4 * g4;
            if (p.n_store > 0 && col >= p.n_store) continue;   // result at its real width
            *(float4*)((float*)p.out + (int64_t)m * p.ldo + col) =
                float4{acc[c][mt][nt][0], acc[c][mt][nt][1], acc[c][mt][nt][2], acc[c][mt][nt][3]};
        }
}

// Persistent, weights-resident kernel.  A workgroup copies the whole [Npad][KPAD] weight matrix into
// LDS once (rows padded by 16 B: conflict-free ds_read_b128 fragment reads) and then walks row tiles
// of ROWS_PER_WG tokens with stride gridDim.x.  After the initial barrier the waves never synchronise
// again: each wave loads the A slab of its 32 rows, sweeps the output channels in chunks of NT n-tiles
// (accumulators of NCH chunks are kept when the LayerNorm epilogue needs the whole row), runs the
// epilogue and stores -- so loads, MFMAs and stores of the 8 waves of a CU overlap freely and the
// weights are fetched from L2 once per CU instead of once per 128 rows.
template <int KSTEPS, int NT, int NCH, int EPI, bool ADD2, int MT, int WV>
__global__ __launch_bounds__(WV * 64) void linear_kernel(GrlLinearArgs p) {
    constexpr int WAVES = WV, ROWS_PER_WAVE = 16 * MT, ROWS_PER_WG = WV * 16 * MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KPAD = KSTEPS * 32;
    constexpr int ROWB = KPAD * 2 + 16;  // padded LDS row (bytes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;

    {   // ---- weights -> LDS, GRP x 16 B global loads in flight per thread ----
        constexpr int SEGS_PER_ROW = KPAD / 8;
        constexpr int GRP = 8;
        const int segs = p.Npad * SEGS_PER_ROW;
        const gemm_t* wsrc = (const gemm_t*)p.w;
        for (int i0 = tid; i0 < segs; i0 += GRP * WAVES * 64) {
            gemm_x8 wv[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                const int i = i0 + j * WAVES * 64;
                if (i < segs) wv[j] = *(const gemm_x8*)(wsrc + (int64_t)i * 8);
            }
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                const int i = i0 + j * WAVES * 64;
                if (i < segs) *(gemm_x8*)(smem + (i / SEGS_PER_ROW) * ROWB + (i % SEGS_PER_ROW) * 16) = wv[j];
            }
        }
    }
    __syncthreads();

    const int ntiles = (p.M + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const int ngroups = p.Npad / (NT * 16 * NCH);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS_PER_WG + wave * ROWS_PER_WAVE;
        if (row0 >= p.M) continue;
        gemm_x8 a[MT][KSTEPS];
        load_a_slab<KSTEPS, MT>(p, row0, lane, a);
        if (p.a16_out != nullptr) {   // the fp16 operand as the layer's weight-gradient GEMM will want it (64 contiguous bytes per row and k-step)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = row0 + 16 * mt + r16;
                if (m < p.M) {
#pragma unroll
                    for (int s = 0; s < KSTEPS; ++s)
                        *(gemm_x8*)((gemm_t*)p.a16_out + (int64_t)m * p.lda16 + 32 * s + 8 * g4) = a[mt][s];
                }
            }
        }
        // opaque per-iteration copy of the lane's column group: keeps the compiler from hoisting the
        // (tile-invariant) bias / gamma / beta / column addresses out of the persistent loop, where
        // ~150 live address registers force the accumulators into scratch
        int g4i = g4;
        asm volatile("" : "+v"(g4i));
        for (int gch = 0; gch < ngroups; ++gch) {
            f32x4 acc[NCH][MT][NT];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int n0 = (gch * NCH + c) * NT * 16;
                const char* wbase = smem + n0 * ROWB;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[c][mt][nt] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const gemm_x8 w = *(const gemm_x8*)(wbase + (nt * 16 + r16) * ROWB + (32 * s + 8 * g4) * 2);
                        // D^T tile: rows = output channel (A operand = W), cols = token (B operand = A slab)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[c][mt][nt] = mfma16_gemm(w, a[mt][s], acc[c][mt][nt]);
                    }
                }
                // bias: lane holds channels n0 + 16*nt + 4*g4 + [0..3] of token row0 + 16*mt + r16
                const float osc = p.out_scale != 0.0f ? p.out_scale : 1.0f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 b4 = *(const float4*)(p.bias + n0 + 16 * nt + 4 * g4i);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[c][mt][nt][0] = fmaf(acc[c][mt][nt][0], osc, b4.x); acc[c][mt][nt][1] = fmaf(acc[c][mt][nt][1], osc, b4.y);
                        acc[c][mt][nt][2] = fmaf(acc[c][mt][nt][2], osc, b4.z); acc[c][mt][nt][3] = fmaf(acc[c][mt][nt][3], osc, b4.w);
                    }
                }
            }
            // the m-tile index is a template argument: a run-time index would push `acc` into scratch.
            // sched_barrier: the epilogue's loads must not be hoisted above the MFMA loop (spills)
            __builtin_amdgcn_sched_barrier(0);
            epilogue<NT, NCH, MT, EPI, ADD2, 0>(p, acc, row0 + r16, row0 + r16 < p.M, gch * NCH * NT * 16, g4i);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MT == 2) {
                epilogue<NT, NCH, MT, EPI, ADD2, 1>(p, acc, row0 + 16 + r16, row0 + 16 + r16 < p.M, gch * NCH * NT * 16, g4i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

constexpr size_t LDS_BUDGET = 160 * 1024;

template <int KSTEPS, int NT, int NCH, int EPI, bool ADD2, int MT, int WV>
int launch_shape(const GrlLinearArgs& p, hipStream_t st) {
    constexpr int ROWS_PER_WG = WV * 16 * MT;
    const size_t lds = (size_t)p.Npad * (KSTEPS * 64 + 16);
    if (lds > LDS_BUDGET) return GRL_ERR_UNSUPPORTED;
    const int ntiles = (p.M + ROWS_PER_WG - 1) / ROWS_PER_WG;
    static const int cap = getenv("GRL_PERSIST_GRID") ? atoi(getenv("GRL_PERSIST_GRID")) : 256;  // tuning knob
    const int grid = ntiles < cap ? ntiles : cap;  // one persistent workgroup per CU
    auto kfn = linear_kernel<KSTEPS, NT, NCH, EPI, ADD2, MT, WV>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WV * 64), lds, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

template <int KSTEPS, int NT, int NCH, int EPI, bool ADD2 = false>
int launch_one(const GrlLinearArgs& p, hipStream_t st) {
    // exactly one workgroup shape per instantiation (if / else chain: every extra shape is another ~110 kernels to compile):
    if constexpr (KSTEPS >= 18) {
        // split-precision layers: the operand slab alone is 4 * KSTEPS VGPRs -> 8 waves (256 VGPRs each)
        return launch_shape<KSTEPS, NT, NCH, EPI, ADD2, 1, 8>(p, st);
    } else if constexpr (EPI == GRL_EPI_LN_RES || KSTEPS >= 12) {
        // LayerNorm epilogues need ~150 VGPRs, K = 384 slabs 48 + the fragment reads in flight -> 12 waves (3 per SIMD,
        // 168 VGPRs; at 16 waves these spill 70-300 registers)
        return launch_shape<KSTEPS, NT, NCH, EPI, ADD2, 1, 12>(p, st);
    } else {
        return launch_shape<KSTEPS, NT, NCH, EPI, ADD2, 1, 16>(p, st);
    }
}

// chunk = NT n-tiles (NT*16 output channels) swept per pass over the A slab; LayerNorm needs the whole
// row: NCH = Npad / (NT*16) chunks of accumulators in registers.
template <int KSTEPS>
int launch_k(const GrlLinearArgs& p, hipStream_t st) {
    const int tiles = p.Npad / 16;
    if (p.epi == GRL_EPI_LN_RES) {
        switch (tiles) {
            case 12: return p.add2 ? launch_one<KSTEPS, 12, 1, GRL_EPI_LN_RES, true>(p, st) : launch_one<KSTEPS, 12, 1, GRL_EPI_LN_RES>(p, st);
            case 8: return p.add2 ? launch_one<KSTEPS, 8, 1, GRL_EPI_LN_RES, true>(p, st) : launch_one<KSTEPS, 8, 1, GRL_EPI_LN_RES>(p, st);
            case 4: return p.add2 ? launch_one<KSTEPS, 4, 1, GRL_EPI_LN_RES, true>(p, st) : launch_one<KSTEPS, 4, 1, GRL_EPI_LN_RES>(p, st);
            default: return GRL_ERR_UNSUPPORTED;
        }
    }
    int nt = 0;
    const int cand[3] = {6, 4, 8};
    for (int i = 0; i < 3; ++i)
        if (tiles % cand[i] == 0) { nt = cand[i]; break; }
#define GRL_LIN_CASE(NTV)                                                                         \
    case NTV:                                                                                     \
        switch (p.epi) {                                                                          \
            case GRL_EPI_PLAIN: return launch_one<KSTEPS, NTV, 1, GRL_EPI_PLAIN>(p, st);          \
            case GRL_EPI_GELU: return launch_one<KSTEPS, NTV, 1, GRL_EPI_GELU>(p, st);            \
            case GRL_EPI_GELU_GRAD: return launch_one<KSTEPS, NTV, 1, GRL_EPI_GELU_GRAD>(p, st);  \
            case GRL_EPI_GROUPNORM: return launch_one<KSTEPS, NTV, 1, GRL_EPI_GROUPNORM>(p, st);  \
            default: return GRL_ERR_UNSUPPORTED;                                                  \
        }
    switch (nt) {
        GRL_LIN_CASE(6)
        GRL_LIN_CASE(4)
        GRL_LIN_CASE(8)
        default: return GRL_ERR_UNSUPPORTED;
    }
#undef GRL_LIN_CASE
}

template <int KSTEPS>
int launch_split(const GrlLinearArgs& p0, hipStream_t st) {
    // weight matrices larger than the LDS budget are processed as several column slabs (whole 32-column
    // groups, so head planes / group norms stay intact); the activations are re-read per slab
    const size_t rowb = KSTEPS * 64 + 16;
    const int max_rows = (int)(LDS_BUDGET / rowb) / 32 * 32;
    if (max_rows <= 0) return GRL_ERR_UNSUPPORTED;
    if ((size_t)p0.Npad * rowb <= LDS_BUDGET) return launch_k<KSTEPS>(p0, st);
    if (p0.epi == GRL_EPI_LN_RES) return GRL_ERR_UNSUPPORTED;
    // equal slabs of whole 32-column groups whose n-tile count launch_k can chunk (multiples of 64 or 96 columns)
    int nslabs = (p0.Npad + max_rows - 1) / max_rows;
    for (;; ++nslabs) {
        if (nslabs > p0.Npad / 32) return GRL_ERR_UNSUPPORTED;
        if (p0.Npad % nslabs) continue;
        const int nc = p0.Npad / nslabs;
        if (nc <= max_rows && (nc % 64 == 0 || nc % 96 == 0)) break;
    }
    const int ncol = p0.Npad / nslabs;
    for (int sidx = 0; sidx < nslabs; ++sidx) {
        GrlLinearArgs p = p0;
        const int c0 = sidx * ncol;
        p.Npad = ncol;
        if (sidx > 0) p.a16_out = nullptr;   // (every slab reads the same A: one copy)
        if (p0.n_store > 0) {   // this slab's share of the real output columns
            const int ns = p0.n_store - c0;
            if (ns <= 0) continue;
            p.n_store = ns >= ncol ? 0 : ns;
        }
        p.w = (const char*)p0.w + (size_t)c0 * KSTEPS * 32 * 2;
        p.bias = p0.bias + c0;
        if (p0.epi == GRL_EPI_GELU_GRAD) p.resid = p0.resid + c0;
        if (p0.gscale) p.gscale = p0.gscale + c0 / 32;
        const size_t esz = p0.out_dtype == GRL_DT_F32 ? 4 : 2;
        if (p0.out_plane_stride > 0) p.out = (char*)p0.out + (size_t)(c0 / 32) * p0.out_plane_stride * esz;
        else p.out = (char*)p0.out + (size_t)c0 * esz;
        if (p0.out_lo != nullptr)
            p.out_lo = (char*)p0.out_lo + (p0.out_plane_stride > 0 ? (size_t)(c0 / 32) * p0.out_plane_stride : (size_t)c0) * esz;
        const int rc = launch_k<KSTEPS>(p, st);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace
