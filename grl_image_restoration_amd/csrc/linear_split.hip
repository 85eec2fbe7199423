// Split-precision token-wise linear layers with the weights stationary in registers (gfx950, round 4): the a_split == 3 path
// of grl_linear_fwd -- every linear layer of precision "high": QKVProjection (mixed_attn_block.py:661-676), MixedAttention.proj
// (mixed_attn_block_efficient.py:379), Mlp fc1 / fc2 (swin_v1_block.py:37-43).
//
//     out[m, :] = epilogue( a_hi[m] . W_hi^T + a_lo[m] . W_hi^T + a_hi[m] . W_lo^T + bias ),   a_hi = fp16(a), a_lo = fp16(a - a_hi)
//
// Why: the generic kernel (csrc/linear_impl.h) keeps the A slab of 16 tokens in registers and streams W through LDS -- with the
// virtual K of 576 / 1152 of the split form the weight matrix no longer fits (6 column slabs, A re-read and re-split six
// times), every 16x16x32 MFMA needs an LDS fragment read of its own (256 B per cycle and CU asked of a 128 B pipe) and the
// kernels spill 20-50 registers: 176 TFLOP/s on the 3x work, 63 % of a GRL-Base deblur forward at checkpoint-like logit scales.
// Here, as in csrc/tail_regs.hip:
//   * a workgroup = 6 compute waves + 2 loader waves and owns a slab of 192 output columns: compute wave w holds the hi and lo
//     A fragments (32x32x16) of columns 32 w .. 32 w + 31 of the slab for the whole K in registers (K = 192: 96 VGPRs, K = 384: 192);
//   * the activations pass through LDS once per slab as 32-token tiles, already split: the loader waves read the fp32 rows
//     two tiles ahead into registers, convert to the hi and the lo plane and write them one tile ahead (double buffer,
//     ONE barrier per tile);
//   * per k-step of 16 a compute wave reads two B fragments (hi, lo: 2 KB) for three MFMAs (96 cycles): 85 B per cycle and CU;
//   * a lane ends up with 16 channels of ONE token: the per-head L2 normalisation of the QKV epilogue is a lane-pair sum,
//     stores are 16 B wide;
//   * the slabs of one token tile run on the same XCD at the same time (blockIdx -> (XCD, slab, walker)), so the tile is
//     fetched from HBM once and from that XCD's L2 by the other slabs.
// Weights: GrlLinearArgs.w_regs (ops.pack_linear_split), layout in include/grl_hip.h.
#include "common.h"
#include "grl_hip_internal.h"
#include "attn_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int LS_T = 32, LS_W = 8, LS_CW = 6, LS_THREADS = LS_W * 64, LS_SLAB = 32 * LS_CW;

template <int KS>
struct LsGeom {
    static constexpr int K = KS * 16;
    static constexpr int ROW = K * 2 + 16;          // fp16 plane row (16 B pad: conflict-free ds_read_b128)
    static constexpr int PLANE = LS_T * ROW;
    static constexpr int BUF = 2 * PLANE;           // hi | lo
    static constexpr int OFF_VEC = 2 * BUF;         // 3 x [192] fp32: bias of the slab | LayerNorm gamma | beta
    static constexpr int OFF_ST = OFF_VEC + 3 * LS_SLAB * 4;   // [6 waves][32 tokens] (mean, M2) of the LayerNorm epilogue
    static constexpr int LDS = OFF_ST + LS_CW * LS_T * 8;
    static constexpr int NLD = K * LS_T / 4 / 128;  // 16-B loads per loader lane and tile
    static constexpr int SEGS = K / 4;              // 16-B segments per fp32 row
};

__device__ __forceinline__ int ls_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

// LN: the LayerNorm + residual epilogue (an instance of its own: its extra barrier and live values stay out of the others)
template <int KS, bool LN>
__global__ __launch_bounds__(LS_THREADS) void linear_split_kernel(GrlLinearArgs p, int nslabs, int walkers) {
    using G = LsGeom<KS>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // blockIdx -> (XCD x, slab, walker): the slabs of a walker's tiles share an XCD (block b runs on XCD b % 8)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slab = q % nslabs, walker = xcd + 8 * (q / nslabs);
    const int ntiles = (p.M + LS_T - 1) / LS_T;
    if (walker >= ntiles) return;
    const int nmine = (ntiles - walker + walkers - 1) / walkers;     // tiles walker, walker + walkers, ...
    const int nt = LS_CW * slab + wave;                              // 32-column tile of this compute wave
    float* vec = (float*)(smem + G::OFF_VEC);
    {
        const int i = 64 * wave + ls_lane();
        if (i < LS_SLAB) vec[i] = LS_SLAB * slab + i < p.Npad ? p.bias[LS_SLAB * slab + i] : 0.f;
        if (LN && i < LS_SLAB) {   // (one slab; pad channels: zero affine, they stay exactly 0)
            vec[LS_SLAB + i] = i < p.n_real ? p.ln_g[i] : 0.f;
            vec[2 * LS_SLAB + i] = i < p.n_real ? p.ln_b[i] : 0.f;
        }
    }
    constexpr bool ln = LN;   // a second barrier per tile: the row statistics cross the waves

    if (wave >= LS_CW) {
        // ---------------- loader waves: fp32 rows -> hi / lo planes, two tiles ahead in registers ----------------
        const int lid = 64 * (wave - LS_CW) + ls_lane();             // 0 .. 127
        // D tiles ahead in registers (K = 384: one -- two would be 192 registers of loads in flight)
        constexpr int D = KS >= 24 ? 1 : 2;
        f32x4 nb[D][G::NLD];
        // load k = GK g + m of a tile: 16-B segment (128 m + lid) of a group of RG rows -- per lane GK (global, LDS) offset pairs,
        // the group moves through compile-time offsets (left to itself the compiler keeps 2 x NLD row / segment pairs alive)
        constexpr int GK = G::SEGS % 3 == 0 ? 3 : 1, RG = GK * 128 / G::SEGS;
        static_assert(G::NLD % GK == 0 && GK * 128 % G::SEGS == 0 && RG * (G::NLD / GK) == LS_T, "loader map");
        int64_t go[GK];
        uint32_t lo_[GK];
#pragma unroll
        for (int m = 0; m < GK; ++m) {
            const int idx = 128 * m + lid, r = idx / G::SEGS, seg = idx - r * G::SEGS;
            go[m] = (int64_t)r * p.lda + 4 * seg;
            lo_[m] = (uint32_t)(r * G::ROW + 8 * seg);
        }
        const int64_t gstep = (int64_t)RG * p.lda;
        auto issue = [&](int it, auto set) {
            constexpr int S = decltype(set)::value;
            const float* a0 = (const float*)p.a + (int64_t)(walker + (int64_t)it * walkers) * LS_T * p.lda;
#pragma unroll
            for (int k = 0; k < G::NLD; ++k) nb[S][k] = *(const f32x4*)(a0 + go[k % GK] + (k / GK) * gstep);
        };
        auto put = [&](int buf, auto set) {
            constexpr int S = decltype(set)::value;
            char* base = smem + buf * G::BUF;
#pragma unroll
            for (int k = 0; k < G::NLD; ++k) {
                const f32x4 v = nb[S][k];
                const f16 h0 = to_f16(v[0]), h1 = to_f16(v[1]), h2 = to_f16(v[2]), h3 = to_f16(v[3]);
                uint2 hi, lo;
                hi.x = __builtin_bit_cast(uint32_t, f16x2{h0, h1});
                hi.y = __builtin_bit_cast(uint32_t, f16x2{h2, h3});
                lo.x = pack_f16_raw(v[0] - (float)h0, v[1] - (float)h1);
                lo.y = pack_f16_raw(v[2] - (float)h2, v[3] - (float)h3);
                char* d = base + lo_[k % GK] + (k / GK) * RG * G::ROW;
                *(uint2*)d = hi;
                *(uint2*)(d + G::PLANE) = lo;
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, D - 1>;
        issue(0, S0{});
        if (D == 2 && nmine > 1) issue(1, S1{});
        put(0, S0{});
        if (nmine > D) issue(D, S0{});
        for (int it = 0; it < nmine; it += 2) {
            // barrier A of tile it: buffer 0 is complete, the compute waves are done with buffer 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (it + 1 < nmine) put(1, S1{});
            if (it + 1 + D < nmine) issue(it + 1 + D, S1{});
            if (ln) __builtin_amdgcn_s_barrier();
            if (it + 1 >= nmine) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (it + 2 < nmine) put(0, S0{});
            if (it + 2 + D < nmine) issue(it + 2 + D, S0{});
            if (ln) __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---------------- compute waves ----------------
    const bool live = 32 * nt < p.Npad;       // (a slab may be partly empty: the wave still takes part in the barriers)
    f16x8 Wh[KS], Wl[KS];
    {
        const f16x8* src = (const f16x8*)p.w_regs + ((int64_t)(LS_CW * slab + wave) * 2 * KS) * 64 + ls_lane();
#pragma unroll
        for (int s = 0; s < KS; ++s) { Wh[s] = src[s * 64]; Wl[s] = src[(KS + s) * 64]; }
    }
    const float gs = p.epi == GRL_EPI_GROUPNORM && live ? p.gscale[nt] : 0.f;
    for (int it = 0; it < nmine; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!live) {
            if (ln) __builtin_amdgcn_s_barrier();
            continue;
        }
        const int lid = ls_lane(), j = lid & 31, half = lid >> 5;
        const int64_t m = (int64_t)(walker + (int64_t)it * walkers) * LS_T + j;
        // LayerNorm epilogue: the residual (and the gated extra branch) of this lane's 16 channels, requested before the k loop
        // (K >= 256: after it -- no registers to park them in)
        constexpr bool EARLY = KS <= 12;
        float4 rr[4], aa[4];
        auto ln_loads = [&] {
            const float* rrow = p.resid + m * p.ldr + 32 * wave + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) rr[g] = *(const float4*)(rrow + 8 * g);
            if (p.add2 != nullptr) {
                const float* grow = p.add2_scale + (int64_t)((walker + (int64_t)it * walkers) * LS_T / p.rows_per_image) * p.Npad + 32 * wave + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 e;
                    if (p.add2_dtype == GRL_DT_F32) {
                        e = *(const float4*)((const float*)p.add2 + m * p.ldadd2 + 32 * wave + 4 * half + 8 * g);
                    } else {
                        const f16x4 h = *(const f16x4*)((const f16*)p.add2 + m * p.ldadd2 + 32 * wave + 4 * half + 8 * g);
                        e = float4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    }
                    const float4 gt = *(const float4*)(grow + 8 * g);
                    aa[g] = float4{e.x * gt.x, e.y * gt.y, e.z * gt.z, e.w * gt.w};
                }
            }
        };
        if constexpr (LN && EARLY) ln_loads();
        const char* bh = smem + (it & 1) * G::BUF + j * G::ROW + 16 * half;
        const char* bl = bh + G::PLANE;
        f32x16 acc;
        {
            const float* bv = vec + 32 * wave + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *(const float4*)(bv + 8 * g);
                acc[4 * g] = t.x; acc[4 * g + 1] = t.y; acc[4 * g + 2] = t.z; acc[4 * g + 3] = t.w;
            }
        }
        // k loop, B fragments two steps ahead (one accumulator chain cannot hide an LDS round trip)
        // (K = 384: one step ahead -- 192 of the 256 registers hold weights)
        constexpr int AHEAD = KS >= 24 ? 1 : 2;
        f16x8 h0 = *(const f16x8*)bh, l0 = *(const f16x8*)bl, h1 = h0, l1 = l0;
        if constexpr (AHEAD == 2) { h1 = *(const f16x8*)(bh + 32); l1 = *(const f16x8*)(bl + 32); }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f16x8 nh = h0, nl = l0;
            if (s + AHEAD < KS) { nh = *(const f16x8*)(bh + 32 * (s + AHEAD)); nl = *(const f16x8*)(bl + 32 * (s + AHEAD)); }
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32_f16(Wh[s], h0, acc);
            acc = mfma32_f16(Wh[s], l0, acc);
            acc = mfma32_f16(Wl[s], h0, acc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AHEAD == 2) { h0 = h1; l0 = l1; h1 = nh; l1 = nl; }
            else { h0 = nh; l0 = nl; }
        }
        // ---- epilogue: register r <-> channel 32 nt + 4 half + (r & 3) + 8 (r >> 2) of token m ----
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
        if constexpr (LN) {
            // out = resid + res_scale * LayerNorm(v) (+ add2 * gate[image]) over the n_real real channels of the row: per-wave
            // (mean, M2) pairs through LDS, combined with Chan's formula (csrc/tail_regs.hip).  Pad channels are exact zeros.
            if constexpr (!EARLY) ln_loads();
            const int c0 = 32 * wave + 4 * half;
            const int nw_i = min(32, max(0, p.n_real - 32 * wave));        // real channels of this wave's tile
            const float inv_nw = nw_i > 0 ? 1.0f / (float)nw_i : 0.f;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += v[r];
            const float mean_w = sum_halves(s) * inv_nw;
            float qd = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dlt = v[r] - mean_w;
                if (c0 + (r & 3) + 8 * (r >> 2) < p.n_real) qd = fmaf(dlt, dlt, qd);
            }
            qd = sum_halves(qd);
            float2* st = (float2*)(smem + G::OFF_ST);
            if (half == 0) st[wave * LS_T + j] = float2{mean_w, qd};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float inv_n = 1.0f / (float)p.n_real;
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < LS_CW; ++w)
                if (32 * w < p.n_real) mean += (float)min(32, p.n_real - 32 * w) * st[w * LS_T + j].x;
            mean *= inv_n;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < LS_CW; ++w)
                if (32 * w < p.n_real) {
                    const float2 e = st[w * LS_T + j];
                    const float dlt = e.x - mean;
                    m2 += e.y + (float)min(32, p.n_real - 32 * w) * dlt * dlt;
                }
            const float rstd = rsqrtf(m2 * inv_n + p.ln_eps);
            const float* gv = vec + LS_SLAB + c0;
            const float* bv2 = vec + 2 * LS_SLAB + c0;
            float* dst = (float*)p.out + m * p.ldo + c0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gg = *(const float4*)(gv + 8 * g), bb = *(const float4*)(bv2 + 8 * g);
                float4 y;
                y.x = rr[g].x + p.res_scale * ((v[4 * g] - mean) * rstd * gg.x + bb.x);
                y.y = rr[g].y + p.res_scale * ((v[4 * g + 1] - mean) * rstd * gg.y + bb.y);
                y.z = rr[g].z + p.res_scale * ((v[4 * g + 2] - mean) * rstd * gg.z + bb.z);
                y.w = rr[g].w + p.res_scale * ((v[4 * g + 3] - mean) * rstd * gg.w + bb.w);
                if (p.add2 != nullptr) { y.x += aa[g].x; y.y += aa[g].y; y.z += aa[g].z; y.w += aa[g].w; }
                const int c = c0 + 8 * g;
                if (c + 0 >= p.n_real) y.x = 0.f;      // keep pad channels 0
                if (c + 1 >= p.n_real) y.y = 0.f;
                if (c + 2 >= p.n_real) y.z = 0.f;
                if (c + 3 >= p.n_real) y.w = 0.f;
                *(float4*)(dst + 8 * g) = y;
            }
            continue;
        }
        if (p.epi == GRL_EPI_GROUPNORM) {
            // per 32-channel group (= one attention head slot): x / max(|x|, 1e-12) * |gscale|; gscale == 0: pass through (v);
            // gscale < 0: additionally 1.0 in column 31 (K plane).  F.normalize eps: efficient.py:85.
            float ss = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ss = fmaf(v[r], v[r], ss);
            ss = sum_halves(ss);
            const float f = gs != 0.0f ? fabsf(gs) / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= f;
            if (gs < 0.0f && half == 1) v[15] = 1.0f;
        } else if (p.epi == GRL_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2v e = gelu_erf2(f32x2v{v[r], v[r + 1]});
                v[r] = e[0]; v[r + 1] = e[1];
            }
        }
        if (p.out_dtype == GRL_DT_F32) {
            float* dst = (float*)p.out + m * p.ldo + 32 * nt + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) *(float4*)(dst + 8 * g) = float4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        } else {
            const int64_t off = p.out_plane_stride > 0 ? (int64_t)nt * p.out_plane_stride + m * 32 : m * p.ldo + 32 * nt;
            if (p.out_lo != nullptr) {
                float w[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) w[r] = v[r] - (float)to_f16(v[r]);
                store_f16_slot((f16*)p.out_lo + off, w, half);
            }
            store_f16_slot((f16*)p.out + off, v, half);
        }
    }
}

template <int KS>
int launch_ls(const GrlLinearArgs& p, hipStream_t st) {
    using G = LsGeom<KS>;
    const int nslabs = (p.Npad + LS_SLAB - 1) / LS_SLAB;
    const int ntiles = (p.M + LS_T - 1) / LS_T;
    static const int cus = getenv("GRL_PERSIST_GRID") ? atoi(getenv("GRL_PERSIST_GRID")) : 256;
    int per_xcd = cus / (8 * nslabs);                       // walkers per XCD: one workgroup per CU
    if (per_xcd < 1) per_xcd = 1;
    if (per_xcd > (ntiles + 7) / 8) per_xcd = (ntiles + 7) / 8;
    const int walkers = 8 * per_xcd;
    auto kfn = p.epi == GRL_EPI_LN_RES ? linear_split_kernel<KS, true> : linear_split_kernel<KS, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3(walkers * nslabs), dim3(LS_THREADS), G::LDS, st, p, nslabs, walkers);
    GRL_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int64_t grl_linear_split_blob_bytes(int32_t Npad, int32_t Ksrc) {
    if (Npad <= 0 || Ksrc <= 0 || (Ksrc % 16)) return 0;
    return (int64_t)((Npad + LS_SLAB - 1) / LS_SLAB) * LS_CW * 2 * (Ksrc / 16) * 1024;
}

// the a_split == 3 layers this kernel takes (everything else stays with the generic family): fp32 A without pooling or
// scaling, plain / GELU / per-head-normalised epilogue
int grl_linear_split_launch(const GrlLinearArgs& p, hipStream_t st) {
    if (p.w_regs == nullptr || p.a_split != 3 || p.a_dtype != GRL_DT_F32 || p.pool_df > 1) return GRL_ERR_UNSUPPORTED;
    if ((p.a_scale != 0.0f && p.a_scale != 1.0f) || (p.out_scale != 0.0f && p.out_scale != 1.0f)) return GRL_ERR_UNSUPPORTED;
    if (p.epi == GRL_EPI_LN_RES) {   // LayerNorm + residual (+ gated extra branch) epilogue: the whole row in one slab, fp32 out
        if (p.Npad > LS_SLAB || p.n_real <= 0 || p.n_real > p.Npad || p.resid == nullptr || p.out_dtype != GRL_DT_F32 || (p.ldr % 4)) return GRL_ERR_UNSUPPORTED;
        if (p.add2 != nullptr && (p.add2_scale == nullptr || p.rows_per_image <= 0 || (p.rows_per_image % LS_T) ||
                                  (p.add2_dtype != GRL_DT_F32 && p.add2_dtype != GRL_DT_F16) || (p.ldadd2 % 4)))
            return GRL_ERR_UNSUPPORTED;
    } else if (p.epi != GRL_EPI_PLAIN && p.epi != GRL_EPI_GELU && p.epi != GRL_EPI_GROUPNORM) return GRL_ERR_UNSUPPORTED;
    if ((p.Kpad % 3) || (p.Npad % 32) || p.M <= 0 || (p.M % LS_T)) return GRL_ERR_UNSUPPORTED;   // (whole 32-token tiles)
    if (p.out_dtype == GRL_DT_F32 ? (p.ldo % 4) != 0 : (p.out_plane_stride <= 0 && (p.ldo % 8) != 0)) return GRL_ERR_BAD_ARG;
    if (p.out_lo != nullptr && p.out_dtype == GRL_DT_F32) return GRL_ERR_BAD_ARG;
    if (p.out_dtype != GRL_DT_F32 && p.out_dtype != GRL_DT_F16) return GRL_ERR_UNSUPPORTED;
    if (p.epi == GRL_EPI_GROUPNORM && p.gscale == nullptr) return GRL_ERR_BAD_ARG;
    if ((p.lda % 4) || (uintptr_t)p.a % 16) return GRL_ERR_BAD_ARG;
    switch (p.Kpad / 3) {
        case 128: return launch_ls<8>(p, st);
        case 192: return launch_ls<12>(p, st);
        case 256: return launch_ls<16>(p, st);
        case 384: return launch_ls<24>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}
