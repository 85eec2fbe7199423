// Block tail with the MLP weights stationary in registers (gfx950) -- DRAFT, NOT YET RUN ON HARDWARE.
//
// Status: written at the end of round 3 with no GPU minutes left.  It compiles (registers / LDS within budget, see
// profiles/r03_kernel_resources.txt) but has never executed; it is reachable only through GRL_TAIL_REGS=1, is not part of any
// parity or performance claim, and its test (tests/test_gpu_kernels.py::test_block_tail_regs_draft) is skipped unless that
// variable is set.  First job of the next round: run that test, then time it against mlp_kernel<6,8,true> (257 us per 4 tiles).
//
// Same contract as grl_block_tail_fwd (include/grl_hip.h; MixedAttention.proj + norm1 + residual + CAB gate + Mlp + norm2 +
// residual, mixed_attn_block_efficient.py:379,543-556, swin_v1_block.py:37-43), GRL-Base shape only (Cpad 192, Hpad 384,
// M and rows_per_image multiples of 32), and the SAME weight blobs (ops.pack_proj / ops.pack_mlp): nothing new to pack.
//
// Why: mlp_kernel streams the 366 KB of proj + fc1 + fc2 weights through LDS for every 128 tokens (18 chunk barriers per tile,
// 2 waves per SIMD, each wave re-reading the whole stream for its 16 tokens) and has no single bottleneck left to remove
// (DESIGN section 4).  The layout that worked for QKV and CAB conv2 this round: weights stationary, activations streaming.
//   * 12 waves (3 per SIMD, <= 168 VGPRs).  Wave w owns hidden channels 32w .. 32w+31 of fc1 (= weight chunk w of the MLP blob:
//     2 groups x 6 k-steps of A fragments, 48 VGPRs) and output channels 16w .. 16w+15 of fc2 (its rows of all 12 chunks:
//     12 k-steps, 48 VGPRs) and of the projection (A fragments read from LDS, where the 78 KB projection stream stays);
//   * a tile is 32 tokens.  Activations meet the weights as MFMA B operands read from LDS tiles: the attention output (LDS-DMA,
//     double buffered, natural channel order like the projection weights), r1 and the hidden activations (written by their
//     producers in the k-slot order of ops.pack_mlp, so a consumer's operand is one 16-B read);
//   * a wave sees only 16 of a token's channels, so the two LayerNorms combine per-wave (mean, M2) pairs through LDS with
//     Chan's formula -- one exchange per norm, no cancellation for rows with |mean| >> std;
//   * 5 barriers per 32 tokens; x (residual), cab and the SE gate are read straight from global memory by the lanes that own
//     the channels; out is written by the same lanes.
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

constexpr int TR_T = 32, TR_W = 12, TR_THREADS = TR_W * 64;
constexpr int TR_CP = 192, TR_HP = 384, TR_KS1 = TR_CP / 32, TR_KS2 = TR_HP / 32;
constexpr int TR_AROW = TR_CP * 2 + 16;            // 400: fp16 row of the att / r1 tiles (and of the W1 / projection rows in the blobs)
constexpr int TR_HROW = TR_HP * 2 + 16;            // 784: fp16 row of the hidden tile
constexpr int TR_W2ROW = 80;                       // blob: fc2 rows of one chunk (32 k-slots + pad)
constexpr int TR_PCH = (32 * TR_AROW + 1023) / 1024 * 1024;                          // 13312: projection chunk image
constexpr int TR_MCH = (32 * TR_AROW + TR_CP * TR_W2ROW + 128 + 1023) / 1024 * 1024; // 28672: MLP chunk image (MlpShape<6>::BUFP)
constexpr int TR_ATT_SEG = TR_AROW / 16;           // 25 16-B segments per tile row (24 + pad)
constexpr int TR_ATT_PIECES = (TR_T * TR_AROW + 1023) / 1024;                        // 13 DMA pieces per att tile
constexpr int TR_OFF_PW = 0;
constexpr int TR_OFF_ATT = TR_OFF_PW + TR_KS1 * TR_PCH;                              // 79872
constexpr int TR_OFF_R1 = TR_OFF_ATT + 2 * TR_ATT_PIECES * 1024;                     // + 26624
constexpr int TR_OFF_H = TR_OFF_R1 + TR_T * TR_AROW;                                 // + 12800
constexpr int TR_OFF_ST = TR_OFF_H + TR_T * TR_HROW;                                 // + 25088
constexpr int TR_OFF_VEC = TR_OFF_ST + 2 * TR_W * TR_T * 8;                          // + 6144
constexpr int TR_VECF = 6 * TR_CP + TR_HP;                                           // pb n1g n1b b2 n2g n2b | b1
constexpr int TR_LDS = TR_OFF_VEC + TR_VECF * 4;                                     // 156672

// per-wave (mean, M2) of a token over this wave's real channels -> LDS; after the barrier every lane combines the 12 pairs
__device__ __forceinline__ void ln_local(const float (&v)[4], int nreal_lane, float n_w, float& mean_w, float& m2_w) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += i < nreal_lane ? v[i] : 0.f;
    s = sum_halves(sum_rows16(s));                 // over the 4 lanes (g4) that hold this token's 16 channels
    mean_w = s / n_w;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float d = v[i] - mean_w;
        q = i < nreal_lane ? fmaf(d, d, q) : q;
    }
    m2_w = sum_halves(sum_rows16(q));
}

__device__ __forceinline__ void ln_combine(const float* st, int token, int n_real, float eps, float& mean, float& rstd) {
    // st: [TR_W][TR_T][2]; wave w contributes n_w = clamp(n_real - 16 w, 0, 16) channels
    float sum = 0.f;
#pragma unroll 4
    for (int w = 0; w < TR_W; ++w) sum = fmaf((float)min(16, max(0, n_real - 16 * w)), st[(w * TR_T + token) * 2], sum);
    mean = sum / (float)n_real;
    float m2 = 0.f;
#pragma unroll 4
    for (int w = 0; w < TR_W; ++w) {        // (the means are read a second time: 12 registers are worth more than 12 LDS reads here)
        const float2 e = *(const float2*)(st + (w * TR_T + token) * 2);
        const float d = e.x - mean;
        m2 += e.y + (float)min(16, max(0, n_real - 16 * w)) * d * d;
    }
    rstd = rsqrtf(m2 / (float)n_real + eps);
}

__global__ __launch_bounds__(TR_THREADS) void tail_regs_kernel(GrlTailArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const char* blob = (const char*)p.blob;
    const int ntiles = p.M / TR_T;
    if ((int)blockIdx.x >= ntiles) return;

    // ---- once per launch: projection stream -> LDS (DMA), fc1 / fc2 A fragments -> registers, vectors -> LDS ----
    for (int q = wave; q < TR_KS1 * (TR_PCH / 1024); q += TR_W) {
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + TR_OFF_PW + q * 1024);
        const char* g = (const char*)p.pblob + (size_t)q * 1024 + lane * 16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
    }
    gemm_x8 A1[2][TR_KS1], A2[TR_KS2];
    {
        const char* c1 = blob + (size_t)wave * TR_MCH + r16 * TR_AROW + 16 * g4;       // chunk `wave`: rows = hidden 32 wave + ..
#pragma unroll
        for (int ga = 0; ga < 2; ++ga)
#pragma unroll
            for (int s = 0; s < TR_KS1; ++s) A1[ga][s] = *(const gemm_x8*)(c1 + ga * 16 * TR_AROW + 64 * s);
        const char* c2 = blob + 32 * TR_AROW + (16 * wave + r16) * TR_W2ROW + 16 * g4; // fc2 rows 16 wave + r16 of every chunk
#pragma unroll
        for (int j = 0; j < TR_KS2; ++j) A2[j] = *(const gemm_x8*)(c2 + (size_t)j * TR_MCH);
    }
    float* vec = (float*)(smem + TR_OFF_VEC);
    for (int i = tid; i < 6 * TR_CP; i += TR_THREADS) {
        const int k = i / TR_CP, c = i - k * TR_CP;
        const float* src = k == 0 ? p.pb : k == 1 ? p.n1_g : k == 2 ? p.n1_b : k == 3 ? p.b2 : k == 4 ? p.n2_g : p.n2_b;
        vec[i] = c < p.n_real ? src[c] : 0.f;      // pad channels: zero bias / affine, so they stay exactly 0 all the way
    }
    for (int i = tid; i < TR_HP; i += TR_THREADS)
        vec[6 * TR_CP + i] = *(const float*)(blob + (size_t)(i >> 5) * TR_MCH + 32 * TR_AROW + TR_CP * TR_W2ROW + 4 * (i & 31));

    // att tile DMA: piece q (1 KiB) = LDS bytes [1024 q, 1024 q + 1024) of the tile image [32 rows][25 segments]
    auto fetch_att = [&](int tile, int buf) {
        for (int q = wave; q < TR_ATT_PIECES; q += TR_W) {
            const int idx = q * 64 + lane;
            int row = idx / TR_ATT_SEG, seg = idx - row * TR_ATT_SEG;
            seg = seg < TR_ATT_SEG - 1 ? seg : TR_ATT_SEG - 2;     // the pad segment repeats the last real one
            row = row < TR_T ? row : TR_T - 1;
            const char* g = (const char*)p.att + ((int64_t)tile * TR_T + row) * p.ldatt * 2 + seg * 16;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + TR_OFF_ATT + buf * (TR_ATT_PIECES * 1024) + q * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
        }
    };
    fetch_att(blockIdx.x, 0);

    const int nreal_lane = min(4, max(0, p.n_real - 16 * wave - 4 * g4));    // real channels among this lane's 4
    const float n_w = (float)min(16, max(1, p.n_real - 16 * wave));
    const int ch0 = 16 * wave + 4 * g4;                                        // this lane's first output channel (proj, fc2)
    float* st1 = (float*)(smem + TR_OFF_ST);
    float* st2 = st1 + TR_W * TR_T * 2;
    // A fragment rows of the projection: chunk wave >> 1, rows 16 (wave & 1) + r16
    const char* pw = smem + TR_OFF_PW + (wave >> 1) * TR_PCH + (16 * (wave & 1) + r16) * TR_AROW + 16 * g4;
    // where this lane's 4 channels go in a k-slot-ordered row: slots 8 g4 + 4 (group parity) of the 32-block
    const int r1_col = (32 * (wave >> 1) + 8 * g4 + 4 * (wave & 1)) * 2;

    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA pieces of this tile (and, first tile, of the projection stream)
        __syncthreads();                                   // B0
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) fetch_att(next, (it + 1) & 1);
        const char* att = smem + TR_OFF_ATT + (it & 1) * (TR_ATT_PIECES * 1024);
        const int64_t m0 = (int64_t)tile * TR_T;
        const int img = (int)(m0 / p.rows_per_image);
        float4 xr[2];
        uint2 cb[2];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const int64_t m = m0 + 16 * tg + r16;
            xr[tg] = *(const float4*)(p.x + m * p.ldx + ch0);
            cb[tg] = *(const uint2*)((const gemm_t*)p.cab + m * p.ldcab + ch0);
        }
        const float4 gt = *(const float4*)(p.gate + (int64_t)img * TR_CP + ch0);

        // ---- P1: projection, norm1, residual, CAB ----
        float v[2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            f32x4 acc = *(const f32x4*)(vec + ch0);        // projection bias
            const char* brow = att + (16 * tg + r16) * TR_AROW + 16 * g4;
#pragma unroll
            for (int s = 0; s < TR_KS1; ++s)
                acc = mfma16_gemm(*(const gemm_x8*)(pw + 64 * s), *(const gemm_x8*)(brow + 64 * s), acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[tg][i] = acc[i];
            float mw, m2;
            ln_local(v[tg], nreal_lane, n_w, mw, m2);
            if (g4 == 0) *(float2*)(st1 + (wave * TR_T + 16 * tg + r16) * 2) = float2{mw, m2};
            __builtin_amdgcn_sched_barrier(0);   // (keeps the two token groups from being interleaved: the register budget is 168)
        }
        __syncthreads();                                   // B1
        float r1[2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            float mean, rstd;
            ln_combine(st1, 16 * tg + r16, p.n_real, p.ln_eps, mean, rstd);
            const f32x4 g1 = *(const f32x4*)(vec + TR_CP + ch0), b1n = *(const f32x4*)(vec + 2 * TR_CP + ch0);
            const float xs[4] = {xr[tg].x, xr[tg].y, xr[tg].z, xr[tg].w};
            const f16x2 c01 = __builtin_bit_cast(f16x2, cb[tg].x), c23 = __builtin_bit_cast(f16x2, cb[tg].y);
            const float cs[4] = {(float)c01[0], (float)c01[1], (float)c23[0], (float)c23[1]};
            const float gs[4] = {gt.x, gt.y, gt.z, gt.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                r1[tg][i] = xs[i] + p.res_scale * ((v[tg][i] - mean) * rstd * g1[i] + b1n[i]) + cs[i] * gs[i];
            uint2 o;
            o.x = pack_f16(r1[tg][0], r1[tg][1]);
            o.y = pack_f16(r1[tg][2], r1[tg][3]);
            *(uint2*)(smem + TR_OFF_R1 + (16 * tg + r16) * TR_AROW + r1_col) = o;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // B2

        // ---- P2: fc1 + GELU -> hidden tile (k-slot order) ----
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            f32x4 h0 = *(const f32x4*)(vec + 6 * TR_CP + 32 * wave + 4 * g4);
            f32x4 h1 = *(const f32x4*)(vec + 6 * TR_CP + 32 * wave + 16 + 4 * g4);
            const char* brow = smem + TR_OFF_R1 + (16 * tg + r16) * TR_AROW + 16 * g4;
#pragma unroll
            for (int s = 0; s < TR_KS1; ++s) {
                const gemm_x8 b = *(const gemm_x8*)(brow + 64 * s);
                h0 = mfma16_gemm(A1[0][s], b, h0);
                h1 = mfma16_gemm(A1[1][s], b, h1);
            }
            const f32x2v a01 = gelu_erf2(f32x2v{h0[0], h0[1]}), a23 = gelu_erf2(f32x2v{h0[2], h0[3]});
            const f32x2v b01 = gelu_erf2(f32x2v{h1[0], h1[1]}), b23 = gelu_erf2(f32x2v{h1[2], h1[3]});
            uint4 o;   // slots 8 g4 + [0..3] = group 0's channels 4 g4 + i, slots 8 g4 + [4..7] = group 1's (hidden 16 + 4 g4 + i)
            o.x = pack_f16(a01[0], a01[1]);
            o.y = pack_f16(a23[0], a23[1]);
            o.z = pack_f16(b01[0], b01[1]);
            o.w = pack_f16(b23[0], b23[1]);
            *(uint4*)(smem + TR_OFF_H + (16 * tg + r16) * TR_HROW + (32 * wave + 8 * g4) * 2) = o;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // B3

        // ---- P3: fc2, norm2, residual ----
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            f32x4 acc = *(const f32x4*)(vec + 3 * TR_CP + ch0);   // fc2 bias
            const char* brow = smem + TR_OFF_H + (16 * tg + r16) * TR_HROW + 16 * g4;
#pragma unroll
            for (int j = 0; j < TR_KS2; ++j) acc = mfma16_gemm(A2[j], *(const gemm_x8*)(brow + 64 * j), acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[tg][i] = acc[i];
            float mw, m2;
            ln_local(v[tg], nreal_lane, n_w, mw, m2);
            if (g4 == 0) *(float2*)(st2 + (wave * TR_T + 16 * tg + r16) * 2) = float2{mw, m2};
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // B4
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            float mean, rstd;
            ln_combine(st2, 16 * tg + r16, p.n_real, p.ln_eps, mean, rstd);
            const f32x4 g2 = *(const f32x4*)(vec + 4 * TR_CP + ch0), b2n = *(const f32x4*)(vec + 5 * TR_CP + ch0);
            float4 o;
            o.x = r1[tg][0] + p.res_scale * ((v[tg][0] - mean) * rstd * g2[0] + b2n[0]);
            o.y = r1[tg][1] + p.res_scale * ((v[tg][1] - mean) * rstd * g2[1] + b2n[1]);
            o.z = r1[tg][2] + p.res_scale * ((v[tg][2] - mean) * rstd * g2[2] + b2n[2]);
            o.w = r1[tg][3] + p.res_scale * ((v[tg][3] - mean) * rstd * g2[3] + b2n[3]);
            *(float4*)(p.out + (m0 + 16 * tg + r16) * p.ldo + ch0) = o;
        }
    }
}

}  // namespace

// draft path of grl_block_tail_fwd (GRL_TAIL_REGS=1); GRL_ERR_UNSUPPORTED for every other shape
int grl_tail_regs_launch(const GrlTailArgs& a, hipStream_t st) {
    if (a.Cpad != TR_CP || a.Hpad != TR_HP || (a.M % TR_T) || a.M <= 0 || (a.rows_per_image % TR_T) || a.n_real <= TR_CP - 16 ||
        a.n_real > TR_CP || (a.ldatt % 8) || (a.ldcab % 4) || (a.ldx % 4) || (a.ldo % 4))
        return GRL_ERR_UNSUPPORTED;
    const int ntiles = a.M / TR_T;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipError_t e = hipFuncSetAttribute((const void*)tail_regs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TR_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tail_regs_kernel, dim3(grid), dim3(TR_THREADS), TR_LDS, st, a);
    GRL_CHECK_LAUNCH();
    return 0;
}
