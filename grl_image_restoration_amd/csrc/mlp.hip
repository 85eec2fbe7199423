// Fused transformer MLP of a GRL block (gfx950):  out = x + res_scale * LayerNorm(fc2(GELU(fc1(x))))
//
// Replaces Mlp.forward (models/common/swin_v1_block.py:37-43) + norm2 + residual
// (models/common/mixed_attn_block_efficient.py:554) in ONE pass over the residual stream: the hidden
// activations (2C channels per token) never leave the CU.  As two kernels the pair moves
// x(4C) + h(2*2C) + h(2*2C) + x(4C) + out(4C) = 20C bytes per token through HBM, fused 8C:
// 969 MB -> 378 MB per launch for 4 tiles of GRL-Base.
//
// Design
//   * a wave owns 16 tokens; their fc1 operand slab (fp16, C x 16) and the fc2 accumulators
//     (16 tokens x C channels, fp32) stay in VGPRs for the whole hidden sweep;
//   * the hidden dimension is swept in chunks of 32 channels:  h_c = GELU(W1_c . x + b1_c)  (2 MFMA n-tiles)
//     is consumed immediately by  acc += W2[:, c] . h_c  (C/16 MFMAs).  The product is computed transposed
//     (D^T = W . X^T, mfma_f32_16x16x32_f16), so the fc1 accumulator fragment of a lane -- 2 x 4 hidden
//     channels of one token -- IS the B operand of the fc2 MFMA once W2's k-slots are stored in the matching
//     order (slot 8g+e <-> hidden channel 4g+e for e < 4, 16+4g+e-4 otherwise; done by the host pack):
//     no shuffle, no LDS round trip for the hidden activations;
//   * W1 + W2 (2 x 2C x C fp16 = 288 KB for Base) exceed the 160 KB LDS, so the weights stream from L2
//     through a double-buffered LDS ring, one 32-channel chunk (W1 rows, W2 columns, b1) per step; the
//     chunk sequence is cyclic, so the ring keeps running across the token tiles of the persistent
//     workgroup.  One barrier per chunk.
// Timing-ablation switches of this file compute WRONG results by construction (they remove work to see what it costs).  They only
// build together with -DGRL_ABLATION, which tools/attn_asm/build_variants_generic.sh passes for its throw-away variant libraries.
#if !defined(GRL_ABLATION) && (defined(MLP_ABL_NOATT) || defined(MLP_ABL_NOXDMA) || defined(MLP_ABL_NOGELU) || defined(MLP_ABL_NOFC2) || defined(MLP_ABL_NOSTORE))
#error "timing-ablation switch without -DGRL_ABLATION: the results of such a build are wrong"
#endif
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int KSTEPS>
struct MlpShape {
    static constexpr int CP = KSTEPS * 32;          // padded channels (= K of fc1 = N of fc2)
    static constexpr int NT2 = CP / 16;             // fc2 n-tiles
    static constexpr int W1ROW = CP * 2 + 16;       // bytes per W1 row (16 B pad: conflict-free ds_read_b128)
    static constexpr int W2ROW = 64 + 16;           // bytes per W2 row of one chunk (32 k-slots + pad)
    static constexpr int W1B = 32 * W1ROW;
    static constexpr int W2B = CP * W2ROW;
    static constexpr int BUF = W1B + W2B + 128;     // + b1 chunk (32 floats)
    static constexpr int BUFP = (BUF + 1023) / 1024 * 1024;   // chunk image, padded to whole 1-KiB DMA pieces
    static constexpr int PIECES = BUFP / 1024;
};

// Fused MLP, see the file header.  WV waves x 16 tokens per workgroup, one persistent workgroup per CU.
//   * weight ring: the blob already IS the LDS image of a chunk (padded rows), so a chunk is copied by
//     PIECES wave-wide global_load_lds_dwordx4 (1 KiB each, no staging registers, no ds_write pass);
//   * the fp32 token slab stays in registers for the residual (no re-read) next to its fp16 MFMA copy, and the
//     next tile's slab is prefetched into a second register set half way through the hidden sweep.
// Internal argument block: the MLP alone (grl_mlp_fwd) or preceded by the attention output projection + norm1 +
// residual + gated CAB branch (grl_block_tail_fwd, PROJ = true).
struct TailP {
    const float* x; int64_t ldx;
    const void* blob;                // MLP chunk images
    int M, Cpad, Hpad;
    const float* b2; const float* ln_g; const float* ln_b;
    int n_real; float ln_eps, res_scale;
    float* out; int64_t ldo;
    // PROJ only
    const gemm_t* att; int64_t ldatt;
    const void* pblob;               // projection chunk images: 32 output rows x (2*Cpad + 16) bytes, padded to 1 KiB
    const float* pb; const float* n1_g; const float* n1_b;
    const gemm_t* cab; int64_t ldcab;
    const float* gate; int rows_per_image;
};

// PROJ: a tile first runs  r1 = x + res_scale * LayerNorm1(att . Wp^T + bp) + cab * gate  (MixedAttention.proj + norm1 +
// residual + CAB, mixed_attn_block_efficient.py:379,543-548) through the same weight ring -- Cpad/32 extra chunk
// iterations -- and feeds r1 to the MLP from registers: the intermediate residual stream never goes to HBM.
template <int KSTEPS, int WV, bool PROJ>
__global__ __launch_bounds__(WV * 64) void mlp_kernel(TailP p) {
    using S = MlpShape<KSTEPS>;
    constexpr int CP = S::CP, NT2 = S::NT2;
    constexpr int THREADS = WV * 64;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int nchunks = p.Hpad / 32;
    const char* blob = (const char*)p.blob;

    // Chunk DMA: this wave's share of the image, pieces wave, wave + WV, ... (1 KiB each, lane * 16 B implicit).
    // Issued as inline asm on purpose: with the builtin the compiler knows the instruction writes LDS and drains
    // vmcnt(0) in front of the next ds_read, i.e. it waits for the prefetch it has just issued.  Hidden from the
    // compiler, the DMA is ordered by hand: a counted s_waitcnt before the barrier that publishes the chunk.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int NP = PROJ ? KSTEPS : 0;                           // projection chunks per tile (32 output channels each)
    constexpr int PPIECES = (32 * S::W1ROW + 1023) / 1024;          // DMA pieces of a projection chunk image
    const char* pblob = (const char*)p.pblob;
    auto fetch = [&](int k, int buf_off) {   // chunk k of the per-tile sequence [NP projection chunks | nchunks MLP chunks]
        const bool isp = PROJ && k < NP;
        // wave-uniform image base in SGPRs + a 32-bit lane offset (saddr form): no per-piece 64-bit VGPR addresses to keep
        const uint64_t srcv = (uint64_t)(isp ? pblob + (size_t)k * (PPIECES * 1024) : blob + (size_t)(k - NP) * S::BUFP);
        const uint64_t src = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(srcv >> 32)) << 32) |
                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)srcv);   // (the builtin returns int)
        const int pieces = isp ? PPIECES : S::PIECES;
#pragma unroll
        for (int q0 = 0; q0 < S::PIECES; q0 += WV) {
            const int q = q0 + wave_u;
            if (q < pieces) {
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + buf_off + q * 1024);
                const uint32_t voff = (uint32_t)(q * 1024 + lane * 16);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(src) : "m0", "memory");
            }
        }
    };

    const int ntiles = (p.M + WV * 16 - 1) / (WV * 16);
    if ((int)blockIdx.x >= ntiles) return;
    // fc2 bias and the norm affine live in LDS behind the ring (read once per tile by every lane)
    float* vec = (float*)(smem + 2 * S::BUFP);
    // Pad channels (>= n_real) of every vector are stored as 0: with the zero pad rows of the packed weights the pad
    // accumulators are then exactly 0, the norm statistics need per-channel masks in the last two n-tiles only, and the pad
    // outputs come out as the pad channels of x (0 in a padded token matrix) without compare/select in the epilogues.
    for (int i = tid; i < 3 * CP; i += THREADS) {
        const float v = i < CP ? p.b2[i] : (i < 2 * CP ? p.ln_g[i - CP] : p.ln_b[i - 2 * CP]);
        vec[i] = (i % CP) < p.n_real ? v : 0.f;
    }
    if constexpr (PROJ)   // [3CP..6CP): projection bias, norm1 weight / bias; [6CP..8CP): SE gate rows of the tile's <= 2 images
        for (int i = tid; i < 3 * CP; i += THREADS) {
            const float v = i < CP ? p.pb[i] : (i < 2 * CP ? p.n1_g[i - CP] : p.n1_b[i - 2 * CP]);
            vec[3 * CP + i] = (i % CP) < p.n_real ? v : 0.f;
        }
    constexpr int VECF = PROJ ? 8 * CP : 3 * CP;

    // Token tile staging: the fp32 rows of the NEXT tile are DMA'd into LDS (rows padded by 16 B) in 1-KiB pieces
    // spread evenly over the chunk iterations of the current tile, so HBM reads run beside the MFMAs instead of in a
    // burst at the tile boundary, and cost no registers.
    constexpr int XROW = CP * 4 + 16, XSEG = XROW / 16;             // bytes / 16-B segments per staged row
    constexpr int XPIECES = (WV * 16 * XROW + 1023) / 1024;
    const int xoff = 2 * S::BUFP + VECF * 4;                         // LDS offset of the token tile
    auto fetch_x = [&](int tile, int piece) {                        // piece: wave-uniform
        const int sigma = piece * 64 + lane;                         // LDS segment this lane fills
        int row = sigma / XSEG, seg = sigma - row * XSEG;
        seg = seg < XSEG - 1 ? seg : XSEG - 2;                       // the pad segment repeats the last real one
        int m = tile * (WV * 16) + row;
        m = m < p.M ? m : p.M - 1;                                   // rows past the end (and past the tile) stay in range
        const float* g = p.x + (int64_t)m * p.ldx + seg * 4;
        const uint32_t m0v = lds0 + xoff + piece * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
    };
    const char* xt = smem + xoff;

    for (int q = wave_u; q < XPIECES; q += WV) fetch_x(blockIdx.x, q);
    fetch(0, 0);
    int it = 0;   // running chunk counter of the ring (chunk = it % nchunks, buffer = it & 1)
    // Chunk-top wait.  The token pieces of the NEXT tile come from HBM (2-3 us under load, longer than a chunk lasts) and are
    // issued BEHIND the chunk's weight pieces, so a counted wait that leaves this wave's newest DMA in flight still covers the
    // weights (loads retire in order) and gives every token piece two chunk periods instead of stalling every chunk on it.
    bool x_in_flight = false;   // wave-uniform: the last DMA this wave issued was a token piece of the next tile
    auto chunk_wait = [&]() {
        if (x_in_flight) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * (WV * 16) + wave * 16 + r16;
        const bool valid = m < p.M;
        // the tile's rows have landed (own pieces: vmcnt, everybody's: barrier).  Lane = token r16; its 8 k-slots of
        // k-step s are the channels 32s + 4*g4 + [0..3] and 32s + 16 + 4*g4 + [0..3] -- the accumulator layout of n-tiles
        // 2s / 2s+1 (W1's columns are packed in the same slot order), so the registers that feed fc1 also provide the
        // residual of the epilogue: x is read from HBM exactly once.
        float4 xs[KSTEPS][2];   // fp32 token rows in accumulator layout: MLP input and final residual
        gemm_x8 a[KSTEPS];
        if constexpr (PROJ) {
            // ---- attention output projection + norm1 + residual + gated CAB  ->  xs (registers) ----
            const int64_t mc0 = valid ? m : (int64_t)p.M - 1;
            const int img0 = (tile * (WV * 16)) / p.rows_per_image;
            for (int i = tid; i < 2 * CP; i += THREADS) {   // gate rows of the (at most two) images this tile touches
                const int im = img0 + i / CP;
                const int last = (p.M - 1) / p.rows_per_image;
                vec[6 * CP + i] = (i % CP) < p.n_real ? p.gate[(int64_t)(im < last ? im : last) * CP + (i % CP)] : 0.f;
            }
            gemm_x8 a0[KSTEPS];
            gemm_x4 cb[NT2];
            {
#ifdef MLP_ABL_NOATT
                const gemm_t* arow = p.att + (mc0 & 127) * p.ldatt + 8 * g4;
#else
                const gemm_t* arow = p.att + mc0 * p.ldatt + 8 * g4;
#endif
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) a0[s] = *(const gemm_x8*)(arow + 32 * s);
            }
            f32x4 pacc[NT2];
#pragma unroll
            for (int j = 0; j < NP; ++j, ++it) {
                chunk_wait();
                __builtin_amdgcn_s_barrier();
                char* cur = smem + (it & 1) * S::BUFP;
                fetch(j + 1, ((it + 1) & 1) * S::BUFP);   // j + 1 == NP is MLP chunk 0
                x_in_flight = false;
                if (j == (NP > 1 ? NP - 2 : 0)) {          // CAB rows: needed after the last projection step, two steps of cover
#ifdef MLP_ABL_NOATT
                    const gemm_t* crow = p.cab + (mc0 & 127) * p.ldcab + 4 * g4;
#else
                    const gemm_t* crow = p.cab + mc0 * p.ldcab + 4 * g4;
#endif
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) cb[nt] = *(const gemm_x4*)(crow + 16 * nt);
                }
                f32x4 h0 = f32x4{0, 0, 0, 0}, h1 = f32x4{0, 0, 0, 0};
                constexpr int KB = KSTEPS > 4 ? KSTEPS / 2 : KSTEPS;   // k-steps per read batch (register budget)
#pragma unroll
                for (int s0 = 0; s0 < KSTEPS; s0 += KB) {
                    gemm_x8 wa[KB], wb[KB];
#pragma unroll
                    for (int s = 0; s < KB; ++s) {
                        wa[s] = *(const gemm_x8*)(cur + r16 * S::W1ROW + (32 * (s0 + s) + 8 * g4) * 2);
                        wb[s] = *(const gemm_x8*)(cur + (16 + r16) * S::W1ROW + (32 * (s0 + s) + 8 * g4) * 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < KB; ++s) {
                        h0 = mfma16_gemm(wa[s], a0[s0 + s], h0);
                        h1 = mfma16_gemm(wb[s], a0[s0 + s], h1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                pacc[2 * j] = h0;
                pacc[2 * j + 1] = h1;
            }
            // the token tile landed before the first of the barriers above (issued a whole tile earlier)
            float s1 = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const float4 b4 = *(const float4*)(vec + 3 * CP + 16 * nt + 4 * g4);
                pacc[nt][0] += b4.x; pacc[nt][1] += b4.y; pacc[nt][2] += b4.z; pacc[nt][3] += b4.w;
                s1 += (pacc[nt][0] + pacc[nt][1]) + (pacc[nt][2] + pacc[nt][3]);   // pad channels are exactly 0
            }
            s1 = sum_halves(sum_rows16(s1));
            const float mean = s1 / (float)p.n_real;
            float s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = pacc[nt][e] - mean;
                    // pad channels (< 32 of them, so only in the last two n-tiles) are left out of the variance: subtracting
                    // their n_pad * mean^2 afterwards would cancel catastrophically for rows with |mean| >> std
                    if (nt < NT2 - 2 || 16 * nt + 4 * g4 + e < p.n_real) s2 = fmaf(d, d, s2);
                }
            s2 = sum_halves(sum_rows16(s2));
            const float rstd = rsqrtf(s2 / (float)p.n_real + p.ln_eps);
            const char* rowp = xt + (wave * 16 + r16) * XROW + 16 * g4;
            const float* grow = vec + 6 * CP + ((int)(mc0 / p.rows_per_image) - img0) * CP;
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const int col = 16 * nt + 4 * g4;
                const float4 res = *(const float4*)(rowp + 64 * nt);
                const float4 g = *(const float4*)(vec + 4 * CP + col);
                const float4 bb = *(const float4*)(vec + 5 * CP + col);
                const float4 gt = *(const float4*)(grow + col);
                float4 o4;
                o4.x = res.x + p.res_scale * ((pacc[nt][0] - mean) * rstd * g.x + bb.x) + (float)cb[nt][0] * gt.x;
                o4.y = res.y + p.res_scale * ((pacc[nt][1] - mean) * rstd * g.y + bb.y) + (float)cb[nt][1] * gt.y;
                o4.z = res.z + p.res_scale * ((pacc[nt][2] - mean) * rstd * g.z + bb.z) + (float)cb[nt][2] * gt.z;
                o4.w = res.w + p.res_scale * ((pacc[nt][3] - mean) * rstd * g.w + bb.w) + (float)cb[nt][3] * gt.w;
                xs[nt >> 1][nt & 1] = o4;
                if ((nt & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bounds the live range of the four vector reads per n-tile
            }
        } else {
            // the tile's rows have landed (own pieces: vmcnt, everybody's: barrier).  Lane = token r16; its 8 k-slots of
            // k-step s are the channels 32s + 4*g4 + [0..3] and 32s + 16 + 4*g4 + [0..3] -- the accumulator layout of n-tiles
            // 2s / 2s+1 (W1's columns are packed in the same slot order), so the registers that feed fc1 also provide the
            // residual of the epilogue: x is read from HBM exactly once.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* rowp = xt + (wave * 16 + r16) * XROW + 16 * g4;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                xs[s][0] = *(const float4*)(rowp + 128 * s);
                xs[s][1] = *(const float4*)(rowp + 128 * s + 64);
            }
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            gemm_x8 v;
            v[0] = to_f16(xs[s][0].x); v[1] = to_f16(xs[s][0].y); v[2] = to_f16(xs[s][0].z); v[3] = to_f16(xs[s][0].w);
            v[4] = to_f16(xs[s][1].x); v[5] = to_f16(xs[s][1].y); v[6] = to_f16(xs[s][1].z); v[7] = to_f16(xs[s][1].w);
            a[s] = v;
        }
        f32x4 acc[NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
        const int next_tile = tile + (int)gridDim.x;   // may lie past the end: rows are clamped, the data is never read

#pragma unroll 1
        for (int c = 0; c < nchunks; ++c, ++it) {
            // Every wave waits for its own DMA pieces (chunk `it` and the token pieces of the previous iteration) before
            // the barrier publishes them; after the barrier nobody reads ring buffer (it+1)&1 any more -- and, at c == 0,
            // everybody has copied its token rows into registers -- so the next DMAs may land.
            chunk_wait();   // (lgkmcnt: my LDS reads of the token tile / last chunk are done)
            __builtin_amdgcn_s_barrier();
            char* cur = smem + (it & 1) * S::BUFP;
            fetch(c + 1 < nchunks ? NP + c + 1 : 0, ((it + 1) & 1) * S::BUFP);
            x_in_flight = false;
#ifndef MLP_ABL_NOXDMA
            for (int q = c * WV + wave_u; q < XPIECES; q += nchunks * WV) { fetch_x(next_tile, q); x_in_flight = true; }
#endif

            // ---- h = GELU(W1_c . x + b1_c): two n-tiles of 16 hidden channels ----
            // LDS fragment reads are issued in batches ahead of the MFMAs that consume them (the compiler
            // otherwise emits read -> wait -> MFMA one by one and fills the MFMA->VALU hazards with s_nop)
            const float4 bA = *(const float4*)(cur + S::W1B + S::W2B + (4 * g4) * 4);
            const float4 bB = *(const float4*)(cur + S::W1B + S::W2B + (16 + 4 * g4) * 4);
            f32x4 h0 = f32x4{0, 0, 0, 0}, h1 = f32x4{0, 0, 0, 0};
#ifdef MLP_KB
            constexpr int KB = MLP_KB;
#else
            constexpr int KB = KSTEPS > 4 ? KSTEPS / 2 : KSTEPS;   // k-steps per read batch (register budget)
#endif
#pragma unroll
            for (int s0 = 0; s0 < KSTEPS; s0 += KB) {
                gemm_x8 wa[KB], wb[KB];
#pragma unroll
                for (int s = 0; s < KB; ++s) {
                    wa[s] = *(const gemm_x8*)(cur + r16 * S::W1ROW + (32 * (s0 + s) + 8 * g4) * 2);
                    wb[s] = *(const gemm_x8*)(cur + (16 + r16) * S::W1ROW + (32 * (s0 + s) + 8 * g4) * 2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < KB; ++s) {
                    h0 = mfma16_gemm(wa[s], a[s0 + s], h0);
                    h1 = mfma16_gemm(wb[s], a[s0 + s], h1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef MLP_QB
            constexpr int QB = MLP_QB, NB = NT2 / QB;
#else
            constexpr int QB = 4, NB = NT2 / QB;
#endif   // fc2 n-tiles per read batch; batch k+1 is in flight while k multiplies (deeper prefetch measured slower: the kernel is LDS-bandwidth bound)
            gemm_x8 w2[2][QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) w2[0][j] = *(const gemm_x8*)(cur + S::W1B + (16 * j + r16) * S::W2ROW + 16 * g4);
            __builtin_amdgcn_sched_barrier(0);
            gemm_x8 hb;   // k-slots 8*g4 + [0..7] of this chunk for token r16 (see the header comment)
#ifdef MLP_ABL_NOGELU
            hb[0] = (f16)h0[0]; hb[1] = (f16)h0[1]; hb[2] = (f16)h0[2]; hb[3] = (f16)h0[3]; hb[4] = (f16)h1[0]; hb[5] = (f16)h1[1]; hb[6] = (f16)h1[2]; hb[7] = (f16)h1[3];
            if (false)
#endif
            {
                const f32x2v g0 = gelu_erf2(f32x2v{h0[0] + bA.x, h0[1] + bA.y}), g1 = gelu_erf2(f32x2v{h0[2] + bA.z, h0[3] + bA.w});
                const f32x2v g2 = gelu_erf2(f32x2v{h1[0] + bB.x, h1[1] + bB.y}), g3 = gelu_erf2(f32x2v{h1[2] + bB.z, h1[3] + bB.w});
                hb[0] = to_f16(g0.x); hb[1] = to_f16(g0.y); hb[2] = to_f16(g1.x); hb[3] = to_f16(g1.y);
                hb[4] = to_f16(g2.x); hb[5] = to_f16(g2.y); hb[6] = to_f16(g3.x); hb[7] = to_f16(g3.y);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- acc += W2[:, chunk] . h ----
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k + 1 < NB) {
#pragma unroll
                    for (int j = 0; j < QB; ++j)
                        w2[(k + 1) & 1][j] = *(const gemm_x8*)(cur + S::W1B + (16 * (QB * (k + 1) + j) + r16) * S::W2ROW + 16 * g4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
#ifndef MLP_ABL_NOFC2
                for (int j = 0; j < QB; ++j) acc[QB * k + j] = mfma16_gemm(w2[k & 1][j], hb, acc[QB * k + j]);
#else
                for (int j = 0; j < QB; ++j) acc[QB * k + j][0] += (float)w2[k & 1][j][0] * (float)hb[0];
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: + b2, LayerNorm over the n_real channels, residual (from the register slab) ----
        float s1 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const float4 b4 = *(const float4*)(vec + 16 * nt + 4 * g4);
            acc[nt][0] += b4.x; acc[nt][1] += b4.y; acc[nt][2] += b4.z; acc[nt][3] += b4.w;
            s1 += (acc[nt][0] + acc[nt][1]) + (acc[nt][2] + acc[nt][3]);   // pad channels are exactly 0
        }
        s1 = sum_halves(sum_rows16(s1));
        const float mean = s1 / (float)p.n_real;
        float s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = acc[nt][e] - mean;
                if (nt < NT2 - 2 || 16 * nt + 4 * g4 + e < p.n_real) s2 = fmaf(d, d, s2);   // (see norm1 above)
            }
        s2 = sum_halves(sum_rows16(s2));
        const float rstd = rsqrtf(s2 / (float)p.n_real + p.ln_eps);
        const int64_t mc = valid ? m : (int64_t)p.M - 1;
        float* orow = p.out + mc * p.ldo;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int col = 16 * nt + 4 * g4;
            const float4 res = xs[nt >> 1][nt & 1];
            const float4 g = *(const float4*)(vec + CP + col);
            const float4 bb = *(const float4*)(vec + 2 * CP + col);
            float4 o4;
            o4.x = res.x + p.res_scale * ((acc[nt][0] - mean) * rstd * g.x + bb.x);
            o4.y = res.y + p.res_scale * ((acc[nt][1] - mean) * rstd * g.y + bb.y);
            o4.z = res.z + p.res_scale * ((acc[nt][2] - mean) * rstd * g.z + bb.z);
            o4.w = res.w + p.res_scale * ((acc[nt][3] - mean) * rstd * g.w + bb.w);
#ifdef MLP_ABL_NOSTORE
            if (valid && o4.x == 12345.678f) *(float4*)(orow + col) = o4;
#else
            if (valid) *(float4*)(orow + col) = o4;
#endif
        }
    }
}

template <int KSTEPS, int WV, bool PROJ>
int launch_mlp(const TailP& p, hipStream_t st) {
    using S = MlpShape<KSTEPS>;
    const size_t lds = 2 * (size_t)S::BUFP + (PROJ ? 8 : 3) * S::CP * sizeof(float) +
                       (size_t)((WV * 16 * (S::CP * 4 + 16) + 1023) / 1024) * 1024;
    const int ntiles = (p.M + WV * 16 - 1) / (WV * 16);
    static const int cap = getenv("GRL_PERSIST_GRID") ? atoi(getenv("GRL_PERSIST_GRID")) : 256;   // tuning knob
    const int grid = ntiles < cap ? ntiles : cap;   // one persistent workgroup per CU
    auto kfn = mlp_kernel<KSTEPS, WV, PROJ>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WV * 64), lds, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

template <bool PROJ>
int launch_tail(const TailP& p, hipStream_t st) {
    switch (p.Cpad / 32) {
        case 2: return launch_mlp<2, 8, PROJ>(p, st);
        case 4: return launch_mlp<4, 8, PROJ>(p, st);
        case 6: return launch_mlp<6, 8, PROJ>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}

int64_t proj_chunk_bytes(int Cpad) { return ((int64_t)32 * (Cpad * 2 + 16) + 1023) / 1024 * 1024; }

int64_t chunk_bytes(int Cpad) {
    switch (Cpad / 32) {
        case 2: return MlpShape<2>::BUFP;
        case 4: return MlpShape<4>::BUFP;
        case 6: return MlpShape<6>::BUFP;
        default: return 0;
    }
}

}  // namespace

extern "C" int64_t grl_mlp_blob_bytes(int32_t Cpad, int32_t Hpad) {
    if (Cpad <= 0 || Hpad <= 0 || (Cpad % 32) || (Hpad % 32) || chunk_bytes(Cpad) == 0) return GRL_ERR_BAD_ARG;
    return (int64_t)(Hpad / 32) * chunk_bytes(Cpad);
}

static bool mlp_args_ok(const void* x, int64_t ldx, const void* blob, const void* out, int64_t ldo, int M, int Cpad, int Hpad, int n_real) {
    if ((Cpad % 32) || (Hpad % 32) || Hpad <= 0 || (ldx % 4) || (ldo % 4) || ldx < Cpad || ldo < Cpad || n_real > Cpad || n_real <= 0) return false;
    if (n_real + 32 <= Cpad) return false;   // Cpad = n_real rounded up to 32: the norm epilogues mask pad channels in the last 32 only
    if (x == nullptr || blob == nullptr || out == nullptr || x == out || ((uintptr_t)blob & 15) != 0) return false;
    return true;
}

extern "C" int grl_mlp_fwd(void* stream, const GrlMlpArgs* args) {
    const GrlMlpArgs& a = *args;
    if (a.M <= 0) return 0;
    if (!mlp_args_ok(a.x, a.ldx, a.blob, a.out, a.ldo, a.M, a.Cpad, a.Hpad, a.n_real)) return GRL_ERR_BAD_ARG;
    TailP p = {};
    p.x = a.x; p.ldx = a.ldx; p.blob = a.blob; p.M = a.M; p.Cpad = a.Cpad; p.Hpad = a.Hpad;
    p.b2 = a.b2; p.ln_g = a.ln_g; p.ln_b = a.ln_b; p.n_real = a.n_real; p.ln_eps = a.ln_eps; p.res_scale = a.res_scale;
    p.out = a.out; p.ldo = a.ldo;
    return launch_tail<false>(p, (hipStream_t)stream);
}

extern "C" int64_t grl_proj_blob_bytes(int32_t Cpad) {
    if (Cpad <= 0 || (Cpad % 32) || chunk_bytes(Cpad) == 0) return GRL_ERR_BAD_ARG;
    return (int64_t)(Cpad / 32) * proj_chunk_bytes(Cpad);
}

// csrc/tail_regs.hip: the register-resident kernel for the GRL-Base shape (GrlTailArgs.rblob)
int grl_tail_regs_launch(const GrlTailArgs& a, hipStream_t st);

extern "C" int grl_block_tail_fwd(void* stream, const GrlTailArgs* args) {
    const GrlTailArgs& a = *args;
    if (a.M <= 0) return 0;
    if (!mlp_args_ok(a.x, a.ldx, a.blob, a.out, a.ldo, a.M, a.Cpad, a.Hpad, a.n_real)) return GRL_ERR_BAD_ARG;
    if (a.att == nullptr || a.cab == nullptr || a.gate == nullptr || a.pblob == nullptr || ((uintptr_t)a.pblob & 15) != 0 ||
        (a.ldatt % 8) || a.ldatt < a.Cpad || (a.ldcab % 4) || a.ldcab < a.Cpad || a.pb == nullptr || a.n1_g == nullptr || a.n1_b == nullptr)
        return GRL_ERR_BAD_ARG;
    if (a.rows_per_image < 128) return GRL_ERR_UNSUPPORTED;   // a 128-token tile may touch at most two images
    if (a.rblob != nullptr) {
        const int rc = grl_tail_regs_launch(a, (hipStream_t)stream);
        if (rc != GRL_ERR_UNSUPPORTED) return rc;
    }
    TailP p = {};
    p.x = a.x; p.ldx = a.ldx; p.blob = a.blob; p.M = a.M; p.Cpad = a.Cpad; p.Hpad = a.Hpad;
    p.b2 = a.b2; p.ln_g = a.n2_g; p.ln_b = a.n2_b; p.n_real = a.n_real; p.ln_eps = a.ln_eps; p.res_scale = a.res_scale;
    p.out = a.out; p.ldo = a.ldo;
    p.att = (const gemm_t*)a.att; p.ldatt = a.ldatt; p.pblob = a.pblob; p.pb = a.pb; p.n1_g = a.n1_g; p.n1_b = a.n1_b;
    p.cab = (const gemm_t*)a.cab; p.ldcab = a.ldcab; p.gate = a.gate; p.rows_per_image = a.rows_per_image;
    return launch_tail<true>(p, (hipStream_t)stream);
}
