// One pass over the residual stream for every attention operand of a GRL block (gfx950):
//   q/k/v head planes   planes[slot][m][0..31] = groupnorm_slot( x[m, :] . W_slot^T + b_slot )                    (fp16)
//   anchor head planes  anc[slot][a][0..31]    = groupnorm_slot( avgpool2x2(x)[a, :] . Wa_slot^T + ba_slot )       (fp16)
// Replaces QKVProjection.forward (models/common/mixed_attn_block.py:661-676), AnchorProjection / AnchorLinear
// (:714-736,739-785: avg_pool2d(df) + Linear C -> C/2) and the F.normalize / logit-scale prologue of Attention.attn
// (models/common/mixed_attn_block_efficient.py:85-90,:39).  Round 2 ran two kernels (csrc/qkv.hip + the pooled variant of
// csrc/linear.hip) that both read x; the anchor launch cost 44 us alone and 168 us inside the two-stream bench.
//
// Why it is built the way it is (measured on the round-2 kernel, tools/attn_asm/build_variants_generic.sh): without any HBM
// traffic that kernel still took 105 of its 160 us -- 193 instructions per (slot, 16 tokens) of which 12 were MFMAs: it was
// bound by instruction issue, not by bandwidth or LDS.  So:
//   * mfma_f32_32x32x16_f16, a wave owns 32 tokens (2 image rows x 16 columns): half the MFMA / LDS-read instructions per
//     token, and the epilogue of a slot serves 32 tokens (packed fp32 math, v_rsq instead of an IEEE division, 16-B stores);
//   * the 2 x 2 average pool commutes with the linear map: the anchor slots are projected per TOKEN like q/k/v and the
//     outputs of the 4 tokens of a pooling cell -- lanes l, l^1, l^16, l^17 of the wave -- are summed in registers (one DPP
//     add + one cross-row add per accumulator register), then bias / normalise / store by the cell's first lane;
//   * tiles are 2 image rows x 64 columns (128 tokens), token pieces and weight chunks arrive by LDS-DMA as in round 2;
//     DMA completion is awaited right before a chunk's stores (vmcnt also counts stores: a wait at the chunk top would sit out
//     the write acknowledgements of the previous chunk every time).
// Timing-ablation switches of this file compute WRONG results by construction (they remove work to see what it costs).  They only
// build together with -DGRL_ABLATION, which tools/attn_asm/build_variants_generic.sh passes for its throw-away variant libraries.
#if !defined(GRL_ABLATION) && (defined(QA_ABL_NOXDMA) || defined(QA_ABL_NOSTORE) || defined(QA_TOKEN_MAJOR) || defined(QS_ABL_NOLO) || defined(QS_ABL_NOFP8))
#error "timing-ablation switch without -DGRL_ABLATION: the results of such a build are wrong"
#endif
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int KS>   // k-steps of 16 channels: Cpad = 16 * KS
struct QaShape {
    static constexpr int CP = KS * 16;
    static constexpr int WROW = CP * 2 + 16;                   // bytes per weight row (16 B pad: conflict-free ds_read_b128)
    static constexpr int SLOT = 32 * WROW + 128 + 16;          // 32 rows | bias (32 fp32) | gscale (fp32, padded to 16 B)
    static constexpr int BUF = 2 * SLOT;                       // a chunk = 2 slots
    static constexpr int BUFP = (BUF + 1023) / 1024 * 1024;
    static constexpr int PIECES = BUFP / 1024;
    static constexpr int XROW = CP * 4 + 16, XSEG = XROW / 16; // staged fp32 token row (16 B pad)
    static constexpr int XPIECES = (128 * XROW + 1023) / 1024;
    static constexpr int LDS = 2 * BUFP + XPIECES * 1024;
};

constexpr int QW = 8;   // waves: wave w and w + 4 share token group w & 3 and take one slot of every chunk each

// 16 channel values of one token (MFMA 32x32 accumulator layout: value r <-> channel (r & 3) + 8 * (r >> 2) + 4 * half) -> the
// token's 64-B fp16 slot.  The half-wave pair exchanges two 8-B pieces so that every lane stores 2 x 16 B.
template <bool SAT>
__device__ __forceinline__ void store_slot(f16* slot, const float (&v)[16], int half, bool active = true) {
    uint2 pk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (SAT) {
            pk[g].x = pack_f16(v[4 * g + 0], v[4 * g + 1]);
            pk[g].y = pack_f16(v[4 * g + 2], v[4 * g + 3]);
        } else {
            typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;
            pk[g].x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v[4 * g + 0], v[4 * g + 1]}, f16x2));
            pk[g].y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v[4 * g + 2], v[4 * g + 3]}, f16x2));
        }
    }
    // lanes l and l + 32 hold the two halves of a token's channels: after the swaps the lower lane has pieces 0 (own) and 0 (of
    // the upper lane), the upper lane pieces 1 (of the lower lane) and 1 (own) -- 16 contiguous bytes each
    swap32(pk[0].x, pk[1].x); swap32(pk[0].y, pk[1].y);
    swap32(pk[2].x, pk[3].x); swap32(pk[2].y, pk[3].y);
    const uint4 lo = uint4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};
    const uint4 hi = uint4{pk[2].x, pk[2].y, pk[3].x, pk[3].y};
    if (active) {   // (the lane exchanges above run with every lane enabled)
        *(uint4*)(slot + 8 * half) = lo;
        *(uint4*)(slot + 8 * half + 16) = hi;
    }
}

template <int KS>
__global__ __launch_bounds__(QW * 64) void qkv_anchor_kernel(GrlQkvAnchorArgs p) {
    using S = QaShape<KS>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wave_u & 3, hs = wave_u >> 2;      // token group (32 tokens: 2 rows x 16 columns), slot of a chunk
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const char* blob = (const char*)p.blob;
    const int tslots = p.nslots + p.nanc;
    const int nchunks = (tslots + 1) >> 1;
    const int tiles_x = p.W >> 6, tiles_img = (p.H >> 1) * tiles_x;
    const int ntiles = p.B * tiles_img;
    if ((int)blockIdx.x >= ntiles) return;

    auto fetch = [&](int chunk, int buf_off) {   // this wave's 1-KiB pieces of the chunk image
        const char* src = blob + (size_t)chunk * S::BUFP + lane * 16;
#pragma unroll
        for (int q0 = 0; q0 < S::PIECES; q0 += QW) {
            const int q = q0 + wave_u;
            if (q < S::PIECES) {
                const uint32_t m0v = lds0 + buf_off + q * 1024;
                const char* g = src + q * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
            }
        }
    };
    // first token (image row 2*y2, column 64*c64) of a tile and the first 2 x 2 pooling cell under it
    auto tile_origin = [&](int tile, int64_t& cell0) -> int64_t {
        const int b = tile / tiles_img, t = tile - b * tiles_img;
        const int y2 = t / tiles_x, c64 = t - y2 * tiles_x;
        cell0 = ((int64_t)b * (p.H >> 1) + y2) * (p.W >> 1) + 32 * c64;
        return ((int64_t)b * p.H + 2 * y2) * p.W + 64 * c64;
    };
    const int xoff = 2 * S::BUFP;
    // token slot ts of the tile (LDS row ts): group ts >> 5, image row (ts >> 4) & 1, column 16 * group + (ts & 15)
    auto fetch_x = [&](int64_t origin, int piece) {
        const int sigma = piece * 64 + lane;
        int row = sigma / S::XSEG, seg = sigma - row * S::XSEG;
        seg = seg < S::XSEG - 1 ? seg : S::XSEG - 2;
        row = row < 128 ? row : 127;
        const int64_t m = origin + (int64_t)((row >> 4) & 1) * p.W + 16 * (row >> 5) + (row & 15);
        const float* g = p.x + m * p.ldx + seg * 4;
        const uint32_t m0v = lds0 + xoff + piece * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
    };
    const char* xt = smem + xoff;

    {
        int64_t c0;
        const int64_t o0 = tile_origin(blockIdx.x, c0);
        for (int q = wave_u; q < S::XPIECES; q += QW) fetch_x(o0, q);
    }
    fetch(0, 0);
    int it = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __builtin_amdgcn_s_barrier();   // the tile's token pieces were awaited inside the previous tile's chunks (first tile: above)
        int64_t cell0, cell_next;
        const int64_t origin = tile_origin(tile, cell0);
        // operand slab: lane = token j of the group; its 8 k-slots of k-step s are the channels 16 s + 8 half + [0..7]
        f16x8 a[KS];
        {
            const char* rowp = xt + (32 * grp + j) * S::XROW + 32 * half;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 v0 = *(const float4*)(rowp + 64 * s), v1 = *(const float4*)(rowp + 64 * s + 16);
                typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
                a[s] = __builtin_bit_cast(f16x8, u32x4{pack_f16(v0.x, v0.y), pack_f16(v0.z, v0.w), pack_f16(v1.x, v1.y), pack_f16(v1.z, v1.w)});
            }
        }
        const int next_tile = tile + (int)gridDim.x;
        const int64_t next_origin = next_tile < ntiles ? tile_origin(next_tile, cell_next) : origin;
        const int64_t m_tok = origin + (int64_t)(j >> 4) * p.W + 16 * grp + (j & 15);                          // this lane's token
        const int64_t m_anc = cell0 + 8 * grp + ((j & 15) >> 1);                                               // its pooling cell
        const bool anc_lane = (j & 17) == 0;   // first lane of a 2 x 2 cell (even column, upper row)

#pragma unroll 1
        for (int c = 0; c < nchunks; ++c, ++it) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's DMA pieces of chunk c were awaited before the previous chunk's stores)
            __builtin_amdgcn_s_barrier();
            const char* cur = smem + (it & 1) * S::BUFP;
            fetch(c + 1 < nchunks ? c + 1 : 0, ((it + 1) & 1) * S::BUFP);
#ifndef QA_ABL_NOXDMA
            for (int q = c * QW + wave_u; q < S::XPIECES; q += nchunks * QW) fetch_x(next_origin, q);
#endif
            const int slot = 2 * c + hs;
            if (slot >= tslots) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; }   // pad slot of an odd total
            const char* wb = cur + hs * S::SLOT;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int KB = KS > 6 ? KS / 2 : KS;   // k-steps per LDS read batch
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += KB) {
                f16x8 wf[KB];
#pragma unroll
                for (int s = 0; s < KB; ++s) wf[s] = *(const f16x8*)(wb + j * S::WROW + (16 * (s0 + s) + 8 * half) * 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < KB; ++s) acc = mfma32_f16(wf[s], a[s0 + s], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            float4 b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b4[g] = *(const float4*)(wb + 32 * S::WROW + (8 * g + 4 * half) * 4);
            const float gs = *(const float*)(wb + 32 * S::WROW + 128);
            float v[16];
            const bool is_anc = slot >= p.nslots;
            if (is_anc) {
                // 2 x 2 average over the lanes l, l^1 (next column), l^16 (next image row), l^17
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = acc[r];
                    t += dpp_move<DPP_QUAD_XOR1>(t);
                    t = sum_rows16(t);
                    v[r] = 0.25f * t;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[r];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) { v[4 * g] += b4[g].x; v[4 * g + 1] += b4[g].y; v[4 * g + 2] += b4[g].z; v[4 * g + 3] += b4[g].w; }
            // per-slot L2 normalisation times |gscale| (F.normalize eps 1e-12, efficient.py:85); gscale 0 = pass through (v);
            // gscale < 0: column 31 of the slot is written as 1.0 (K planes: partner of the attention kernel's offset slot)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) { s0 = fmaf(v[4 * g], v[4 * g], s0); s1 = fmaf(v[4 * g + 1], v[4 * g + 1], s1); s2 = fmaf(v[4 * g + 2], v[4 * g + 2], s2); s3 = fmaf(v[4 * g + 3], v[4 * g + 3], s3); }
            float ss = (s0 + s1) + (s2 + s3);
            ss = sum_halves(ss);
            const float f = gs != 0.0f ? fabsf(gs) * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)) : 1.0f;   // |gs| / max(sqrt(ss), 1e-12)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= f;
            if (gs < 0.0f && half) v[15] = 1.0f;   // channel (15 & 3) + 8 * (15 >> 2) + 4 = 31
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next chunk + token pieces landed; older stores long done
#ifdef QA_ABL_NOSTORE
            if (v[0] != 12345.678f) continue;
#endif
            if (is_anc) {
                f16* o = (f16*)p.anc + (int64_t)(slot - p.nslots) * p.anc_plane_stride + m_anc * 32;
                if (gs != 0.0f) store_slot<false>(o, v, half, anc_lane); else store_slot<true>(o, v, half, anc_lane);
            } else {
#ifdef QA_TOKEN_MAJOR   // timing experiment: token-major output rows [m][nslots * 32]
                f16* o = (f16*)p.out + m_tok * (p.nslots * 32) + slot * 32;
#else
                f16* o = (f16*)p.out + (int64_t)slot * p.out_plane_stride + m_tok * 32;
#endif
                if (gs != 0.0f) store_slot<false>(o, v, half); else store_slot<true>(o, v, half);
            }
        }
    }
}

template <int KS>
int launch_qa(const GrlQkvAnchorArgs& p, hipStream_t st) {
    using S = QaShape<KS>;
    const int ntiles = p.B * (p.H >> 1) * (p.W >> 6);
    static const int cap = getenv("GRL_PERSIST_GRID") ? atoi(getenv("GRL_PERSIST_GRID")) : 256;   // tuning knob
    const int grid = ntiles < cap ? ntiles : cap;   // persistent workgroups
    auto kfn = qkv_anchor_kernel<KS>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(QW * 64), S::LDS, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// GRL-Base shape (Cpad = 192, 18 q/k/v + 3 anchor slots): the WEIGHTS stay in registers, the tokens stream.
//
// The kernel above keeps a wave's 32 tokens in registers and streams the 21 weight slots through LDS: 11 chunk barriers per
// 128 tokens, and each barrier drains the MFMA pipe of a workgroup that is alone on its CU (187 us per 4 tiles, 105 of them
// without any HBM traffic).  The whole weight set is 21 x 32 x 192 fp16 = 258 KB -- it fits the register file of one CU.  So:
//   * compute wave w (7 of the 8 waves) owns slots 3w .. 3w+2: 3 x 12 A fragments = 144 VGPRs, loaded once per launch from
//     the same blob the streaming kernel uses;
//   * the token tile (2 image rows x 64 columns, fp16, 400-B rows) is the B operand of everybody, double buffered in LDS;
//     ONE barrier per 128 tokens;
//   * the 8th wave is the loader: it reads the next tile (96 float4 per lane, 4 batches of 24 with the following batch in
//     flight -- the registers the compute waves spend on weights are free here), converts to fp16 and writes the other LDS
//     buffer.  (A first version let every thread fetch one float4 per phase: each phase then waited out an HBM round trip.)
//   * accumulator layout, 2 x 2 anchor pooling across lanes, normalisation and stores are those of the kernel above.
constexpr int QR_W = 8, QR_KS = 12, QR_SLOTS = 21;
constexpr int QR_XROW = 192 * 2 + 16;                 // fp16 token row in LDS (16 B pad: conflict-free ds_read_b128)
constexpr int QR_XBUF = 128 * QR_XROW;                // 51200
constexpr int QR_VEC = QR_SLOTS * 36 * 4;             // bias (32) + gscale (1, padded to 4) floats per slot
constexpr int QR_LDS = 2 * QR_XBUF + QR_VEC;

__global__ __launch_bounds__(QR_W * 64) void qkv_regs_kernel(GrlQkvAnchorArgs p) {
    using S = QaShape<QR_KS>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool compute = wave_u < 7;
    const char* blob = (const char*)p.blob;
    const int tiles_x = p.W >> 6, tiles_img = (p.H >> 1) * tiles_x;
    const int ntiles = p.B * tiles_img;
    if ((int)blockIdx.x >= ntiles) return;

    float* vec = (float*)(smem + 2 * QR_XBUF);     // [slot][36]: bias, gscale
    for (int i = tid; i < QR_SLOTS * 33; i += QR_W * 64) {
        const int slot = i / 33, c = i - 33 * slot;
        vec[slot * 36 + c] = *(const float*)(blob + (size_t)(slot >> 1) * S::BUFP + (slot & 1) * S::SLOT + 32 * S::WROW + 4 * c);
    }

    auto tile_origin = [&](int tile, int64_t& cell0) -> int64_t {
        const int b = tile / tiles_img, t = tile - b * tiles_img;
        const int y2 = t / tiles_x, c64 = t - y2 * tiles_x;
        cell0 = ((int64_t)b * (p.H >> 1) + y2) * (p.W >> 1) + 32 * c64;
        return ((int64_t)b * p.H + 2 * y2) * p.W + 64 * c64;
    };
    // float4 number idx (0 .. 6143) of a tile: token slot idx / 48 (group ts >> 5, image row (ts >> 4) & 1, column
    // 16 * group + (ts & 15)), channels 4 * (idx % 48) ..
    auto x_src = [&](int64_t origin, int idx) -> const float4* {
        const int ts = idx / 48, c4 = idx - 48 * ts;
        const int64_t m = origin + (int64_t)((ts >> 4) & 1) * p.W + 16 * (ts >> 5) + (ts & 15);
        return (const float4*)(p.x + m * p.ldx + 4 * c4);
    };
    auto x_put = [&](char* buf, int idx, float4 v) {
        const int ts = idx / 48, c4 = idx - 48 * ts;
        uint2 o;
        o.x = pack_f16(v.x, v.y);
        o.y = pack_f16(v.z, v.w);
        *(uint2*)(buf + ts * QR_XROW + 8 * c4) = o;
    };

    {   // first tile: all 12 float4 of every thread at once
        int64_t c0;
        const int64_t o0 = tile_origin(blockIdx.x, c0);
#pragma unroll
        for (int ph = 0; ph < 12; ++ph) x_put(smem, ph * 512 + tid, *x_src(o0, ph * 512 + tid));
    }

    if (!compute) {
        // ---- loader wave: the next tile, 96 float4 per lane in 4 batches of 24; batch b + 1 is in flight while b is converted.
        // 4 tokens x 48 float4 = 3 wave-wide loads: load r of token quad q covers float4 number 64 r + lane of the quad.
        int goff[3], woff[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = 64 * r + lane, d = e / 48, c4 = e - 48 * d;
            goff[r] = d * (int)p.ldx + 4 * c4;          // floats from the quad's first token
            woff[r] = d * QR_XROW + 8 * c4;
        }
        int it = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
            __syncthreads();
            const int next_tile = tile + (int)gridDim.x;
            if (next_tile >= ntiles) continue;
            int64_t cn;
            const int64_t next_origin = tile_origin(next_tile, cn);
            char* xn = smem + ((it + 1) & 1) * QR_XBUF;
            // token quad q = token slots 4 q .. 4 q + 3: group q >> 3, image row (q >> 2) & 1, columns 16 * group + 4 * (q & 3) ..
            auto quad = [&](int q) { return p.x + (next_origin + (int64_t)((q >> 2) & 1) * p.W + 16 * (q >> 3) + 4 * (q & 3)) * p.ldx; };
            float4 nb[2][24];
#pragma unroll
            for (int k = 0; k < 24; ++k) nb[0][k] = *(const float4*)(quad(k / 3) + goff[k % 3]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < 3) {
#pragma unroll
                    for (int k = 0; k < 24; ++k) nb[(b + 1) & 1][k] = *(const float4*)(quad(8 * (b + 1) + k / 3) + goff[k % 3]);
                }
#pragma unroll
                for (int k = 0; k < 24; ++k) {
                    const float4 v = nb[b & 1][k];
                    uint2 o;
                    o.x = pack_f16(v.x, v.y);
                    o.y = pack_f16(v.z, v.w);
                    *(uint2*)(xn + (8 * b + k / 3) * 4 * QR_XROW + woff[k % 3]) = o;
                }
            }
        }
        return;
    }

    // ---- this wave's three slots: A fragments (lane = weight row j, k-slots 16 s + 8 half + [0..7]) from the chunk images
    f16x8 A[3][QR_KS];
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int slot = 3 * wave_u + i;
            const char* wb = blob + (size_t)(slot >> 1) * S::BUFP + (slot & 1) * S::SLOT + j * S::WROW + 16 * half;
#pragma unroll
            for (int s = 0; s < QR_KS; ++s) A[i][s] = *(const f16x8*)(wb + 32 * s);
        }
    }
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        __syncthreads();      // tile `it` is complete in its buffer; everybody is done reading the other one
        const char* xt = smem + (it & 1) * QR_XBUF;
        int64_t cell0;
        const int64_t origin = tile_origin(tile, cell0);
        const bool anc_lane = (j & 17) == 0;   // first lane of a 2 x 2 cell (even column, upper row)

#pragma unroll 1
        for (int grp = 0; grp < 4; ++grp) {
            const int64_t m_tok = origin + (int64_t)(j >> 4) * p.W + 16 * grp + (j & 15);      // this lane's token
            const int64_t m_anc = cell0 + 8 * grp + ((j & 15) >> 1);                           // its pooling cell
            const char* rowp = xt + (32 * grp + j) * QR_XROW + 16 * half;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int slot = 3 * wave_u + i;
                const float* vb = vec + slot * 36;
                // the bias is the accumulators' initial value (value r <-> channel (r & 3) + 8 * (r >> 2) + 4 * half); anchor
                // slots: the sum over a 2 x 2 cell then holds 4 x bias, like 4 x the pooled projection
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = *(const float4*)(vb + 8 * g + 4 * half);
                    acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
                }
                const float gs = vb[32];
#pragma unroll
                for (int s = 0; s < QR_KS; ++s) acc = mfma32_f16(A[i][s], *(const f16x8*)(rowp + 32 * s), acc);
                float v[16];
                const bool is_anc = slot >= p.nslots;
                if (is_anc) {
                    // 4 x the 2 x 2 average: lanes l, l^1 (next column), l^16 (next image row), l^17
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float t = acc[r];
                        t += dpp_move<DPP_QUAD_XOR1>(t);
                        v[r] = sum_rows16(t);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[r];
                }
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) { s0 = fmaf(v[4 * g], v[4 * g], s0); s1 = fmaf(v[4 * g + 1], v[4 * g + 1], s1); s2 = fmaf(v[4 * g + 2], v[4 * g + 2], s2); s3 = fmaf(v[4 * g + 3], v[4 * g + 3], s3); }
                const float ss = sum_halves((s0 + s1) + (s2 + s3));
                // |gs| / max(||v||, 1e-12); the anchor sums are 4 x too large: clamp 16 x higher, or scale by 1/4 when there is no norm
                const float f = gs != 0.0f ? fabsf(gs) * __builtin_amdgcn_rsqf(fmaxf(ss, is_anc ? 16e-24f : 1e-24f)) : (is_anc ? 0.25f : 1.0f);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= f;
                if (gs < 0.0f && half) v[15] = 1.0f;
                if (is_anc) {
                    f16* o = (f16*)p.anc + (int64_t)(slot - p.nslots) * p.anc_plane_stride + m_anc * 32;
                    if (gs != 0.0f) store_slot<false>(o, v, half, anc_lane); else store_slot<true>(o, v, half, anc_lane);
                } else {
                    f16* o = (f16*)p.out + (int64_t)slot * p.out_plane_stride + m_tok * 32;
                    if (gs != 0.0f) store_slot<false>(o, v, half); else store_slot<true>(o, v, half);
                }
            }
        }
    }
}

int launch_qr(const GrlQkvAnchorArgs& p, hipStream_t st) {
    const int ntiles = p.B * (p.H >> 1) * (p.W >> 6);
    const int grid = ntiles < 256 ? ntiles : 256;   // persistent workgroups, one per CU (the weights fill its register file)
    hipError_t e = hipFuncSetAttribute((const void*)qkv_regs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QR_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(qkv_regs_kernel, dim3(grid), dim3(QR_W * 64), QR_LDS, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Split-precision variant of the kernel above (round 4; GrlQkvAnchorArgs.lo_blob): blocks whose logit scales sit near the clamp.
//
// The q.k logits multiply the fp16 rounding of x and W by up to 144 (scale 100 x log2 e), and BOTH roundings matter (CPU
// emulation of every rounding point on the clamp-scale fixture: weights exact or inputs exact alone change nothing measurable,
// both exact 8.4e-4 -> 5.9e-4).  Round 3 sent such blocks through the generic three-term split linear + a separate anchor
// launch (x1.57 per step).  Here the low parts ride on what the register-resident design already has:
//   * x_lo = fp16(x - x_hi) is a second token tile in LDS and meets the SAME A fragments (W_hi in registers): one more
//     fp16 MFMA per k-step, no more registers;
//   * W_lo only has to be known to a few bits: e4m3((W - W_hi) 2^(e+4)), 96 KB for the 15 normalised slots, resident in LDS
//     for the whole launch, multiplied with x_8 = e4m3(x_hi / 16) -- converted in registers from the fp16 B operand, the fp8
//     32x32x16 MFMA has the same lane <-> k layout -- at twice the fp16 MFMA rate;
//   * the split slots are all normalised per token (q, k, anchors), so a common factor is free: W_hi and the bias are scaled by
//     2^e in registers once per launch (exact; e <= 14 chosen by the host so that nothing overflows) and the main term
//     accumulates at the scale 2^e of the two low terms -- ONE accumulator per slot, no fold; v slots (gscale 0, not
//     normalised, not multiplied by a logit scale) are not split.
// LDS: 96 KB of W_lo8 leave room for 32-token tiles (hi + lo, double buffered: 51 KB), so a tile is one token group of the
// kernel above and there is one barrier per 32 tokens.  Wave w computes a PAIR of split slots together (the B fragments and
// their fp8 conversion are shared) and then a single slot; wave 3 -- it shares its SIMD with the loader wave -- takes the three
// anchor slots, all split.
constexpr int QS_XT = 32 * QR_XROW;                   // one 32-token fp16 plane (hi or lo)
constexpr int QS_X8ROW = 192 + 16;                    // e4m3 token row (16 B pad: conflict-free ds_read_b128)
constexpr int QS_XBUF = 2 * QS_XT + 32 * QS_X8ROW;    // hi | lo | x8: 32256
constexpr int QS_NLO = 15, QS_LROW = 192, QS_LSLOT = 32 * QS_LROW;   // W_lo8 rows are not padded but XOR-swizzled (below)
constexpr int QS_OFF_LO = 2 * QS_XBUF;                // 64512
constexpr int QS_OFF_VEC = QS_OFF_LO + QS_NLO * QS_LSLOT;
constexpr int QS_LDS = QS_OFF_VEC + QR_VEC;           // 159696

typedef __attribute__((__vector_size__(2 * sizeof(short)))) short s16x2;
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;

// the low-part contraction over 64 channels: one v_mfma_f32_32x32x64_f8f6f4 (e4m3 x e4m3, no block scales).  A and B index k
// the same way, so any (lane half, register, byte) <-> k assignment is right as long as both operands use the same one: here
// registers 2u, 2u+1 of half h hold k = 16 u + 8 h + [0..7] of the 64-block -- the k order of the fp16 fragments -- and
// ops.pack_qkv_anchor_lo / the loader wave store rows with byte 64 c + 32 h + 8 u + t = channel 64 c + 16 u + 8 h + t, so a lane
// half's operand is 32 contiguous bytes.  W_lo8 rows have no pad: 16-B segment s of row j sits at s ^ ((j >> 2) & 3).
__device__ __forceinline__ f32x16 mfma64_fp8(i32x8 a, i32x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

#ifdef QS_DEBUG   // timing probes (s_memtime ticks, 10 ns): [wave 0..7][region 0..7], summed over workgroups and tiles
__device__ unsigned long long qs_dbg[64];
#define QS_T(x) const long long x = __builtin_amdgcn_s_memtime()
#define QS_ADD(i, v) qs_acc[i] += (unsigned long long)(v)
#else
#define QS_T(x)
#define QS_ADD(i, v)
#endif

__global__ __launch_bounds__(QR_W * 64) void qkv_split_kernel(GrlQkvAnchorArgs p) {
    using S = QaShape<QR_KS>;
#ifdef QS_DEBUG
    unsigned long long qs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = lane >> 5, j = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool compute = wave_u < 7;
    const char* blob = (const char*)p.blob;
    const int tiles_x = p.W >> 6, tiles_img = (p.H >> 1) * tiles_x;
    const int ntiles = p.B * tiles_img;
    if ((int)blockIdx.x >= ntiles) return;
    const int nsub = 4 * ((ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);   // 32-token tiles of this workgroup
    const float two_e = ((const float*)p.lo_blob)[1];   // 2^e: scale of the low terms, and of W_hi / the bias of the split slots

    float* vec = (float*)(smem + QS_OFF_VEC);      // [slot][36]: bias, gscale
    for (int i = tid; i < QR_SLOTS * 33; i += QR_W * 64) {
        const int slot = i / 33, c = i - 33 * slot;
        vec[slot * 36 + c] = *(const float*)(blob + (size_t)(slot >> 1) * S::BUFP + (slot & 1) * S::SLOT + 32 * S::WROW + 4 * c);
    }
    for (int i = tid; i < QS_NLO * QS_LSLOT / 16; i += QR_W * 64)
        *(uint4*)(smem + QS_OFF_LO + 16 * i) = *(const uint4*)((const char*)p.lo_blob + 16 + 16 * i);

    // 32-token tile number `it` of this workgroup: token group it & 3 of its 2-row x 64-column tile number it >> 2
    auto sub_origin = [&](int it, int64_t& cell0) -> int64_t {
        const int tile = (int)blockIdx.x + (it >> 2) * (int)gridDim.x, grp = it & 3;
        const int b = tile / tiles_img, t = tile - b * tiles_img;
        const int y2 = t / tiles_x, c64 = t - y2 * tiles_x;
        cell0 = ((int64_t)b * (p.H >> 1) + y2) * (p.W >> 1) + 32 * c64 + 8 * grp;
        return ((int64_t)b * p.H + 2 * y2) * p.W + 64 * c64 + 16 * grp;
    };
    // 4 channels of a token (fp32) -> the hi / lo fp16 planes at byte offset `off`, the e4m3 plane at `off8` (x_hi / 16, clamped:
    // the fp8 conversion does not saturate -- probed, tools/probes/fp8_probe.hip: beyond +-448 it yields NaN)
    auto x_put = [&](char* buf, int off, int off8, float4 v) {
        typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;
        // x_hi: saturating fp32 -> fp16 (common.h, sat16), two values per conversion
        const f16x2 a = __builtin_convertvector(f32x2{sat16(v.x), sat16(v.y)}, f16x2), b = __builtin_convertvector(f32x2{sat16(v.z), sat16(v.w)}, f16x2);
        const uint32_t h0 = __builtin_bit_cast(uint32_t, a), h1 = __builtin_bit_cast(uint32_t, b);
        // x_lo = x - x_hi, one mixed-precision FMA each (the half is read in place), not scaled: it meets the same (scaled) A
        // fragments as x_hi.  Residuals below 2^-14 (|x| < 1/4) are fp16 subnormals: at worst lost, never wrong.  No saturation needed.
        auto resid = [](uint32_t h, float x, bool hi_half) {
            float r;
            if (hi_half) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
            else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
            return r;
        };
        const f16x2 l0 = __builtin_convertvector(f32x2{resid(h0, v.x, false), resid(h0, v.y, true)}, f16x2);
        const f16x2 l1 = __builtin_convertvector(f32x2{resid(h1, v.z, false), resid(h1, v.w, true)}, f16x2);
        *(uint2*)(buf + off) = uint2{h0, h1};
        *(uint2*)(buf + QS_XT + off) = uint2{__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1)};
        // x_8 = e4m3(x_hi / 16) straight from the packed halves, clamped to +-7168 first
        const f16x2 top = {(f16)7168.0f, (f16)7168.0f}, bot = {(f16)-7168.0f, (f16)-7168.0f};
        s16x2 w8 = {0, 0};
        w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w8, __builtin_elementwise_max(__builtin_elementwise_min(a, top), bot), 16.0f, false);
        w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w8, __builtin_elementwise_max(__builtin_elementwise_min(b, top), bot), 16.0f, true);
        *(int*)(buf + 2 * QS_XT + off8) = __builtin_bit_cast(int, w8);
    };
    // byte offset of channels 4 c4 .. 4 c4 + 3 in an e4m3 token row: 64 c + 32 h + 8 u + t for channel 64 c + 16 u + 8 h + t
    auto x8_col = [](int c4) { const int k = 4 * c4; return (k & ~63) + 32 * ((k >> 3) & 1) + 8 * ((k >> 4) & 3) + (k & 7); };

    {   // first tile: 3 float4 per thread (token slot ts: image row ts >> 4, column ts & 15)
        int64_t c0;
        const int64_t o0 = sub_origin(0, c0);
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int idx = ph * 512 + tid, ts = idx / 48, c4 = idx - 48 * ts;
            const int64_t m = o0 + (int64_t)(ts >> 4) * p.W + (ts & 15);
            x_put(smem, ts * QR_XROW + 8 * c4, ts * QS_X8ROW + x8_col(c4), *(const float4*)(p.x + m * p.ldx + 4 * c4));
        }
    }

    if (!compute) {
        // ---- loader wave: 24 float4 per lane and tile (token quad q = slots 4 q .. 4 q + 3: image row q >> 2, columns 4 (q & 3) ..;
        // 4 tokens x 48 float4 = 3 wave-wide loads).  Tile it + 2 is in flight while the compute waves work on tile it + 1's
        // predecessor: its loads are issued right after tile it + 1 has been converted and have a whole tile period to land.
        int goff[3], woff[3], woff8[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = 64 * r + lane, d = e / 48, c4 = e - 48 * d;
            goff[r] = d * (int)p.ldx + 4 * c4;          // floats from the quad's first token
            woff[r] = d * QR_XROW + 8 * c4;
            woff8[r] = d * QS_X8ROW + x8_col(c4);
        }
        // The loads are inline asm on purpose: the compiler's own wait-count bookkeeping put "s_waitcnt vmcnt(9)" right behind
        // the 24 loads of a tile (it wants a fixed number in flight at the loop head), i.e. the loader sat out an HBM round trip
        // per tile and everybody waited for it at the barrier.  Hidden from the compiler, the loads are ordered by hand: issued
        // at the end of a tile period, awaited (vmcnt(0), tied to the registers) at the start of the next one.  The ISA has been
        // checked for copies of these registers between issue and wait (there are none: tools/kernel_resources.sh).
        f32x4 nb[24];
        auto issue = [&](int it) {
            int64_t cn;
            const int64_t o = sub_origin(it, cn);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                const int q = k / 3;
                const float* src = p.x + (o + (int64_t)(q >> 2) * p.W + 4 * (q & 3)) * p.ldx + goff[k % 3];
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nb[k]) : "v"(src) : "memory");
            }
        };
        auto landed = [&]() {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]), "+v"(nb[4]), "+v"(nb[5]), "+v"(nb[6]), "+v"(nb[7]),
                           "+v"(nb[8]), "+v"(nb[9]), "+v"(nb[10]), "+v"(nb[11]), "+v"(nb[12]), "+v"(nb[13]), "+v"(nb[14]), "+v"(nb[15]),
                           "+v"(nb[16]), "+v"(nb[17]), "+v"(nb[18]), "+v"(nb[19]), "+v"(nb[20]), "+v"(nb[21]), "+v"(nb[22]), "+v"(nb[23])
                         :: "memory");
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler-tracked loads of the prologue)
        if (nsub > 1) issue(1);
        for (int it = 0; it < nsub; ++it) {
            QS_T(l0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            QS_T(l1);
            QS_ADD(0, l1 - l0);
            if (it + 1 >= nsub) continue;
            char* xn = smem + ((it + 1) & 1) * QS_XBUF;
            landed();
            QS_T(l2);
            QS_ADD(1, l2 - l1);
#pragma unroll
            for (int k = 0; k < 24; ++k)
                x_put(xn, (k / 3) * 4 * QR_XROW + woff[k % 3], (k / 3) * 4 * QS_X8ROW + woff8[k % 3], float4{nb[k][0], nb[k][1], nb[k][2], nb[k][3]});
            QS_T(l3);
            QS_ADD(2, l3 - l2);
            if (it + 2 < nsub) issue(it + 2);
            QS_T(l4);
            QS_ADD(3, l4 - l3);
        }
#ifdef QS_DEBUG
        if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&qs_dbg[8 * wave_u + i], qs_acc[i]);
#endif
        return;
    }

    // ---- this wave's slots: a pair of split slots (sa, sb) and a single slot sc (split for the anchor wave only) ----
    int sa, sc;
    bool c_split;
    if (wave_u == 3) { sa = 18; sc = 20; c_split = true; }
    else {
        const int w = wave_u < 3 ? wave_u : wave_u - 1, br = w / 3, i = w - 3 * br;   // branch (window | stripe), head
        sa = 9 * br + 2 * i; sc = 9 * br + 6 + i; c_split = false;                    // q/k slots 9 br .. 9 br + 5, v slots 9 br + 6 ..
    }
    sa = __builtin_amdgcn_readfirstlane(sa); sc = __builtin_amdgcn_readfirstlane(sc);
    const int sb = sa + 1;
    auto lo_index = [](int slot) { return slot < 6 ? slot : (slot < 15 ? slot - 3 : slot - 6); };   // rank among the normalised slots
    f16x8 A[3][QR_KS];
    {
        const int sl[3] = {sa, sb, sc};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const char* wb = blob + (size_t)(sl[i] >> 1) * S::BUFP + (sl[i] & 1) * S::SLOT + j * S::WROW + 16 * half;
            const f16 ws = (f16)((i < 2 || c_split) ? two_e : 1.0f);
#pragma unroll
            for (int s = 0; s < QR_KS; ++s) A[i][s] = *(const f16x8*)(wb + 32 * s) * ws;
        }
    }
    // this lane's two 16-B segments of a 64-channel block of W_lo8 row j (swizzled, see mfma64_fp8): + 64 c per block
    const int sw = (j >> 2) & 3;
    const int seg0 = 16 * ((2 * half) ^ sw), seg1 = 16 * ((2 * half + 1) ^ sw);
    const char* la = smem + QS_OFF_LO + lo_index(sa) * QS_LSLOT + j * QS_LROW;
    const char* lb = la + QS_LSLOT;
    const char* lc = smem + QS_OFF_LO + lo_index(c_split ? sc : sa) * QS_LSLOT + j * QS_LROW;
    typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;
    auto ld32 = [](const char* p0, const char* p1) {   // two 16-B LDS reads -> one 32-byte MFMA operand
        const i32x4 a = *(const i32x4*)p0, b = *(const i32x4*)p1;
        return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    const bool anc_lane = (j & 17) == 0;   // first lane of a 2 x 2 cell (even column, upper row)

    auto bias_init = [&](int slot, float ws) -> f32x16 {
        const float* vb = vec + slot * 36;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *(const float4*)(vb + 8 * g + 4 * half);
            acc[4 * g] = b4.x * ws; acc[4 * g + 1] = b4.y * ws; acc[4 * g + 2] = b4.z * ws; acc[4 * g + 3] = b4.w * ws;
        }
        return acc;
    };
    // pooling (anchor slots), per-slot L2 norm, store: the epilogue of qkv_regs_kernel; `ws2` = square of the factor the slot's
    // values carry (it only moves the 1e-12 floor of the norm)
    auto finish = [&](const f32x16& acc, int slot, float ws2, int64_t m_tok, int64_t m_anc) {
        const float gs = vec[slot * 36 + 32];
        float v[16];
        const bool is_anc = slot >= p.nslots;
        if (is_anc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[r];
                t += dpp_move<DPP_QUAD_XOR1>(t);
                v[r] = sum_rows16(t);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r];
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) { s0 = fmaf(v[4 * g], v[4 * g], s0); s1 = fmaf(v[4 * g + 1], v[4 * g + 1], s1); s2 = fmaf(v[4 * g + 2], v[4 * g + 2], s2); s3 = fmaf(v[4 * g + 3], v[4 * g + 3], s3); }
        const float ss = sum_halves((s0 + s1) + (s2 + s3));
        const float f = gs != 0.0f ? fabsf(gs) * __builtin_amdgcn_rsqf(fmaxf(ss, (is_anc ? 16e-24f : 1e-24f) * ws2)) : (is_anc ? 0.25f : 1.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= f;
        if (gs < 0.0f && half) v[15] = 1.0f;
        if (is_anc) {
            f16* o = (f16*)p.anc + (int64_t)(slot - p.nslots) * p.anc_plane_stride + m_anc * 32;
            if (gs != 0.0f) store_slot<false>(o, v, half, anc_lane); else store_slot<true>(o, v, half, anc_lane);
        } else {
            f16* o = (f16*)p.out + (int64_t)slot * p.out_plane_stride + m_tok * 32;
            if (gs != 0.0f) store_slot<false>(o, v, half); else store_slot<true>(o, v, half);
        }
    };
    const float two_2e = two_e * two_e;
    // The two split slots sa, sb go together (they share the token fragments).  The k loops are software-pipelined by hand: the fragments of k-step s + 1 (and, at the head of a 64-channel block, the
    // block's fp8 operands) are requested before the MFMAs of step s, and a scheduling barrier keeps that order -- left alone the
    // compiler issued reads and MFMAs in alternating batches with "s_waitcnt lgkmcnt(0)" in between (LDS latency exposed ~12
    // times per slot pair, measured 4.2 k cycles for 1.9 k cycles of MFMA work).
    auto do_pair = [&](const char* rowp, const char* row8, int64_t m_tok, int64_t m_anc) {
        f32x16 acc0 = bias_init(sa, two_e), acc1 = bias_init(sb, two_e);
        f16x8 bh = *(const f16x8*)(rowp), bl = *(const f16x8*)(rowp + QS_XT);
        i32x8 x8, wa8, wb8;
#pragma unroll
        for (int s = 0; s < QR_KS; ++s) {
            const int c = s >> 2, u = s & 3;
            f16x8 nh = bh, nl = bl;
            if (s + 1 < QR_KS) { nh = *(const f16x8*)(rowp + 32 * (s + 1)); nl = *(const f16x8*)(rowp + QS_XT + 32 * (s + 1)); }
#ifndef QS_ABL_NOFP8
            if (u == 0) {
                x8 = ld32(row8 + 64 * c, row8 + 64 * c + 16);
                wa8 = ld32(la + 64 * c + seg0, la + 64 * c + seg1);
                wb8 = ld32(lb + 64 * c + seg0, lb + 64 * c + seg1);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            acc0 = mfma32_f16(A[0][s], bh, acc0);
            acc1 = mfma32_f16(A[1][s], bh, acc1);
#ifndef QS_ABL_NOLO
            acc0 = mfma32_f16(A[0][s], bl, acc0);
            acc1 = mfma32_f16(A[1][s], bl, acc1);
#endif
#ifndef QS_ABL_NOFP8
            if (u == 3) {
                acc0 = mfma64_fp8(wa8, x8, acc0);
                acc1 = mfma64_fp8(wb8, x8, acc1);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            bh = nh; bl = nl;
        }
#ifdef QS_DEBUG
        asm volatile("s_nop 0" : "+v"(acc0), "+v"(acc1));   // the MFMA results have arrived
#endif
        QS_T(c2);
        finish(acc0, sa, two_2e, m_tok, m_anc);
        QS_T(c3);
        finish(acc1, sb, two_2e, m_tok, m_anc);
        QS_T(c4);
        QS_ADD(2, c3 - c2);
        QS_ADD(3, c4 - c3);
    };
    auto do_single = [&](const char* rowp, const char* row8, int64_t m_tok, int64_t m_anc) {
        if (c_split) {
            f32x16 acc = bias_init(sc, two_e);
            f16x8 bh = *(const f16x8*)(rowp), bl = *(const f16x8*)(rowp + QS_XT);
            i32x8 x8, wc8;
#pragma unroll
            for (int s = 0; s < QR_KS; ++s) {
                const int c = s >> 2, u = s & 3;
                f16x8 nh = bh, nl = bl;
                if (s + 1 < QR_KS) { nh = *(const f16x8*)(rowp + 32 * (s + 1)); nl = *(const f16x8*)(rowp + QS_XT + 32 * (s + 1)); }
#ifndef QS_ABL_NOFP8
                if (u == 0) {
                    x8 = ld32(row8 + 64 * c, row8 + 64 * c + 16);
                    wc8 = ld32(lc + 64 * c + seg0, lc + 64 * c + seg1);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma32_f16(A[2][s], bh, acc);
#ifndef QS_ABL_NOLO
                acc = mfma32_f16(A[2][s], bl, acc);
#endif
#ifndef QS_ABL_NOFP8
                if (u == 3) acc = mfma64_fp8(wc8, x8, acc);
#endif
                __builtin_amdgcn_sched_barrier(0);
                bh = nh; bl = nl;
            }
            finish(acc, sc, two_2e, m_tok, m_anc);
        } else {
            f32x16 acc = bias_init(sc, 1.0f);
            f16x8 bh = *(const f16x8*)(rowp);
#pragma unroll
            for (int s = 0; s < QR_KS; ++s) {
                f16x8 nh = bh;
                if (s + 1 < QR_KS) nh = *(const f16x8*)(rowp + 32 * (s + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma32_f16(A[2][s], bh, acc);
                __builtin_amdgcn_sched_barrier(0);
                bh = nh;
            }
            finish(acc, sc, 1.0f, m_tok, m_anc);
        }
    };
#ifdef QS_ORDER
    const bool single_first = wave_u >= 4;   // the two compute waves of a SIMD run their MFMA-heavy and VALU-heavy phases out of step
#else
    const bool single_first = false;
#endif

#pragma unroll 1
    for (int it = 0; it < nsub; ++it) {
        QS_T(c0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (not vmcnt: the stores of the previous tile need not have landed)
        __builtin_amdgcn_s_barrier();   // tile `it` is complete in its buffer; everybody is done reading the other one
        const char* xt = smem + (it & 1) * QS_XBUF;
        int64_t cell0;
        const int64_t origin = sub_origin(it, cell0);
        const int64_t m_tok = origin + (int64_t)(j >> 4) * p.W + (j & 15);      // this lane's token
        const int64_t m_anc = cell0 + ((j & 15) >> 1);                          // its pooling cell
        const char* rowp = xt + j * QR_XROW + 16 * half;
        const char* row8 = xt + 2 * QS_XT + j * QS_X8ROW + 32 * half;
        QS_T(c1);
        QS_ADD(0, c1 - c0);
        if (single_first) {
            do_single(rowp, row8, m_tok, m_anc);
            do_pair(rowp, row8, m_tok, m_anc);
        } else {
            do_pair(rowp, row8, m_tok, m_anc);
            QS_T(c5);
            QS_ADD(1, c5 - c1);     // the whole pair phase (regions 2, 3 = its two epilogues)
            do_single(rowp, row8, m_tok, m_anc);
            QS_T(c6);
            QS_ADD(4, c6 - c5);
        }
    }
#ifdef QS_DEBUG
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&qs_dbg[8 * wave_u + i], qs_acc[i]);
#endif
}

#ifdef QS_DEBUG
extern "C" int grl_qs_debug(unsigned long long* out64, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out64, HIP_SYMBOL(qs_dbg), sizeof(unsigned long long) * 64);
    if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(qs_dbg), z, sizeof(z)); }
    return 0;
}
#endif

int launch_qs(const GrlQkvAnchorArgs& p, hipStream_t st) {
    const int ntiles = p.B * (p.H >> 1) * (p.W >> 6);
    const int grid = ntiles < 256 ? ntiles : 256;   // persistent workgroups, one per CU
    hipError_t e = hipFuncSetAttribute((const void*)qkv_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QS_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(qkv_split_kernel, dim3(grid), dim3(QR_W * 64), QS_LDS, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

int64_t qa_chunk_bytes(int Cpad) {
    switch (Cpad / 16) {
        case 4: return QaShape<4>::BUFP;
        case 8: return QaShape<8>::BUFP;
        case 12: return QaShape<12>::BUFP;
        default: return 0;
    }
}

}  // namespace

extern "C" int64_t grl_qkv_anchor_blob_bytes(int32_t Cpad, int32_t nslots, int32_t nanc) {
    if (Cpad <= 0 || nslots <= 0 || nanc < 0 || (Cpad % 32)) return GRL_ERR_BAD_ARG;
    const int64_t cb = qa_chunk_bytes(Cpad);
    if (cb == 0) return GRL_ERR_BAD_ARG;
    return (int64_t)((nslots + nanc + 1) / 2) * cb;
}

extern "C" int64_t grl_qkv_anchor_lo_blob_bytes(int32_t Cpad, int32_t nsplit) {
    if (Cpad != 192 || nsplit != QS_NLO) return GRL_ERR_UNSUPPORTED;
    return 16 + (int64_t)QS_NLO * QS_LSLOT;
}

extern "C" int grl_qkv_anchor_fwd(void* stream, const GrlQkvAnchorArgs* args) {
    const GrlQkvAnchorArgs& p = *args;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0) return 0;
    if ((p.H & 1) || (p.W & 63)) return GRL_ERR_UNSUPPORTED;   // tiles are 2 image rows x 64 columns
    if ((p.Cpad % 32) || p.nslots <= 0 || p.nanc < 0 || (p.ldx % 4) || p.ldx < p.Cpad) return GRL_ERR_BAD_ARG;
    const int64_t M = (int64_t)p.B * p.H * p.W;
    if (p.x == nullptr || p.blob == nullptr || p.out == nullptr || ((uintptr_t)p.blob & 15) != 0 || p.out_plane_stride < M * 32) return GRL_ERR_BAD_ARG;
    if (p.nanc > 0 && (p.anc == nullptr || p.anc_plane_stride < M / 4 * 32)) return GRL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (p.lo_blob != nullptr) {   // split-precision variant: GRL-Base shape with the q,q,q,k,k,k,v,v,v x 2 + 3 anchor slot order only
        if (p.Cpad != 192 || p.nslots != 18 || p.nanc != 3 || ((uintptr_t)p.lo_blob & 15) != 0) return GRL_ERR_UNSUPPORTED;
        return launch_qs(p, st);
    }
    static const bool regs_off = getenv("GRL_QKV_REGS") && atoi(getenv("GRL_QKV_REGS")) == 0;
    if (p.Cpad == 192 && p.nslots + p.nanc == QR_SLOTS && !regs_off) return launch_qr(p, st);
    switch (p.Cpad / 16) {
        case 4: return launch_qa<4>(p, st);
        case 8: return launch_qa<8>(p, st);
        case 12: return launch_qa<12>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}
