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

extern "C" int grl_qkv_anchor_fwd(void* stream, const GrlQkvAnchorArgs* args) {
    const GrlQkvAnchorArgs& p = *args;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0) return 0;
    if ((p.H & 1) || (p.W & 63)) return GRL_ERR_UNSUPPORTED;   // tiles are 2 image rows x 64 columns
    if ((p.Cpad % 32) || p.nslots <= 0 || p.nanc < 0 || (p.ldx % 4) || p.ldx < p.Cpad) return GRL_ERR_BAD_ARG;
    const int64_t M = (int64_t)p.B * p.H * p.W;
    if (p.x == nullptr || p.blob == nullptr || p.out == nullptr || ((uintptr_t)p.blob & 15) != 0 || p.out_plane_stride < M * 32) return GRL_ERR_BAD_ARG;
    if (p.nanc > 0 && (p.anc == nullptr || p.anc_plane_stride < M / 4 * 32)) return GRL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    static const bool regs_off = getenv("GRL_QKV_REGS") && atoi(getenv("GRL_QKV_REGS")) == 0;
    if (p.Cpad == 192 && p.nslots + p.nanc == QR_SLOTS && !regs_off) return launch_qr(p, st);
    switch (p.Cpad / 16) {
        case 4: return launch_qa<4>(p, st);
        case 8: return launch_qa<8>(p, st);
        case 12: return launch_qa<12>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}
