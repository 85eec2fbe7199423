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
__device__ __forceinline__ void store_slot(f16* slot, const float (&v)[16], int half) {
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
    const uint2 sa = half ? pk[0] : pk[1], sb = half ? pk[2] : pk[3];
    uint2 ra, rb;
    ra.x = __shfl_xor(sa.x, 32, 64); ra.y = __shfl_xor(sa.y, 32, 64);
    rb.x = __shfl_xor(sb.x, 32, 64); rb.y = __shfl_xor(sb.y, 32, 64);
    const uint4 lo = half ? uint4{ra.x, ra.y, pk[1].x, pk[1].y} : uint4{pk[0].x, pk[0].y, ra.x, ra.y};
    const uint4 hi = half ? uint4{rb.x, rb.y, pk[3].x, pk[3].y} : uint4{pk[2].x, pk[2].y, rb.x, rb.y};
    *(uint4*)(slot + 8 * half) = lo;
    *(uint4*)(slot + 8 * half + 16) = hi;
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
                    t += __shfl_xor(t, 16, 64);
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
            ss += xhalf(ss);
            const float f = gs != 0.0f ? fabsf(gs) * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)) : 1.0f;   // |gs| / max(sqrt(ss), 1e-12)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= f;
            if (gs < 0.0f && half) v[15] = 1.0f;   // channel (15 & 3) + 8 * (15 >> 2) + 4 = 31
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next chunk + token pieces landed; older stores long done
#ifdef QA_ABL_NOSTORE
            if (v[0] != 12345.678f) continue;
#endif
            if (is_anc) {
                if (anc_lane) {
                    f16* o = (f16*)p.anc + (int64_t)(slot - p.nslots) * p.anc_plane_stride + m_anc * 32;
                    if (gs != 0.0f) store_slot<false>(o, v, half); else store_slot<true>(o, v, half);
                }
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
    switch (p.Cpad / 16) {
        case 4: return launch_qa<4>(p, st);
        case 8: return launch_qa<8>(p, st);
        case 12: return launch_qa<12>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}
