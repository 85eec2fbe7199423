// Streaming QKV projection with the cosine-attention prologue fused (gfx950).
//
//   planes[slot][m][0..31] = groupnorm_slot( x[m, :] . W_slot^T + b_slot )        fp16 head planes
//
// Replaces QKVProjection.forward (models/common/mixed_attn_block.py:669-676) + the F.normalize / logit-scale of
// Attention.attn (models/common/mixed_attn_block_efficient.py:85-90, :39) exactly like the GRL_EPI_GROUPNORM path of
// csrc/linear.hip, but in ONE pass over the residual stream: the weights-resident kernel needs two column slabs
// (576 x 192 fp16 > 160 KB LDS) and therefore reads x twice.  Here
//   * the fp32 token tile (128 rows) is DMA'd into LDS in 1-KiB pieces while the previous tile is multiplied;
//   * the weights stream from L2 through a double-buffered LDS ring, SPC head slots (32 output columns each) per
//     chunk; the chunk image in global memory already is the padded LDS layout, so a chunk is a handful of
//     global_load_lds_dwordx4 (no staging registers, no ds_write pass);
//   * a wave owns 16 tokens: its fp16 operand slab stays in VGPRs for the whole sweep, every chunk is
//     2*SPC n-tiles x K/32 MFMAs, the per-slot L2 normalisation needs two cross-lane adds, and the store of a slot
//     is 1 KiB contiguous per wave (head-plane layout).
// DMA ordering is by hand (inline asm, counted s_waitcnt before the publishing barrier) as in csrc/mlp.hip.
// Timing-ablation switches of this file compute WRONG results by construction (they remove work to see what it costs).  They only
// build together with -DGRL_ABLATION, which tools/attn_asm/build_variants_generic.sh passes for its throw-away variant libraries.
#if !defined(GRL_ABLATION) && (defined(QKV_ABL_NOXDMA) || defined(QKV_ABL_NOMFMA) || defined(QKV_ABL_NOSTORE))
#error "timing-ablation switch without -DGRL_ABLATION: the results of such a build are wrong"
#endif
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int KSTEPS, int SPC>
struct QkvShape {
    static constexpr int CP = KSTEPS * 32;
    static constexpr int WROW = CP * 2 + 16;                   // bytes per weight row (16 B pad)
    static constexpr int SLOT = 32 * WROW + 128 + 16;          // 32 rows | bias (32 fp32) | gscale (fp32, padded to 16 B)
    static constexpr int BUF = SPC * SLOT;
    static constexpr int BUFP = (BUF + 1023) / 1024 * 1024;
    static constexpr int PIECES = BUFP / 1024;
};

// WV waves per workgroup.  TOK = token groups (16 tokens each) per tile; with WV == 2 * TOK (SPC == 2) the waves w and
// w + TOK share a token group and take one head slot of every chunk each: twice the waves per SIMD to overlap the LDS
// read, MFMA, normalise and store phases of a chunk.
template <int KSTEPS, int SPC, int WV, int TOK>
__global__ __launch_bounds__(WV * 64) void qkv_kernel(GrlQkvArgs p) {
    static_assert(WV == TOK || (WV == 2 * TOK && SPC == 2), "waves either own a token group or half of its slots");
    using S = QkvShape<KSTEPS, SPC>;
    constexpr int CP = S::CP;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int nchunks = p.nslots / SPC;
    const char* blob = (const char*)p.blob;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    auto fetch = [&](int chunk, int buf_off) {   // this wave's 1-KiB pieces of the chunk image
        const char* src = blob + (size_t)chunk * S::BUFP + lane * 16;
#pragma unroll
        for (int q0 = 0; q0 < S::PIECES; q0 += WV) {
            const int q = q0 + wave_u;
            if (q < S::PIECES) {
                const uint32_t m0v = lds0 + buf_off + q * 1024;
                const char* g = src + q * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
            }
        }
    };
    constexpr int XROW = CP * 4 + 16, XSEG = XROW / 16;
    constexpr int XPIECES = (TOK * 16 * XROW + 1023) / 1024;
    const int xoff = 2 * S::BUFP;
    auto fetch_x = [&](int tile, int piece) {
        const int sigma = piece * 64 + lane;
        int row = sigma / XSEG, seg = sigma - row * XSEG;
        seg = seg < XSEG - 1 ? seg : XSEG - 2;
        int m = tile * (TOK * 16) + row;
        m = m < p.M ? m : p.M - 1;
        const float* g = p.x + (int64_t)m * p.ldx + seg * 4;
        const uint32_t m0v = lds0 + xoff + piece * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
    };
    const char* xt = smem + xoff;

    const int ntiles = (p.M + TOK * 16 - 1) / (TOK * 16);
    if ((int)blockIdx.x >= ntiles) return;
    const int tg = wave % TOK, sl0 = wave / TOK;          // token group of this wave; first slot it computes
    constexpr int SLS = WV / TOK;                          // slot stride
    for (int q = wave_u; q < XPIECES; q += WV) fetch_x(blockIdx.x, q);
    fetch(0, 0);
    int it = 0;
    // DMA completion is awaited just BEFORE a chunk's output stores, not at the top of the next chunk: vmcnt also counts
    // stores, so a wait at the chunk top would sit out the write acknowledgements of the stores issued a moment earlier
    // (and the HBM latency of the token pieces of the next tile) once per chunk -- the kernel was latency bound on it.
    static_assert(true, "");

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * (TOK * 16) + tg * 16 + r16;
        const bool valid = m < p.M;
        __builtin_amdgcn_s_barrier();   // the tile's token pieces were awaited inside the previous tile's chunks (first tile: above)
        // operand slab: lane = token r16; k-slots 8*g4 + [0..7] of k-step s = channels 32s + 4*g4 + [0..3] and
        // 32s + 16 + 4*g4 + [0..3] (the weight columns are packed in the same order)
        gemm_x8 a[KSTEPS];
        {
            const char* rowp = xt + (tg * 16 + r16) * XROW + 16 * g4;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const float4 v0 = *(const float4*)(rowp + 128 * s), v1 = *(const float4*)(rowp + 128 * s + 64);
                gemm_x8 v;
                v[0] = to_f16(v0.x); v[1] = to_f16(v0.y); v[2] = to_f16(v0.z); v[3] = to_f16(v0.w);
                v[4] = to_f16(v1.x); v[5] = to_f16(v1.y); v[6] = to_f16(v1.z); v[7] = to_f16(v1.w);
                a[s] = v;
            }
        }
        const int next_tile = tile + (int)gridDim.x;
        f16* orow = (f16*)p.out + (int64_t)(valid ? m : 0) * 32 + 4 * g4;

#pragma unroll 1
        for (int c = 0; c < nchunks; ++c, ++it) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's DMA pieces of chunk c were awaited before the previous chunk's stores)
            __builtin_amdgcn_s_barrier();
            const char* cur = smem + (it & 1) * S::BUFP;
            fetch(c + 1 < nchunks ? c + 1 : 0, ((it + 1) & 1) * S::BUFP);
#ifndef QKV_ABL_NOXDMA
            for (int q = c * WV + wave_u; q < XPIECES; q += nchunks * WV) fetch_x(next_tile, q);
#endif
#pragma unroll
            for (int si = 0; si < SPC / SLS; ++si) {
                const int sl = sl0 + si * SLS;
                const char* wb = cur + sl * S::SLOT;
                gemm_x8 wa[KSTEPS], wb_[KSTEPS];
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    wa[s] = *(const gemm_x8*)(wb + r16 * S::WROW + (32 * s + 8 * g4) * 2);
                    wb_[s] = *(const gemm_x8*)(wb + (16 + r16) * S::WROW + (32 * s + 8 * g4) * 2);
                }
                const float4 bA = *(const float4*)(wb + 32 * S::WROW + (4 * g4) * 4);
                const float4 bB = *(const float4*)(wb + 32 * S::WROW + (16 + 4 * g4) * 4);
                const float gs = *(const float*)(wb + 32 * S::WROW + 128);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 h0 = f32x4{0, 0, 0, 0}, h1 = f32x4{0, 0, 0, 0};
#pragma unroll
#ifdef QKV_ABL_NOMFMA
                for (int s = 0; s < 1; ++s) {
#else
                for (int s = 0; s < KSTEPS; ++s) {
#endif
                    h0 = mfma16_gemm(wa[s], a[s], h0);
                    h1 = mfma16_gemm(wb_[s], a[s], h1);
                }
                __builtin_amdgcn_sched_barrier(0);
                h0[0] += bA.x; h0[1] += bA.y; h0[2] += bA.z; h0[3] += bA.w;
                h1[0] += bB.x; h1[1] += bB.y; h1[2] += bB.z; h1[3] += bB.w;
                // per-slot L2 normalisation times |gscale| (F.normalize eps 1e-12, efficient.py:85); gscale 0 = pass through (v);
                // gscale < 0: column 31 of the slot is written as 1.0 (K planes: partner of the attention kernel's offset slot)
                float ss = h0[0] * h0[0] + h0[1] * h0[1] + h0[2] * h0[2] + h0[3] * h0[3] + h1[0] * h1[0] + h1[1] * h1[1] +
                           h1[2] * h1[2] + h1[3] * h1[3];
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                const float f = gs != 0.0f ? fabsf(gs) / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
                const float c31 = (gs < 0.0f && g4 == 3) ? 1.0f : h1[3] * f;
                uint2 lo, hi;
                lo.x = pack_f16(h0[0] * f, h0[1] * f); lo.y = pack_f16(h0[2] * f, h0[3] * f);
                hi.x = pack_f16(h1[0] * f, h1[1] * f); hi.y = pack_f16(h1[2] * f, c31);
                if (si == SPC / SLS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next chunk + token pieces landed; older stores long done
#ifdef QKV_ABL_NOSTORE
                if (valid && lo.x == 0x12345678u) {
#else
                if (valid) {
#endif
                    f16* o = orow + (int64_t)(c * SPC + sl) * p.out_plane_stride;
                    *(uint2*)(o) = lo;           // channels 4*g4 + [0..3]
                    *(uint2*)(o + 16) = hi;      // channels 16 + 4*g4 + [0..3]
                }
            }
        }
    }
}

template <int KSTEPS, int SPC, int WV, int TOK = WV>
int launch_qkv(const GrlQkvArgs& p, hipStream_t st) {
    using S = QkvShape<KSTEPS, SPC>;
    const size_t lds = 2 * (size_t)S::BUFP + (size_t)((TOK * 16 * (S::CP * 4 + 16) + 1023) / 1024) * 1024;
    const int ntiles = (p.M + TOK * 16 - 1) / (TOK * 16);
    static const int cap0 = getenv("GRL_PERSIST_GRID") ? atoi(getenv("GRL_PERSIST_GRID")) : 256;   // tuning knob
    const int cap = cap0 * (WV <= 4 ? 2 : 1);       // small workgroups: two per CU (or one beside a workgroup of another kernel)
    const int grid = ntiles < cap ? ntiles : cap;   // persistent workgroups
    auto kfn = qkv_kernel<KSTEPS, SPC, WV, TOK>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WV * 64), lds, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

template <int KSTEPS>
int launch_qkv_k(const GrlQkvArgs& p, hipStream_t st) {
    static const int small = getenv("GRL_QKV_SMALL") ? atoi(getenv("GRL_QKV_SMALL")) : 0;   // experiment: 64-token workgroups
    static const int w16 = getenv("GRL_QKV_W16") ? atoi(getenv("GRL_QKV_W16")) : 1;       // 16 waves, slot-split (-5 % vs 8 waves)
    if (p.nslots % 2 == 0 && w16) return launch_qkv<KSTEPS, 2, 16, 8>(p, st);
    if (p.nslots % 2 == 0) return small ? launch_qkv<KSTEPS, 2, 4>(p, st) : launch_qkv<KSTEPS, 2, 8>(p, st);
    return launch_qkv<KSTEPS, 1, 8>(p, st);
}

int64_t slot_chunk_bytes(int Cpad, int spc) {
    switch (Cpad / 32) {
        case 2: return spc == 2 ? QkvShape<2, 2>::BUFP : QkvShape<2, 1>::BUFP;
        case 4: return spc == 2 ? QkvShape<4, 2>::BUFP : QkvShape<4, 1>::BUFP;
        case 6: return spc == 2 ? QkvShape<6, 2>::BUFP : QkvShape<6, 1>::BUFP;
        default: return 0;
    }
}

}  // namespace

extern "C" int64_t grl_qkv_blob_bytes(int32_t Cpad, int32_t nslots) {
    if (Cpad <= 0 || nslots <= 0 || (Cpad % 32)) return GRL_ERR_BAD_ARG;
    const int spc = nslots % 2 == 0 ? 2 : 1;
    const int64_t cb = slot_chunk_bytes(Cpad, spc);
    if (cb == 0) return GRL_ERR_BAD_ARG;
    return (int64_t)(nslots / spc) * cb;
}

extern "C" int grl_qkv_fwd(void* stream, const GrlQkvArgs* args) {
    const GrlQkvArgs& p = *args;
    if (p.M <= 0) return 0;
    if ((p.Cpad % 32) || p.nslots <= 0 || (p.ldx % 4) || p.ldx < p.Cpad || p.out_plane_stride < (int64_t)p.M * 32) return GRL_ERR_BAD_ARG;
    if (p.x == nullptr || p.blob == nullptr || p.out == nullptr || ((uintptr_t)p.blob & 15) != 0) return GRL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (p.Cpad / 32) {
        case 2: return launch_qkv_k<2>(p, st);
        case 4: return launch_qkv_k<4>(p, st);
        case 6: return launch_qkv_k<6>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}
