// 3x3 convolution 192 -> 192 channels on fp32 token matrices (TransformerStage.conv + residual, conv_after_body + residual:
// models/networks/grl.py:137,164-170,348,516) for gfx950 -- round 5.  Same contract as conv3x3_kernel (csrc/conv.hip, grl_conv3x3_fwd)
// for the shape the stages of GRL-Base use: CinP = CoutP = 192, fp32 in / fp32 out, bias, optional residual, nothing else.
//
// Why a second kernel: the generic one computes 16x16x32 tiles with every wave reading ALL weight fragments of a tap (28 LDS reads per
// 48 small MFMAs), streams the weights through registers into LDS, and synchronises once per tap: 27 barriers and 663 KB of weight
// traffic per 256-pixel tile.  Measured 280 us per 4 tiles = 0.22 of the MFMA peak (profiles/r04_bench_kernels.txt), 6 % of a step.
// Here
//   * 32x32x16 MFMAs, 8 waves as 4 (pixel rows) x 2 (channel halves): a wave owns 2 image rows x 32 pixels x 96 output channels =
//     six 32x32 accumulators; a k-step reads 3 weight + 2 pixel fragments for 6 MFMAs (5 KB of LDS reads per 192 matrix-pipe cycles
//     instead of 28 KB per 768);
//   * input channels in chunks of 16 = ONE k-step per tap: the weights of all nine taps of a chunk (9 x 192 rows x 32 B = 54 KB) come
//     in by LDS-DMA straight from the packed [9][CoutP][CinP] array grl_conv3x3_fwd already receives (a row's 32 bytes are contiguous
//     there), double buffered -- one barrier per chunk (12 per tile), 54 MFMAs per wave between two barriers;
//   * the (8+2) x (32+2) halo tile (fp32 -> fp16 on the way, zeros outside the image) goes through registers in GROUPS of 64 channels
//     (128 contiguous bytes per pixel), requested a group ahead and left in flight for two weight chunks: the activations come from HBM, the weights
//     from L2, and with one workgroup per CU nothing else hides that latency;
//   * weight rows are 32 bytes, so a b128 fragment read has a 32-byte lane stride: the two 16-byte halves of row n sit at position
//     h ^ ((n >> 3) & 1) (source-side for the DMA) -- 16 rows with distinct n mod 16 cover 16 distinct 16-byte bank groups; the 64-byte
//     pixel rows are swizzled by (pix >> 2) & 3 to the same effect (conflict-free for every tap shift).
// Timing-ablation switches (C9_ABL_*) compute WRONG results by construction; they only build together with -DGRL_ABLATION.
#if !defined(GRL_ABLATION) && (defined(C9_ABL_NOMFMA) || defined(C9_ABL_NOSTORE) || defined(C9_ABL_NORESID) || defined(C9_ABL_NOINPUT) || defined(C9_ABL_NODMA))
#error "timing-ablation switch without -DGRL_ABLATION: the results of such a build are wrong"
#endif
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

constexpr int C9_TH = 8, C9_TW = 32, C9_HW = C9_TW + 2, C9_HH = C9_TH + 2, C9_NPIX = C9_HW * C9_HH;   // 340 halo pixels
constexpr int C9_C = 192, C9_KC = 16, C9_GC = 32;          // weight chunk: 16 input channels (one k-step per tap); input group: 32
constexpr int C9_WBUF = 9 * C9_C * 32;          // 55296: weights of one chunk, all taps
constexpr int C9_IBUF = C9_NPIX * C9_GC * 2;    // 21760: the halo tile of one input group, fp16, 64-byte rows (swizzled)
constexpr int C9_LDS = 2 * C9_WBUF + C9_IBUF;   // 132352
constexpr int C9_WPIECES = C9_WBUF / 1024;      // 54 DMA pieces of 1 KiB
constexpr int C9_THREADS = 512;
constexpr int C9_INV = (C9_NPIX * (C9_GC / 4) + C9_THREADS - 1) / C9_THREADS;   // float4 input pieces per thread and group (6)

__global__ __launch_bounds__(C9_THREADS) void conv192_kernel(GrlConvArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 3, wn = wave >> 2;
    const int x0 = blockIdx.x * C9_TW, y0 = blockIdx.y * C9_TH, b = blockIdx.z;
    const int ngroups = p.CinP / C9_GC;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // ---- weight DMA: piece i = LDS bytes [1024 i, +1024) of a weight buffer = flattened rows 32 i .. 32 i + 31 (row R = tap * 192 + cout) ----
    typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
    u32x4 wsrd;
    {
        const uint64_t a = (uint64_t)p.w;
        wsrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        wsrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        wsrd[2] = 0xffffffffu;
        wsrd[3] = 0x00020000u;
    }
    constexpr int WPW = (C9_WPIECES + 7) / 8;     // pieces per wave (7)
    uint32_t wvoff[WPW];                          // per-lane byte offset of the piece's source (chunk 0)
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
        const int piece = wave + 8 * k;
        const int R = 32 * piece + (lane >> 1), pos = lane & 1;
        const int tap = R / C9_C, r = R - tap * C9_C;
        const int h = pos ^ ((R >> 3) & 1);
        wvoff[k] = (uint32_t)(((int64_t)tap * p.w_tap_stride + (int64_t)r * p.CinP + 8 * h) * 2);
    }
    auto dma_piece = [&](int k, int chunk, int buf) {
        const int piece = wave + 8 * k;
        if (piece < C9_WPIECES) {
            const uint32_t m0v = lds0 + (uint32_t)buf * C9_WBUF + piece * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(wvoff[k]), "s"(wsrd),
                         "s"((uint32_t)(chunk * C9_KC * 2)) : "memory");
        }
    };
    auto dma_weights = [&](int chunk, int buf) {
#pragma unroll
        for (int k = 0; k < WPW; ++k) dma_piece(k, chunk, buf);
    };

    // ---- input halo, one GROUP of 64 channels at a time: thread s -> halo pixel s >> 4, channels 4 (s & 15) .. + 3 of the group.
    // The fp32 rows come from HBM (the weights from L2): they are requested a group ahead into registers and stay in flight for two
    // weight chunks (the memory counter retires in order, see the chunk loop).
    auto in_row = [&](int k, bool& ok) -> int64_t {
        const int s = tid + k * C9_THREADS;
        const int pix = s / (C9_GC / 4);
        const int hy = pix / C9_HW, hx = pix - hy * C9_HW;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        ok = pix < C9_NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        return (((int64_t)b * p.H + gy) * p.W + gx) * p.ldx + 4 * (s % (C9_GC / 4));
    };
    float4 iv[C9_INV];
    auto load_input = [&](int group) {
#pragma unroll
        for (int k = 0; k < C9_INV; ++k) {
            // (unconditional loads from a clamped address, zeroed afterwards: every wave issues exactly C9_INV of them, which the
            // s_waitcnt vmcnt(C9_INV) of the chunk loop counts on)
            bool ok;
            const int64_t off = in_row(k, ok);
            const float4 v = *(const float4*)((const float*)p.x + (ok ? off : (int64_t)(4 * (tid & 7))) + group * C9_GC);
            iv[k] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // row of pixel `pix`: four 16-byte segments, segment g at position g ^ ((pix >> 2) & 3): with 64-byte rows (16 banks of the 64) a
    // b128 read of 16 lanes with distinct pix mod 16 then covers 16 distinct (pix mod 4, position) pairs -- conflict free for every tap
    auto store_input = [&]() {
#pragma unroll
        for (int k = 0; k < C9_INV; ++k) {
            const int s = tid + k * C9_THREADS;
            const int pix = s / (C9_GC / 4), q = s % (C9_GC / 4);
            if (pix < C9_NPIX) {
                uint2 pk;
                pk.x = pack_f16(iv[k].x, iv[k].y);
                pk.y = pack_f16(iv[k].z, iv[k].w);
                *(uint2*)(smem + 2 * C9_WBUF + pix * (C9_GC * 2) + (((q >> 1) ^ ((pix >> 2) & 3)) << 4) + ((q & 1) << 3)) = pk;
            }
        }
    };

    // ---- fragment addresses ----
    // weights: row tap * 192 + 96 wn + 32 i + l31 -> the swizzle bit only depends on l31 (96 wn + 32 i and 192 tap are multiples of 16)
    const uint32_t a_lane = (uint32_t)((96 * wn + l31) * 32 + ((half ^ ((l31 >> 3) & 1)) << 4));
    // pixels: halo index (2 wm + rs) * 34 + dx + l31 for the 4 row shifts rs = j + dy and the 3 column shifts dx; the 16-byte segment
    // of chunk cc (0..3) of the group and k-half `half` is 2 cc + half
    int b_pix[4][3];
#pragma unroll
    for (int rs = 0; rs < 4; ++rs)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) b_pix[rs][dx] = (2 * wm + rs) * C9_HW + dx + l31;

    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: weights of chunk 0, input group 0 ----
    dma_weights(0, 0);
    load_input(0);
    store_input();

#pragma unroll 1
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll 1      // (unrolled, the four chunks' fragment addresses live in registers at once: 256 VGPRs + scratch)
        for (int cc = 0; cc < C9_GC / C9_KC; ++cc) {
            const int c = g * (C9_GC / C9_KC) + cc, buf = c & 1;
            // own DMA pieces of chunk c landed (and, at cc = 0, own LDS writes).  The memory counter retires in order: at cc = 1 the
            // C9_INV halo loads of the next group were issued BEHIND the pieces of this chunk and may stay in flight (they get two
            // chunks, ~7 k cycles, before the wait of cc = 2 retires them with the next pieces)
#if defined(C9_ABL_NOINPUT) || defined(C9_ABL_NODMA)
            if (false) {}
#else
            if (cc == 1 && g + 1 < ngroups) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C9_INV) : "memory");
#endif
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // everybody's; all readers of the other weight buffer are done
            // The seven DMA pieces of the next chunk are NOT issued here in one burst: an LDS-DMA instruction holds the issuing wave for
            // ~100 cycles, longer when all eight waves queue at the texture-address unit right behind the barrier (ablation: the
            // weight DMA cost 54 of 275 us).  Piece k goes out in front of tap k: the partner wave of the SIMD runs MFMAs meanwhile.
            // The halo loads of the next group go out behind the LAST piece (tap 7): the memory-counter waits above rely on that order.
            const bool more = c + 1 < ngroups * (C9_GC / C9_KC);
            const char* wb = smem + buf * C9_WBUF + a_lane;
            const char* ib = smem + 2 * C9_WBUF;
            // fragments of tap t + 1 are requested BEFORE the six MFMAs of tap t (192 matrix-pipe cycles of cover): left to the compiler
            // a read sat 2..4 MFMAs ahead of its use and the wave stalled on lgkmcnt at every tap (MFMA-only ablation: 52 % pipe use)
            const int seg = 2 * cc + half;
            auto frags = [&](int tap, f16x8 (&a)[3], f16x8 (&bb)[2]) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                const int p0 = b_pix[dy][dx], p1 = b_pix[dy + 1][dx];
                bb[0] = *(const f16x8*)(ib + p0 * (C9_GC * 2) + ((seg ^ ((p0 >> 2) & 3)) << 4));
                bb[1] = *(const f16x8*)(ib + p1 * (C9_GC * 2) + ((seg ^ ((p1 >> 2) & 3)) << 4));
#pragma unroll
                for (int i = 0; i < 3; ++i) a[i] = *(const f16x8*)(wb + (tap * C9_C + 32 * i) * 32);
            };
            f16x8 fa[2][3], fb[2][2];
            frags(0, fa[0], fb[0]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = tap & 1;
#ifndef C9_ABL_NODMA
                if (tap < WPW && more) dma_piece(tap, c + 1, buf ^ 1);
#endif
#ifndef C9_ABL_NOINPUT
                if (tap == WPW && cc == 0 && g + 1 < ngroups) load_input(g + 1);
#endif
                if (tap + 1 < 9) frags(tap + 1, fa[cur ^ 1], fb[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#ifndef C9_ABL_NOMFMA
                    acc[i][0] = mfma32_f16(fa[cur][i], fb[cur][0], acc[i][0]);
                    acc[i][1] = mfma32_f16(fa[cur][i], fb[cur][1], acc[i][1]);
#else
                    acc[i][0][0] += (float)fa[cur][i][0] + (float)fb[cur][0][0]; acc[i][1][0] += (float)fa[cur][i][1] + (float)fb[cur][1][0];
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (g + 1 < ngroups) {
            // the next group's halo (in registers since the start of this group) replaces this one: every wave must be done reading it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            store_input();
        }
    }

    // ---- epilogue.  A lane owns channels 96 wn + 32 i + 8 g + 4 half + [0..3] of pixel (y0 + 2 wm + j, x0 + l31): stored from there,
    // a wave instruction touches 32 token rows with 32 bytes each -- residual read and output write in that pattern were 93 of the
    // kernel's 313 us (ablation, profiles/r05_conv192.txt).  So the accumulators (+ bias) of one image row go through a wave-private
    // LDS tile [32 pixels][96 channels] and leave row-wise: 384 contiguous bytes per pixel, 16 bytes per lane. ----
    const float osc = p.out_scale != 0.0f ? p.out_scale : 1.0f;
    constexpr int EROW = 96 * 4 + 16;                    // bytes per pixel row of the tile (16 B pad)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // every wave is done with the weight / halo buffers
    char* et = smem + wave * (32 * EROW);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = 32 * i + 8 * g + 4 * half;            // channel inside the wave's 96
                const float4 b4 = *(const float4*)(p.bias + 96 * wn + cl);
                *(float4*)(et + l31 * EROW + cl * 4) = float4{fmaf(acc[i][j][4 * g], osc, b4.x), fmaf(acc[i][j][4 * g + 1], osc, b4.y),
                                                               fmaf(acc[i][j][4 * g + 2], osc, b4.z), fmaf(acc[i][j][4 * g + 3], osc, b4.w)};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (wave-private tile: no barrier)
        const int gy = y0 + 2 * wm + j;
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int f = it * 64 + lane;                            // float4 index inside the tile: pixel f / 24, channel quad f % 24
            const int px = f / 24, q = f - px * 24;
            const int gx = x0 + px;
            float4 v = *(const float4*)(et + px * EROW + q * 16);
            if (gy < p.H && gx < p.W) {
#ifndef C9_ABL_NORESID
                const int64_t row = ((int64_t)b * p.H + gy) * p.W + gx;
                if (p.resid != nullptr) {
                    const float4 r4 = *(const float4*)(p.resid + row * p.ldr + 96 * wn + 4 * q);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
#else
                const int64_t row = ((int64_t)b * p.H + gy) * p.W + gx;
#endif
#ifdef C9_ABL_NOSTORE
                if (v.x == 12345.678f)
#endif
                *(float4*)((float*)p.out + row * p.ldo + 96 * wn + 4 * q) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the tile is rewritten for the next image row
    }
}

}  // namespace

// 1 when grl_conv3x3_fwd's arguments are the stage-conv shape this kernel serves
bool grl_conv192_supported(const GrlConvArgs& p) {
    return p.CinP == C9_C && p.CoutP == C9_C && (p.CinP % C9_GC) == 0 && p.x_dtype == GRL_DT_F32 && p.out_dtype == GRL_DT_F32 && p.x_split <= 1 && p.act == 0 &&
           p.shuffle_r <= 1 && p.pool_partial == nullptr && (p.x_scale == 0.0f || p.x_scale == 1.0f) && (p.ldx % 4) == 0 && (p.ldo % 4) == 0 &&
           (p.resid == nullptr || (p.ldr % 4) == 0) && p.w_tap_stride == (int64_t)C9_C * C9_C && p.x_cols == 0 && p.n_store == 0;
}

int grl_conv192_launch(const GrlConvArgs& p, hipStream_t st) {
    const dim3 grid((p.W + C9_TW - 1) / C9_TW, (p.H + C9_TH - 1) / C9_TH, p.B);
    hipError_t e = hipFuncSetAttribute((const void*)conv192_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C9_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv192_kernel, grid, dim3(C9_THREADS), C9_LDS, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
