// 3x3 convolutions of the GRL path as fp16-operand MFMA implicit GEMM over an LDS halo tile (gfx950).
//
// Replaces:
//   CAB: conv3x3 C->C/4, GELU, conv3x3 C/4->C + the global-average pool of ChannelAttention
//        (models/common/mixed_attn_block.py:948-983)
//   TransformerStage.conv / conv_after_body (+ residual)   models/networks/grl.py:137,168,348,516
//   conv_first, conv_before_upsample(+LeakyReLU), Upsample convs (+PixelShuffle), conv_last
//        models/networks/grl.py:293,352-379 ; models/common/upsample.py:16-19,45-46
//
// Layout: activations are channels-last token matrices [B*H*W, Cpad]; weights are pre-packed to
// fp16 [9 taps][CoutP][CinP] (K contiguous).  One workgroup (8 waves) produces an 8 x 32 pixel
// tile for up to 192 output channels.  Input channels are processed in chunks of KC (64/32):
// the (8+2) x (32+2) halo tile of the chunk is converted to fp16 once and staged in LDS, and for
// each of the 9 taps the [CoutP][KC] weight slice streams through a double-buffered LDS slot
// (global loads for tap t+1 are issued before the MFMAs of tap t).  The product is computed
// transposed (rows = output channel, cols = pixel) exactly like csrc/linear.hip so a lane owns 4
// consecutive output channels of one pixel.  Epilogue: bias, GELU / LeakyReLU, residual, optional
// per-workgroup channel sums (deterministic two-stage global average pool for the SE block) and an
// optional pixel-shuffle store.
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 32;       // output tile (pixels); wave w owns tile row w
constexpr int CWAVES = 8;
constexpr int HALO_W = TW + 2, HALO_H = TH + 2;

// (Keeping the weight slices of all 9 taps of a channel chunk resident -- two barriers per chunk instead of ten -- was
// measured 25-35 % SLOWER on the CAB convs: the larger LDS footprint leaves one workgroup per CU.  Not kept.)
template <int KC, int NT>
__global__ __launch_bounds__(CWAVES * 64) void conv3x3_kernel(GrlConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = KC * 2 + 16;                 // padded row (bytes) for pixels and weight rows
    constexpr int KS = KC / 32;                       // MFMA k-steps per chunk
    constexpr int IN_BYTES = HALO_H * HALO_W * ROWB;
    constexpr int WT_BYTES = NT * 16 * ROWB;          // one tap's weight slice
    constexpr int SEG_ROW = KC / 8;                   // 16-B segments per row
    constexpr int WSEGS = NT * 16 * SEG_ROW;
    constexpr int WPT = (WSEGS + CWAVES * 64 - 1) / (CWAVES * 64);  // weight segments per thread
    char* in_s = smem;
    char* wt_s = smem + IN_BYTES;                     // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int nkc = p.CinP / KC;

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

    gemm_x8 wpre[WPT];
    auto load_w = [&](int tap, int kc) {
        const gemm_t* src = (const gemm_t*)p.w + (int64_t)tap * p.w_tap_stride + kc * KC;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int s = tid + i * CWAVES * 64;
            if (s < WSEGS) {
                const int rr = s / SEG_ROW, cc = s % SEG_ROW;
                wpre[i] = *(const gemm_x8*)(src + (int64_t)rr * p.CinP + cc * 8);
            }
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int s = tid + i * CWAVES * 64;
            if (s < WSEGS) {
                const int rr = s / SEG_ROW, cc = s % SEG_ROW;
                *(gemm_x8*)(wt_s + buf * WT_BYTES + rr * ROWB + cc * 16) = wpre[i];
            }
        }
    };

    // x_split == 3: the virtual input channels are [hi(x) | lo(x) | hi(x)] over CinP / 3 source channels (grl_hip.h);
    // x_split == 2: [hi(x) | lo(x)] over CinP / 2 -- the activations split, the weights twice as they are
    const int csrc = p.x_split == 3 ? p.CinP / 3 : (p.x_split == 2 ? p.CinP / 2 : p.CinP);
    const float xsc = p.x_scale != 0.0f ? p.x_scale : 1.0f;   // backward pass: gradients pre-scaled into fp16 range
    const float osc = p.out_scale != 0.0f ? p.out_scale : 1.0f;
    auto stage_input = [&](int kc) {
#pragma unroll 2
        for (int s = tid; s < HALO_H * HALO_W * SEG_ROW; s += CWAVES * 64) {
            const int pix = s / SEG_ROW, cc = s % SEG_ROW;
            const int vch = kc * KC + cc * 8;             // first virtual channel of this 16-B segment
            int part = 0, ch0 = vch;
            if (p.x_split >= 2) {                         // (uniform; keeps the integer division out of the common path)
                part = vch / csrc;
                ch0 = vch - part * csrc;
            }
            const int hy = pix / HALO_W, hx = pix % HALO_W;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            gemm_x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                const int64_t row = ((int64_t)b * p.H + gy) * p.W + gx;
                if (p.x_dtype == GRL_DT_F16) {
                    v = *(const gemm_x8*)((const gemm_t*)p.x + row * p.ldx + ch0);
                } else {
                    const float4* q = (const float4*)((const float*)p.x + row * p.ldx + ch0);
                    float4 a0, a1;
                    if (p.x_cols > 0) {   // input at its real width: channels >= x_cols read as zeros
                        a0 = ch0 < p.x_cols ? q[0] : float4{0, 0, 0, 0};
                        a1 = ch0 + 4 < p.x_cols ? q[1] : float4{0, 0, 0, 0};
                    } else {
                        a0 = q[0]; a1 = q[1];
                    }
                    const float e[8] = {a0.x * xsc, a0.y * xsc, a0.z * xsc, a0.w * xsc, a1.x * xsc, a1.y * xsc, a1.z * xsc, a1.w * xsc};
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = to_f16(e[i]);
                    if (p.x_split >= 2 && part == 1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = (f16)(e[i] - (float)v[i]);
                    }
                }
            }
            *(gemm_x8*)(in_s + pix * ROWB + cc * 16) = v;
        }
    };
    auto mfma_tap = [&](int tap, const char* wb) {
        const int dy = tap / 3, dx = tap - dy * 3;
        gemm_x8 af[2][KS];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                af[mt][ks] = *(const gemm_x8*)(in_s + ((wave + dy) * HALO_W + (16 * mt + r16 + dx)) * ROWB + (32 * ks + 8 * g4) * 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const gemm_x8 wf = *(const gemm_x8*)(wb + (nt * 16 + r16) * ROWB + (32 * ks + 8 * g4) * 2);
                acc[0][nt] = mfma16_gemm(wf, af[0][ks], acc[0][nt]);
                acc[1][nt] = mfma16_gemm(wf, af[1][ks], acc[1][nt]);
            }
        }
    };

    for (int kc = 0; kc < nkc; ++kc) {
        load_w(0, kc);
        __syncthreads();  // previous chunk's readers are done with in_s / wt_s
        stage_input(kc);
        store_w(0);
        __syncthreads();
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) load_w(tap + 1, kc);  // in flight during the MFMAs below
            mfma_tap(tap, wt_s + (tap & 1) * WT_BYTES);
            if (tap < 8) {
                store_w((tap + 1) & 1);  // the other buffer: its last readers finished before the previous barrier
                __syncthreads();
            }
        }
    }

    // ---- epilogue: lane owns channels 16*nt + 4*g4 + [0..3] of pixel (y0+wave, x0+16*mt+r16) ----
    const int gy = y0 + wave;
    float psum[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) psum[nt][e] = 0.f;

    // 16-bit outputs without pixel shuffle (CAB convs, conv_before_upsample): the tile is packed into LDS as
    // [pixel][CoutP] rows and written back with whole-row 16-B stores -- a tile row (32 pixels) is contiguous in the
    // token matrix.  Storing from the accumulator layout instead costs 2*NT instructions of 16 x 32-B pieces per wave
    // (measured: 55 % of the CAB conv2 kernel).
    const bool staged = p.out_dtype != GRL_DT_F32 && p.shuffle_r <= 1 && p.resid == nullptr;
    // Channels CoutP .. round_up(CoutP, 32) of the output rows are written as zeros when the row is wide enough
    // (ldo): the consumer's K dimension is padded to 32, so its input needs no separate zero fill.
    constexpr int ZSEGS = ((NT * 16 + 31) / 32 * 32 - NT * 16) / 8;   // 16-B zero segments behind the real channels
    constexpr int OROW = (NT * 2 + ZSEGS) * 16 + 16;   // bytes per staged pixel row (16 B pad: conflict-free 8-B writes)
    if (staged) {
        __syncthreads();   // all MFMA-phase readers of in_s / wt_s are done: the space is reused for the tile
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int gx = x0 + 16 * mt + r16;
            const bool valid = gy < p.H && gx < p.W;
            char* prow = smem + (wave * TW + 16 * mt + r16) * OROW;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = 16 * nt + 4 * g4;
                const float4 b4 = *(const float4*)(p.bias + c);
                float v[4] = {fmaf(acc[mt][nt][0], osc, b4.x), fmaf(acc[mt][nt][1], osc, b4.y), fmaf(acc[mt][nt][2], osc, b4.z), fmaf(acc[mt][nt][3], osc, b4.w)};
                if (p.act == 1) {
{
                        const f32x2v ga = gelu_erf2(f32x2v{v[0], v[1]}), gb = gelu_erf2(f32x2v{v[2], v[3]});
                        v[0] = ga.x; v[1] = ga.y; v[2] = gb.x; v[3] = gb.y;
                    }
                } else if (p.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                }
                if (valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) psum[nt][e] += v[e];
                }
                uint2 pk;
                pk.x = pack_f16(v[0], v[1]);
                pk.y = pack_f16(v[2], v[3]);
                *(uint2*)(prow + c * 2) = pk;
            }
        }
        const int zs = (p.ldo >= (NT * 16 + 31) / 32 * 32) ? ZSEGS : 0;
        if (ZSEGS > 0 && tid < TH * TW) {
#pragma unroll
            for (int z = 0; z < ZSEGS; ++z) *(uint4*)(smem + tid * OROW + (NT * 2 + z) * 16) = uint4{0, 0, 0, 0};
        }
        __syncthreads();
        const int SEGS = NT * 2 + zs;   // 16-B segments per pixel row
        for (int i = tid; i < TH * TW * SEGS; i += CWAVES * 64) {
            const int px = i / SEGS, sg = i - px * SEGS;
            const int ty = px / TW, tx = px - ty * TW;
            const int oy = y0 + ty, ox = x0 + tx;
            if (oy < p.H && ox < p.W) {
                const int64_t row = ((int64_t)b * p.H + oy) * p.W + ox;
                *(uint4*)((gemm_t*)p.out + row * p.ldo + sg * 8) = *(const uint4*)(smem + px * OROW + sg * 16);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < (staged ? 0 : 2); ++mt) {
        const int gx = x0 + 16 * mt + r16;
        const bool valid = gy < p.H && gx < p.W;
        const int64_t row = ((int64_t)b * p.H + gy) * p.W + gx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = 16 * nt + 4 * g4;
            const float4 b4 = *(const float4*)(p.bias + c);
            float v[4] = {fmaf(acc[mt][nt][0], osc, b4.x), fmaf(acc[mt][nt][1], osc, b4.y), fmaf(acc[mt][nt][2], osc, b4.z), fmaf(acc[mt][nt][3], osc, b4.w)};
            if (p.act == 1) {
{
                    const f32x2v ga = gelu_erf2(f32x2v{v[0], v[1]}), gb = gelu_erf2(f32x2v{v[2], v[3]});
                    v[0] = ga.x; v[1] = ga.y; v[2] = gb.x; v[3] = gb.y;
                }
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
            }
            if (valid) {
                if (p.pool_partial != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) psum[nt][e] += v[e];
                }
                if (p.resid != nullptr) {
                    const float4 r4 = *(const float4*)(p.resid + row * p.ldr + c);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                int64_t orow = row;
                int oc = c;
                bool keep = true;
                if (p.shuffle_r > 1) {
                    // PixelShuffle(r) with output channels pre-permuted to (i, j, cc) order at pack time
                    const int r = p.shuffle_r, cg = p.shuffle_cg;
                    const int q = c / cg;
                    const int ij = p.shuffle_ij0 + q;
                    oc = c - q * cg;
                    keep = ij < r * r;  // channel slots beyond r*r sub-pixels are padding
                    orow = ((int64_t)b * p.H * r + (gy * r + ij / r)) * (p.W * r) + (gx * r + ij % r);
                }
                if (!keep || (p.n_store > 0 && oc >= p.n_store)) {   // (n_store: result at its real width)
                } else if (p.out_dtype != GRL_DT_F32) {
                    uint2 pk;
                    pk.x = pack_f16(v[0], v[1]);
                    pk.y = pack_f16(v[2], v[3]);
                    *(uint2*)((gemm_t*)p.out + orow * p.ldo + oc) = pk;
                } else {
                    *(float4*)((float*)p.out + orow * p.ldo + oc) = float4{v[0], v[1], v[2], v[3]};
                }
            }
        }
    }
    if (p.pool_partial != nullptr) {
        // deterministic per-workgroup channel sums: lanes (xor 1,2,4,8) -> wave -> LDS -> one row per WG
        __syncthreads();
        float* red = (float*)smem;  // [CWAVES][NT*16]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = row16_sum(psum[nt][e]);
                if (r16 == 0) red[wave * (NT * 16) + 16 * nt + 4 * g4 + e] = s;
            }
        __syncthreads();
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int c = tid; c < NT * 16; c += CWAVES * 64) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < CWAVES; ++w) s += red[w * (NT * 16) + c];
            p.pool_partial[(int64_t)wg * p.pool_stride + c] = s;
        }
    }
}

template <int KC, int NT>
int launch_conv(const GrlConvArgs& p, hipStream_t st) {
    const dim3 grid((p.W + TW - 1) / TW, (p.H + TH - 1) / TH, p.B);
    const size_t rowb = KC * 2 + 16;
    const size_t lds_in = (size_t)HALO_H * HALO_W * rowb;
    size_t lds = lds_in + 2 * (size_t)NT * 16 * rowb;
    const size_t tile_b = (size_t)TH * TW * (NT * 32 + 32 + 16);   // LDS-staged 16-bit output tile (epilogue), incl. zero segments
    if (lds < tile_b) lds = tile_b;
    auto kfn = conv3x3_kernel<KC, NT>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, grid, dim3(CWAVES * 64), lds, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

template <int KC>
int launch_conv_nt(const GrlConvArgs& p, hipStream_t st) {
    switch (p.CoutP / 16) {
        case 1: return launch_conv<KC, 1>(p, st);
        case 2: return launch_conv<KC, 2>(p, st);
        case 3: return launch_conv<KC, 3>(p, st);
        case 4: return launch_conv<KC, 4>(p, st);
        case 6: return launch_conv<KC, 6>(p, st);
        case 8: return launch_conv<KC, 8>(p, st);
        case 12: return launch_conv<KC, 12>(p, st);
        default: return GRL_ERR_UNSUPPORTED;
    }
}

// SE excitation: scale[b][c] = sigmoid(W2 . relu(W1 . mean_b + b1) + b2)   (mixed_attn_block.py:956-963)
// one 1024-thread workgroup per image: 4 threads per channel split the partial-sum rows (fixed order).
__global__ __launch_bounds__(1024) void se_kernel(const float* __restrict__ partial, int wgs_per_image, int CP, int C,
                                                  int Cmid, float inv_hw, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float* __restrict__ scale) {
    __shared__ float part[16][256];
    __shared__ float mean[256];
    __shared__ float mid[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    {   // 16 row-slices x 64 lanes of float4: partial sums in a fixed order (deterministic)
        const int c4 = (tid & 63) * 4, q = tid >> 6;
        if (c4 < CP) {
            float4 s = float4{0, 0, 0, 0};
            const float* pp = partial + (int64_t)b * wgs_per_image * CP + c4;
#pragma unroll 4
            for (int i = q; i < wgs_per_image; i += 16) {
                const float4 v = *(const float4*)(pp + (int64_t)i * CP);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            part[q][c4] = s.x; part[q][c4 + 1] = s.y; part[q][c4 + 2] = s.z; part[q][c4 + 3] = s.w;
        }
    }
    __syncthreads();
    if (tid < CP) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += part[q][tid];
        mean[tid] = s * inv_hw;
    }
    __syncthreads();
    {   // one wave per hidden unit
        const int lane = tid & 63, wv = tid >> 6;
        for (int j = wv; j < Cmid; j += 16) {
            float s = 0.f;
            for (int k = lane; k < C; k += 64) s += w1[j * C + k] * mean[k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) mid[j] = fmaxf(s + b1[j], 0.f);
        }
    }
    __syncthreads();
    if (tid < CP) {
        float o = 0.f;
        if (tid < C) {
            float s = b2[tid];
            for (int j = 0; j < Cmid; ++j) s += w2[tid * Cmid + j] * mid[j];
            o = 1.0f / (1.0f + __expf(-s));
        }
        scale[(int64_t)b * CP + tid] = o;
    }
}

}  // namespace

// csrc/conv192.hip (round 5): the 192 -> 192 fp32 stage / after-body convolution on 32x32x16 MFMAs, one barrier per 16-channel chunk
bool grl_conv192_supported(const GrlConvArgs& p);
int grl_conv192_launch(const GrlConvArgs& p, hipStream_t st);

extern "C" int grl_conv3x3_fwd(void* stream, const GrlConvArgs* args) {
    const GrlConvArgs& p = *args;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0) return GRL_ERR_BAD_ARG;
    if (p.CinP % 32 || p.CoutP % 16 || p.CoutP > 192 || (p.ldx % (p.x_cols > 0 ? 4 : 8)) || (p.ldo % 4)) return GRL_ERR_BAD_ARG;
    if (p.x_cols < 0 || p.n_store < 0) return GRL_ERR_BAD_ARG;
    if (p.x_cols > 0 && ((p.x_cols % 4) || p.x_cols > p.CinP || p.ldx < p.x_cols || p.x_dtype != GRL_DT_F32 || p.x_split >= 2)) return GRL_ERR_BAD_ARG;
    if (p.n_store > 0 && ((p.n_store % 4) || p.n_store > p.CoutP || p.ldo < p.n_store || p.out_dtype != GRL_DT_F32 || p.shuffle_r > 1)) return GRL_ERR_BAD_ARG;
    if (p.x_split < 0 || p.x_split > 3) return GRL_ERR_BAD_ARG;
    if (p.x_split >= 2 && (p.x_dtype != GRL_DT_F32 || p.CinP % (8 * p.x_split) != 0)) return GRL_ERR_BAD_ARG;   // parts are whole 8-channel segments
    if (p.pool_partial != nullptr && p.pool_stride < p.CoutP) return GRL_ERR_BAD_ARG;
    if (p.shuffle_r > 1 && (p.shuffle_cg <= 0 || (p.shuffle_cg % 4) || (p.CoutP % p.shuffle_cg))) return GRL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    static const int c192_off = getenv("GRL_CONV192") ? atoi(getenv("GRL_CONV192")) == 0 : 0;   // 0: generic kernel (A/B)
    if (!c192_off && grl_conv192_supported(p)) return grl_conv192_launch(p, st);
    static const int kc_env = getenv("GRL_CONV_KC") ? atoi(getenv("GRL_CONV_KC")) : 0;   // tuning knob: 32 forces the 32-channel chunks
    if (p.CinP % 64 == 0 && kc_env != 32) return launch_conv_nt<64>(p, st);
    return launch_conv_nt<32>(p, st);
}

extern "C" int grl_conv3x3_num_workgroups(int32_t B, int32_t H, int32_t W) {
    return B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

extern "C" int grl_se_scale_fwd(void* stream, const float* pool_partial, int32_t B, int32_t wgs_per_image, int32_t CP,
                                int32_t C, int32_t Cmid, int32_t HW, const float* w1, const float* b1, const float* w2,
                                const float* b2, float* scale) {
    if (CP > 256 || Cmid > 64 || C > CP) return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(se_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, pool_partial, wgs_per_image, CP, C, Cmid,
                       1.0f / (float)HW, w1, b1, w2, b2, scale);
    GRL_CHECK_LAUNCH();
    return 0;
}
