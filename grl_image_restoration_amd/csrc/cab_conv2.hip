// Second 3x3 convolution of the CAB branch (C/4 -> C channels) with the weights held in registers (gfx950).
//
// Replaces CAB.cab[2] + the global-average pool of ChannelAttention (models/common/mixed_attn_block.py:948-983) for the
// GRL-Base shape on the 16-bit path: <= 48 input channels (45 real), <= 192 output channels (180 real).
//
// Why a kernel of its own: K = 9 taps x 45 channels is short and N = 180 is wide, so the generic implicit GEMM of
// csrc/conv.hip (weight slices of one tap streamed through LDS, a barrier per tap, the output tile staged through LDS)
// spends most of a workgroup's life in staging and barriers: 120 us per 4 tiles at about 20 % MFMA-busy.  Here
//   * the whole filter bank, K packed tap-major x 48 channels (432 -> 14 k-steps of 32), 192 x 448 fp16 = 168 KB, lives in
//     the VGPRs of a persistent 8-wave workgroup: wave w holds the A fragments of output channels 48*(w&3) .. +47
//     (3 groups of 16 x 14 k-steps x 4 VGPRs = 168 VGPRs), loaded once per launch straight from the packed blob;
//   * the (8+2) x (32+2) pixel halo tile of the next output tile is brought in by LDS-DMA (buffer_load ... lds, 16 B per
//     lane, out-of-image pixels read as zero through the descriptor's bounds check) while the current one multiplies:
//     one barrier per 256 output pixels;
//   * the im2col operand of a k-step is a single ds_read_b128 per lane from the halo tile (8 consecutive channels of one
//     tap of one pixel); D^T = W . X^T (mfma_f32_16x16x32_f16), so a lane ends up with 4 consecutive output channels of
//     one pixel: bias is the accumulator's initial value, the result goes out as one 8-byte store per 16-channel group;
//   * the channel sums of the SE pool are kept per lane over the workgroup's life and written once, as one row of
//     partial sums per workgroup (deterministic two-stage average, consumed by grl_se_scale_fwd).
// Algorithmic bytes per pixel: 96 (mid, fp16, read ~1.3x with the halo) + 384 (out, fp16); 2 * 405 * 180 flops.
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

constexpr int C2_TH = 8, C2_TW = 32;                  // output pixels per tile; a wave owns 4 rows x 32 pixels
constexpr int C2_HH = C2_TH + 2, C2_HW = C2_TW + 2;   // halo tile
constexpr int C2_SEG = 7;                             // 16-B segments per halo pixel: 48 channels + 1 (pitch 112 B: conflict-free b128)
constexpr int C2_PXB = C2_SEG * 16;
constexpr int C2_NPIECE = C2_HH * C2_HW * C2_SEG;     // 2380 DMA pieces of 16 B
constexpr int C2_WAVES = 8, C2_THREADS = C2_WAVES * 64;
constexpr int C2_ROUNDS = (C2_NPIECE + C2_THREADS - 1) / C2_THREADS;   // 5
constexpr int C2_HALO_B = C2_ROUNDS * C2_THREADS * 16;                // 40960
constexpr int C2_KSTEPS = 14, C2_NG = 3;              // k-steps of 32 (tap-major x 48), 16-channel groups per wave
constexpr int C2_KREAL = 9 * 48;

typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t c2_u32x4;

__global__ __launch_bounds__(C2_THREADS) void cab_conv2_kernel(GrlCabConv2Args p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, g4 = lane >> 4;
    const int cs = wave_u & 3, ph = wave_u >> 2;        // channel set (48 channels), pixel half (tile rows 4*ph .. 4*ph+3)
    const int img = blockIdx.x / p.wgs_per_image, wgi = blockIdx.x % p.wgs_per_image;
    const int tiles_x = (p.W + C2_TW - 1) / C2_TW, tiles_y = (p.H + C2_TH - 1) / C2_TH, ntiles = tiles_x * tiles_y;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // ---- the filter bank of this wave's 48 output channels: 3 x 14 fragments, 16 B per lane each, straight from the blob
    gemm_x8 A[C2_NG][C2_KSTEPS];
    {
        const gemm_x8* src = (const gemm_x8*)p.blob + (int64_t)(3 * cs) * C2_KSTEPS * 64 + lane;
#pragma unroll
        for (int g = 0; g < C2_NG; ++g)
#pragma unroll
            for (int s = 0; s < C2_KSTEPS; ++s) A[g][s] = src[(g * C2_KSTEPS + s) * 64];
    }
    f32x4 pool[C2_NG];
#pragma unroll
    for (int g = 0; g < C2_NG; ++g) pool[g] = f32x4{0, 0, 0, 0};
    // bias: the accumulators' initial value, re-read per pixel row from LDS (behind the two halo buffers) to save 12 VGPRs
    float* bias_s = (float*)(smem + 2 * C2_HALO_B);
    if (tid < 192) bias_s[tid] = p.bias[tid];
    __syncthreads();
    const float* bias_l = bias_s + 48 * cs + 4 * g4;
    // ---- im2col: k-step s feeds lane (r16, g4) the 8 channels k = 32 s + 8 g4 .. + 7 of pixel r16, k = tap * 48 + c.  The
    // address is (pixel + 16 g4) + a compile-time offset, except where the 32 k of a step straddle two taps (s % 3 == 1,
    // lanes g4 >= 2 belong to the next tap): three more lane addresses cover those (next tap one pixel right; next tap at
    // the start of the next halo row; the zero-weight tail of K, which re-reads channels of the last tap).
    const bool hi2 = g4 >= 2;

    // ---- halo DMA: piece i = (halo pixel i / 7, segment i % 7); this lane's pieces are i = tid + 512 j
    int h0;    // piece tid: hy << 16 | hx << 4 | seg
    {
        const int px = tid / C2_SEG, seg = tid - px * C2_SEG, hy = px / C2_HW, hx = px - hy * C2_HW;
        h0 = hy << 16 | hx << 4 | seg;
    }
    c2_u32x4 xsrd;
    {
        const uint64_t a = (uint64_t)((const gemm_t*)p.x + (int64_t)img * p.H * p.W * p.ldx);
        xsrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        xsrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        xsrd[2] = (uint32_t)((int64_t)p.H * p.W * p.ldx * 2);   // bytes of this image: everything else reads 0
        xsrd[3] = 0x00020000u;
    }
    auto prefetch = [&](int tile, int buf) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * C2_TH - 1, x0 = tx * C2_TW - 1;
        int hy = h0 >> 16, hx = (h0 >> 4) & 0xfff, seg = h0 & 15;
#pragma unroll 1   // (unrolled, the five pieces' addresses were hoisted out of the tile loop and spilled)
        for (int j = 0; j < C2_ROUNDS; ++j) {
            if (j > 0) {   // 512 pieces on = 73 pixels + 1 segment = 2 halo rows + 5 pixels + 1 segment
                static_assert(C2_THREADS == 73 * C2_SEG + 1 && 73 == 2 * C2_HW + 5, "piece stepping");
                seg += 1; hx += 5; hy += 2;
                if (seg >= C2_SEG) { seg -= C2_SEG; hx += 1; }
                if (hx >= C2_HW) { hx -= C2_HW; hy += 1; }
            }
            const int gy = y0 + hy, gx = x0 + hx;
            const bool ok = hy < C2_HH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const uint32_t voff = ok ? (uint32_t)(((int64_t)gy * p.W + gx) * p.ldx * 2 + seg * 16) : 0xfffffff0u;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + buf * C2_HALO_B + (j * C2_THREADS + wave_u * 64) * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(xsrd) : "memory");
        }
    };

    int tile = wgi;
    if (tile < ntiles) prefetch(tile, 0);
    for (int it = 0; tile < ntiles; ++it, tile += p.wgs_per_image) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // the tile is complete; everybody is done reading the other buffer
        if (tile + p.wgs_per_image < ntiles) prefetch(tile + p.wgs_per_image, (it + 1) & 1);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const char* hb = smem + (it & 1) * C2_HALO_B;
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * ph + rr;
            const int y = ty * C2_TH + r;
            // two pixel groups (the row's halves) at a time: 6 independent accumulator chains
            const char* u0 = hb + (r * C2_HW + r16) * C2_PXB + 16 * g4;
            const char* uA = u0 + (hi2 ? C2_PXB - 96 : 0);
            const char* uB = u0 + (hi2 ? (C2_HW - 2) * C2_PXB - 96 : 0);
            const char* uC = u0 + (hi2 ? -32 : 0);
            f32x4 acc0[C2_NG], acc1[C2_NG];
#pragma unroll
            for (int g = 0; g < C2_NG; ++g) { acc0[g] = *(const f32x4*)(bias_l + 16 * g); acc1[g] = acc0[g]; }
#pragma unroll
            for (int s = 0; s < C2_KSTEPS; ++s) {
                constexpr int TAPOFF[9] = {0, C2_PXB, 2 * C2_PXB, C2_HW * C2_PXB, (C2_HW + 1) * C2_PXB, (C2_HW + 2) * C2_PXB,
                                           2 * C2_HW * C2_PXB, (2 * C2_HW + 1) * C2_PXB, (2 * C2_HW + 2) * C2_PXB};
                const int tap0 = 32 * s / 48, c00 = 32 * s - 48 * tap0;
                const char* ua = c00 != 32 ? u0 : (tap0 == 8 ? uC : (tap0 % 3 == 2 ? uB : uA));
                const gemm_x8 x0 = *(const gemm_x8*)(ua + TAPOFF[tap0] + 2 * c00);
                const gemm_x8 x1 = *(const gemm_x8*)(ua + TAPOFF[tap0] + 2 * c00 + 16 * C2_PXB);
#pragma unroll
                for (int g = 0; g < C2_NG; ++g) {
                    acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g][s], x0, acc0[g], 0, 0, 0);
                    acc1[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g][s], x1, acc1[g], 0, 0, 0);
                }
            }
            const int xa = tx * C2_TW + r16, xb = xa + 16;
            const bool va = y < p.H && xa < p.W, vb = y < p.H && xb < p.W;
            if (va) {
                gemm_t* orow = (gemm_t*)p.out + ((int64_t)(img * p.H + y) * p.W + xa) * p.ldo + 48 * cs + 4 * g4;
#pragma unroll
                for (int g = 0; g < C2_NG; ++g) {
                    pool[g] += acc0[g];
                    uint2 o;
                    o.x = pack_f16(acc0[g][0], acc0[g][1]);
                    o.y = pack_f16(acc0[g][2], acc0[g][3]);
                    *(uint2*)(orow + 16 * g) = o;
                }
            }
            if (vb) {
                gemm_t* orow = (gemm_t*)p.out + ((int64_t)(img * p.H + y) * p.W + xb) * p.ldo + 48 * cs + 4 * g4;
#pragma unroll
                for (int g = 0; g < C2_NG; ++g) {
                    pool[g] += acc1[g];
                    uint2 o;
                    o.x = pack_f16(acc1[g][0], acc1[g][1]);
                    o.y = pack_f16(acc1[g][2], acc1[g][3]);
                    *(uint2*)(orow + 16 * g) = o;
                }
            }
        }
    }

    // ---- channel sums of this workgroup: over the 16 pixel lanes, then the two pixel halves (fixed order) -> one row
    if (p.pool_partial != nullptr) {
#pragma unroll
        for (int g = 0; g < C2_NG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = pool[g][i];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                pool[g][i] = v;
            }
        __builtin_amdgcn_s_barrier();      // all waves are past their last halo reads
        float* red = (float*)smem;         // [2 halves][192]
        if (r16 == 0)
#pragma unroll
            for (int g = 0; g < C2_NG; ++g) *(f32x4*)(red + ph * 192 + 48 * cs + 16 * g + 4 * g4) = pool[g];
        __syncthreads();
        if (tid < 192) p.pool_partial[(int64_t)blockIdx.x * p.pool_stride + tid] = red[tid] + red[192 + tid];

        // ---- squeeze-excite gate by the LAST workgroup of the image (ChannelAttention, mixed_attn_block.py:956-963):
        //   gate[c] = sigmoid(W2 . relu(W1 . mean + b1) + b2).  As a launch of its own (grl_se_scale_fwd) the 10-us kernel sat in
        //   its stream for ~90 us behind the other tile group's CU-filling kernels, 640 times per step.
        if (p.gate != nullptr) {
            __shared__ int s_last;
            __threadfence();                      // this workgroup's row is visible device-wide before it is counted
            __syncthreads();
            if (tid == 0) s_last = atomicAdd(p.se_counter + img, 1) == p.wgs_per_image - 1;
            __syncthreads();
            if (s_last) {
                __threadfence();                  // acquire: the other workgroups' rows
                float* mean = red + 384;          // [192]
                float* hid = red + 576;           // [64]
                if (tid < 192) {
                    const float* pp = p.pool_partial + (int64_t)img * p.wgs_per_image * p.pool_stride + tid;
                    float sum = 0.f;
                    for (int i = 0; i < p.wgs_per_image; ++i) sum += __builtin_nontemporal_load(pp + (int64_t)i * p.pool_stride);   // fixed order
                    mean[tid] = sum * p.inv_hw;
                }
                __syncthreads();
                for (int jj = wave_u; jj < p.se_mid; jj += C2_WAVES) {      // one wave per hidden unit
                    float sum = 0.f;
                    for (int k = lane; k < p.se_c; k += 64) sum += p.se_w1[jj * p.se_c + k] * mean[k];
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (lane == 0) hid[jj] = fmaxf(sum + p.se_b1[jj], 0.f);
                }
                __syncthreads();
                if (tid < 192) {
                    float o = 0.f;
                    if (tid < p.se_c) {
                        float sum = p.se_b2[tid];
                        for (int jj = 0; jj < p.se_mid; ++jj) sum += p.se_w2[tid * p.se_mid + jj] * hid[jj];
                        o = 1.0f / (1.0f + __expf(-sum));
                    }
                    p.gate[(int64_t)img * 192 + tid] = o;
                }
                if (tid == 0) p.se_counter[img] = 0;   // ready for the next launch on this stream
            }
        }
    }
}

}  // namespace

extern "C" int64_t grl_cab_conv2_blob_bytes(void) { return (int64_t)12 * C2_KSTEPS * 64 * 16; }

extern "C" int grl_cab_conv2_fwd(void* stream, const GrlCabConv2Args* args) {
    const GrlCabConv2Args& p = *args;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.wgs_per_image <= 0) return GRL_ERR_BAD_ARG;
    if (p.ldx < 56 || (p.ldx % 8) || p.ldo < 192 || (p.ldo % 4)) return GRL_ERR_BAD_ARG;   // 7 segments of every pixel row are read
    if ((int64_t)p.H * p.W * p.ldx * 2 >= 0xfffffff0ll) return GRL_ERR_UNSUPPORTED;          // 32-bit buffer offsets per image
    if (p.pool_partial != nullptr && p.pool_stride < 192) return GRL_ERR_BAD_ARG;
    if (p.gate != nullptr && (p.pool_partial == nullptr || p.se_counter == nullptr || p.se_w1 == nullptr || p.se_b1 == nullptr ||
                              p.se_w2 == nullptr || p.se_b2 == nullptr || p.se_c <= 0 || p.se_c > 192 || p.se_mid <= 0 || p.se_mid > 64))
        return GRL_ERR_BAD_ARG;
    hipError_t e = hipFuncSetAttribute((const void*)cab_conv2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C2_HALO_B + 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cab_conv2_kernel, dim3(p.B * p.wgs_per_image), dim3(C2_THREADS), 2 * C2_HALO_B + 1024, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
