// Small memory-bound kernels of the GRL path: row LayerNorm (norm_start / norm_end,
// models/networks/grl.py:494,501) and library self-description.
#include "common.h"
#include "grl_hip_internal.h"

namespace {

// one wave per token row; n_pad <= 256 channels; pad channels written as 0
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                        int64_t ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, int n_real, int n_pad,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * ldx;
    const int c = lane * 4;
    float4 v = float4{0, 0, 0, 0};
    if (c < n_pad) v = *(const float4*)(xr + c);
    float e[4] = {v.x, v.y, v.z, v.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (c + i) < n_real ? e[i] : 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)n_real;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float d = e[i] - mean;
        q += (c + i) < n_real ? d * d : 0.f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)n_real + eps);
    if (c < n_pad) {
        float o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o4[i] = (c + i) < n_real ? (e[i] - mean) * rstd * gamma[c + i] + beta[c + i] : 0.f;
        *(float4*)(y + (int64_t)row * ldy + c) = float4{o4[0], o4[1], o4[2], o4[3]};
    }
}

// y = resid + res_scale * LayerNorm(x) (+ add2 * gate[image])  -- the un-fused form of the linear kernel's LN_RES epilogue,
// used by the split-precision path (its projections are slab-split and cannot carry the row norm in their epilogue)
__global__ __launch_bounds__(256) void layernorm_res_kernel(GrlLnResArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int c = lane * 4;
    const bool in = c < p.n_pad;
    float4 v = float4{0, 0, 0, 0};
    if (in) v = *(const float4*)(p.x + (int64_t)row * p.ldx + c);
    float e[4] = {v.x, v.y, v.z, v.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (c + i) < p.n_real ? e[i] : 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)p.n_real;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float d = e[i] - mean;
        q += (c + i) < p.n_real ? d * d : 0.f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)p.n_real + p.eps);
    if (!in) return;
    const float4 r4 = *(const float4*)(p.resid + (int64_t)row * p.ldr + c);
    const float r[4] = {r4.x, r4.y, r4.z, r4.w};
    float a[4] = {0, 0, 0, 0};
    if (p.add2 != nullptr) {
        const float4 g4 = *(const float4*)(p.add2_scale + (int64_t)(row / p.rows_per_image) * p.n_pad + c);
        if (p.add2_dtype == GRL_DT_F32) {
            const float4 a4 = *(const float4*)((const float*)p.add2 + (int64_t)row * p.ldadd2 + c);
            a[0] = a4.x * g4.x; a[1] = a4.y * g4.y; a[2] = a4.z * g4.z; a[3] = a4.w * g4.w;
        } else {
            const f16x4 a4 = *(const f16x4*)((const f16*)p.add2 + (int64_t)row * p.ldadd2 + c);
            a[0] = (float)a4[0] * g4.x; a[1] = (float)a4[1] * g4.y; a[2] = (float)a4[2] * g4.z; a[3] = (float)a4[3] * g4.w;
        }
    }
    float o4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o4[i] = (c + i) < p.n_real ? r[i] + p.res_scale * ((e[i] - mean) * rstd * p.gamma[c + i] + p.beta[c + i]) + a[i] : 0.f;
    *(float4*)(p.y + (int64_t)row * p.ldy + c) = float4{o4[0], o4[1], o4[2], o4[3]};
}

}  // namespace

extern "C" int grl_layernorm_res_fwd(void* stream, const GrlLnResArgs* args) {
    const GrlLnResArgs& p = *args;
    if (p.M <= 0) return 0;
    if (p.n_pad > 256 || (p.n_pad & 3) || p.n_real > p.n_pad || (p.ldx & 3) || (p.ldy & 3) || (p.ldr & 3) || p.resid == nullptr)
        return GRL_ERR_BAD_ARG;
    if (p.add2 != nullptr && (p.add2_scale == nullptr || p.rows_per_image <= 0 || (p.ldadd2 & 3) ||
                              (p.add2_dtype != GRL_DT_F32 && p.add2_dtype != GRL_DT_F16)))
        return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(layernorm_res_kernel, dim3((p.M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_layernorm_fwd(void* stream, const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma,
                                 const float* beta, int32_t M, int32_t n_real, int32_t n_pad, float eps) {
    if (M <= 0) return 0;
    if (n_pad > 256 || (n_pad & 3) || n_real > n_pad || (ldx & 3) || (ldy & 3)) return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, gamma,
                       beta, M, n_real, n_pad, eps);
    GRL_CHECK_LAUNCH();
    return 0;
}

// torch conv weight [Cout][Cin][3][3] (fp32) -> the layout grl_conv3x3_fwd reads, fp16 [9 taps][rows_pad][cols_pad] (tap = ky*3+kx),
// zero padded, and the padded fp32 bias, in ONE launch.  flip_t: the DATA-GRADIENT form of the same convolution -- taps flipped,
// channel roles swapped: out[tap][ci][co] = w[co][ci][2-ky][2-kx].  The training path packs every convolution's weights twice per
// step (they move every step); as torch code (permute + copy, zeros, slice assignment, fp16 copy; flip + transpose + copy for
// the gradient form) that was 4-7 launches each, ~700 of a captured step's ~10 k kernel nodes.
namespace {
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ b, f16* __restrict__ ow,
                                                        float* __restrict__ ob, int Cout, int Cin, int rows_pad, int cols_pad, int flip_t) {
    const int total = 9 * rows_pad * cols_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % cols_pad, row = (i / cols_pad) % rows_pad, tap = i / (cols_pad * rows_pad);
        const int co = flip_t ? col : row, ci = flip_t ? row : col;
        const int src_tap = flip_t ? 8 - tap : tap;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * 9 + src_tap];
        ow[i] = (f16)v;
    }
    if (ob != nullptr)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows_pad; i += gridDim.x * blockDim.x)
            ob[i] = (b != nullptr && i < Cout) ? b[i] : 0.f;
}
}  // namespace

extern "C" int grl_pack_conv3x3(void* stream, const float* w, const float* b, void* out_w, float* out_b, int32_t Cout, int32_t Cin,
                                int32_t rows_pad, int32_t cols_pad, int32_t flip_t) {
    if (!w || !out_w || Cout <= 0 || Cin <= 0 || rows_pad < (flip_t ? Cin : Cout) || cols_pad < (flip_t ? Cout : Cin)) return GRL_ERR_BAD_ARG;
    const int total = 9 * rows_pad * cols_pad;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), dim3(256), 0, (hipStream_t)stream, w, b,
                       (f16*)out_w, out_b, Cout, Cin, rows_pad, cols_pad, flip_t);
    GRL_CHECK_LAUNCH();
    return 0;
}

// nn.Linear weight [N][K] (fp32) -> the operand of grl_linear_fwd, fp16 [Np][Kp] zero padded, AND its transpose [Kp][Np] (the
// operand of the data-gradient launch dX = dY W) in one launch: the training path refreshes both once per step and parameter
// (torch: a converting slice copy + .t().contiguous(), two launches per linear layer, 400 per step).
namespace {
__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ w, f16* __restrict__ wp, f16* __restrict__ wt, int N, int K,
                                                          int Np, int Kp, const float* __restrict__ b, float* __restrict__ bp) {
    __shared__ f16 tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    if (bp != nullptr && blockIdx.x == 0 && threadIdx.x < 32 && n0 + threadIdx.x < Np)       // the zero-padded fp32 bias [Np] rides along
        bp[n0 + threadIdx.x] = (b != nullptr && n0 + threadIdx.x < N) ? b[n0 + threadIdx.x] : 0.f;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        const f16 v = (n < N && k < K) ? (f16)w[(int64_t)n * K + k] : (f16)0.f;
        tile[r][tx] = v;
        if (n < Np && k < Kp) wp[(int64_t)n * Kp + k] = v;
    }
    __syncthreads();
    if (wt != nullptr)
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, n = n0 + tx;
            if (k < Kp && n < Np) wt[(int64_t)k * Np + n] = tile[tx][r];
        }
}
}  // namespace

extern "C" int grl_pack_linear(void* stream, const float* w, const float* b, void* out_w, void* out_wt, float* out_b, int32_t N, int32_t K,
                               int32_t Np, int32_t Kp) {
    if (!w || !out_w || N <= 0 || K <= 0 || Np < N || Kp < K) return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((Kp + 31) / 32, (Np + 31) / 32), dim3(256), 0, (hipStream_t)stream, w, (f16*)out_w, (f16*)out_wt,
                       N, K, Np, Kp, b, out_b);
    GRL_CHECK_LAUNCH();
    return 0;
}

// out = a + b (+ c) (+ d) on flat fp32 arrays (16-byte aligned, n a multiple of 4): the gradient of a tensor with several consumers in ONE
// pass -- autograd adds the contributions pairwise as they arrive (a block's input feeds the QKV projection, the anchor pooling, the CAB
// and the residual: three launches of 71 MB each for what one launch of 118 MB does).
namespace {
__global__ __launch_bounds__(256) void sum4_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                                   const float4* __restrict__ d, float4* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = a[i];
        const float4 t = b[i];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        if (c != nullptr) { const float4 u = c[i]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
        if (d != nullptr) { const float4 u = d[i]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
        out[i] = s;
    }
}
}  // namespace

extern "C" int grl_sum4(void* stream, const float* a, const float* b, const float* c, const float* d, float* out, int64_t n) {
    if (!a || !b || !out || n <= 0 || (n & 3) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)out) & 15)) return GRL_ERR_BAD_ARG;
    const int64_t n4 = n >> 2, wgs = (n4 + 255) / 256;
    hipLaunchKernelGGL(sum4_kernel, dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b,
                       (const float4*)c, (const float4*)d, (float4*)out, n4);
    GRL_CHECK_LAUNCH();
    return 0;
}

// Debug aid (GRL_DIRTY_LDS=1 in the Python wrappers calls it before every C-ABI launch): overwrites the LDS of every CU with
// 0xFF bytes (fp32 / fp16 NaN).  LDS is not cleared between workgroups, so a kernel that reads a location it has not written sees
// whatever the previous workgroup on that CU left there -- zeros or finite numbers most of the time, which hides the bug and makes
// it depend on what ran before.  After this launch such a read yields NaN in the parity tests.
namespace {
__global__ __launch_bounds__(256) void dirty_lds_kernel(int words) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile uint32_t* s = (volatile uint32_t*)smem;
    for (int i = threadIdx.x; i < words; i += blockDim.x) s[i] = 0xFFFFFFFFu;
    __syncthreads();
}
}  // namespace

extern "C" int grl_debug_dirty_lds(void* stream) {
    const int bytes = 160 * 1024;          // the whole LDS of a CU: one workgroup per CU at a time
    hipError_t e = hipFuncSetAttribute((const void*)dirty_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dirty_lds_kernel, dim3(2048), dim3(256), bytes, (hipStream_t)stream, bytes / 4);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_abi_version(void) { return GRL_ABI_VERSION; }
extern "C" const char* grl_build_info(void) { return "grl_hip gfx950 (MI355X) " __DATE__ " " __TIME__; }
