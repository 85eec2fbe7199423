// Squeeze-excite gate of the CAB in a TRAINING step (gfx950): the two-layer MLP on the pooled channel means, forward and backward.
//
// Replaces ChannelAttention.attention[1..4] (models/common/mixed_attn_block.py:956-963: Conv2d(C, C/r, 1) -> ReLU -> Conv2d(C/r, C, 1) ->
// Sigmoid on the [B, C] output of AdaptiveAvgPool2d) and autograd through it.  The tensors are tiny ([8, 180] and [8, 10] for GRL-Base at
// batch 8) and as torch code the chain was 4 launches forward (two BLAS GEMMs of ~8 us each, ReLU, sigmoid) and 8 backward (four BLAS
// GEMMs, two column sums, the two activation adjoints): ~3 ms of a 100-ms captured step for a few kFLOP -- a captured step costs the SUM of
// its kernels' durations, however small the work.
//   forward : hid = relu(W1 pool + b1);  gate = sigmoid(W2 hid + b2)                       one workgroup per image
//   backward: dz2 = d_gate gate (1 - gate);  dW2 += dz2 hid^T;  db2 += dz2;  dh = W2^T dz2;  dz1 = dh [hid > 0];
//             dW1 += dz1 pool^T;  db1 += dz1;  d_pool = W1^T dz1                            one workgroup per image + atomics into zeroed
//             outputs (GrlSeMlpArgs.parallel), or ONE workgroup that walks the images (no atomics, bit-reproducible)
#include "common.h"
#include "grl_hip_internal.h"

namespace {

constexpr int SE_T = 256;          // threads: one per channel (C <= 256)
constexpr int SE_MID = 64;         // hidden units (C / reduction) <= 64

__device__ __forceinline__ float se_wave_sum(float v) { return sum_halves(sum_rows16(row16_sum(v))); }

__global__ __launch_bounds__(SE_T) void se_mlp_fwd_kernel(GrlSeMlpArgs p) {
    __shared__ float pool[SE_T], hid[SE_MID];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < p.C) pool[tid] = p.pool[(int64_t)b * p.C + tid];
    __syncthreads();
    for (int j = wave; j < p.Cmid; j += SE_T / 64) {
        float s = 0.f;
        for (int c = lane; c < p.C; c += 64) s = fmaf(p.w1[j * p.C + c], pool[c], s);
        s = se_wave_sum(s);
        if (lane == 0) {
            const float h = fmaxf(s + p.b1[j], 0.f);
            hid[j] = h;
            p.hidden[(int64_t)b * p.Cmid + j] = h;
        }
    }
    __syncthreads();
    if (tid < p.C) {
        float s = p.b2[tid];
        for (int j = 0; j < p.Cmid; ++j) s = fmaf(p.w2[tid * p.Cmid + j], hid[j], s);
        p.gate[(int64_t)b * p.C + tid] = 1.0f / (1.0f + __expf(-s));
    }
}

// PAR: one workgroup per image, the parameter gradients (sums over the batch) accumulated with atomics into arrays the CALLER zeroed
// (B adds per address: no contention to speak of); !PAR: one workgroup walks the images, every element owned by one thread -- no atomics,
// bit-reproducible, but a serial chain of B x (three barriers + global read-modify-writes): 42 us for 8 images against 7.
template <bool PAR>
__global__ __launch_bounds__(SE_T) void se_mlp_bwd_kernel(GrlSeMlpArgs p) {
    __shared__ float dz2[SE_T], dz1[SE_MID], hid[SE_MID];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool own = tid < p.C;
    float db2 = 0.f;
    if constexpr (!PAR) {
        for (int i = tid; i < p.Cmid * p.C; i += SE_T) { p.d_w1[i] = 0.f; p.d_w2[i] = 0.f; }
        if (tid < p.Cmid) p.d_b1[tid] = 0.f;
        __syncthreads();
    }
    auto add = [](float* dst, float v) {
        if constexpr (PAR) unsafeAtomicAdd(dst, v);
        else *dst += v;
    };
    const int b0 = PAR ? blockIdx.x : 0, b1 = PAR ? blockIdx.x + 1 : p.B;
    for (int b = b0; b < b1; ++b) {
        float pl = 0.f;
        if (own) {
            const float g = p.gate[(int64_t)b * p.C + tid];
            const float z = p.d_gate[(int64_t)b * p.C + tid] * g * (1.0f - g);
            dz2[tid] = z;
            db2 += z;
            pl = p.pool[(int64_t)b * p.C + tid];
        }
        if (tid < p.Cmid) hid[tid] = p.hidden[(int64_t)b * p.Cmid + tid];
        __syncthreads();
        if (own)                                                                   // dW2 [C][Cmid]: row tid is this thread's
            for (int j = 0; j < p.Cmid; ++j) add(p.d_w2 + tid * p.Cmid + j, dz2[tid] * hid[j]);
        for (int j = wave; j < p.Cmid; j += SE_T / 64) {                           // dh = W2^T dz2, through the ReLU
            float s = 0.f;
            for (int c = lane; c < p.C; c += 64) s = fmaf(p.w2[c * p.Cmid + j], dz2[c], s);
            s = se_wave_sum(s);
            if (lane == 0) {
                const float z = hid[j] > 0.f ? s : 0.f;
                dz1[j] = z;
                add(p.d_b1 + j, z);
            }
        }
        __syncthreads();
        if (own) {
            float dp = 0.f;
            for (int j = 0; j < p.Cmid; ++j) {
                add(p.d_w1 + j * p.C + tid, dz1[j] * pl);                          // dW1 [Cmid][C]: column tid is this thread's
                dp = fmaf(p.w1[j * p.C + tid], dz1[j], dp);
            }
            p.d_pool[(int64_t)b * p.C + tid] = dp;
        }
        __syncthreads();
    }
    if (own) {
        if constexpr (PAR) unsafeAtomicAdd(p.d_b2 + tid, db2);
        else p.d_b2[tid] = db2;
    }
}

// ---- the two passes over the token matrix around the MLP ---------------------------------------------------------------------------
// colsum: out[b][c] = scale * sum over the rows of image b of a[row][c] (* b2[row][c]): the average pool of ChannelAttention
//         (AdaptiveAvgPool2d(1): a = the CAB's conv output, scale = 1 / HW) and, in the backward pass, the gate's gradient sum_rows dy * u.
// apply : out[row][c] = a[row][c] * g[b][c] (+ f[row][c]) (+ k * h[b][c]): forward x1 + u * gate; backward d_u = dy * gate + d_pool / HW.
// C <= 256, a multiple of 4; rows_per_image rows per image; all matrices [M, ld] fp32 with 16-byte aligned rows.
constexpr int SE_ROWS = 64;        // rows per workgroup of the colsum kernel (4 waves x 16 rows)

__global__ __launch_bounds__(256) void se_colsum_kernel(GrlSeRowsArgs p) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane * 4;
    const bool in = c < p.C;
    const int wgs_per_image = (p.rows_per_image + SE_ROWS - 1) / SE_ROWS;
    const int img = blockIdx.x / wgs_per_image, part = blockIdx.x - img * wgs_per_image;
    const int r0 = part * SE_ROWS, r1 = min(p.rows_per_image, r0 + SE_ROWS);
    float4 s = float4{0, 0, 0, 0};
    if (in)
        for (int r = r0 + wave; r < r1; r += 4) {
            const int64_t row = (int64_t)img * p.rows_per_image + r;
            float4 v = *(const float4*)(p.a + row * p.lda + c);
            if (p.f != nullptr) {
                const float4 w = *(const float4*)(p.f + row * p.ldf + c);
                v = float4{v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w};
            }
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    *(float4*)&red[wave][c] = s;
    __syncthreads();
    const int col = threadIdx.x;
    if (col < p.C)      // (wgs_per_image adds per address: 64 for a 64 x 64 image -- no replicas needed, see GrlLnTrainArgs.stat_replicas)
        unsafeAtomicAdd(p.out + (int64_t)img * p.C + col, p.k * (red[0][col] + red[1][col] + red[2][col] + red[3][col]));
}

__global__ __launch_bounds__(256) void se_apply_kernel(GrlSeRowsArgs p) {
    const int c4n = p.C >> 2;
    const int64_t total = (int64_t)p.M * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / c4n;
        const int c = (int)(i - row * c4n) * 4;
        const int img = (int)(row / p.rows_per_image);
        const float4 a = *(const float4*)(p.a + row * p.lda + c);
        const float4 g = *(const float4*)(p.g + (int64_t)img * p.C + c);
        float4 o = float4{a.x * g.x, a.y * g.y, a.z * g.z, a.w * g.w};
        if (p.f != nullptr) {
            const float4 f = *(const float4*)(p.f + row * p.ldf + c);
            o.x += f.x; o.y += f.y; o.z += f.z; o.w += f.w;
        }
        if (p.h != nullptr) {
            const float4 h = *(const float4*)(p.h + (int64_t)img * p.C + c);
            o.x = fmaf(p.k, h.x, o.x); o.y = fmaf(p.k, h.y, o.y); o.z = fmaf(p.k, h.z, o.z); o.w = fmaf(p.k, h.w, o.w);
        }
        *(float4*)(p.out + row * p.ldo + c) = o;
    }
}

bool se_rows_ok(const GrlSeRowsArgs& p) {
    return p.M > 0 && p.C > 0 && p.C <= 256 && (p.C & 3) == 0 && p.rows_per_image > 0 && p.M % p.rows_per_image == 0 && p.a && p.out &&
           (p.lda & 3) == 0 && p.lda >= p.C && (p.f == nullptr || ((p.ldf & 3) == 0 && p.ldf >= p.C));
}

bool se_args_ok(const GrlSeMlpArgs& p) {
    return p.B > 0 && p.C > 0 && p.C <= SE_T && p.Cmid > 0 && p.Cmid <= SE_MID && p.pool && p.w1 && p.b1 && p.w2 && p.b2 && p.gate && p.hidden;
}

}  // namespace

extern "C" int grl_se_mlp_fwd(void* stream, const GrlSeMlpArgs* args) {
    const GrlSeMlpArgs& p = *args;
    if (!se_args_ok(p)) return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(p.B), dim3(SE_T), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_se_mlp_bwd(void* stream, const GrlSeMlpArgs* args) {
    const GrlSeMlpArgs& p = *args;
    if (!se_args_ok(p) || !p.d_gate || !p.d_pool || !p.d_w1 || !p.d_b1 || !p.d_w2 || !p.d_b2) return GRL_ERR_BAD_ARG;
    if (p.parallel) hipLaunchKernelGGL(se_mlp_bwd_kernel<true>, dim3(p.B), dim3(SE_T), 0, (hipStream_t)stream, p);   // parallel: zeroed outputs
    else hipLaunchKernelGGL(se_mlp_bwd_kernel<false>, dim3(1), dim3(SE_T), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_se_colsum(void* stream, const GrlSeRowsArgs* args) {
    const GrlSeRowsArgs& p = *args;
    if (!se_rows_ok(p)) return GRL_ERR_BAD_ARG;
    const int wgs = (p.M / p.rows_per_image) * ((p.rows_per_image + SE_ROWS - 1) / SE_ROWS);
    hipLaunchKernelGGL(se_colsum_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_se_apply(void* stream, const GrlSeRowsArgs* args) {
    const GrlSeRowsArgs& p = *args;
    if (!se_rows_ok(p) || !p.g || (p.ldo & 3) || p.ldo < p.C) return GRL_ERR_BAD_ARG;
    const int64_t total = (int64_t)p.M * (p.C >> 2);
    const int64_t wgs = (total + 255) / 256;
    hipLaunchKernelGGL(se_apply_kernel, dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
