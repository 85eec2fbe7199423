// Row LayerNorm of the TRAINING path (gfx950): forward with saved statistics, and backward.
//
// Replaces nn.LayerNorm on the token matrices of a training step -- norm1 / norm2 of every block
// (models/common/mixed_attn_block_efficient.py:543-556), norm_start / norm_end (models/networks/grl.py:494,501) -- and autograd
// through them.  As torch kernels they were 82 launches of 37 us forward and 82 x (33 + 32 + 5) us backward per step of BASELINE
// config 5 (profiles/r06_train_kernel_stats.txt: 1.3 TB/s on a [32 768, 180] fp32 matrix); the rows are 720 bytes, one wave per row
// with 16-byte accesses streams them at the HBM rate.
//   forward : y = (x - mean) rstd gamma + beta;  mean / rstd per row kept for the backward pass
//             with `resid` (round 6): y = resid + c_row ((x - mean) rstd gamma + beta),  c_row = alpha * row_scale[row / rows_per_image]
//             -- the post-norm residual of a block with its residual scale and DropPath keep mask (one Bernoulli draw per image,
//             scale_by_keep; mixed_attn_block_efficient.py:543-556) in the same pass: as torch code the addcmul after every norm was one
//             more 47 MB pass forward and two multiplies backward; the backward kernel takes dL/dy and applies c_row itself
//             (dL/dresid = dL/dy: no launch at all)
//   backward: g = dy gamma;  dx = rstd (g - mean_c(g) - xhat mean_c(g xhat));  dgamma = sum_rows dy xhat;  dbeta = sum_rows dy
//             (column sums: per-lane partials over a wave's rows, the four waves of a workgroup through LDS, one atomic per column
//             and workgroup into the zeroed outputs)
#include "common.h"
#include "grl_hip_internal.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) { return sum_halves(sum_rows16(row16_sum(v))); }

// Both kernels keep TWO rows per wave in flight (their loads are issued before the first reduction starts) and derive the row index
// from the wave number in an SGPR: one row at a time they were latency chains of load -> reduce -> reduce -> store, 47 us for the
// backward pass of a [32 768, 180] matrix that the HBM streams in under 20.
__device__ __forceinline__ float row_factor(const GrlLnTrainArgs& p, int row) {
    return p.row_scale != nullptr ? p.alpha * p.row_scale[row / p.rows_per_image] : p.alpha;
}

__global__ __launch_bounds__(256) void ln_train_fwd_kernel(GrlLnTrainArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane * 4;
    const bool in = c < p.n;                       // (n is a multiple of 4)
    float4 gm = float4{0, 0, 0, 0}, bt = gm;
    if (in) { gm = *(const float4*)(p.gamma + c); bt = *(const float4*)(p.beta + c); }
    const float inv_n = 1.0f / (float)p.n;
    const int nw = gridDim.x * 4;                  // waves of the launch; wave w takes rows w, w + nw, ... two at a time
    for (int row0 = blockIdx.x * 4 + wave; row0 < p.M; row0 += 2 * nw) {
        const int row1 = row0 + nw;
        const bool two = row1 < p.M;               // (wave-uniform)
        const int r1 = two ? row1 : row0;
        float4 v0 = float4{0, 0, 0, 0}, v1 = v0, q0 = v0, q1 = v0;
        float c0 = 0.f, c1 = 0.f;
        if (in) {
            v0 = *(const float4*)(p.x + (int64_t)row0 * p.ldx + c);
            v1 = *(const float4*)(p.x + (int64_t)r1 * p.ldx + c);
            if (p.resid != nullptr) {
                q0 = *(const float4*)(p.resid + (int64_t)row0 * p.ldr + c);
                q1 = *(const float4*)(p.resid + (int64_t)r1 * p.ldr + c);
            }
        }
        if (p.resid != nullptr) { c0 = row_factor(p, row0); c1 = row_factor(p, r1); }
        auto one = [&](const float4& v, const float4& q, float cr, int row) {
            const float mean = wave_sum(v.x + v.y + v.z + v.w) * inv_n;
            const float d0 = in ? v.x - mean : 0.f, d1 = in ? v.y - mean : 0.f, d2 = in ? v.z - mean : 0.f, d3 = in ? v.w - mean : 0.f;
            const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * inv_n + p.eps);
            if (in) {
                float4 o = float4{d0 * rstd * gm.x + bt.x, d1 * rstd * gm.y + bt.y, d2 * rstd * gm.z + bt.z, d3 * rstd * gm.w + bt.w};
                if (p.resid != nullptr) o = float4{fmaf(cr, o.x, q.x), fmaf(cr, o.y, q.y), fmaf(cr, o.z, q.z), fmaf(cr, o.w, q.w)};
                *(float4*)(p.y + (int64_t)row * p.ldy + c) = o;
            }
            if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
        };
        one(v0, q0, c0, row0);
        if (two) one(v1, q1, c1, row1);
    }
}

__global__ __launch_bounds__(256) void ln_train_bwd_kernel(GrlLnTrainArgs p) {
    __shared__ float red[2][4][256];               // [dgamma | dbeta][wave][column]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane * 4;
    const bool in = c < p.n;
    float4 gm = float4{0, 0, 0, 0};
    if (in) gm = *(const float4*)(p.gamma + c);
    const float inv_n = 1.0f / (float)p.n;
    float4 sg = float4{0, 0, 0, 0}, sb = sg;
    const int nw = gridDim.x * 4;
    for (int row0 = blockIdx.x * 4 + wave; row0 < p.M; row0 += 2 * nw) {
        const int row1 = row0 + nw;
        const bool two = row1 < p.M;
        const int r1 = two ? row1 : row0;
        float4 v0 = float4{0, 0, 0, 0}, v1 = v0, e0 = v0, e1 = v0;
        if (in) {
            v0 = *(const float4*)(p.x + (int64_t)row0 * p.ldx + c); e0 = *(const float4*)(p.dy + (int64_t)row0 * p.lddy + c);
            v1 = *(const float4*)(p.x + (int64_t)r1 * p.ldx + c);   e1 = *(const float4*)(p.dy + (int64_t)r1 * p.lddy + c);
        }
        const float m0 = p.mean[row0], s0 = p.rstd[row0], m1 = p.mean[r1], s1r = p.rstd[r1];
        float c0 = 1.0f, c1 = 1.0f;
        if (p.alpha != 0.0f) { c0 = row_factor(p, row0); c1 = row_factor(p, r1); }   // fused residual: the norm's output entered y times c_row
        auto one = [&](const float4& v, float4 d, float cr, float mean, float rstd, int row) {
            d.x *= cr; d.y *= cr; d.z *= cr; d.w *= cr;
            const float h0 = in ? (v.x - mean) * rstd : 0.f, h1 = in ? (v.y - mean) * rstd : 0.f, h2 = in ? (v.z - mean) * rstd : 0.f,
                        h3 = in ? (v.w - mean) * rstd : 0.f;
            const float g0 = d.x * gm.x, g1 = d.y * gm.y, g2 = d.z * gm.z, g3 = d.w * gm.w;
            const float s1 = wave_sum(g0 + g1 + g2 + g3) * inv_n;
            const float s2 = wave_sum(g0 * h0 + g1 * h1 + g2 * h2 + g3 * h3) * inv_n;
            if (in)
                *(float4*)(p.dx + (int64_t)row * p.lddx + c) =
                    float4{rstd * (g0 - s1 - h0 * s2), rstd * (g1 - s1 - h1 * s2), rstd * (g2 - s1 - h2 * s2), rstd * (g3 - s1 - h3 * s2)};
            sg.x = fmaf(d.x, h0, sg.x); sg.y = fmaf(d.y, h1, sg.y); sg.z = fmaf(d.z, h2, sg.z); sg.w = fmaf(d.w, h3, sg.w);
            sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
        };
        one(v0, e0, c0, m0, s0, row0);
        if (two) one(v1, e1, c1, m1, s1r, row1);
    }
    *(float4*)&red[0][wave][c] = sg;
    *(float4*)&red[1][wave][c] = sb;
    __syncthreads();
    const int col = threadIdx.x;                   // 256 threads: one column each
    if (col < p.n) {
        const int rep = p.stat_replicas > 1 ? (int)(blockIdx.x % (unsigned)p.stat_replicas) * p.n : 0;    // (atomics on one address serialise)
        unsafeAtomicAdd(p.dgamma + rep + col, red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col]);
        unsafeAtomicAdd(p.dbeta + rep + col, red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col]);
    }
}

bool ln_args_ok(const GrlLnTrainArgs& p) {
    return p.M > 0 && p.n > 0 && p.n <= 256 && (p.n & 3) == 0 && p.x && p.gamma && p.mean && p.rstd && (p.ldx & 3) == 0 && p.ldx >= p.n;
}

}  // namespace

extern "C" int grl_layernorm_train_fwd(void* stream, const GrlLnTrainArgs* args) {
    const GrlLnTrainArgs& p = *args;
    if (!ln_args_ok(p) || !p.y || !p.beta || (p.ldy & 3) || p.ldy < p.n) return GRL_ERR_BAD_ARG;
    if (p.resid != nullptr && ((p.ldr & 3) || p.ldr < p.n)) return GRL_ERR_BAD_ARG;
    if (p.row_scale != nullptr && p.rows_per_image <= 0) return GRL_ERR_BAD_ARG;
    const int grid = (p.M + 7) / 8 < 2048 ? (p.M + 7) / 8 : 2048;
    hipLaunchKernelGGL(ln_train_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_layernorm_bwd(void* stream, const GrlLnTrainArgs* args) {
    const GrlLnTrainArgs& p = *args;
    if (!ln_args_ok(p) || !p.dy || !p.dx || !p.dgamma || !p.dbeta || (p.lddy & 3) || (p.lddx & 3) || p.lddy < p.n || p.lddx < p.n) return GRL_ERR_BAD_ARG;
    if (p.row_scale != nullptr && (p.rows_per_image <= 0 || p.alpha == 0.0f)) return GRL_ERR_BAD_ARG;
    const int grid = (p.M + 7) / 8 < 2048 ? (p.M + 7) / 8 : 2048;
    hipLaunchKernelGGL(ln_train_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
