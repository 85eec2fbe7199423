// Relative-position bias tables of the training path (gfx950): the continuous-position-bias MLP of AffineTransform
// (models/common/mixed_attn_block_efficient.py:23-58:  table = 16 * sigmoid(cpb_mlp(coords)),  cpb_mlp = Linear(2, 512) -> ReLU ->
// Linear(512, nh, bias=False)) for MANY transforms at once, forward and backward, WITHOUT the hidden layer in memory.
//
// Why a kernel: the tables depend on weights only, so the training path evaluates all blocks' MLPs in one batched chain
// (GRL._train_tables).  As torch code that chain materialises h = relu(coords W1^T + b1) of shape [transforms, rows, 512]: for GRL-Base
// at the checkpoint geometry 80 stripe transforms x 9 025 rows x 512 = 1.5 GB of fp32 (plus 0.33 GB for the 40 window transforms),
// written by two addcmul, read by the relu, the bmm and again by all their backward nodes -- ~9 ms of a 145 ms training step in
// kernels that do nothing but move that tensor (profiles/r06_train_kernel_stats.txt: threshold_backward 1.6 ms per launch).  Here a
// thread owns a table row, walks the 512 hidden units with the weights of its transform in LDS (broadcast reads) and keeps h in a
// register.  The backward pass recomputes h the same way and reduces dW1 / db1 / dW2 over the rows: per hidden unit a DPP wave
// reduction of (3 + nh) values, one ds_add per value and wave into LDS accumulators, one atomic per value and workgroup to memory.
//
// Output layout = what the attention kernels read (tables.kernel_table): out[g][n][i] = 16 log2(e) * sigmoid(pre[g][n][rows-1-i])
// for i < rows (REVERSED rows, exp2 domain), and for the pad entries i = rows .. rows4-1 the value of source row 0 (no (query, key)
// pair addresses them).
#include "common.h"
#include "grl_hip_internal.h"

namespace {

constexpr int CPB_H = 512;          // hidden width of cpb_mlp (mixed_attn_block_efficient.py:29)
constexpr int CPB_T = 256;          // threads per workgroup
constexpr int CPB_RPT = 4;          // table rows per thread in the backward kernel
constexpr float CPB_A = 16.0f * LOG2E_F;

template <int NH>
__device__ __forceinline__ void cpb_load_weights(float* w1x, float* w1y, float* bb, float* w2, const float* W1, const float* b1,
                                                 const float* W2, int g, int tid) {
    for (int j = tid; j < CPB_H; j += CPB_T) {
        const float2 w = *(const float2*)(W1 + ((int64_t)g * CPB_H + j) * 2);
        w1x[j] = w.x; w1y[j] = w.y;
        bb[j] = b1[(int64_t)g * CPB_H + j];
#pragma unroll
        for (int n = 0; n < NH; ++n) w2[n * CPB_H + j] = W2[((int64_t)g * NH + n) * CPB_H + j];
    }
}

// pre-activation of the output layer for one table row (c0, c1): acc[n] = sum_j W2[n][j] relu(W1[j] . c + b1[j])
template <int NH>
__device__ __forceinline__ void cpb_row(const float* w1x, const float* w1y, const float* bb, const float* w2, float c0, float c1, float (&acc)[NH]) {
#pragma unroll
    for (int n = 0; n < NH; ++n) acc[n] = 0.f;
    for (int j = 0; j < CPB_H; j += 4) {
        const float4 ax = *(const float4*)(w1x + j), ay = *(const float4*)(w1y + j), ab = *(const float4*)(bb + j);
        // (b1 + c0 w1x) + c1 w1y: the order of the torch expression this replaces
        const float h0 = fmaxf(0.f, fmaf(c1, ay.x, fmaf(c0, ax.x, ab.x))), h1 = fmaxf(0.f, fmaf(c1, ay.y, fmaf(c0, ax.y, ab.y)));
        const float h2 = fmaxf(0.f, fmaf(c1, ay.z, fmaf(c0, ax.z, ab.z))), h3 = fmaxf(0.f, fmaf(c1, ay.w, fmaf(c0, ax.w, ab.w)));
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const float4 w = *(const float4*)(w2 + n * CPB_H + j);
            acc[n] = fmaf(w.w, h3, fmaf(w.z, h2, fmaf(w.y, h1, fmaf(w.x, h0, acc[n]))));
        }
    }
}

template <int NH>
__global__ __launch_bounds__(CPB_T) void cpb_fwd_kernel(GrlCpbArgs p) {
    __shared__ __attribute__((aligned(16))) float w1x[CPB_H], w1y[CPB_H], bb[CPB_H], w2[NH * CPB_H];
    const int tid = threadIdx.x, g = blockIdx.y;
    cpb_load_weights<NH>(w1x, w1y, bb, w2, p.w1, p.b1, p.w2, g, tid);
    __syncthreads();
    const int r = blockIdx.x * CPB_T + tid;
    if (r >= p.rows) return;
    const float2 c = *(const float2*)(p.coords + (int64_t)r * 2);
    float acc[NH];
    cpb_row<NH>(w1x, w1y, bb, w2, c.x, c.y, acc);
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        const float v = CPB_A / (1.0f + __expf(-acc[n]));
        float* o = p.out + ((int64_t)g * NH + n) * p.rows4;
        o[p.rows - 1 - r] = v;
        if (r == 0)
            for (int i = p.rows; i < p.rows4; ++i) o[i] = v;
    }
}

template <int NH>
__global__ __launch_bounds__(CPB_T) void cpb_bwd_kernel(GrlCpbArgs p) {
    __shared__ __attribute__((aligned(16))) float w1x[CPB_H], w1y[CPB_H], bb[CPB_H], w2[NH * CPB_H];
    __shared__ float a1x[CPB_H], a1y[CPB_H], ab[CPB_H], a2[NH * CPB_H];       // gradient accumulators of this workgroup
    const int tid = threadIdx.x, g = blockIdx.y, lane = tid & 63;
    cpb_load_weights<NH>(w1x, w1y, bb, w2, p.w1, p.b1, p.w2, g, tid);
    for (int j = tid; j < CPB_H; j += CPB_T) {
        a1x[j] = 0.f; a1y[j] = 0.f; ab[j] = 0.f;
#pragma unroll
        for (int n = 0; n < NH; ++n) a2[n * CPB_H + j] = 0.f;
    }
    __syncthreads();
    // this thread's rows, and for each the gradient of the output layer's pre-activations
    float c0[CPB_RPT], c1[CPB_RPT], dpre[CPB_RPT][NH];
#pragma unroll
    for (int q = 0; q < CPB_RPT; ++q) {
        const int r = (blockIdx.x * CPB_RPT + q) * CPB_T + tid;
        const bool ok = r < p.rows;
        const float2 c = ok ? *(const float2*)(p.coords + (int64_t)r * 2) : float2{0.f, 0.f};
        c0[q] = c.x; c1[q] = c.y;
        float acc[NH];
        cpb_row<NH>(w1x, w1y, bb, w2, c.x, c.y, acc);
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const float* go = p.d_out + ((int64_t)g * NH + n) * p.rows4;
            float d = ok ? go[p.rows - 1 - r] : 0.f;
            if (r == 0)
                for (int i = p.rows; i < p.rows4; ++i) d += go[i];     // the pad entries repeat source row 0
            const float s = 1.0f / (1.0f + __expf(-acc[n]));
            dpre[q][n] = d * CPB_A * s * (1.0f - s);
        }
    }
    // hidden units: recompute h, reduce (d b1, d W1, d W2) over the workgroup's rows
    for (int j = 0; j < CPB_H; ++j) {
        const float wx = w1x[j], wy = w1y[j], b = bb[j];
        float wn[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) wn[n] = w2[n * CPB_H + j];
        float gb = 0.f, gx = 0.f, gy = 0.f, g2[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) g2[n] = 0.f;
#pragma unroll
        for (int q = 0; q < CPB_RPT; ++q) {
            const float pre = fmaf(c1[q], wy, fmaf(c0[q], wx, b));
            const float h = fmaxf(0.f, pre);
            float dh = 0.f;
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                dh = fmaf(wn[n], dpre[q][n], dh);
                g2[n] = fmaf(dpre[q][n], h, g2[n]);
            }
            dh = pre > 0.f ? dh : 0.f;
            gb += dh; gx = fmaf(dh, c0[q], gx); gy = fmaf(dh, c1[q], gy);
        }
        gb = sum_halves(sum_rows16(row16_sum(gb)));
        gx = sum_halves(sum_rows16(row16_sum(gx)));
        gy = sum_halves(sum_rows16(row16_sum(gy)));
#pragma unroll
        for (int n = 0; n < NH; ++n) g2[n] = sum_halves(sum_rows16(row16_sum(g2[n])));
        if (lane == 0) {
            atomicAdd(&ab[j], gb); atomicAdd(&a1x[j], gx); atomicAdd(&a1y[j], gy);
#pragma unroll
            for (int n = 0; n < NH; ++n) atomicAdd(&a2[n * CPB_H + j], g2[n]);
        }
    }
    __syncthreads();
    for (int j = tid; j < CPB_H; j += CPB_T) {
        unsafeAtomicAdd(p.d_w1 + ((int64_t)g * CPB_H + j) * 2, a1x[j]);
        unsafeAtomicAdd(p.d_w1 + ((int64_t)g * CPB_H + j) * 2 + 1, a1y[j]);
        unsafeAtomicAdd(p.d_b1 + (int64_t)g * CPB_H + j, ab[j]);
#pragma unroll
        for (int n = 0; n < NH; ++n) unsafeAtomicAdd(p.d_w2 + ((int64_t)g * NH + n) * CPB_H + j, a2[n * CPB_H + j]);
    }
}

template <int NH>
int cpb_launch(const GrlCpbArgs& p, hipStream_t st, bool bwd) {
    if (bwd) {
        const dim3 grid((p.rows + CPB_T * CPB_RPT - 1) / (CPB_T * CPB_RPT), p.G);
        hipLaunchKernelGGL(cpb_bwd_kernel<NH>, grid, dim3(CPB_T), 0, st, p);
    } else {
        const dim3 grid((p.rows + CPB_T - 1) / CPB_T, p.G);
        hipLaunchKernelGGL(cpb_fwd_kernel<NH>, grid, dim3(CPB_T), 0, st, p);
    }
    GRL_CHECK_LAUNCH();
    return 0;
}

int cpb_dispatch(const GrlCpbArgs& p, hipStream_t st, bool bwd) {
    switch (p.nh) {
        case 1: return cpb_launch<1>(p, st, bwd);
        case 2: return cpb_launch<2>(p, st, bwd);
        case 3: return cpb_launch<3>(p, st, bwd);
        case 4: return cpb_launch<4>(p, st, bwd);
        case 6: return cpb_launch<6>(p, st, bwd);
        case 8: return cpb_launch<8>(p, st, bwd);
        default: return GRL_ERR_UNSUPPORTED;
    }
}

bool cpb_args_ok(const GrlCpbArgs& p) {
    return p.G > 0 && p.G <= 65535 && p.rows > 0 && p.rows4 >= p.rows && (p.rows4 & 3) == 0 && p.hidden == CPB_H && p.coords && p.w1 && p.b1 &&
           p.w2;
}

}  // namespace

extern "C" int grl_cpb_table_fwd(void* stream, const GrlCpbArgs* args) {
    const GrlCpbArgs& p = *args;
    if (!cpb_args_ok(p) || !p.out) return GRL_ERR_BAD_ARG;
    return cpb_dispatch(p, (hipStream_t)stream, false);
}

extern "C" int grl_cpb_table_bwd(void* stream, const GrlCpbArgs* args) {
    const GrlCpbArgs& p = *args;
    if (!cpb_args_ok(p) || !p.d_out || !p.d_w1 || !p.d_b1 || !p.d_w2) return GRL_ERR_BAD_ARG;
    return cpb_dispatch(p, (hipStream_t)stream, true);
}
