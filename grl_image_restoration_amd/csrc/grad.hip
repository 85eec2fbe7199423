// Weight-gradient GEMM and the multi-tensor AdamW step of the GRL training path (gfx950).
//
//   grl_gemm_tn : C[N, K] += sum_m A[m, n] * B[row(m), k]        (reduction over tokens / pixels)
//       replaces the weight gradients autograd computes for F.linear / F.conv2d in the reference's training step
//       (engines/base.py:221-236 backward through models/common/mixed_attn_block_efficient.py, mixed_attn_block.py,
//       swin_v1_block.py, networks/grl.py): dW = dY^T X for the token-wise linears (QKV, anchor, proj, fc1, fc2) and,
//       with the image-shift mode, the nine taps of a 3x3 convolution's weight gradient
//       dW[tap][co][ci] = sum_pixels dY[p, co] * X[p + (dy, dx), ci]   (zero outside the image).
//   grl_adamw_step : torch.optim.AdamW (config/optimizer/adamw.yaml) over a list of tensors in ONE launch
//       (1390 parameter tensors for GRL-Base: a per-tensor launch would be host-bound).
//
// gemm_tn design: the reduction dimension (tokens) is the long one (M ~ 10^4..10^6, N, K <= 576), i.e. both operands
// are tall-skinny matrices read once per output tile column/row.  A workgroup owns a 64 x 64 output tile and a slab of
// M; it converts 32-row pieces of A (fp32 gradients, pre-scaled by `a_scale` into fp16 range) and B (fp32 or fp16
// activations) to fp16 in LDS, row-major, and reads them back TRANSPOSED with ds_read_b64_tr_b16 -- the MFMA wants the
// reduction index in its k-slots -- so nothing is transposed in registers.  Four waves = 2 x 2 tiles of
// mfma_f32_32x32x16_f16; partial tiles of the M slabs are combined with fp32 atomics (C must be zeroed by the caller).
#include "common.h"
#include <stdlib.h>
#include "grl_hip_internal.h"

namespace {

constexpr int GT = 64;        // output tile side
constexpr int GM = 32;        // rows (reduction) per LDS piece
constexpr int GROW = GT * 2;  // bytes per LDS row (64 fp16), XOR-swizzled in 16-B segments by row

// 8 operand values from column `off` on; `nv` of them exist (8, 4 or 0: widths are multiples of 4, nothing is read beyond a row's end)
__device__ __forceinline__ f16x8 load8_f16(const void* base, int dtype, int64_t off, float scale, int nv) {
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (nv <= 0) return v;
    if (dtype == GRL_DT_F16) {
        v = *(const f16x8*)((const f16*)base + off);
    } else {
        const float4* q = (const float4*)((const float*)base + off);
        const float4 a0 = q[0], a1 = nv > 4 ? q[1] : float4{0, 0, 0, 0};
        v[0] = to_f16(a0.x * scale); v[1] = to_f16(a0.y * scale); v[2] = to_f16(a0.z * scale); v[3] = to_f16(a0.w * scale);
        v[4] = to_f16(a1.x * scale); v[5] = to_f16(a1.y * scale); v[6] = to_f16(a1.z * scale); v[7] = to_f16(a1.w * scale);
    }
    return v;
}

// (csrc/attn_common.h: xcd_remap) XCD x works on a contiguous range of the n work items; bijective for any n
__device__ __forceinline__ int xcd_remap_tn(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GrlGemmTnArgs p) {
    __shared__ __attribute__((aligned(16))) char As[GM * GROW];
    __shared__ __attribute__((aligned(16))) char Bs[GM * GROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // Work item = (slab of M, tap; output tile).  The tiles of one slab read the same rows of a and b -- a is re-read by every k tile,
    // b by every n tile -- so they are kept on ONE XCD, next to each other in time: its L2 then serves the re-reads (the dispatcher
    // places workgroup i on XCD i % 8; with the plain blockIdx order the 27 tiles of a QKV slab were spread over all eight L2s and
    // every one of them fetched its own copy).  GRL_GEMM_TN_XCD=0 at launch time restores the plain order (A/B timing).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.reserved0 == 0) {
        const int T = gridDim.x * gridDim.y;
        const int w = xcd_remap_tn(bx + gridDim.x * (by + gridDim.y * bz), T * gridDim.z);
        const int tile = w % T;
        bz = w / T; by = tile / gridDim.x; bx = tile - by * gridDim.x;
    }
    const int n0 = bx * GT, k0 = by * GT;
    const int tap = bz / p.splits, slab = bz - tap * p.splits;
    const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap % 3 - 1 : 0;
    // M slab of this workgroup (multiple of GM rows)
    const int pieces = (p.M + GM - 1) / GM;
    const int per = (pieces + p.splits - 1) / p.splits;
    const int pc0 = slab * per, pc1 = min(pieces, pc0 + per);
    const int wn = wave & 1, wk = wave >> 1;
    const bool a_f16 = p.a_dtype == GRL_DT_F16;       // (already a_scale-d: GrlLinearArgs.a16_out of the data-gradient launch)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // staging: thread -> (row = tid >> 3, 16-B segment = tid & 7) of a 32 x 64 piece
    const int srow = tid >> 3, sseg = tid & 7;
    for (int pc = pc0; pc < pc1; ++pc) {
        const int m = pc * GM + srow;
        f16x8 av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = av;
        if (m < p.M) {
            av = load8_f16(p.a, a_f16 ? GRL_DT_F16 : GRL_DT_F32, (int64_t)m * p.lda + n0 + sseg * 8, a_f16 ? 1.0f : p.a_scale, p.N - (n0 + sseg * 8));
            int64_t brow = m;
            bool inside = true;
            if (p.taps == 9) {   // image shift: B row of pixel (y + dy, x + dx), zero outside the image
                const int x = m % p.W, y = (m / p.W) % p.H;
                inside = (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
                brow = (int64_t)m + dy * p.W + dx;
            }
            const int kc = k0 + sseg * 8;
            if (inside) {
                bv = load8_f16(p.b, p.b_dtype, brow * p.ldb + kc, 1.0f, p.K - kc);
                if (p.b_ones) {   // the virtual ones column (bias gradient); K and kc are multiples of 4 / 8: slot 0 or 4
                    if (p.K == kc) bv[0] = (f16)1.0f;
                    else if (p.K == kc + 4) bv[4] = (f16)1.0f;
                }
            }
        }
        __syncthreads();   // previous piece's readers are done
        *(f16x8*)(As + srow * GROW + ((sseg ^ (srow & 7)) << 4)) = av;
        *(f16x8*)(Bs + srow * GROW + ((sseg ^ (srow & 7)) << 4)) = bv;
        __syncthreads();
        // operand fragments: lane (col = l31 of this wave's 32 columns, half) gets, for k-step s, reduction rows
        // 16*s + 4*half + e (e < 4) and 16*s + 8 + 4*half + (e - 4): the same permutation on both operands.
        // ds_read_b64_tr_b16: within a 16-lane group lane i points at row i>>2, columns 4*(i&3) of a [4 rows][16 cols]
        // block and receives column (i & 15), rows 0..3.
        typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;
        auto frag = [&](const char* base, int col0, int s) -> f16x8 {
            // column group of this lane: 16 * ((lane >> 4) & 1) + 4 * (lane & 3) within the wave's 32 columns
            const int c = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const int seg = c >> 3, sub = (c & 7) * 2;
            s16x4 lo, hi;
            {
                const int r = 16 * s + 4 * half + ((lane & 15) >> 2);
                lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + r * GROW + ((seg ^ (r & 7)) << 4) + sub));
            }
            {
                const int r = 16 * s + 8 + 4 * half + ((lane & 15) >> 2);
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + r * GROW + ((seg ^ (r & 7)) << 4) + sub));
            }
            const s16x8 both = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(f16x8, both);
        };
#pragma unroll
        for (int s = 0; s < GM / 16; ++s) {
            const f16x8 af = frag(As, 32 * wn, s);   // A operand: rows = output n, k-slots = reduction rows
            const f16x8 bf = frag(Bs, 32 * wk, s);   // B operand: cols = output k
            acc = mfma32_f16(af, bf, acc);
        }
    }
    // accumulator: row n = n0 + 32*wn + (r&3) + 8*(r>>2) + 4*half, col k = k0 + 32*wk + l31
    float* c = p.c + (int64_t)tap * p.c_tap_stride;
    const int k = k0 + 32 * wk + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * wn + mfma32_row(r, half);
        if (n < p.N && k < p.K) {
            if (p.c_fix != nullptr)     // deterministic: 64-bit fixed point, integer atomics (any order gives the same sum)
                atomicAdd((unsigned long long*)(p.c_fix + (int64_t)tap * p.c_tap_stride + (int64_t)n * p.ldc + k),
                          (unsigned long long)(long long)__float2ll_rn(acc[r] * 1073741824.0f));
            else
                unsafeAtomicAdd(c + (int64_t)n * p.ldc + k, acc[r] * p.out_scale);
        } else if (n < p.N && k == p.K && p.b_ones && (p.taps == 1 || tap == 4)) {   // column sums of a: the bias gradient
            if (p.c_bias_fix != nullptr)
                atomicAdd((unsigned long long*)(p.c_bias_fix + n), (unsigned long long)(long long)__float2ll_rn(acc[r] * 1073741824.0f));
            else
                unsafeAtomicAdd(p.c_bias + n, acc[r] * p.out_scale);
        }
    }
}

// ---- multi-tensor AdamW ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(GrlAdamWArgs p) {
    // chunk -> (tensor, offset): one workgroup per 4096-element chunk of one tensor
    const int ch = blockIdx.x;
    const int t = p.chunk_tensor[ch];
    const int64_t off = (int64_t)p.chunk_offset[ch];
    const int64_t n = p.numel[t];
    float* w = (float*)p.params[t];
    const float* g = (const float*)p.grads[t];
    float* m = (float*)p.exp_avg[t];
    float* v = (float*)p.exp_avg_sq[t];
    // (captured launches: learning rate and weight decay come from device memory, refreshed by the host before each replay)
    const float lr = p.hyper_dev != nullptr ? p.hyper_dev[0] : p.lr;
    const float wd_all = p.hyper_dev != nullptr ? p.hyper_dev[1] : p.weight_decay;
    const float wd = p.weight_decay_flags != nullptr && p.weight_decay_flags[t] == 0 ? 0.f : wd_all;
    float bc1 = p.bias_correction1, bc2s = p.bias_correction2_sqrt;
    if (p.bias_corrections_dev != nullptr) {   // graph replay: the step count, and what depends on it, live on the device
        bc1 = p.bias_corrections_dev[0];
        bc2s = p.bias_corrections_dev[1];
    }
#pragma unroll 4
    for (int64_t i = off + threadIdx.x; i < min(n, off + 4096); i += 256) {
        const float gi = g[i] * p.grad_scale;
        float wi = w[i];
        wi *= 1.0f - lr * wd;                               // decoupled weight decay (torch.optim.AdamW)
        const float mi = p.beta1 * m[i] + (1.0f - p.beta1) * gi;
        const float vi = p.beta2 * v[i] + (1.0f - p.beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + p.eps;
        w[i] = wi - (lr / bc1) * (mi / denom);
    }
}

}  // namespace

extern "C" int grl_gemm_tn(void* stream, const GrlGemmTnArgs* args) {
    const GrlGemmTnArgs& p = *args;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 0;
    if (p.b_dtype != GRL_DT_F32 && p.b_dtype != GRL_DT_F16) return GRL_ERR_BAD_ARG;
    if (p.a_dtype != 0 && p.a_dtype != GRL_DT_F32 && p.a_dtype != GRL_DT_F16) return GRL_ERR_BAD_ARG;
    const int aq = p.a_dtype == GRL_DT_F16 ? 8 : 4, bq = p.b_dtype == GRL_DT_F16 ? 8 : 4;   // elements per 16-byte piece
    if ((p.lda % aq) || (p.ldb % bq) || (p.N % 4) || (p.K % 4) || p.ldc < p.K) return GRL_ERR_BAD_ARG;
    if (p.lda < (p.N + aq - 1) / aq * aq || p.ldb < (p.K + bq - 1) / bq * bq) return GRL_ERR_BAD_ARG;   // (whole pieces are read)
    if (p.b_ones && p.c_bias == nullptr && p.c_bias_fix == nullptr) return GRL_ERR_BAD_ARG;
    if ((p.c_fix != nullptr) != (p.c_bias_fix != nullptr) && p.b_ones) return GRL_ERR_BAD_ARG;
    if (p.taps != 1 && p.taps != 9) return GRL_ERR_BAD_ARG;
    if (p.taps == 9 && (p.H <= 0 || p.W <= 0 || p.M % (p.H * p.W) != 0)) return GRL_ERR_BAD_ARG;
    if (p.splits <= 0 || p.splits * p.taps > 65535 || (p.c == nullptr && p.c_fix == nullptr)) return GRL_ERR_BAD_ARG;
    const dim3 grid((p.N + GT - 1) / GT, (p.K + (p.b_ones ? 1 : 0) + GT - 1) / GT, p.splits * p.taps);
    GrlGemmTnArgs q = p;
    static const int plain_order = getenv("GRL_GEMM_TN_XCD") ? atoi(getenv("GRL_GEMM_TN_XCD")) == 0 : 0;
    q.reserved0 = plain_order;         // (kernel-internal use of the reserved field: 1 = blockIdx order)
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, q);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_adamw_step(void* stream, const GrlAdamWArgs* args) {
    const GrlAdamWArgs& p = *args;
    if (p.num_chunks <= 0) return 0;
    if (!p.params || !p.grads || !p.exp_avg || !p.exp_avg_sq || !p.numel || !p.chunk_tensor || !p.chunk_offset) return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(adamw_kernel, dim3(p.num_chunks), dim3(256), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
