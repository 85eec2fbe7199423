// Head planes of the training path (gfx950): projection output -> the attention kernels' operands, forward and backward.
//
// Replaces, in a training step, the element-wise chain between QKVProjection / AnchorLinear and Attention.attn
// (models/common/mixed_attn_block_efficient.py:36-47,85-90: F.normalize(q) * exp(min(logit_scale, ln 100)), F.normalize(k); the
// reshape / permute of :147-150,:240-250) and autograd through it: per (token, slot, head) vector of d <= 32 channels
//     y = x * scale[slot][head] / max(|x|, 1e-12)      (normalised slots: q with the clamped logit scale * log2 e, k with 1)
//     y = x                                            (raw slots: v)
// written as fp32 head planes [slot][head][token][32] with the constants the attention kernels want in the pad columns (1.0 in
// column `one_col`: k slot 31 = partner of the running softmax offset, v column d = the softmax denominator) AND as their fp16
// copy, in one pass over x.  As torch code (GRL._block_planes: norm, clamp, where, mul, permute + cat, fp16 copy -- and in the
// backward pass their adjoints plus last-dim reductions for the scale gradients) this was ~18 launches forward + backward per
// chain, two chains per block: the largest remaining block of glue in the captured training step (profiles/r06_train_kernel_stats.txt).
//
// Mapping: 16 lanes (one DPP row) own one vector, 2 channels per lane (8-byte loads: a vector of d = 30 floats starts on an 8-byte
// boundary only); sums over a vector are DPP row reductions.  Several output slots may read the same input slot (`src`: the
// anchors serve as queries, scaled, and as keys): the backward kernel sums their contributions into one dx.
#include "common.h"
#include "grl_hip_internal.h"

namespace {

constexpr int PL_T = 256;            // threads per workgroup = 16 vectors at a time

// (Both kernels keep PL_U vectors per 16-lane group in flight: the loads of all of them are issued before the first is used.  One
// vector at a time the kernels were latency bound -- 35 / 72 us per launch for 36 / 100 MB of traffic.)
constexpr int PL_U = 4;

__global__ __launch_bounds__(PL_T) void planes_fwd_kernel(GrlPlanesArgs p) {
    const int sub = threadIdx.x & 15;                  // lane inside the vector's DPP row: channels 2 sub, 2 sub + 1
    const int c = 2 * sub;
    const int64_t nvec = (int64_t)p.T * p.S_out * p.nh;
    const int64_t stride = (int64_t)gridDim.x * (PL_T / 16);
    for (int64_t v0 = (int64_t)blockIdx.x * (PL_T / 16) + (threadIdx.x >> 4); v0 < nvec; v0 += stride * PL_U) {
        // v = (t * S_out + s) * nh + h: consecutive vectors read consecutive memory of x
        float2 a[PL_U];
        int hh[PL_U], ss[PL_U];
        int64_t tt[PL_U];
#pragma unroll
        for (int u = 0; u < PL_U; ++u) {
            const int64_t v = v0 + u * stride;
            a[u] = float2{0.f, 0.f};
            hh[u] = 0; ss[u] = 0; tt[u] = 0;
            if (v < nvec) {
                hh[u] = (int)(v % p.nh);
                ss[u] = (int)((v / p.nh) % p.S_out);
                tt[u] = v / ((int64_t)p.nh * p.S_out);
                const float* xv = p.x + ((tt[u] * p.S_in + p.src[ss[u]]) * p.nh + hh[u]) * p.d;
                if (c + 1 < p.d) a[u] = *(const float2*)(xv + c);
                else if (c < p.d) a[u].x = xv[c];
            }
        }
#pragma unroll
        for (int u = 0; u < PL_U; ++u) {
            const int64_t v = v0 + u * stride;
            const int s = ss[u], h = hh[u];
            float f = 1.0f;
            const float nrm = sqrtf(row16_sum(a[u].x * a[u].x + a[u].y * a[u].y));      // (all lanes take part in the row reduction)
            if (!p.raw[s]) f = p.scale[s * p.nh + h] / fmaxf(nrm, 1e-12f);
            float y0 = a[u].x * f, y1 = a[u].y * f;
            if (c == p.one_col[s]) y0 = 1.0f;
            if (c + 1 == p.one_col[s]) y1 = 1.0f;
            if (v < nvec) {
                const int64_t o = (((int64_t)s * p.nh + h) * p.T + tt[u]) * 32 + c;
                if (p.out32 != nullptr) *(float2*)(p.out32 + o) = float2{y0, y1};
                *(uint32_t*)((f16*)p.out16 + o) = pack_f16(y0, y1);
            }
        }
    }
}

__global__ __launch_bounds__(PL_T) void planes_bwd_kernel(GrlPlanesArgs p) {
    __shared__ float ds[8 * 8];                        // scale gradients of this workgroup [S_out <= 8][nh <= 8]
    const int sub = threadIdx.x & 15;
    if (threadIdx.x < 64) ds[threadIdx.x] = 0.f;
    __syncthreads();
    const int c = 2 * sub;
    const int L = p.S_in * p.nh;                       // vectors per token
    const int64_t nvec = (int64_t)p.T * L;
    const int64_t stride = (int64_t)gridDim.x * (PL_T / 16);
    // The host picks the grid so that the stride is a multiple of L: a 16-lane group then meets the SAME (input slot, head) in every
    // iteration, its scale-gradient terms add up in registers and reach the LDS histogram once (one LDS atomic per vector and slot
    // before: 16 of them per iteration on a dozen addresses), and the index arithmetic leaves the loop.
    const int64_t vfirst = (int64_t)blockIdx.x * (PL_T / 16) + (threadIdx.x >> 4);
    const int h = (int)(vfirst % p.nh), i = (int)((vfirst / p.nh) % p.S_in);
    constexpr int U = 2;
    float tot[8];
    bool feeds[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { tot[s] = 0.f; feeds[s] = s < p.S_out && p.src[s] == i && p.dy[s] != nullptr; }
    for (int64_t v0 = vfirst; v0 < nvec; v0 += stride * U) {
        // v = (t * S_in + i) * nh + h: one INPUT vector; every output slot fed by it contributes
        float2 a[U], dd[U][8];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            live[u] = v < nvec;
            const int64_t vc = live[u] ? v : vfirst;
            const int64_t t = vc / L;
            const float* xv = p.x + vc * p.d;
            a[u] = float2{0.f, 0.f};
            if (c + 1 < p.d) a[u] = *(const float2*)(xv + c);
            else if (c < p.d) a[u].x = xv[c];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                dd[u][s] = float2{0.f, 0.f};
                if (feeds[s]) dd[u][s] = *(const float2*)(p.dy[s] + ((int64_t)h * p.T + t) * 32 + c);     // (pad columns carry no gradient)
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            const float nrm = fmaxf(sqrtf(row16_sum(a[u].x * a[u].x + a[u].y * a[u].y)), 1e-12f);
            const float u0 = a[u].x / nrm, u1 = a[u].y / nrm;
            float g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (!feeds[s]) continue;                                                // (group-uniform: the row reduction below is safe)
                const float d0 = c < p.d ? dd[u][s].x : 0.f, d1 = c + 1 < p.d ? dd[u][s].y : 0.f;
                if (p.raw[s]) { g0 += d0; g1 += d1; continue; }
                const float dot = row16_sum(d0 * u0 + d1 * u1);                         // dy . u  (= the scale gradient's term)
                const float f = p.scale[s * p.nh + h] / nrm;
                g0 = fmaf(f, d0 - u0 * dot, g0);
                g1 = fmaf(f, d1 - u1 * dot, g1);
                if (live[u]) tot[s] += dot;
            }
            if (live[u]) {
                if (c + 1 < p.d) *(float2*)(p.dx + v * p.d + c) = float2{g0, g1};
                else if (c < p.d) p.dx[v * p.d + c] = g0;
            }
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < p.S_out && p.want_dscale[s] && tot[s] != 0.f) atomicAdd(&ds[s * 8 + h], tot[s]);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int s = threadIdx.x >> 3, hh = threadIdx.x & 7;
        const int rep = p.dscale_replicas > 1 ? (int)(blockIdx.x % (unsigned)p.dscale_replicas) * p.S_out * p.nh : 0;   // (see GrlLnTrainArgs.stat_replicas)
        if (s < p.S_out && hh < p.nh && p.want_dscale[s] && ds[threadIdx.x] != 0.f) unsafeAtomicAdd(p.dscale + rep + s * p.nh + hh, ds[threadIdx.x]);
    }
}

bool planes_ok(const GrlPlanesArgs& p) {
    if (p.T <= 0 || p.S_in <= 0 || p.S_in > 8 || p.S_out <= 0 || p.S_out > 8 || p.nh <= 0 || p.nh > 8 || p.d <= 0 || p.d > 32 || (p.d & 1)) return false;
    for (int s = 0; s < p.S_out; ++s)
        if (p.src[s] < 0 || p.src[s] >= p.S_in || p.one_col[s] >= 32) return false;
    return p.x != nullptr && p.scale != nullptr;
}

}  // namespace

extern "C" int grl_head_planes_fwd(void* stream, const GrlPlanesArgs* args) {
    const GrlPlanesArgs& p = *args;
    if (!planes_ok(p) || !p.out16) return GRL_ERR_BAD_ARG;      // (out32 optional: the fp32 planes are only autograd's handle on the operands)
    const int64_t nvec = (int64_t)p.T * p.S_out * p.nh;
    const int64_t wgs = (nvec + PL_T / 16 - 1) / (PL_T / 16);
    hipLaunchKernelGGL(planes_fwd_kernel, dim3((unsigned)(wgs < 8192 ? wgs : 8192)), dim3(PL_T), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

extern "C" int grl_head_planes_bwd(void* stream, const GrlPlanesArgs* args) {
    const GrlPlanesArgs& p = *args;
    if (!planes_ok(p) || !p.dx || !p.dscale) return GRL_ERR_BAD_ARG;
    const int64_t nvec = (int64_t)p.T * p.S_in * p.nh;
    int64_t wgs = (nvec + PL_T / 16 - 1) / (PL_T / 16);
    if (wgs > 2048) wgs = 2048;
    // the kernel's precondition: the vector stride (16 per workgroup) is a multiple of the vectors per token, or the launch covers
    // everything in one pass (L <= 64, so at least one such grid exists below 2048)
    const int L = p.S_in * p.nh;
    int g = L, x = PL_T / 16;
    while (x) { const int r = g % x; g = x; x = r; }   // gcd(L, 16)
    const int m = L / g;                               // the grid must be a multiple of m
    if (wgs * (PL_T / 16) < nvec) wgs = wgs / m * m;   // (several passes: round down; never to zero: wgs = 2048 >= m here)
    if (wgs <= 0) return GRL_ERR_BAD_ARG;
    hipLaunchKernelGGL(planes_bwd_kernel, dim3((unsigned)wgs), dim3(PL_T), 0, (hipStream_t)stream, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
