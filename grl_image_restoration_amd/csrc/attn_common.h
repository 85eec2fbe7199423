// Index arithmetic shared by the attention forward and backward kernels (gfx950): window / stripe partition, cyclic shift,
// shifted-window region labels (models/common/ops.py:36-157) as address math on a GrlTokenGrid.
#pragma once
#include "common.h"

namespace {

constexpr float MASK_L2 = -100.0f * LOG2E_F;   // the reference's -100 mask value in the log2 domain
constexpr float NEG_BIG = -1.0e30f;

// Workgroup -> work item map that keeps consecutive work items (the query blocks of one window and
// head, which share K/V) on ONE XCD: the dispatcher places block b on XCD b % 8 and every XCD has a
// private L2 (MI355X guide, T1).  Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// global -> LDS copy of one head's bias table with 4 x 16 B loads in flight per thread
__device__ __forceinline__ void load_table(float* tab, const float* src, int trows, int tid, int nthreads) {
    const int n4 = (trows + 3) >> 2;  // the per-head stride is padded to a multiple of 4 floats
    const float4* s4 = (const float4*)src;
    float4* d4 = (float4*)tab;
    for (int i0 = tid; i0 < n4; i0 += 4 * nthreads) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = s4[min(i0 + j * nthreads, n4 - 1)];   // (conditional loads put v[] into scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * nthreads < n4) d4[i0 + j * nthreads] = v[j];
    }
}

__device__ __forceinline__ int region1d(int p, int n, int s, int sh) {
    // ops.py:76-100: labels 0 | 1 | 2 split at n-s and n-sh; a zero shift labels the whole axis alike
    if (sh == 0) return 0;
    return p < n - s ? 0 : (p < n - sh ? 1 : 2);
}

// token n of window (wy,wx) of image b -> (row index in the token matrix, region id)
__device__ __forceinline__ void locate(const GrlTokenGrid& g, int b, int wy, int wx, int n, int64_t& row, int& rid) {
    const int hq = n / g.ww, wq = n - hq * g.ww;
    const int ry = wy * g.wh + hq, rx = wx * g.ww + wq;
    int oy = ry + g.shy; if (oy >= g.Himg) oy -= g.Himg;
    int ox = rx + g.shx; if (ox >= g.Wimg) ox -= g.Wimg;
    row = g.transposed ? ((int64_t)b * g.Wimg + ox) * g.Himg + oy : ((int64_t)b * g.Himg + oy) * g.Wimg + ox;
    rid = 3 * region1d(ry, g.Himg, g.wh, g.shy) + region1d(rx, g.Wimg, g.ww, g.shx);
}

// 16 x fp16 of one (query, head) slot half -> global as two 16-B stores.  A lane holds 4 x 4 consecutive head dims (8-B
// pieces), its partner 32 lanes away the interleaved ones: the pair swaps two pieces each so that every lane owns 2 x 8
// consecutive dims (the epilogue is store-issue bound: a token's 64-B slot is written by 2 lanes x 2 instructions, not 2 x 4).
__device__ __forceinline__ void store_f16_slot(f16* slot, const float (&v)[16], int half) {
    uint2 pk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        pk[g].x = pack_f16(v[4 * g + 0], v[4 * g + 1]);
        pk[g].y = pack_f16(v[4 * g + 2], v[4 * g + 3]);
    }
    // half 0: dims 0-7 = own g0 | partner g0, dims 16-23 = own g2 | partner g2;  half 1: dims 8-15 = partner g1 | own g1, 24-31 likewise
    swap32(pk[0].x, pk[1].x); swap32(pk[0].y, pk[1].y);   // (upper half of the first <-> lower half of the second operand)
    swap32(pk[2].x, pk[3].x); swap32(pk[2].y, pk[3].y);
    const uint4 lo = uint4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};
    const uint4 hi = uint4{pk[2].x, pk[2].y, pk[3].x, pk[3].y};
    *(uint4*)(slot + 8 * half) = lo;
    *(uint4*)(slot + 8 * half + 16) = hi;
}

// normalised O^T fragment of one query tile -> global (lane holds head dims 8*g + 4*half + [0..3] of query l31)
__device__ __forceinline__ void store_o(const GrlAttnArgs& p, const f32x16& O, float inv, int64_t qrow, int head, int half) {
    if (p.out_dtype == GRL_DT_F32) {
        float* dst = (float*)p.o.ptr + qrow * p.o.ld + p.o.col0 + head * p.o.hstride + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(dst + 8 * g) = float4{O[4 * g + 0] * inv, O[4 * g + 1] * inv, O[4 * g + 2] * inv, O[4 * g + 3] * inv};
    } else {
        const int64_t off = qrow * p.o.ld + p.o.col0 + head * p.o.hstride;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = O[r] * inv;
        store_f16_slot((f16*)p.o.ptr + off, v, half);
        if (p.o_lo != nullptr) {   // rounding residual: the low half of a split-precision operand (precision "high")
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] -= (float)to_f16(v[r]);
            store_f16_slot((f16*)p.o_lo + off, v, half);
        }
    }
}

// row `oc` of an O^T fragment (the ones-column of V: the softmax denominator), valid in both half-waves
__device__ __forceinline__ float ones_row(const f32x16& O, int oc, int half) {
    const int base_row = oc & ~4;  // row index with the half bit cleared
    float cand = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (mfma32_row(r, 0) == base_row) cand = O[r];
    const float other = xhalf(cand);
    return (half == ((oc >> 2) & 1)) ? cand : other;
}

}  // namespace
