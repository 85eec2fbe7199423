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
        for (int j = 0; j < 4; ++j)
            if (i0 + j * nthreads < n4) v[j] = s4[i0 + j * nthreads];
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
    row = ((int64_t)b * g.Himg + oy) * g.Wimg + ox;
    rid = 3 * region1d(ry, g.Himg, g.wh, g.shy) + region1d(rx, g.Wimg, g.ww, g.shx);
}

}  // namespace
