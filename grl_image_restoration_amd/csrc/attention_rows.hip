// Row-streaming cosine attention for 32-aligned windows on gfx950 (MI355X): the fast path of grl_attention_fwd since
// round 3.  Same mathematics and operand layouts as csrc/attention.hip (window, anchor->window and window->anchor
// attention of mixed_attn_block_efficient.py:77-94,128-165,215-270; fp16 operands, bias as accumulator init, lazy
// softmax offset in head-dim slot 31, ones column of V as the denominator) -- what changes is how the key loop runs:
//
//   * measured (tools/ubench/pipes.hip, tools/attn_asm/proto.hip): at head_dim 32 a 32x32 score tile is 4 MFMAs
//     (128 cycles) against 16 v_exp_f32 + 8 packed converts + the overflow test, and ONE wave issues a VALU instruction
//     only every ~5-8 cycles however independent its instructions are; the transcendental / VALU issue is the binding
//     resource and it is only saturated by many waves per SIMD (2 waves: ~265 cycles per tile, 4 waves: ~190).  The
//     round-2 kernel needed 245 VGPRs (2 waves per SIMD) and spent its time in single-wave issue stalls;
//   * so the key loop is written by hand (tools/attn_asm/gen_attn_loop.py -> attn_rows_asm.inc) on a fixed budget of
//     64 transient VGPRs: logits, weights and PV operands are computed IN PLACE (bias fragment -> logits -> exp ->
//     packed fp16 in the low half of the same registers), the overflow test is 9 packed 3-input maxima for two tiles,
//     and the whole kernel stays under 128 VGPRs: 4 workgroups x 4 waves per CU, the hardware interleaves 4 waves per
//     SIMD instead of a software pipeline inside each wave;
//   * 40 KB of LDS per workgroup: K and V chunks of 4 key rows (128 keys) double buffered by LDS-DMA, and the
//     relative-position table as a sliding window of the rows this chunk can reach (<= 4 KB, double buffered), instead
//     of the whole per-workgroup slice (27 KB for anchor->window, which forced one workgroup per CU);
//   * a weight that reaches 2^14 leaves the statement BEFORE the row's PV product; the offsets of the queries
//     concerned are raised from the exact row maximum (compiler-generated cold code), O is rescaled and the row is
//     redone; the maxima are taken over the rest of the chunk (round 4), so a chunk trips at most once.  The offsets start at
//     the logit floor: the first chunk of a wave goes through the same code before its first row ("prime").
// Timing-ablation switches of this file compute WRONG results by construction (they remove work to see what it costs).  They only
// build together with -DGRL_ABLATION, which tools/attn_asm/build_variants_generic.sh passes for its throw-away variant libraries.
#if !defined(GRL_ABLATION) && (defined(ROWS_ABL_NOBARRIER) || defined(ROWS_ABL_NODMA) || defined(ROWS_ABL_REPEAT))
#error "timing-ablation switch without -DGRL_ABLATION: the results of such a build are wrong"
#endif
#include "common.h"
#include "grl_hip_internal.h"
#include "attn_common.h"
#include "attn_rows_asm.inc"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int RW = 4;                 // waves per workgroup; a wave owns 2 query tiles (2 rows x one 32-wide segment)
constexpr int RROWS = 4;              // key rows per chunk
constexpr int RKC = RROWS * 32;       // keys per chunk
constexpr int KBUF = RKC * 64;        // bytes of one K (or V) chunk buffer
constexpr int TBUF = 4096;            // bytes of one table-window buffer
constexpr int TBUF_D32 = 8192;        // ... of the head_dim-32 instance (dn geometry: 16x32 anchors against 64x128 stripes need 6.6 KB)
constexpr int rows_tbuf(bool d32) { return d32 ? TBUF_D32 : TBUF; }
constexpr int rows_lds(bool d32) { return 4 * KBUF + 2 * rows_tbuf(d32); }
constexpr float ROWS_REST = 4.0f;     // after an offset move the row maximum sits in (2^3, 2^4]
constexpr float ROWS_EXTRA = 3.0f;    // ... unless up to this much more reaches the level where the overflow test can go (msafe)

struct RowsGeom {
    int qseg, units, upw, nqs;
};
__host__ __device__ inline RowsGeom rows_geom(const GrlAttnArgs& p) {
    RowsGeom g;
    g.qseg = p.q.ww >> 5;
    g.units = (p.q.wh / 2) * g.qseg;
    g.upw = RW < g.units ? RW : g.units;
    g.nqs = (g.units + g.upw - 1) / g.upw;
    return g;
}
// query rows [hqa, hqb] and 32-wide segments [sga, sgb] of workgroup qs
__host__ __device__ inline void rows_span(const RowsGeom& g, int qs, int& hqa, int& hqb, int& sga, int& sgb) {
    const int u0 = qs * g.upw, u1 = (u0 + g.upw < g.units ? u0 + g.upw : g.units) - 1;
    hqa = 2 * (u0 / g.qseg);
    hqb = 2 * (u1 / g.qseg) + 1;
    if (u0 / g.qseg == u1 / g.qseg) { sga = u0 % g.qseg; sgb = u1 % g.qseg; }
    else { sga = 0; sgb = g.qseg - 1; }
}

#ifdef ROWS_DEBUG
__device__ unsigned long long rows_dbg[8];
#define DBG_T(x) const long long x = __builtin_amdgcn_s_memtime()
#define DBG_ADD(i, v) dbg_acc[i] += (unsigned long long)(v)
#define DBG_FLUSH() do { if (lane_id() == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&rows_dbg[i_], dbg_acc[i_]); } while (0)
#else
#define DBG_T(x)
#define DBG_ADD(i, v)
#define DBG_FLUSH()
#endif

// D32: head_dim 32 (GRL-Small).  All 32 head-dim slots of Q / K / V are data, so the running offset lives in two VGPRs
// (nm0 / nm1 = -m of the lane's query in tile 0 / 1) that the statement adds to the logits, and the softmax denominator is
// summed from the packed fp16 weights (l0 / l1, one partial sum per half-wave); 48 KB of LDS -> 3 workgroups per CU.
template <bool D32>
__global__ __launch_bounds__(RW * 64, D32 ? 3 : 4) void attn_rows_kernel(GrlAttnArgs p) {
    constexpr int TB = rows_tbuf(D32);
    extern __shared__ __attribute__((aligned(1024))) char smem[];
#ifdef ROWS_DEBUG
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DBG_T(t_start);
    const int wave = threadIdx.x >> 6;
    // the lane id is recomputed (2 VALU) wherever it is needed instead of being kept in a VGPR across the key loop
    // (volatile asm: the compiler would hoist a builtin out of the chunk loop and spill what it derives from it)
    auto lane_id = [] {
        int x;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
        return x;
    };
    const RowsGeom g = rows_geom(p);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qs = bid % g.nqs; bid /= g.nqs;
    const int head = bid % p.nh; bid /= p.nh;
    const int wx = bid % p.nwx; bid /= p.nwx;
    const int wy = bid % p.nwy;
    const int b = bid / p.nwy;
    const int D = p.q.ww + p.k.ww - 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // LDS: K[2][KBUF] | V[2][KBUF] | T[2][TBUF]
    constexpr uint32_t VOFF = 2 * KBUF, TOFF = 4 * KBUF;

    const bool border = p.masked && ((p.q.shy > 0 && wy == p.nwy - 1) || (p.q.shx > 0 && wx == p.nwx - 1));
    int hqa, hqb, sga, sgb;
    rows_span(g, qs, hqa, hqb, sga, sgb);
    // reversed table entry of (query (hq, wq), key (hk, wk)) = R0 + (hk - hq) * D + (wk - wq)
    const int R0 = p.trows - 1 - (p.k.wh - 1) * D - (p.k.ww - 1);
    const int nrc = p.k.wh / RROWS, nch = (p.k.ww >> 5) * nrc;

    // ---- this wave's unit: query rows 2*pr, 2*pr+1, segment sg ----
    int unit = qs * g.upw + wave_u;                       // (wave-uniform values stay in SGPRs: the VGPR budget is 128)
    const bool active = wave_u < g.upw && unit < g.units;
    if (!active) unit = qs * g.upw;
    const int pr = unit / g.qseg, sg = unit - pr * g.qseg;
    const int hq0 = 2 * pr;

    f16x8 q00, q01, q10, q11;
    int idq0, idq1;
    {
        int64_t row;
        const int lane = lane_id(), half = lane >> 5, l31 = lane & 31;
        const int wq = 32 * sg + l31;
        locate(p.q, b, wy, wx, hq0 * p.q.ww + wq, row, idq0);
        const f16* src = (const f16*)p.q.ptr + row * p.q.ld + p.q.col0 + head * p.q.hstride + 8 * half;
        q00 = *(const f16x8*)(src);
        q01 = *(const f16x8*)(src + 16);
        locate(p.q, b, wy, wx, (hq0 + 1) * p.q.ww + wq, row, idq1);
        src = (const f16*)p.q.ptr + row * p.q.ld + p.q.col0 + head * p.q.hstride + 8 * half;
        q10 = *(const f16x8*)(src);
        q11 = *(const f16x8*)(src + 16);
        // offsets start at the floor of the head's logits (every weight >= 1): the first row raises them
        // (integer valued, |m| < 2048: exact in fp16 -- slot 31 of the upper half-wave IS the running offset, no second copy)
        const float mq = p.lazy_floor[head];
        if (!D32 && half) { q01[7] = (f16)(-mq); q11[7] = (f16)(-mq); }   // head-dim slot 31 (K holds 1.0 there)
    }
    // (pinned here: left alone the region labels are re-derived inside each instance of the chunk loop, and the row / column values
    // they come from stay alive -- in spill slots -- across the other instance)
    asm volatile("" : "+v"(idq0), "+v"(idq1));
    float nm0 = -p.lazy_floor[head], nm1 = nm0, l0 = 0.f, l1 = 0.f;   // (D32 only)
    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }

    // Per-chunk work outside the statement is kept small on purpose (every wave repeats it for every 4 key rows; the first version
    // spent 165 instructions there against 360 in the rows): chunks are addressed by (strip sk, first key row hk0) carried along
    // the loop -- no divisions; the DMA sources are buffer descriptors (one per tensor) + a per-lane byte offset that only
    // depends on the strip + a scalar row offset; the lane parts of the LDS addresses live in 3 VGPRs, buffer parity and key
    // row enter inside the statement as scalars.
    typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
    auto srd = [](const void* ptr, uint32_t bytes) {   // raw buffer descriptor: base, stride 0, num_records, gfx950 data format word
        const uint64_t a = (uint64_t)ptr;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        r[2] = bytes;
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 ksrd = srd((const f16*)p.k.ptr + (int64_t)b * p.k.Himg * p.k.Wimg * p.k.ld + p.k.col0 + head * p.k.hstride, 0xffffffffu);
    const u32x4 vsrd = srd((const f16*)p.v.ptr + (int64_t)b * p.v.Himg * p.v.Wimg * p.v.ld + p.v.col0 + head * p.v.hstride, 0xffffffffu);
    const u32x4 tsrd = srd(p.table + (int64_t)head * p.tstride, (uint32_t)p.tstride * 4u);   // reads past the head's table return 0
    // bytes per image row / per image column step (transposed view: a row step is one token, a column step Himg tokens)
    const int ktx = p.k.transposed ? p.k.Himg : 1;
    const uint32_t krb = (uint32_t)((p.k.transposed ? 1 : p.k.Wimg) * (int)p.k.ld * 2), vrb = (uint32_t)((p.k.transposed ? 1 : p.k.Wimg) * (int)p.v.ld * 2);
    const int kcol = wave_u & 1, krow = wave_u >> 1;     // this wave's DMA pieces: key rows krow and krow + 2 of a chunk, column half kcol
    // s_nop 4: SGPR operands may come straight from SALU / readfirstlane (5 wait states before a VMEM instruction reads them)
#define ROWS_DMA(m0v, voff, rsrc, soff) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
    // first (4-float aligned) reversed table entry of the window of chunk (sk, hk0)
    auto window_lo = [&](int sk, int hk0) { return (R0 + (hk0 - hqb) * D + 32 * (sk - sgb) - 31) & ~3; };
    const int win_n4 = (((hqb - hqa + RROWS - 1) * D + 32 * (sgb - sga) + 62 + 3) >> 2) + 1;   // 16-B pieces of a window (upper bound)
    // chunk (sk, hk0) -> LDS buffers `par` by DMA: K, V (row major, read back with the hardware transpose), table window
    auto prefetch = [&](int sk, int hk0, int par) {
        const uint32_t bo = (uint32_t)par * KBUF;
        // per-lane byte offsets of the K / V pieces inside an image row (recomputed per chunk: ~12 VALU; kept in VGPRs across
        // the rows they were spilled, and the reload waits drained the DMA queue)
        const int ln = lane_id();
        const int kx = 16 * kcol + (ln >> 2), s3 = ln & 3;                        // key column inside the 32-wide strip, 16-B segment
        int ox = wx * p.k.ww + 32 * sk + kx + p.k.shx; if (ox >= p.k.Wimg) ox -= p.k.Wimg;
        const uint32_t vk = (uint32_t)((ox * ktx * (int)p.k.ld + (s3 ^ ((kx >> 2) & 3)) * 8) * 2);    // XOR swizzle on the source side
        const uint32_t vv = (uint32_t)((ox * ktx * (int)p.v.ld + s3 * 8) * 2);
#pragma unroll
        for (int j = 0; j < KBUF / (RW * 1024); ++j) {
            const int q = wave_u + j * RW;                                           // piece: keys 16*q .. 16*q+15 = row q >> 1, half q & 1
            int oy = wy * p.k.wh + hk0 + krow + 2 * j + p.k.shy; if (oy >= p.k.Himg) oy -= p.k.Himg;
            const uint32_t mk = lds0 + bo + q * 1024, mv = lds0 + VOFF + bo + q * 1024;
            ROWS_DMA(mk, vk, ksrd, (uint32_t)oy * krb);
            ROWS_DMA(mv, vv, vsrd, (uint32_t)oy * vrb);
        }
#pragma unroll
        for (int j = 0; j < TB / (RW * 1024); ++j) {
            const int piece = wave_u + j * RW;
            if (piece * 64 < win_n4) {
                const uint32_t vt = (uint32_t)ln * 16u;
                const uint32_t mt = lds0 + TOFF + (uint32_t)par * TB + piece * 1024;
                ROWS_DMA(mt, vt, tsrd, (uint32_t)(window_lo(sk, hk0) * 4 + piece * 1024));
            }
        }
    };
#ifdef ROWS_STAGGER
    __builtin_amdgcn_s_sleep(1);
    for (int i_ = 0; i_ < (int)(blockIdx.x >> 3 & 3) * 4; ++i_) __builtin_amdgcn_s_sleep(8);   // experiment: de-phase the workgroups of a CU
#endif
    prefetch(0, 0, 0);

    // lane parts of the LDS addresses.  K fragment of key l31 of a row: 16-B segment (2 * kstep + half) ^ sw of its 64-B row
    // (k-step 1: ^ 32, inside the statement); V^T through ds_read_b64_tr_b16; bias of (tile 0, key row): table entry
    // R0 + (hk - hq0) * D + 32 * (sk - sg) + i - l31 with the lane's rows i = 4 * half + {0..3} + 8 * {0..3}
    uint32_t ka0, va, bl;
    {
        const int ln = lane_id();
        const int half = ln >> 5, l31 = ln & 31, sw = (l31 >> 2) & 3;
        ka0 = lds0 + l31 * 64 + ((half ^ sw) << 4);
        va = lds0 + VOFF + (4 * half + ((ln & 15) >> 2)) * 64 + (16 * ((ln >> 4) & 1) + 4 * (ln & 3)) * 2;
        bl = lds0 + TOFF + 4 * (4 * half - l31);
    }
    const int d4 = __builtin_amdgcn_readfirstlane(4 * D);
    // offsets at or above msafe: logit - offset <= 13.5 for every key of the head (GrlAttnArgs.lazy_ceil)
    const int msafe_i = __builtin_amdgcn_readfirstlane(p.lazy_ceil != nullptr ? (int)__builtin_ceilf(p.lazy_ceil[head] - 13.5f) : 0x40000000);

    // BORDER is a compile-time tag: each instance of the chunk loop holds ONE asm statement (with both variants in one loop
    // the register allocator shuffled and spilled the O / Q operands around every statement)
    int poison = 0;  // wave-uniform: a row kept tripping (non-finite logits): the outputs of this wave's queries become NaN
    auto chunks = [&](auto border_tag) {
    constexpr bool BORDER = decltype(border_tag)::value;
    int sk = 0, hk0 = 0;
    int nochk = 0;   // wave-uniform: every query's offset is within 13.5 of the head's logit bound, no weight can reach 2^14 any more
    DBG_T(t_loop);
    DBG_ADD(0, t_loop - t_start);
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        DBG_T(t_c0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA pieces of chunk ch landed, own LDS reads retired
        DBG_T(t_c1);
#ifndef ROWS_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();                                  // everybody's; and all are done with the buffers of chunk ch-1
#endif
        DBG_T(t_c2);
        DBG_ADD(1, t_c1 - t_c0);
        DBG_ADD(2, t_c2 - t_c1);
        const int sk_c = sk, hk_c = hk0;
        hk0 += RROWS;
        if (hk0 == p.k.wh) { hk0 = 0; ++sk; }
#ifndef ROWS_ABL_NODMA
        if (ch + 1 < nch) prefetch(sk, hk0, (ch + 1) & 1);
#endif
        if (!active) continue;
        const uint32_t par = (uint32_t)(ch & 1) * KBUF;
        // scalar part of the bias address of (tile 0, key row hk_c), advanced by the statement row by row
        uint32_t sb = (uint32_t)(ch & 1) * TB + 4 * (R0 + (hk_c - hq0) * D + 32 * (sk_c - sg) - window_lo(sk_c, hk_c));
        // region labels of the 2 x 16-key bands of the chunk's key rows (ops.py:76-157; bands are aligned to 16 on this path)
        uint32_t ids = 0;
        if constexpr (BORDER) {
#pragma unroll
            for (int r = 0; r < RROWS; ++r) {
                const int ry = wy * p.k.wh + hk_c + r, rx = wx * p.k.ww + 32 * sk_c;
                const int iy = 3 * region1d(ry, p.k.Himg, p.k.wh, p.k.shy);
                ids |= (uint32_t)(iy + region1d(rx, p.k.Wimg, p.k.ww, p.k.shx)) << (8 * r);
                ids |= (uint32_t)(iy + region1d(rx + 16, p.k.Wimg, p.k.ww, p.k.shx)) << (8 * r + 4);
            }
            ids = __builtin_amdgcn_readfirstlane(ids);
        }
        int rs = 0, last_trip = -1;
#ifdef ROWS_ABL_REPEAT   // timing experiment: the rows of chunk 0 ROWS_ABL_REPEAT times, no other chunks (results are wrong)
        if (ch > 0) continue;
        for (int rep_ = 0; rep_ < ROWS_ABL_REPEAT; ++rep_) {
        rs = 0; sb = (uint32_t)(ch & 1) * TB + 4 * (R0 + (hk_c - hq0) * D + 32 * (sk_c - sg) - window_lo(sk_c, hk_c));
#endif
        DBG_T(t_c3);
        DBG_ADD(3, t_c3 - t_c2);
        // The offsets start at the logit floor, so the very first row of a wave always "tripped": half a row of wasted work, the exit
        // from the statement and the repair -- 2.5 rows of 32 (measured).  The first chunk now enters the repair code directly
        // (`prime`): its exact maxima set the offsets before any exponential is taken.
        bool prime = ch == 0;
        while (true) {
            int done = 0;
            const int rs_u = __builtin_amdgcn_readfirstlane(rs), nochk_u = __builtin_amdgcn_readfirstlane(nochk);
            if (prime) {
            } else if constexpr (BORDER && D32) {
                uint32_t t0, t1, t2;
                asm volatile(ATTN_ROWS4_MASK1_D32
                             : [o0] "+v"(O0), [o1] "+v"(O1), [l0] "+v"(l0), [l1] "+v"(l1), [sb] "+s"(sb), [done] "=s"(done), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2)
                             : [ka0] "v"(ka0), [va] "v"(va), [bl] "v"(bl), [q00] "v"(q00), [q01] "v"(q01), [q10] "v"(q10), [q11] "v"(q11), [nm0] "v"(nm0), [nm1] "v"(nm1),
                               [d4] "s"(d4), [rs] "s"(rs_u), [par] "s"(par), [nochk] "s"(nochk_u), [ids] "s"(ids), [idq0] "v"(idq0), [idq1] "v"(idq1)
                             : ATTN_ROWS_CLOBBER);
            } else if constexpr (D32) {
                asm volatile(ATTN_ROWS4_MASK0_D32
                             : [o0] "+v"(O0), [o1] "+v"(O1), [l0] "+v"(l0), [l1] "+v"(l1), [sb] "+s"(sb), [done] "=s"(done)
                             : [ka0] "v"(ka0), [va] "v"(va), [bl] "v"(bl), [q00] "v"(q00), [q01] "v"(q01), [q10] "v"(q10), [q11] "v"(q11), [nm0] "v"(nm0), [nm1] "v"(nm1),
                               [d4] "s"(d4), [rs] "s"(rs_u), [par] "s"(par), [nochk] "s"(nochk_u)
                             : ATTN_ROWS_CLOBBER);
            } else if constexpr (BORDER) {
                uint32_t t0, t1, t2;
                asm volatile(ATTN_ROWS4_MASK1
                             : [o0] "+v"(O0), [o1] "+v"(O1), [sb] "+s"(sb), [done] "=s"(done), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2)
                             : [ka0] "v"(ka0), [va] "v"(va), [bl] "v"(bl), [q00] "v"(q00), [q01] "v"(q01), [q10] "v"(q10), [q11] "v"(q11),
                               [d4] "s"(d4), [rs] "s"(rs_u), [par] "s"(par), [nochk] "s"(nochk_u), [ids] "s"(ids), [idq0] "v"(idq0), [idq1] "v"(idq1)
                             : ATTN_ROWS_CLOBBER);
            } else {
                asm volatile(ATTN_ROWS4_MASK0
                             : [o0] "+v"(O0), [o1] "+v"(O1), [sb] "+s"(sb), [done] "=s"(done)
                             : [ka0] "v"(ka0), [va] "v"(va), [bl] "v"(bl), [q00] "v"(q00), [q01] "v"(q01), [q10] "v"(q10), [q11] "v"(q11),
                               [d4] "s"(d4), [rs] "s"(rs_u), [par] "s"(par), [nochk] "s"(nochk_u)
                             : ATTN_ROWS_CLOBBER);
            }
            if (!prime && done == RROWS) { DBG_T(t_c4); DBG_ADD(4, t_c4 - t_c3); break; }
            DBG_ADD(6, 1);
            // ---- cold: a weight of key row `done` reached 2^14.  Exact row maxima -> raise the offsets, rescale O, redo the row ----
            if (!prime && done == last_trip) {
                // the row tripped again right after its offsets were raised: only non-finite logits do that.  Poison the
                // outputs (the reference yields NaN as well) and move on.
                poison = 1;
                rs = done + 1;
                sb += 4 * D;
                if (rs == RROWS) break;
                continue;
            }
            last_trip = prime ? -1 : done;
            prime = false;
            {
                const int ln = lane_id();
                const int half = ln >> 5, l31 = ln & 31, sw = (l31 >> 2) & 3;
                const char* Kc = smem + __builtin_amdgcn_readfirstlane((int)((ch & 1) * KBUF));   // (recomputed: `par` as a VGPR across the statement was a spill slot)
                // (Nothing the repair path needs may be kept in a spill slot across the statement: a scratch reload here is "pending" at
                // the head of the loop on every path, and the compiler then puts s_waitcnt vmcnt(0) in front of the statement -- which
                // also waits for the DMA of the next chunk, issued a few instructions earlier.  Hence the opaque copy of bl, which keeps
                // bl - lds0 from being hoisted out of the chunk loop, and msafe as an integer in an SGPR.)
                uint32_t blx = bl;
                asm volatile("" : "+v"(blx));
                // Exact maxima of the tripping row AND of the rows of the chunk still to come (round 4): at checkpoint-like logit
                // scales 20 of a wave's 32 key rows tripped and a trip costs 2.5 rows (measured, tools/attn_asm/dbg_rows.py); the
                // chunk's K rows and table window are in LDS anyway, so one repair now covers the whole chunk.
                float mx0 = NEG_BIG, mx1 = NEG_BIG;
#pragma unroll 1
                for (int rr = done; rr < RROWS; ++rr) {
                    const int kk = 32 * rr + l31;
                    const f16x8 kf0 = *(const f16x8*)(Kc + kk * 64 + (((0 + half) ^ sw) << 4));
                    const f16x8 kf1 = *(const f16x8*)(Kc + kk * 64 + (((2 + half) ^ sw) << 4));
                    const float* t0p = (const float*)(smem + (blx + sb - lds0)) + (rr - done) * D;   // tile 0's fragment of row rr; tile 1's is one table row below
                    f32x16 S0, S1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { S0[r] = t0p[(r & 3) + 8 * (r >> 2)]; S1[r] = t0p[(r & 3) + 8 * (r >> 2) - D]; }
                    S0 = mfma32_f16(kf0, q00, S0);
                    S1 = mfma32_f16(kf0, q10, S1);
                    S0 = mfma32_f16(kf1, q01, S0);
                    S1 = mfma32_f16(kf1, q11, S1);
                    if constexpr (BORDER) {
                        const int id_lo = (ids >> (8 * rr)) & 15, id_hi = (ids >> (8 * rr + 4)) & 15;
                        const float a_lo = id_lo != idq0 ? MASK_L2 : 0.f, a_hi = id_hi != idq0 ? MASK_L2 : 0.f;
                        const float b_lo = id_lo != idq1 ? MASK_L2 : 0.f, b_hi = id_hi != idq1 ? MASK_L2 : 0.f;
#pragma unroll
                        for (int r = 0; r < 8; ++r) { S0[r] += a_lo; S0[8 + r] += a_hi; S1[r] += b_lo; S1[8 + r] += b_hi; }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) { mx0 = fmaxf(mx0, S0[r]); mx1 = fmaxf(mx1, S1[r]); }
                }
                if constexpr (D32) { mx0 += nm0; mx1 += nm1; }   // (the fragments carry no offset slot: relative to the current offsets)
                mx0 = fmaxf(mx0, xhalf(mx0));
                mx1 = fmaxf(mx1, xhalf(mx1));
                float d0 = fmaxf(0.f, __builtin_ceilf(mx0) - ROWS_REST), d1 = fmaxf(0.f, __builtin_ceilf(mx1) - ROWS_REST);
                {
                    // offsets within ROWS_EXTRA of the level where the overflow test becomes unnecessary go there right away
                    // (the row maximum then rests at 2^(ROWS_REST - ROWS_EXTRA) at worst: ample for fp16 weights)
                    const float msafe = (float)msafe_i;
                    float m0, m1;   // the new offsets
                    if constexpr (D32) {
                        m0 = d0 - nm0; m1 = d1 - nm1;
                    } else {
                        const float t0 = (float)q01[7], t1 = (float)q11[7], u0 = xhalf(t0), u1 = xhalf(t1);
                        m0 = d0 - (half ? t0 : u0); m1 = d1 - (half ? t1 : u1);
                    }
                    if (m0 < msafe && m0 >= msafe - ROWS_EXTRA) { d0 += msafe - m0; m0 = msafe; }
                    if (m1 < msafe && m1 >= msafe - ROWS_EXTRA) { d1 += msafe - m1; m1 = msafe; }
                    nochk = __builtin_amdgcn_ballot_w64(!(m0 >= msafe && m1 >= msafe)) == 0;
                }
                if constexpr (D32) { nm0 -= d0; nm1 -= d1; }
                else if (half) { q01[7] = (f16)((float)q01[7] - d0); q11[7] = (f16)((float)q11[7] - d1); }   // slot 31 holds -m
                const float f0 = __builtin_amdgcn_exp2f(-d0), f1 = __builtin_amdgcn_exp2f(-d1);
#pragma unroll
                for (int r = 0; r < 16; ++r) { O0[r] *= f0; O1[r] *= f1; }
                if constexpr (D32) { l0 *= f0; l1 *= f1; }
            }
            rs = done;
        }
#ifdef ROWS_ABL_REPEAT
        }
#endif
    }
    };
    DBG_T(t_l0);
    if (border) chunks(std::true_type{});
    else chunks(std::false_type{});
    DBG_T(t_l1);
    DBG_ADD(5, t_l1 - t_l0);
    DBG_ADD(7, 1);
    DBG_FLUSH();
    if (!active) return;

    {
        int64_t row;
        int rid;
        const int lane = lane_id(), half = lane >> 5, l31 = lane & 31;
        const int wq = 32 * sg + l31;
        const float mq0 = D32 ? -nm0 : -xhalf((float)q01[7]), mq1 = D32 ? -nm1 : -xhalf((float)q11[7]);   // lower half-wave <- the upper one's slot 31
        // (an opaque copy of the window width: otherwise the reciprocal and the row / column values of the prologue's locate() are
        // kept alive across the whole key loop for this one -- five registers in spill slots on a 128-register budget)
        GrlTokenGrid gq = p.q;
        asm volatile("" : "+s"(gq.ww));
        locate(gq, b, wy, wx, hq0 * gq.ww + wq, row, rid);
        float l = D32 ? l0 + xhalf(l0) : ones_row(O0, p.ones_col, half);
        if (poison) l = __builtin_nanf("");
        store_o(p, O0, 1.0f / l, row, head, half);
        if (p.lse != nullptr && half == 0) p.lse[(int64_t)head * p.lse_stride + row] = mq0 + __builtin_amdgcn_logf(l);
        locate(gq, b, wy, wx, (hq0 + 1) * gq.ww + wq, row, rid);
        l = D32 ? l1 + xhalf(l1) : ones_row(O1, p.ones_col, half);
        if (poison) l = __builtin_nanf("");
        store_o(p, O1, 1.0f / l, row, head, half);
        if (p.lse != nullptr && half == 0) p.lse[(int64_t)head * p.lse_stride + row] = mq1 + __builtin_amdgcn_logf(l);
    }
}

}  // namespace

// 1 when the geometry is served by the row-streaming kernel (the caller has already checked the lazy-offset preconditions)
bool grl_attn_rows_supported(const GrlAttnArgs& p) {
    const bool d32 = p.head_dim == 32;   // no spare slot: offset and denominator on the VALU (attn_rows_kernel<true>)
    if ((p.q.ww % 32) || (p.k.ww % 32) || (p.q.wh % 2) || (p.k.wh % RROWS) || (p.ones_col < 0) != d32) return false;
    if (p.masked && ((p.k.shx & 15) || (p.q.shx & 15))) return false;
    const RowsGeom g = rows_geom(p);
    const int D = p.q.ww + p.k.ww - 1;
    for (int qs = 0; qs < g.nqs; ++qs) {   // every workgroup's table window must fit one buffer
        int hqa, hqb, sga, sgb;
        rows_span(g, qs, hqa, hqb, sga, sgb);
        const int n = (hqb - hqa + RROWS - 1) * D + 32 * (sgb - sga) + 63 + 3;
        if (n > rows_tbuf(d32) / 4) return false;
    }
    return true;
}

#ifdef ROWS_DEBUG
extern "C" int grl_attn_rows_debug(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(rows_dbg), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(rows_dbg), z, sizeof(z)); }
    return 0;
}
#endif

int grl_attn_rows_launch(const GrlAttnArgs& p, hipStream_t st) {
    const RowsGeom g = rows_geom(p);
    const int64_t grid = (int64_t)g.nqs * p.nh * p.nwx * p.nwy * p.B;
    if (grid > 0x7fffffff) return GRL_ERR_BAD_ARG;
    const bool d32 = p.head_dim == 32;
    auto kfn = d32 ? attn_rows_kernel<true> : attn_rows_kernel<false>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds(d32));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3((int)grid), dim3(RW * 64), rows_lds(d32), st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}
