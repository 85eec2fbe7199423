// Cosine window / anchored-stripe attention for GRL on gfx950 (MI355X).
//
// One kernel family serves the three attentions of a GRL block:
//   window   : q,k,v = window tokens                 (mixed_attn_block_efficient.py:128-165)
//   a2w      : q = anchors,  k,v = stripe tokens     (:256-258)
//   w2a      : q = stripe tokens, k = anchors, v = a2w output   (:259)
// Roll, window/stripe partition + reverse, head split, relative-position index and the shifted
// window masks of the reference (ops.py:36-157,352-375) never touch memory here: they are index
// arithmetic on the token grids described by GrlTokenGrid.
//
// Formulation (flash-style, nothing of size Nq x Nk is materialised), all operands fp16, fp32 accumulation:
//   S^T tile (32 keys x 32 queries) = mfma_32x32x16_f16(K_tile, Q_tile), accumulator *initialised with the
//   relative-position bias* gathered from an LDS table, so that S^T = log2e * (scale * cos(q,k) + bias) - offset
//   comes straight out of the matrix core (scale*log2e is folded into q by the QKV epilogue).  P = exp2(S^T) is
//   packed to fp16 in registers in exactly the order the PV product wants its B operand (a permutation of the key
//   index inside a tile, mirrored when V^T fragments are read), and O^T += V^T P^T runs on the matrix core again.
//   With `ones_col`, the softmax denominator is row `ones_col` of O^T.
//
// The softmax offset.  P is an fp16 MFMA operand and sub-normal halves are lost to the matrix core, so the weights of a
// row must sit high in the fp16 range: largest weight of a row between 2^4 and 2^14 leaves >= 18 binades (12 nats) below it.
// The logit scale is clamped at 100 (efficient.py:39), i.e. logits span +-144 in the log2 domain, and q / k are different
// projections, so no a-priori bound of the row maximum is tight enough: the offset must follow the data.
//   lazy   (fast kernel): a running per-query offset m, integer valued, kept in the spare head-dim slot 31 of the Q
//           fragment (K carries 1.0 there), i.e. it costs no instruction in the loop: S^T = K.Q + bias - m comes out of the
//           matrix core.  A tile whose largest packed weight reaches 2^14 (packed 16-bit max over the tile + one compare)
//           takes a rare wave-uniform slow path that raises m so that the tile maximum lands at 2^4, rescales O and redoes
//           the tile -- flash-attention's running maximum, evaluated only when needed;
//   online (generic kernel): ordinary running maximum per tile, weights = exp2(S - max + 14).
//
// Work decomposition: one workgroup = (window, head, block of up to 256 queries); each wave owns 2 query tiles (64
// queries) whose Q fragments and O^T accumulators stay in registers; K and V chunks of 256 keys are staged in LDS.
#include "common.h"
#include <type_traits>
#include "grl_hip_internal.h"
#include "attn_common.h"
#include <stdlib.h>

namespace {

constexpr int QT = 2;            // query tiles per wave
constexpr int KC = 256;          // keys per LDS chunk
constexpr int VROW = KC * 2 + 8; // V^T row stride in bytes of the round-2 fast kernel's transposed staging (pad: conflict-free ds_read_b64)
constexpr float P_TOP = 14.0f;           // log2 of the largest fp16 weight the kernels produce (fp16 max is 2^16)
constexpr float LAZY_REST = 4.0f;        // lazy offset: after a rescale the tile maximum sits in (2^3, 2^4]
                                         // (measured: 6 -> 4 costs nothing in parity and saves 9 % at logit scale 100; 2 loses parity margin)
constexpr unsigned short LAZY_TRIP = 0x7400;   // fp16 bit pattern of 2^14: a packed weight >= this moves the offset
constexpr unsigned short F16_INF = 0x7C00;

// ------------------------------------------------------------------------------------------------
// Generic kernel: any window shape (12x12 windows, 8-anchor stripes, ragged key counts, head_dim 32), online softmax.
// ------------------------------------------------------------------------------------------------
// SPLIT: split-precision operands (precision "high"): S = q_hi k_hi + q_lo k_hi + q_hi k_lo, O = P_hi v_hi + P_lo v_hi + P_hi v_lo with the fp16
// residual planes q_lo / k_lo / v_lo and the rounding residual of the weights (3 + 3 MFMA terms instead of 1 + 1: ~22-bit operands).
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef __fp16 tr_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// V^T fragments of a 32-key tile staged row major (64 B per key) through the LDS transpose read (same lane map as
// csrc/attention_rows.hip): rows = head dim l31; k-slot e of step s <-> key 16*s + 8*(e>>2) + 4*half + (e&3)
__device__ __forceinline__ void vt_frags(const char* tile, int lane, f16x8 (&vf)[2]) {
    const int half = lane >> 5;
    const char* a = tile + (4 * half + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    typedef __attribute__((address_space(3))) tr_h4* lp;
    const tr_h4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(a)), r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(a + 512));
    const tr_h4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(a + 1024)), r3 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(a + 1536));
    vf[0] = __builtin_bit_cast(f16x8, __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7));
    vf[1] = __builtin_bit_cast(f16x8, __builtin_shufflevector(r2, r3, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool ONES, bool KW4, bool SPLIT>
__global__ __launch_bounds__(256) void attn_kernel(GrlAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    const int half = lane >> 5, l31 = lane & 31;
    const int Nq = p.q.wh * p.q.ww, Nk = p.k.wh * p.k.ww;
    const int qblk = (nthreads >> 6) * (QT * 32);
    const int nqs = (Nq + qblk - 1) / qblk;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qs = bid % nqs; bid /= nqs;
    const int head = bid % p.nh; bid /= p.nh;
    const int wx = bid % p.nwx; bid /= p.nwx;
    const int wy = bid % p.nwy;
    const int b = bid / p.nwy;
    const int D = p.q.ww + p.k.ww - 1;

    // ---- LDS carve ----
    float* tab = (float*)smem;                                   // trows
    const int kcap = min(KC, (Nk + 31) & ~31);                   // key slots staged at a time (small windows: more workgroups per CU)
    char* Ks = smem + (((size_t)p.trows * 4 + 15) & ~(size_t)15); // kcap x 64 B (16-B slots XOR-swizzled)
    char* Vs = Ks + kcap * 64;                                   // kcap x 64 B row major: V^T fragments come out of ds_read_b64_tr_b16
    int* koff = (int*)(Vs + kcap * 64);                          // kcap
    unsigned char* kreg = (unsigned char*)(koff + kcap);         // kcap
    char* Ks2 = (char*)(kreg + kcap);                            // SPLIT: residual planes, same layouts
    char* Vs2 = Ks2 + kcap * 64;

    load_table(tab, p.table + (int64_t)head * p.tstride, p.trows, tid, nthreads);

    // does this window need the mask path at all?  (window touches the wrapped border, or ragged keys)
    const bool border = p.masked && ((p.q.shy > 0 && wy == p.nwy - 1) || (p.q.shx > 0 && wx == p.nwx - 1));

    // ---- per-lane query state ----
    int U[QT], idq[QT];
    int64_t qrow[QT];
    bool qvalid[QT];
    f16x8 qf[QT][2], ql[QT][2];
    f32x16 O[QT];
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int n = qs * qblk + wave * (QT * 32) + t * 32 + l31;
        qvalid[t] = n < Nq;
        if (!qvalid[t]) n = Nq - 1;
        locate(p.q, b, wy, wx, n, qrow[t], idq[t]);
        const int hq = n / p.q.ww, wq = n - hq * p.q.ww;
        U[t] = p.trows - 1 - (hq * D + wq + (p.k.wh - 1) * D + (p.k.ww - 1));  // reversed table: index = U + v
        const f16* src = (const f16*)p.q.ptr + qrow[t] * p.q.ld + p.q.col0 + head * p.q.hstride + 8 * half;
        qf[t][0] = *(const f16x8*)(src);
        qf[t][1] = *(const f16x8*)(src + 16);
        if constexpr (SPLIT) {
            ql[t][0] = ql[t][1] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (p.q_lo != nullptr) {
                const f16* sl = (const f16*)p.q_lo + qrow[t] * p.q.ld + p.q.col0 + head * p.q.hstride + 8 * half;
                ql[t][0] = *(const f16x8*)(sl);
                ql[t][1] = *(const f16x8*)(sl + 16);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
        mrun[t] = NEG_BIG;
        lrun[t] = 0.f;
    }

    const int nchunks = (Nk + KC - 1) / KC;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int k0 = ch * KC;
        const int klen = min(KC, Nk - k0);
        const int ntiles = (klen + 31) >> 5;
        __syncthreads();
        // ---- stage K rows (swizzled 16-B slots) and V^T; per-key table offset + region id ----
        // (one thread per key: a single locate() -- two integer divisions -- per key instead of per 16-B piece; V goes in row
        // major like K, one 16-B write per piece: the transposed copy cost eight 2-byte LDS writes per piece)
        for (int kk = tid; kk < ntiles * 32; kk += nthreads) {
            const int n = k0 + kk;
            const bool valid = n < Nk;
            int64_t row; int rid;
            locate(p.k, b, wy, wx, valid ? n : 0, row, rid);
            const f16* ksrc = (const f16*)p.k.ptr + row * p.k.ld + p.k.col0 + head * p.k.hstride;
            const f16* vsrc = (const f16*)p.v.ptr + row * p.v.ld + p.v.col0 + head * p.v.hstride;
            // (a dead key slot -- past the window's last key -- holds key 0's data: its logit is masked to -1e30 through kreg,
            // its weight is exactly 0, so any finite K / V will do and the loads need no predicate)
            f16x8 kv[4], vv[4];
#pragma unroll
            for (int seg = 0; seg < 4; ++seg) {
                kv[seg] = *(const f16x8*)(ksrc + seg * 8);
                vv[seg] = *(const f16x8*)(vsrc + seg * 8);
            }
#pragma unroll
            for (int seg = 0; seg < 4; ++seg) {
                *(f16x8*)(Ks + kk * 64 + ((seg ^ ((kk >> 2) & 3)) << 4)) = kv[seg];
                *(f16x8*)(Vs + kk * 64 + seg * 16) = vv[seg];
            }
            if constexpr (SPLIT) {
                const int64_t ko = row * p.k.ld + p.k.col0 + head * p.k.hstride, vo = row * p.v.ld + p.v.col0 + head * p.v.hstride;
#pragma unroll
                for (int seg = 0; seg < 4; ++seg) {
                    const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                    kv[seg] = p.k_lo != nullptr ? *(const f16x8*)((const f16*)p.k_lo + ko + seg * 8) : zero;
                    vv[seg] = p.v_lo != nullptr ? *(const f16x8*)((const f16*)p.v_lo + vo + seg * 8) : zero;
                }
#pragma unroll
                for (int seg = 0; seg < 4; ++seg) {
                    *(f16x8*)(Ks2 + kk * 64 + ((seg ^ ((kk >> 2) & 3)) << 4)) = kv[seg];
                    *(f16x8*)(Vs2 + kk * 64 + seg * 16) = vv[seg];
                }
            }
            const int nn = valid ? n : 0;
            const int hk = nn / p.k.ww, wk = nn - hk * p.k.ww;
            koff[kk] = hk * D + wk;
            kreg[kk] = valid ? (unsigned char)rid : (unsigned char)255;
        }
        __syncthreads();

        for (int kt = 0; kt < ntiles; ++kt) {
            const int kb = kt * 32;
            // K fragments: A operand rows = keys kb + l31, k-slots = head dims 8*half (+16 for step 1)
            f16x8 kf[2];
            {
                const int kk = kb + l31;
                const int sw = (kk >> 2) & 3;
                kf[0] = *(const f16x8*)(Ks + kk * 64 + (((0 + half) ^ sw) << 4));
                kf[1] = *(const f16x8*)(Ks + kk * 64 + (((2 + half) ^ sw) << 4));
            }
            // bias gather -> accumulator init
            f32x16 S[QT];
            if constexpr (KW4) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int v0 = koff[kb + 8 * g + 4 * half];
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const float* tp = tab + (U[t] + v0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) S[t][4 * g + e] = tp[e];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int v = koff[kb + mfma32_row(r, half)];
#pragma unroll
                    for (int t = 0; t < QT; ++t) S[t][r] = tab[U[t] + v];
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                S[t] = mfma32_f16(kf[0], qf[t][0], S[t]);
                S[t] = mfma32_f16(kf[1], qf[t][1], S[t]);
            }
            if constexpr (SPLIT) {
                f16x8 k2[2];
                const int kk = kb + l31;
                const int sw = (kk >> 2) & 3;
                k2[0] = *(const f16x8*)(Ks2 + kk * 64 + (((0 + half) ^ sw) << 4));
                k2[1] = *(const f16x8*)(Ks2 + kk * 64 + (((2 + half) ^ sw) << 4));
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    S[t] = mfma32_f16(kf[0], ql[t][0], S[t]);
                    S[t] = mfma32_f16(kf[1], ql[t][1], S[t]);
                    S[t] = mfma32_f16(k2[0], qf[t][0], S[t]);
                    S[t] = mfma32_f16(k2[1], qf[t][1], S[t]);
                }
            }
            // (wave-uniform: only a border window needs the region mask, only the window's last tile has dead key slots)
            if (border || k0 + kb + 32 > Nk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t ids = *(const uint32_t*)(kreg + kb + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idk = (ids >> (8 * e)) & 255;
#pragma unroll
                        for (int t = 0; t < QT; ++t) {
                            float s = S[t][4 * g + e];
                            if (idk == 255) s = NEG_BIG;
                            else if (border && idk != idq[t]) s += MASK_L2;
                            S[t][4 * g + e] = s;
                        }
                    }
                }
            }
            // online softmax; numerators packed straight into the PV B-operand order
            f16x8 pb[QT][2];
            f16x8 pl[SPLIT ? QT : 1][2];   // SPLIT: rounding residual of the weights, p - fp16(p) (round 5: the third PV term)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float mx = S[t][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[t][r]);
                mx = fmaxf(mx, xhalf(mx));
                const float mnew = fmaxf(mrun[t], mx);
                if (__builtin_amdgcn_ballot_w64(mnew > mrun[t]) != 0) {   // (wave-uniform) some query's running maximum moved: rescale
                    const float alpha = __builtin_amdgcn_exp2f(mrun[t] - mnew);
                    const f32x2v a2 = {alpha, alpha};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2v o2 = f32x2v{O[t][r], O[t][r + 1]} * a2;
                        O[t][r] = o2[0]; O[t][r + 1] = o2[1];
                    }
                    lrun[t] *= alpha;
                    mrun[t] = mnew;
                }
                float ps = 0.f;
                const float nsub = P_TOP - mnew;   // weights <= 2^14: 28 binades of fp16 normals below the row maximum
                const f32x2v ns2 = {nsub, nsub};
                uint32_t pw[8], pwl[SPLIT ? 8 : 1];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2v e2 = f32x2v{S[t][r], S[t][r + 1]} + ns2;
                    const float p0 = __builtin_amdgcn_exp2f(e2[0]), p1 = __builtin_amdgcn_exp2f(e2[1]);
                    if constexpr (!ONES) ps += p0 + p1;
                    pw[r >> 1] = pack_f16_raw(p0, p1);
                    if constexpr (SPLIT) pwl[r >> 1] = pack_f16_raw(p0 - (float)(f16)p0, p1 - (float)(f16)p1);
                }
                pb[t][0] = __builtin_bit_cast(f16x8, u32x4v{pw[0], pw[1], pw[2], pw[3]});
                pb[t][1] = __builtin_bit_cast(f16x8, u32x4v{pw[4], pw[5], pw[6], pw[7]});
                if constexpr (SPLIT) {
                    pl[t][0] = __builtin_bit_cast(f16x8, u32x4v{pwl[0], pwl[1], pwl[2], pwl[3]});
                    pl[t][1] = __builtin_bit_cast(f16x8, u32x4v{pwl[4], pwl[5], pwl[6], pwl[7]});
                }
                if constexpr (!ONES) lrun[t] += ps;
            }
            // V^T fragments: rows = head dim l31; k-slot e <-> key kb + 16*s + 8*(e>>2) + 4*half + (e&3)
            f16x8 vf[2];
            vt_frags(Vs + kb * 64, lane, vf);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                O[t] = mfma32_f16(vf[0], pb[t][0], O[t]);
                O[t] = mfma32_f16(vf[1], pb[t][1], O[t]);
            }
            if constexpr (SPLIT) {
                // O = P_hi v_hi + P_lo v_hi + P_hi v_lo.  The P_lo term is new in round 5: with the weights rounded to fp16 the
                // everything-split path left 5.8e-3 on the ill-conditioned clamp-scale fixture of GRL-Small (CPU emulation of that
                // one rounding alone: 6.4e-3; the reference's own fp32 arithmetic is 2.4e-4 from its float64 run there), with it 4.6e-4
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    O[t] = mfma32_f16(vf[0], pl[t][0], O[t]);
                    O[t] = mfma32_f16(vf[1], pl[t][1], O[t]);
                }
                f16x8 v2[2];
                vt_frags(Vs2 + kb * 64, lane, v2);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    O[t] = mfma32_f16(v2[0], pb[t][0], O[t]);
                    O[t] = mfma32_f16(v2[1], pb[t][1], O[t]);
                }
            }
        }
    }

    // ---- normalise and store ----
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l;
        if constexpr (ONES) l = ones_row(O[t], p.ones_col, half);
        else l = lrun[t] + xhalf(lrun[t]);
        if (qvalid[t]) {
            store_o(p, O[t], 1.0f / l, qrow[t], head, half);
            if (p.lse != nullptr && half == 0) p.lse[(int64_t)head * p.lse_stride + qrow[t]] = mrun[t] - P_TOP + __builtin_amdgcn_logf(l);
        }
    }
}

template <bool ONES, bool SPLIT>
int launch_kw(const GrlAttnArgs& p, int grid, int block, size_t lds, hipStream_t st) {
    const bool kw4 = (p.k.ww % 4) == 0;
#define GRL_ATTN_GO(KW4V)                                                                                   \
    {                                                                                                       \
        auto kfn = attn_kernel<ONES, KW4V, SPLIT>;                                                          \
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                           (int)lds);                                                       \
        if (e != hipSuccess) return (int)e;                                                                 \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(block), lds, st, p);                                       \
    }
    if (kw4) GRL_ATTN_GO(true) else GRL_ATTN_GO(false)
#undef GRL_ATTN_GO
    GRL_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Fast path: 32-aligned windows (q.ww % 32 == 0, k.ww % 32 == 0, q.wh % 2 == 0, k.wh % 8 == 0), ones column, lazy offset.  Covers the released-checkpoint geometries (window 32; stripes 64x64 / 64x128 and their anchor grids)
// -- the MFMA-bound regime of SURVEY 8(d).
//
//   * 4 waves per workgroup share one K / V chunk (8 key rows x 32 keys of one 32-wide strip) and the bias table;
//     K chunks go global -> LDS by DMA into a double buffer one chunk ahead (XOR swizzle applied on the source side);
//     V either the same way, read back with the hardware transpose (ds_read_b64_tr_b16; KDMA 2), or -- where the table
//     slice leaves no room for four chunk buffers at two workgroups per CU -- through registers, written transposed
//     as key pairs (KDMA 1);
//   * a wave owns 2 query tiles: the same 32-wide column segment of 2 consecutive window rows, so K / V^T fragments
//     are read once per 2 tiles and 2 independent MFMA chains are in flight per wave;
//   * a key tile is a 32-wide segment of one key row: the bias of (query lane, key row i) is tab[U + i] with the table
//     stored reversed, i.e. plain ascending LDS reads that initialise the accumulator.  The bias fragment of (query
//     row hq+1, key row hk+1) equals that of (hq, hk), so tile 1 takes the fragment tile 0 gathered one key row
//     earlier: two fragment registers HA / HB swap roles every row and LDS bias reads are halved.
// ------------------------------------------------------------------------------------------------
constexpr int FW = 4, QTN = 2, FROWS = 8;
constexpr int FKC = FROWS * 32;
constexpr int FVROW = FKC * 2 + 8;

// Upper bound of the bias-table entries one workgroup of the fast kernel needs: its queries span at most
// QTN * (units_per_wg / segments_per_row + 2) rows, every key row of the window, all column offsets.
__host__ __device__ inline int fast_table_floats(const GrlAttnArgs& p) {
    const int qseg = p.q.ww >> 5;
    const int units = (p.q.wh / QTN) * qseg;
    const int upw = FW < units ? FW : units;
    const int rows = QTN * (upw / qseg + 2);
    const int D = p.q.ww + p.k.ww - 1;
    const int n = (rows - 1 + p.k.wh) * D + 4;
    const int all = (p.trows + 3) & ~3;
    // whole 1-KiB DMA pieces: the last piece of the table copy must not reach into the K buffer behind the table,
    // whose own DMA is in flight at the same time
    return ((n < all ? n : all) + 255) & ~255;
}

template <int KDMA>
__global__ __launch_bounds__(FW * 64, 2) void attn_fast_kernel(GrlAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int qseg = p.q.ww >> 5;                      // 32-wide segments per query row
    const int units = (p.q.wh / QTN) * qseg;           // (row group, segment) units per window
    const int upw = min(FW, units);                    // units (= active waves) per workgroup
    const int nqs = (units + upw - 1) / upw;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qs = bid % nqs; bid /= nqs;
    const int head = bid % p.nh; bid /= p.nh;
    const int wx = bid % p.nwx; bid /= p.nwx;
    const int wy = bid % p.nwy;
    const int b = bid / p.nwy;
    const int D = p.q.ww + p.k.ww - 1;

    // LDS: bias table slice (the rows this workgroup's queries can reach) | two K chunks | V (KDMA 2: two row-major chunks,
    // KDMA 1: one transposed chunk [32][FVROW]) | key region ids
    float* tab = (float*)smem;
    const int tab_floats = fast_table_floats(p);
    char* Ks0 = smem + (((size_t)tab_floats * 4 + 15) & ~(size_t)15);
    char* Vt = Ks0 + 2 * FKC * 64;
    unsigned char* kreg0 = (unsigned char*)(Vt + (KDMA == 2 ? 2 * FKC * 64 : 32 * FVROW));   // KDMA 2: two buffers of FKC bytes

    // the table arrives REVERSED (see grl_hip.h) so that a lane's 16 key rows read ascending addresses.
    // It is DMA'd (global_load_lds, 1 KiB per wave-instruction, no staging registers): all pieces are in flight at
    // once and overlap the first K/V chunk's loads; the barrier in front of the first LDS use waits for them.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int tab_lo = 0;   // first table entry (reversed order, multiple of 4) held in LDS
    {
        // queries of this workgroup: rows hq_lo .. hq_hi  ->  reversed entries trows - (hq_hi + k.wh) * D .. trows - 1 - hq_lo * D
        const int u0 = qs * upw, u1 = min(u0 + upw, units) - 1;
        const int hq_lo = QTN * (u0 / qseg), hq_hi = QTN * (u1 / qseg) + QTN - 1;
        const int pos_lo = p.trows - (hq_hi + p.k.wh) * D, pos_hi = p.trows - 1 - hq_lo * D;
        tab_lo = pos_lo & ~3;
        const int first4 = tab_lo >> 2;
        const int n4 = ((pos_hi - tab_lo) >> 2) + 1;
        const int last4 = ((p.trows + 3) >> 2) - 1;
        const float4* s4 = (const float4*)(p.table + (int64_t)head * p.tstride);
        for (int q = wave_u; q * 64 < n4; q += FW) {
            int i = first4 + q * 64 + lane;
            i = i < last4 ? i : last4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s4 + i),
                                             (__attribute__((address_space(3))) void*)(tab + q * 256), 16, 0, 0);
        }
    }
    const bool border = p.masked && ((p.q.shy > 0 && wy == p.nwy - 1) || (p.q.shx > 0 && wx == p.nwx - 1));
    const bool band16 = (p.k.shx & 15) == 0;   // key-column region bands are aligned to 16 (k.ww % 32 == 0 on this path)

    // ---- this wave's unit: query rows QTN*pr .. QTN*pr+QTN-1, segment sg ----
    int unit = qs * upw + wave;
    const bool active = wave < upw && unit < units;
    if (!active) unit = 0;
    const int pr = unit / qseg, sg = unit - pr * qseg;
    int Ub[QTN], idq[QTN];
    int64_t qrow[QTN];
    f16x8 qf[QTN][2];
    f32x16 O[QTN];
    float mq[QTN];   // the query's softmax offset (integer valued, log2 domain); mirrored as -mq in q slot 31
#pragma unroll
    for (int t = 0; t < QTN; ++t) {
        const int hq = QTN * pr + t, wq = 32 * sg + l31;
        locate(p.q, b, wy, wx, hq * p.q.ww + wq, qrow[t], idq[t]);
        // table index of (query, key (hk, wk)) = U - hk*D - wk, reversed: (trows-1-U) + hk*D + wk;
        // lane's key rows are wk = 32*sk + i, i = (r&3) + 8*(r>>2) + 4*half
        Ub[t] = p.trows - 1 - (hq * D + wq + (p.k.wh - 1) * D + (p.k.ww - 1)) + 4 * half - tab_lo;
        const f16* src = (const f16*)p.q.ptr + qrow[t] * p.q.ld + p.q.col0 + head * p.q.hstride + 8 * half;
        qf[t][0] = *(const f16x8*)(src);
        qf[t][1] = *(const f16x8*)(src + 16);
        // start at the floor of the head's logits (every weight >= 1): the slow path raises it as larger logits appear
        mq[t] = p.lazy_floor[head];
        if (half) qf[t][1][7] = (f16)(-mq[t]);   // head-dim slot 31 (K holds 1.0 there)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    }

    f32x16 HA, HB;  // bias fragments carried across key rows
#pragma unroll
    for (int r = 0; r < 16; ++r) { HA[r] = 0.f; HB[r] = 0.f; }
    const int kseg = p.k.ww >> 5;
    const int nrc = p.k.wh / FROWS;  // chunks per strip
    const int nch = kseg * nrc;

    // ---- K / V staging ----
    constexpr int SPT = (FKC * 4) / (FW * 64);   // (key, 16-B segment) slots per thread and chunk
    static_assert(SPT * FW * 64 == FKC * 4 && (SPT % 2) == 0, "chunk must divide evenly over the workgroup in key pairs");

    // key kk of chunk ch -> row of the token matrix (+ region id of the key)
    auto key_row = [&](int ch, int kk, int& rid) -> int64_t {
        const int sk = ch / nrc, hk0 = (ch - sk * nrc) * FROWS;
        const int ry = wy * p.k.wh + hk0 + (kk >> 5), rx = wx * p.k.ww + 32 * sk + (kk & 31);
        int oy = ry + p.k.shy; if (oy >= p.k.Himg) oy -= p.k.Himg;
        int ox = rx + p.k.shx; if (ox >= p.k.Wimg) ox -= p.k.Wimg;
        rid = 3 * region1d(ry, p.k.Himg, p.k.wh, p.k.shy) + region1d(rx, p.k.Wimg, p.k.ww, p.k.shx);
        return ((int64_t)b * p.k.Himg + oy) * p.k.Wimg + ox;
    };
    f16x8 pv_[SPT];
    int prid[SPT];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // The K chunk goes global -> LDS by DMA (LDS segment sigma holds segment (sigma & 3) ^ ((kk >> 2) & 3) of key
    // kk = sigma >> 2); both K and V are issued one chunk ahead, while the previous chunk is in the matrix cores.
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int j = 0; j < (FKC * 64) / (FW * 1024); ++j) {
            const int q = wave_u + j * FW;
            const int sigma = q * 64 + lane, kk = sigma >> 2, seg = (sigma & 3) ^ ((kk >> 2) & 3);
            int rid;
            const int64_t row = key_row(ch, kk, rid);
            const f16* g = (const f16*)p.k.ptr + row * p.k.ld + p.k.col0 + head * p.k.hstride + seg * 8;
            const uint32_t m0v = lds0 + (uint32_t)(Ks0 - smem) + (ch & 1) * FKC * 64 + q * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
        }
        if constexpr (KDMA == 2) {
            // V the same way, row-major and unswizzled: the PV operand is fetched with ds_read_b64_tr_b16 (hardware
            // transpose across 16 lanes), so nothing passes through registers and there is no commit phase at all
#pragma unroll
            for (int j = 0; j < (FKC * 64) / (FW * 1024); ++j) {
                const int q = wave_u + j * FW;
                const int sigma = q * 64 + lane, kk = sigma >> 2, seg = sigma & 3;
                int rid;
                const int64_t row = key_row(ch, kk, rid);
                const f16* g = (const f16*)p.v.ptr + row * p.v.ld + p.v.col0 + head * p.v.hstride + seg * 8;
                const uint32_t m0v = lds0 + (uint32_t)(Vt - smem) + (ch & 1) * FKC * 64 + q * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
            }
            if (p.masked) {
                for (int kk = tid; kk < FKC; kk += FW * 64) {
                    int rid;
                    key_row(ch, kk, rid);
                    kreg0[(ch & 1) * FKC + kk] = (unsigned char)rid;
                }
            }
        } else {
            // V: a thread owns (key pair, segment) slots so that the transposed LDS writes are 4 bytes (two keys) wide
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int i2 = tid + (j >> 1) * FW * 64;           // pair slot: pair = i2 >> 2, segment = i2 & 3
                const int64_t row = key_row(ch, 2 * (i2 >> 2) + (j & 1), prid[j]);
                pv_[j] = *(const f16x8*)((const f16*)p.v.ptr + row * p.v.ld + p.v.col0 + head * p.v.hstride + (i2 & 3) * 8);
            }
        }
    };
    prefetch(0);

#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int sk = ch / nrc, hk0 = (ch - sk * nrc) * FROWS;
        const char* Ks = Ks0 + (ch & 1) * FKC * 64;
        const char* Vs = Vt + (ch & 1) * FKC * 64;                              // KDMA 2 only
        unsigned char* kreg = kreg0 + (KDMA == 2 ? (ch & 1) * FKC : 0);
        if constexpr (KDMA == 2) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA pieces of chunk ch landed, own LDS traffic retired
            __builtin_amdgcn_s_barrier();                                  // everybody's; and all are done with the buffers of chunk ch-1
            if (ch + 1 < nch) prefetch(ch + 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA pieces (table, K chunk) and V loads have landed
            __builtin_amdgcn_s_barrier();                       // ... everybody's; and all are done reading the previous chunk
#pragma unroll
            for (int j = 0; j < SPT; j += 2) {
                const int i2 = tid + (j >> 1) * FW * 64;
                const int kk0 = 2 * (i2 >> 2), seg = i2 & 3;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f16x2 two;
                    two[0] = pv_[j][e];
                    two[1] = pv_[j + 1][e];
                    *(f16x2*)(Vt + (seg * 8 + e) * FVROW + kk0 * 2) = two;
                }
                if (seg == 0) *(unsigned short*)(kreg + kk0) = (unsigned short)(prid[j] | (prid[j + 1] << 8));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // V^T / region ids visible
            if (ch + 1 < nch) prefetch(ch + 1);                 // issued behind the barrier: its address math delays nobody
        }
        if (!active) continue;

        // LDS reads of one key tile: K fragments, V^T fragments, key region ids
        auto frags = [&](int kt, f16x8 (&kf)[2], f16x8 (&vf)[2], uint32_t (&ids)[4]) {
            const int kb = kt * 32;
            const int kk = kb + l31;
            const int sw = (kk >> 2) & 3;
            kf[0] = *(const f16x8*)(Ks + kk * 64 + (((0 + half) ^ sw) << 4));
            kf[1] = *(const f16x8*)(Ks + kk * 64 + (((2 + half) ^ sw) << 4));
            if constexpr (KDMA == 2) {
                // lane (d = l31, half): elements e < 4 = keys 16*s2 + 4*half + e, e >= 4 = keys 16*s2 + 8 + 4*half + e-4.
                // ds_read_b64_tr_b16: within a 16-lane group lane i points at row i>>2, columns 4*(i&3) of a [4 keys][16 d]
                // block and receives column (i & 15), rows 0..3.
                typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
                typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
                const char* vb = Vs + (kb + 4 * half + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + (16 * s2) * 64));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + (16 * s2 + 8) * 64));
                    typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;
                    const s16x8 both = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    vf[s2] = __builtin_bit_cast(f16x8, both);
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const char* vp = Vt + l31 * FVROW + (kb + 16 * s2 + 4 * half) * 2;
                    const f16x4 lo = *(const f16x4*)(vp);
                    const f16x4 hi = *(const f16x4*)(vp + 16);
                    vf[s2] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            if (border) {
#pragma unroll
                for (int g = 0; g < 4; ++g) ids[g] = *(const uint32_t*)(kreg + kb + 8 * g + 4 * half);
            }
        };
        auto gather = [&](int t, int hk, f32x16& dst) {
            const float* tp = tab + (Ub[t] + hk * D + 32 * sk);
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = tp[(r & 3) + 8 * (r >> 2)];
        };
        // BORDER is a compile-time tag: with a run-time `if (border)` around the mask the two versions of S meet in 32 register
        // copies (+ the MFMA drain in front of them) on the common unmasked path
        auto logits = [&](auto border_tag, const f16x8 (&kf)[2], const f32x16& C0, const f32x16& C1, const uint32_t (&ids)[4], f32x16 (&S)[2]) {
            S[0] = mfma32_f16(kf[0], qf[0][0], C0);
            S[1] = mfma32_f16(kf[0], qf[1][0], C1);
            S[0] = mfma32_f16(kf[1], qf[0][1], S[0]);
            S[1] = mfma32_f16(kf[1], qf[1][1], S[1]);
            if constexpr (decltype(border_tag)::value) {
                if (band16) {
                    // region labels change only at multiples of 16 key columns (shift and window width are multiples of 16):
                    // keys 0..15 of the tile (accumulator registers 0..7) share one label, keys 16..31 (registers 8..15) another
                    const int id_lo = ids[0] & 255, id_hi = ids[2] & 255;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float m_lo = id_lo != idq[t] ? MASK_L2 : 0.f, m_hi = id_hi != idq[t] ? MASK_L2 : 0.f;
#pragma unroll
                        for (int r = 0; r < 8; ++r) { S[t][r] += m_lo; S[t][8 + r] += m_hi; }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int idk = (ids[r >> 2] >> (8 * (r & 3))) & 255;
                            S[t][r] += idk != idq[t] ? MASK_L2 : 0.f;
                        }
                }
            }
        };
        auto pair = [&](auto border_tag, f16x8 (&kf)[2], f16x8 (&vf)[2], const f32x16& C0, const f32x16& C1, uint32_t (&ids)[4]) {
            f32x16 S[2];
            logits(border_tag, kf, C0, C1, ids, S);
            // weights as packed fp16 pairs (word i of a tile = accumulator registers 2i, 2i+1: the PV B-operand order)
            typedef __attribute__((__vector_size__(2 * sizeof(unsigned short)))) unsigned short u16x2;
            typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
            uint32_t pw[2][8];
            auto weights = [&](float sub0, float sub1) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float sub = t ? sub1 : sub0;
                        typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;
                        const f32x2 e2 = {__builtin_amdgcn_exp2f(S[t][2 * i] - sub), __builtin_amdgcn_exp2f(S[t][2 * i + 1] - sub)};
                        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(e2, f16x2));   // one v_cvt_pk_f16_f32
                        asm("" : "+v"(w));   // opaque: keeps the compiler from converting every half a second time for the max below
                        pw[t][i] = w;
                    }
            };
            weights(0.f, 0.f);
            // largest packed weight of each tile: 7 v_pk_max_u16 (positive halves order like integers) + 1 to fold the pair
            unsigned short top[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u16x2 m4[4];   // tree, not a chain: seven dependent v_pk_max_u16 cost a stall each
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    m4[i] = __builtin_elementwise_max(__builtin_bit_cast(u16x2, pw[t][2 * i]), __builtin_bit_cast(u16x2, pw[t][2 * i + 1]));
                const u16x2 m2 = __builtin_elementwise_max(__builtin_elementwise_max(m4[0], m4[1]), __builtin_elementwise_max(m4[2], m4[3]));
                top[t] = m2[0] > m2[1] ? m2[0] : m2[1];
            }
            bool post = false;   // wave-uniform: raise the offsets after this tile's PV product
            if (__builtin_amdgcn_ballot_w64(top[0] >= LAZY_TRIP || top[1] >= LAZY_TRIP) != 0ull) {
                if (__builtin_amdgcn_ballot_w64(top[0] >= F16_INF || top[1] >= F16_INF) != 0ull) {
                    // a weight overflowed fp16: raise the offsets of the queries concerned so that their tile maximum lands at
                    // 2^REST, rescale their accumulators, redo the tile's weights
                    logits(border_tag, kf, C0, C1, ids, S);
                    float delta[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float mx = S[t][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[t][r]);
                        mx = fmaxf(mx, xhalf(mx));
                        delta[t] = fmaxf(0.f, __builtin_ceilf(mx) - LAZY_REST);
                        mq[t] += delta[t];
                        if (half) qf[t][1][7] = (f16)(-mq[t]);
                        const float f = __builtin_amdgcn_exp2f(-delta[t]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[t][r] *= f;
                    }
                    weights(delta[0], delta[1]);
                } else {
                    post = true;   // weights in [2^14, 2^16): still exact -- use them, move the offset afterwards
                }
            }
            f16x8 pb[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    pb[t][j] = __builtin_bit_cast(f16x8, u32x4{pw[t][4 * j], pw[t][4 * j + 1], pw[t][4 * j + 2], pw[t][4 * j + 3]});
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                O[t] = mfma32_f16(vf[0], pb[t][0], O[t]);
                O[t] = mfma32_f16(vf[1], pb[t][1], O[t]);
            }
            if (post) {
                // cheap offset move: the row's largest weight is known from its fp16 exponent; bring it down to 2^REST
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int mine = top[t];
                    const int row = max(mine, __shfl_xor(mine, 32, 64));
                    const int e = (row >> 10) - 15;                          // floor(log2(largest weight)); sub-normals give < -14
                    const float delta = (float)max(0, e - (int)LAZY_REST);
                    mq[t] += delta;
                    if (half) qf[t][1][7] = (f16)(-mq[t]);
                    const float f = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[t][r] *= f;
                }
            }
        };
        auto rows = [&](auto border_tag) {
#pragma unroll 1
            for (int kt = 0; kt < FROWS; kt += 2) {
                f16x8 kf[2], vf[2];
                uint32_t ids[4] = {0, 0, 0, 0};
                frags(kt, kf, vf, ids);
                if (hk0 + kt == 0) gather(1, 0, HB);      // strip start: tile 1 has no predecessor fragment
                gather(0, hk0 + kt, HA);
                pair(border_tag, kf, vf, HA, HB, ids);
                frags(kt + 1, kf, vf, ids);
                gather(0, hk0 + kt + 1, HB);              // HB (tile 1 @ row kt) is consumed
                pair(border_tag, kf, vf, HB, HA, ids);    // tile 1 @ row kt+1 == tile 0 @ row kt
            }
        };
        if (border) rows(std::true_type{});
        else rows(std::false_type{});
    }
    if (!active) return;

#pragma unroll
    for (int t = 0; t < QTN; ++t) {
        const float l = ones_row(O[t], p.ones_col, half);
        store_o(p, O[t], 1.0f / l, qrow[t], head, half);
        if (p.lse != nullptr && half == 0) p.lse[(int64_t)head * p.lse_stride + qrow[t]] = mq[t] + __builtin_amdgcn_logf(l);
    }
}

size_t fast_lds_bytes(const GrlAttnArgs& p, int kdma) {
    const size_t kc = (size_t)FKC;
    const size_t tab = ((size_t)fast_table_floats(p) * 4 + 15) & ~(size_t)15;
    if (kdma == 2) return tab + 4 * kc * 64 + 2 * kc;
    return tab + 2 * kc * 64 + 32 * (kc * 2 + 8) + kc;
}

template <int KDMA>
int launch_fast_v(const GrlAttnArgs& p, hipStream_t st) {
    const int units = (p.q.wh / QTN) * (p.q.ww >> 5);
    const int upw = min(FW, units);
    const int nqs = (units + upw - 1) / upw;
    const int64_t grid = (int64_t)nqs * p.nh * p.nwx * p.nwy * p.B;
    if (grid > 0x7fffffff) return GRL_ERR_BAD_ARG;
    const size_t lds = fast_lds_bytes(p, KDMA);
    if (lds > 160 * 1024) return GRL_ERR_UNSUPPORTED;
    auto kfn = attn_fast_kernel<KDMA>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kfn, dim3((int)grid), dim3(FW * 64), lds, st, p);
    GRL_CHECK_LAUNCH();
    return 0;
}

int launch_fast(const GrlAttnArgs& p, hipStream_t st) {
    // K and V by DMA into double buffers and V through the hardware transpose read when two workgroups still fit a CU
    // (window, window->anchor), else K by DMA + V through registers (anchor->window: its 27 KB table slice + four 16 KB
    // buffers would leave one workgroup per CU)
    return fast_lds_bytes(p, 2) <= 80 * 1024 ? launch_fast_v<2>(p, st) : launch_fast_v<1>(p, st);
}

}  // namespace

// csrc/attention_rows.hip: hand-scheduled row-streaming kernel for the 32-aligned geometries
bool grl_attn_rows_supported(const GrlAttnArgs& p);
int grl_attn_rows_launch(const GrlAttnArgs& p, hipStream_t st);
// csrc/attention_pipe.hip (round 5): the same geometries at head_dim <= 30 with the key-row loop software-pipelined inside each wave
bool grl_attn_pipe_supported(const GrlAttnArgs& p);
int grl_attn_pipe_launch(const GrlAttnArgs& p, hipStream_t st);

extern "C" int grl_attention_rows_geometry_ok(const GrlAttnArgs* args) { return grl_attn_rows_supported(*args) ? 1 : 0; }

extern "C" int grl_attention_fwd(void* stream, const GrlAttnArgs* args) {
    const GrlAttnArgs& p = *args;
    const int Nq = p.q.wh * p.q.ww, Nk = p.k.wh * p.k.ww;
    if (Nq <= 0 || Nk <= 0 || p.B <= 0 || p.nh <= 0) return GRL_ERR_BAD_ARG;
    if (p.q.Himg != p.nwy * p.q.wh || p.q.Wimg != p.nwx * p.q.ww) return GRL_ERR_BAD_ARG;
    if (p.k.Himg != p.nwy * p.k.wh || p.k.Wimg != p.nwx * p.k.ww) return GRL_ERR_BAD_ARG;
    if (p.trows != (p.q.wh + p.k.wh - 1) * (p.q.ww + p.k.ww - 1) || p.tstride < p.trows || (p.tstride & 3)) return GRL_ERR_BAD_ARG;
    if (p.head_dim > 32 || p.ones_col >= 32) return GRL_ERR_BAD_ARG;
    if (p.out_dtype != GRL_DT_F32 && p.out_dtype != GRL_DT_F16) return GRL_ERR_BAD_ARG;
    if ((p.q.ld % 8) || (p.k.ld % 8) || (p.v.ld % 8) || (p.o.ld % 4) || (p.q.col0 % 8) || (p.k.col0 % 8) ||
        (p.v.col0 % 8) || (p.o.col0 % 4) || (p.q.hstride % 8) || (p.k.hstride % 8) || (p.v.hstride % 8) || (p.o.hstride % 4))
        return GRL_ERR_BAD_ARG;
    if (p.lse != nullptr && p.lse_stride <= 0) return GRL_ERR_BAD_ARG;
    if (p.q.transposed != p.k.transposed || p.q.transposed != p.v.transposed || p.q.transposed != p.o.transposed) return GRL_ERR_BAD_ARG;
    if (p.out_dtype == GRL_DT_F16 && ((p.o.ld % 8) || (p.o.col0 % 8) || (p.o.hstride % 8))) return GRL_ERR_BAD_ARG;   // 16-B stores
    hipStream_t st = (hipStream_t)stream;
    // fast path: 32-aligned geometry with the ones column and the lazy running offset (needs the spare head-dim slot 31:
    // K carries 1.0 there, `k_one31`)
    const bool split = p.q_lo != nullptr || p.k_lo != nullptr || p.v_lo != nullptr;
    if (p.o_lo != nullptr && p.out_dtype != GRL_DT_F16) return GRL_ERR_BAD_ARG;
    const bool lazy_ok = !split && p.k_one31 && p.head_dim <= 30 && p.ones_col != 31 && p.lazy_floor != nullptr;
    static const int rows_off = getenv("GRL_ATTN_ROWS") ? atoi(getenv("GRL_ATTN_ROWS")) == 0 : 0;   // 0: round-2 fast kernels (A/B)
    // head_dim 32 (GRL-Small): no spare slot; the row-streaming kernel carries offset and denominator on the VALU instead
    const bool d32_ok = !split && p.head_dim == 32 && p.ones_col < 0 && p.lazy_floor != nullptr;
    // opt-in (GRL_ATTN_PIPE=1): measured in the whole network it does not beat the row-streaming kernel yet
    static const int pipe_on = getenv("GRL_ATTN_PIPE") ? atoi(getenv("GRL_ATTN_PIPE")) != 0 : 0;
    if (lazy_ok && pipe_on && !rows_off && !getenv("GRL_ATTN_GENERIC") && grl_attn_pipe_supported(p)) return grl_attn_pipe_launch(p, st);
    if ((lazy_ok || d32_ok) && !rows_off && !getenv("GRL_ATTN_GENERIC") && grl_attn_rows_supported(p)) return grl_attn_rows_launch(p, st);
    if (lazy_ok && !p.q.transposed && p.ones_col >= 0 && (p.q.ww % 32) == 0 && (p.k.ww % 32) == 0 && (p.q.wh % 2) == 0 &&
        (p.k.wh % 8) == 0 && fast_lds_bytes(p, 1) <= 160 * 1024 && !getenv("GRL_ATTN_GENERIC"))
        return launch_fast(p, st);
    // query blocks of equal size: 288 anchors are 2 x 3 waves (192 + 96 queries), not 4 waves + a workgroup that stages every
    // key chunk for 32 queries
    const int units = (Nq + QT * 32 - 1) / (QT * 32);
    const int waves = (units + (units + 3) / 4 - 1) / ((units + 3) / 4);
    const int qblk = waves * QT * 32;
    const int nqs = (Nq + qblk - 1) / qblk;
    const int64_t grid = (int64_t)nqs * p.nh * p.nwx * p.nwy * p.B;
    if (grid > 0x7fffffff) return GRL_ERR_BAD_ARG;
    const size_t kcap = (size_t)min(KC, (Nk + 31) & ~31);
    const size_t lds = (((size_t)p.trows * 4 + 15) & ~(size_t)15) + 2 * kcap * 64 + kcap * 4 + kcap + (split ? 2 * kcap * 64 : 0);
    if (lds > 160 * 1024) return GRL_ERR_UNSUPPORTED;
    if (split)
        return p.ones_col >= 0 ? launch_kw<true, true>(p, (int)grid, waves * 64, lds, st) : launch_kw<false, true>(p, (int)grid, waves * 64, lds, st);
    return p.ones_col >= 0 ? launch_kw<true, false>(p, (int)grid, waves * 64, lds, st) : launch_kw<false, false>(p, (int)grid, waves * 64, lds, st);
}
