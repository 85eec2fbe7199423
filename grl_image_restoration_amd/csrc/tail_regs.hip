// Block tail with ALL weights stationary in registers (gfx950, round 4): the GRL-Base path of grl_block_tail_fwd.
//
//     r1  = x + res_scale * LayerNorm1(att . Wp^T + bp) + cab * gate[image]
//     out = r1 + res_scale * LayerNorm2(fc2(GELU(fc1(r1))))
// (MixedAttention.proj + norm1 + residual + CAB gate + Mlp + norm2 + residual: mixed_attn_block_efficient.py:379,543-556,
// swin_v1_block.py:37-43; contract in include/grl_hip.h.)  Shape: Cpad 192, Hpad 384, M and rows_per_image multiples of 32;
// weights from GrlTailArgs.rblob (ops.pack_tail_regs).
//
// Why: mlp_kernel<6,8,true> streams the 366 KB of proj + fc1 + fc2 weights through LDS for every 128 tokens -- 18 chunk
// barriers per tile, every wave re-reading the whole stream for its 16 tokens: 254 us per 4 tiles, 2.4 TB/s.  Counting
// instructions (s_memtime probes, round 4) both that kernel and the first register-resident versions are bound by instruction
// ISSUE (~4.8 cycles per instruction and SIMD): 231 wave-instructions per token for the streaming kernel, 506 for a 12-wave
// 16x16x32 version (a lane = 4 channels of 2 tokens: everything per-token is repeated 48 times; correct, 376 us -- recorded
// under tools/attn_asm/dead_ends/).  So the layout is chosen for FEW INSTRUCTIONS:
//   * 32x32x16 MFMAs: a lane = 16 channels of ONE token -- per-token work (norm statistics, row addresses) once per lane pair,
//     epilogues amortised over 16 values, half the B-operand LDS reads of 16x16x32;
//   * 8 waves.  The 24 row tiles of the three matrices (6 x proj, 12 x fc1, 6 x fc2 with K = 384) are A fragments in registers:
//     wave w < 6 holds proj tile w, fc1 tile w, fc2 tile w (48 + 48 + 96 VGPRs: the channels 32 w .. 32 w + 31 of r1 and out
//     belong to ONE wave and one lane per token, so the fp32 r1 simply overwrites the lane's x values in the LDS tile until
//     the output epilogue needs it); waves 6, 7 hold fc1 tiles 6..8 / 9..11 (144 VGPRs);
//   * NO weights in LDS: it holds the activations of a 32-token tile -- att, x (fp32) and cab, double buffered, brought in one
//     tile ahead by the two fc1-only waves through their spare registers (see load_next); r1 and the hidden activations (fp16)
//     are written by their producers in the natural K order of the consumers' B fragments;
//   * a wave sees 32 of a token's 192 channels: the LayerNorms combine per-wave (mean, M2) pairs through LDS (Chan);
//   * 192 of 256 VGPRs hold weights.  Everything else is written to keep few values alive: lane-derived offsets are recomputed
//     at the head of every phase from an opaque lane id, epilogues run four channels at a time between scheduling barriers,
//     k loops are software-pipelined by hand one step deep (left alone the compiler alternates read and MFMA batches with
//     lgkmcnt(0) in between, csrc/qkv_anchor.hip).
#include "common.h"
#include "grl_hip_internal.h"
#include <stdlib.h>

namespace {

constexpr int TR_T = 32, TR_W = 8, TR_THREADS = TR_W * 64;
constexpr int TR_CP = 192, TR_HP = 384, TR_KS1 = TR_CP / 16, TR_KS2 = TR_HP / 16;   // k-steps of 16
constexpr int TR_AROW = TR_CP * 2 + 16;            // 400: fp16 row of the att / cab / r1 tiles (16 B pad: conflict-free ds_read_b128)
constexpr int TR_HROW = TR_HP * 2 + 16;            // 784: fp16 row of the hidden tile
constexpr int TR_XROW = TR_CP * 4 + 16;            // 784: fp32 row of the x tile
constexpr int TR_ASEG = TR_AROW / 16, TR_XSEG = TR_XROW / 16;
constexpr int TR_APIECES = (TR_T * TR_AROW + 1023) / 1024;   // 13 DMA pieces (1 KiB) per att / cab tile
constexpr int TR_XPIECES = (TR_T * TR_XROW + 1023) / 1024;   // 25 per x tile
constexpr int TR_ABUF = TR_APIECES * 1024, TR_XBUF = TR_XPIECES * 1024;
constexpr int TR_OFF_ATT = 0;                                 // [2][TR_ABUF]
constexpr int TR_OFF_CAB = TR_OFF_ATT + 2 * TR_ABUF;          // [2][TR_ABUF]
constexpr int TR_OFF_X = TR_OFF_CAB + 2 * TR_ABUF;            // [2][TR_XBUF]
constexpr int TR_OFF_R1 = TR_OFF_X + 2 * TR_XBUF;
constexpr int TR_OFF_H = TR_OFF_R1 + TR_T * TR_AROW;
constexpr int TR_OFF_ST = TR_OFF_H + TR_T * TR_HROW;          // [2 norms][6 waves][32 tokens] float2
constexpr int TR_OFF_VEC = TR_OFF_ST + 2 * 6 * TR_T * 8;
constexpr int TR_VECF = 7 * TR_CP + TR_HP;                    // pb n1g n1b b2 n2g n2b gate | b1
constexpr int TR_LDS = TR_OFF_VEC + TR_VECF * 4;              // 152 KB
constexpr int TR_FRAGS = 48;                                  // A fragments (1 KiB each) per wave in the blob
static_assert(TR_LDS <= 160 * 1024, "LDS budget");

#ifdef TR_DEBUG   // timing probes (s_memtime ticks): [wave][region], summed over workgroups and tiles
__device__ unsigned long long tr_dbg[64];
#define TR_TIME(x) const long long x = __builtin_amdgcn_s_memtime()
#define TR_ADD(i, v) tr_acc[i] += (unsigned long long)(v)
#else
#define TR_TIME(x)
#define TR_ADD(i, v)
#endif

#define TR_SB() __builtin_amdgcn_sched_barrier(0)

// The lane id, recomputed (2 VALU) at the head of every phase instead of living in a VGPR -- together with everything derived
// from it -- across the whole tile loop (volatile asm: a builtin would be hoisted out of the loop again).
__device__ __forceinline__ int tr_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

__device__ __forceinline__ void tr_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS only: the DMA / store counters are handled where they matter)
    __builtin_amdgcn_s_barrier();
}

// acc += W_tile . B^T over KS k-steps: A fragments in registers, B fragments (32 tokens x 16 channels) from the LDS tile whose
// row `brow` (this lane's token, + 16 B for the upper half-wave) is given.  Pipelined by hand, see the file header.
template <int KS>
__device__ __forceinline__ void tr_gemm(const f16x8* A, const char* brow, f32x16& acc) {
    f16x8 b0 = *(const f16x8*)brow, b1 = *(const f16x8*)(brow + 32);   // two steps ahead: one accumulator chain cannot hide an LDS round trip
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 nb = b0;
        if (s + 2 < KS) nb = *(const f16x8*)(brow + 32 * (s + 2));
        TR_SB();
        acc = mfma32_f16(A[s], b0, acc);
        TR_SB();
        b0 = b1; b1 = nb;
    }
}

// accumulator <- per-channel fp32 vector in LDS (this lane's channels c0 + 8 g + [0..3])
__device__ __forceinline__ void tr_bias(f32x16& acc, const float* v, int c0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = *(const float4*)(v + c0 + 8 * g);
        acc[4 * g] = t.x; acc[4 * g + 1] = t.y; acc[4 * g + 2] = t.z; acc[4 * g + 3] = t.w;
    }
}

// per-wave (mean, M2) of this lane's token over the wave's real channels -> st_wave[token].  MASKED: the wave holds pad channels
// (wave 5: channels 160 + ...; pad values are exact zeros, so only M2 needs the mask: bit r of realmask = register r is real)
template <bool MASKED>
__device__ __forceinline__ void tr_ln_local(const f32x16& v, uint32_t realmask, float inv_n, float2* st_wave, int j, int half) {
    float s0 = (v[0] + v[1]) + (v[2] + v[3]), s1 = (v[4] + v[5]) + (v[6] + v[7]);
    float s2 = (v[8] + v[9]) + (v[10] + v[11]), s3 = (v[12] + v[13]) + (v[14] + v[15]);
    const float mean_w = sum_halves((s0 + s1) + (s2 + s3)) * inv_n;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float d0 = v[r] - mean_w, d1 = v[r + 1] - mean_w;
        if (!MASKED || ((realmask >> r) & 1)) q0 = fmaf(d0, d0, q0);
        if (!MASKED || ((realmask >> (r + 1)) & 1)) q1 = fmaf(d1, d1, q1);
    }
    const float q = sum_halves(q0 + q1);
    if (half == 0) st_wave[j] = float2{mean_w, q};
}

// Chan's combination of the six per-wave pairs of token j (waves 0..4 hold 32 real channels, wave 5 n5 = n_real - 160)
__device__ __forceinline__ void tr_ln_combine(const float2* st, int j, float n5, float inv_n, float eps, float& mean, float& rstd) {
    const float2 e0 = st[j], e1 = st[TR_T + j], e2 = st[2 * TR_T + j], e3 = st[3 * TR_T + j], e4 = st[4 * TR_T + j], e5 = st[5 * TR_T + j];
    mean = (32.f * ((e0.x + e1.x) + (e2.x + e3.x) + e4.x) + n5 * e5.x) * inv_n;
    const float d0 = e0.x - mean, d1 = e1.x - mean, d2 = e2.x - mean, d3 = e3.x - mean, d4 = e4.x - mean, d5 = e5.x - mean;
    const float m2 = ((e0.y + e1.y) + (e2.y + e3.y) + (e4.y + e5.y)) + 32.f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + d4 * d4) + n5 * d5 * d5;
    rstd = rsqrtf(m2 * inv_n + eps);
}

__global__ __launch_bounds__(TR_THREADS) void tail_regs_kernel(GrlTailArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
#ifdef TR_DEBUG
    unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool cw = wave < 6;                      // channel wave: proj / fc2 tile `wave`, fc1 tile `wave`; else fc1 tiles 6 + 3 (wave - 6) ..
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int ntiles = p.M / TR_T;
    if ((int)blockIdx.x >= ntiles) return;

    // ---- once per launch: A fragments -> registers, vectors -> LDS ----
    f16x8 Wt[TR_FRAGS];   // (loaded at the head of each role's loop below: a common load is a common live range across the branch)
    float* vec = (float*)(smem + TR_OFF_VEC);
    {
        const int tid = 64 * wave + tr_lane();
        for (int i = tid; i < 6 * TR_CP; i += TR_THREADS) {
            const int k = i / TR_CP, c = i - k * TR_CP;
            const float* src = k == 0 ? p.pb : k == 1 ? p.n1_g : k == 2 ? p.n1_b : k == 3 ? p.b2 : k == 4 ? p.n2_g : p.n2_b;
            vec[i] = c < p.n_real ? src[c] : 0.f;  // pad channels: zero bias / affine, so they stay exactly 0 all the way
        }
        for (int i = tid; i < TR_HP; i += TR_THREADS)
            vec[7 * TR_CP + i] = ((const float*)((const char*)p.rblob + (size_t)TR_W * TR_FRAGS * 1024))[i];
    }
    float* gate_l = vec + 6 * TR_CP;
    int gate_img = -1;

    // tile DMA: att and cab pieces are [32 rows][25 x 16 B] images (24 real segments + the pad, which repeats the last one),
    // the x piece [32 rows][49 x 16 B]; 51 pieces of 1 KiB per tile, dealt round-robin over the waves.  Buffer loads: the tensor
    // base sits in a descriptor, the tile in the scalar offset, and a lane's 32-bit offset inside the tile costs ~8 instructions
    // (with flat 64-bit addresses a piece cost 25, i.e. ~300 cycles of a wave's issue time: 3 k cycles per tile).
    typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
    auto srd = [](const void* ptr) {   // raw buffer descriptor: base, stride 0, all of memory, gfx950 data format word
        const uint64_t a = (uint64_t)ptr;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        r[2] = 0xffffffffu;
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 att_srd = srd(p.att), cab_srd = srd(p.cab), x_srd = srd(p.x);
    const uint32_t att_rb = (uint32_t)p.ldatt * 2u, cab_rb = (uint32_t)p.ldcab * 2u, x_rb = (uint32_t)p.ldx * 4u;   // bytes per row
    // s_nop 4: SGPR operands may come straight from SALU / readfirstlane (5 wait states before a VMEM instruction reads them)
#define TR_DMA(m0v, voff, rsrc, soff) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
    auto fetch = [&](int tile, int buf) {
        const int lane = tr_lane();
        for (int q0 = (wave + 2) & 7; q0 < 2 * TR_APIECES + TR_XPIECES; q0 += TR_W) {   // (wave-uniform)
            if (q0 < 2 * TR_APIECES) {
                const bool is_cab = q0 >= TR_APIECES;
                const int q = is_cab ? q0 - TR_APIECES : q0;
                const int idx = q * 64 + lane;
                int row = idx / TR_ASEG, seg = idx - row * TR_ASEG;
                seg = min(seg, TR_ASEG - 2);
                row = min(row, TR_T - 1);
                const uint32_t rb = is_cab ? cab_rb : att_rb;
                const uint32_t voff = (uint32_t)row * rb + (uint32_t)seg * 16u;
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (is_cab ? TR_OFF_CAB : TR_OFF_ATT) + buf * TR_ABUF + q * 1024);
                const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)tile * (TR_T * rb));
                if (is_cab) TR_DMA(m0v, voff, cab_srd, soff);
                else TR_DMA(m0v, voff, att_srd, soff);
            } else {
                const int q = q0 - 2 * TR_APIECES;
                const int idx = q * 64 + lane;
                int row = idx / TR_XSEG, seg = idx - row * TR_XSEG;
                seg = min(seg, TR_XSEG - 2);
                row = min(row, TR_T - 1);
                const uint32_t voff = (uint32_t)row * x_rb + (uint32_t)seg * 16u;
                TR_DMA(__builtin_amdgcn_readfirstlane(lds0 + TR_OFF_X + buf * TR_XBUF + q * 1024), voff, x_srd, __builtin_amdgcn_readfirstlane((uint32_t)tile * (TR_T * x_rb)));
            }
        }
    };
    fetch(blockIdx.x, 0);
    // After the first tile the two fc1-only waves bring the next tile in through REGISTERS: 51 LDS-DMA instructions cost every wave
    // ~2.3 k cycles of issue time at the head of a tile (an LDS-DMA of a wave does not overlap with its next one: measured ~300
    // cycles apiece whoever issues them), while plain loads are all in flight at once.  Waves 6, 7 are idle during the projection
    // and the first norm: each issues 12 loads of 16 B per lane after B0 (att, cab) and, once those are copied to the LDS buffers,
    // 12 more after B1 (x), copied after B2 -- 48 registers, which these waves can spare.  Inline asm as in qkv_split_kernel (the compiler's
    // wait-count bookkeeping would sit the loads out right after issuing them).
    // Only the real 16-B segments travel, in groups of 192 = 3 wave-wide loads: 8 rows of att / cab (24 segments each) or 4 rows of
    // x (48): a lane's offsets inside a group are three constants per tensor kind (recomputed per call: ~30 instructions), the
    // group itself moves through the scalar offset -- 3 instructions per load instead of ~15 (the loader waves issue one
    // instruction per ~10 cycles, and everybody waits for them at B1 / B2).  Round 0: wave 6 loads att, wave 7 cab (4 groups
    // of 8 rows); round 1: x, wave 6 rows 0..15, wave 7 rows 16..31 (4 groups of 4 rows).
    f32x4 nb[12];
    auto lane_maps = [&](int segs, uint32_t rb, uint32_t pitch, uint32_t (&vo)[3], uint32_t (&lo)[3]) {
        const int lane = tr_lane();
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int idx = 64 * m + lane;
            const int row = segs == 24 ? idx / 24 : idx / 48, seg = idx - segs * row;
            vo[m] = (uint32_t)row * rb + (uint32_t)seg * 16u;
            lo[m] = (uint32_t)row * pitch + (uint32_t)seg * 16u;
        }
    };
#define TR_LOAD(dst, voff, rsrc, soff) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
    auto load_next = [&](int tile, int round) {       // waves 6, 7 only
        uint32_t vo[3], lo[3];
        if (round == 0) {
            const uint32_t rb = wave == 6 ? att_rb : cab_rb;
            lane_maps(24, rb, TR_AROW, vo, lo);
            const uint32_t s0 = __builtin_amdgcn_readfirstlane((uint32_t)tile * (TR_T * rb));
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const uint32_t soff = s0 + (uint32_t)(k / 3) * 8u * rb;
                if (wave == 6) TR_LOAD(nb[k], vo[k % 3], att_srd, soff);
                else TR_LOAD(nb[k], vo[k % 3], cab_srd, soff);
            }
        } else {
            lane_maps(48, x_rb, TR_XROW, vo, lo);
            const uint32_t s0 = __builtin_amdgcn_readfirstlane((uint32_t)tile * (TR_T * x_rb) + (uint32_t)(wave - 6) * 16u * x_rb);
#pragma unroll
            for (int k = 0; k < 12; ++k) TR_LOAD(nb[k], vo[k % 3], x_srd, s0 + (uint32_t)(k / 3) * 4u * x_rb);
        }
    };
    auto store_next = [&](int buf, int round) {       // ... after the data has landed
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]), "+v"(nb[4]), "+v"(nb[5]), "+v"(nb[6]), "+v"(nb[7]), "+v"(nb[8]),
                       "+v"(nb[9]), "+v"(nb[10]), "+v"(nb[11])
                     :: "memory");
        uint32_t vo[3], lo[3];
        if (round == 0) {
            lane_maps(24, 0u, TR_AROW, vo, lo);
            char* base = smem + (wave == 6 ? TR_OFF_ATT : TR_OFF_CAB) + buf * TR_ABUF;
#pragma unroll
            for (int k = 0; k < 12; ++k) *(f32x4*)(base + (k / 3) * 8 * TR_AROW + lo[k % 3]) = nb[k];
        } else {
            lane_maps(48, 0u, TR_XROW, vo, lo);
            char* base = smem + TR_OFF_X + buf * TR_XBUF + (wave - 6) * 16 * TR_XROW;
#pragma unroll
            for (int k = 0; k < 12; ++k) *(f32x4*)(base + (k / 3) * 4 * TR_XROW + lo[k % 3]) = nb[k];
        }
    };

    const float n5 = (float)(p.n_real - 160);
    const float inv_n = 1.0f / (float)p.n_real, inv_nw = wave < 5 ? 1.0f / 32.f : 1.0f / n5;
    float2* st1 = (float2*)(smem + TR_OFF_ST);
    float2* st2 = st1 + 6 * TR_T;
    // per phase: j = token (MFMA column), half = upper / lower half-wave; ch0 = this lane's first channel (accumulator register r
    // <-> channel ch0 + (r & 3) + 8 (r >> 2))
#define TR_LANE() const int ln_ = tr_lane(), j = ln_ & 31, half = ln_ >> 5, ch0 = 32 * wave + 4 * half; (void)j; (void)ch0
    auto realmask_of = [&](int ch0) {
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= (uint32_t)(ch0 + (r & 3) + 8 * (r >> 2) < p.n_real) << r;
        return m;
    };

    // SE gate row of the image (changes once per rows_per_image / 32 tiles of a workgroup): every wave runs this with the same result
    auto stage_gate = [&](int img) {
        if (img != gate_img) {
            tr_barrier();
            const int i = 64 * wave + tr_lane();
            if (i < TR_CP) gate_l[i] = i < p.n_real ? p.gate[(int64_t)img * TR_CP + i] : 0.f;
            gate_img = img;
        }
    };
    auto fc1_tile = [&](const f16x8* A, int ht, int j, int half) {   // hidden channels 32 ht .. of the tile's 32 tokens
        f32x16 h;
        tr_bias(h, vec + 7 * TR_CP, 32 * ht + 4 * half);
        tr_gemm<TR_KS1>(A, smem + TR_OFF_R1 + j * TR_AROW + 16 * half, h);
        char* hr = smem + TR_OFF_H + j * TR_HROW + (32 * ht + 4 * half) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x2v a = gelu_erf2(f32x2v{h[4 * g], h[4 * g + 1]}), b = gelu_erf2(f32x2v{h[4 * g + 2], h[4 * g + 3]});
            uint2 o;
            o.x = pack_f16(a[0], a[1]);
            o.y = pack_f16(b[0], b[1]);
            *(uint2*)(hr + 16 * g) = o;
            TR_SB();   // (one group of four at a time: interleaving all sixteen GELUs costs ~40 registers)
        }
    };

    // The fc1-only waves have three fc1 tiles where a channel wave has one, and a tile is mostly GELU (16 per lane: ~150 of its ~190
    // instructions).  So they only run the MFMAs and park the fp32 PRE-activations (bias included) of their 2 x 96 hidden channels in
    // the att / cab buffers of the current tile -- dead since the first norm -- and after B3a all eight waves apply the GELU: a
    // fc1-only wave to the first 48 channels of its own region (24 values per lane), channel wave w to 16 of the remaining 2 x 48
    // (8 per lane, after its first fc2 half).  Phase lengths 2.7 k + 2.1 k cycles instead of 2.7 k + 3.8 k.
    auto fc1_pre = [&](const f16x8* A, int t, char* region, int j, int half) {   // hidden channels 192 + 96 R + 32 t ..: pre-activations
        f32x16 h;
        tr_bias(h, vec + 7 * TR_CP, 192 + 96 * (wave - 6) + 32 * t + 4 * half);
        tr_gemm<TR_KS1>(A, smem + TR_OFF_R1 + j * TR_AROW + 16 * half, h);
        char* dst = region + j * TR_AROW + (32 * t + 4 * half) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(dst + 32 * g) = float4{h[4 * g], h[4 * g + 1], h[4 * g + 2], h[4 * g + 3]};
    };
    // GELU of N4 x 4 consecutive pre-activations of token j starting at channel c0 of region R -> hidden tile (fp16)
    // (`free_sched`: the fc1-only waves have registers to spare -- their 24 GELUs may interleave, which hides the latency of the
    // reciprocals / exponentials that two waves per SIMD cannot; the channel waves keep one group of four at a time)
    auto gelu_slice = [&](int R, int c0, int n4, int j, int buf, bool free_sched) {
        const char* src = smem + (R ? TR_OFF_CAB : TR_OFF_ATT) + buf * TR_ABUF + j * TR_AROW + c0 * 4;
        char* dst = smem + TR_OFF_H + j * TR_HROW + (192 + 96 * R + c0) * 2;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q < n4) {
                const float4 v = *(const float4*)(src + 16 * q);
                const f32x2v a = gelu_erf2(f32x2v{v.x, v.y}), b = gelu_erf2(f32x2v{v.z, v.w});
                uint2 o;
                o.x = pack_f16(a[0], a[1]);
                o.y = pack_f16(b[0], b[1]);
                *(uint2*)(dst + 8 * q) = o;
                if (!free_sched) TR_SB();
            }
        }
    };

    // The channel waves and the fc1-only waves run SEPARATE tile loops with the same barrier sequence (gate, B0, B1, B2, B3a, B3b,
    // B4): the register allocator works per program point, so in one shared loop the 48 fragment registers the fc1-only waves do
    // not use and the 48 registers of data their loader keeps in flight both counted against the channel waves' code.
    if (cw) {
        {
            const char* src = (const char*)p.rblob + ((size_t)wave * TR_FRAGS) * 1024 + tr_lane() * 16;
#pragma unroll
            for (int f = 0; f < TR_FRAGS; ++f) Wt[f] = *(const f16x8*)(src + (size_t)f * 1024);
        }
        int it = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
            TR_TIME(t0);
            const int buf = it & 1;
            const int64_t m0 = (int64_t)tile * TR_T;
            stage_gate((int)(m0 / p.rows_per_image));
            if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first tile's DMA pieces
            tr_barrier();                                  // B0: tile `it` is in its buffers; the other buffers are free
            TR_TIME(t1);
            TR_ADD(0, t1 - t0);
            f32x16 acc;
            {
                // ---- P1: projection + per-wave norm statistics ----
                TR_LANE();
                tr_bias(acc, vec, ch0);
                tr_gemm<TR_KS1>(Wt, smem + TR_OFF_ATT + buf * TR_ABUF + j * TR_AROW + 16 * half, acc);
                if (wave < 5) tr_ln_local<false>(acc, 0, inv_nw, st1 + wave * TR_T, j, half);
                else tr_ln_local<true>(acc, realmask_of(ch0), inv_nw, st1 + wave * TR_T, j, half);
            }
            TR_TIME(t2);
            TR_ADD(1, t2 - t1);
            tr_barrier();                                  // B1
            {
                TR_LANE();
                float mean, rstd;
                tr_ln_combine(st1, j, n5, inv_n, p.ln_eps, mean, rstd);
                rstd *= p.res_scale;
                char* xrow = smem + TR_OFF_X + buf * TR_XBUF + j * TR_XROW + ch0 * 4;
                const char* crow = smem + TR_OFF_CAB + buf * TR_ABUF + j * TR_AROW + ch0 * 2;
                char* r1row = smem + TR_OFF_R1 + j * TR_AROW + ch0 * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // channels ch0 + 8 g + [0..3]
                    const float4 xs = *(const float4*)(xrow + 32 * g);
                    const uint2 cb = *(const uint2*)(crow + 16 * g);
                    const float4 g1 = *(const float4*)(vec + TR_CP + ch0 + 8 * g), b1n = *(const float4*)(vec + 2 * TR_CP + ch0 + 8 * g);
                    const float4 gt = *(const float4*)(gate_l + ch0 + 8 * g);
                    const f16x2 c01 = __builtin_bit_cast(f16x2, cb.x), c23 = __builtin_bit_cast(f16x2, cb.y);
                    float4 r1;   // x + res_scale ((v - mean) rstd g + b) + cab gate
                    r1.x = fmaf((float)c01[0], gt.x, fmaf(p.res_scale, b1n.x, fmaf((acc[4 * g] - mean) * rstd, g1.x, xs.x)));
                    r1.y = fmaf((float)c01[1], gt.y, fmaf(p.res_scale, b1n.y, fmaf((acc[4 * g + 1] - mean) * rstd, g1.y, xs.y)));
                    r1.z = fmaf((float)c23[0], gt.z, fmaf(p.res_scale, b1n.z, fmaf((acc[4 * g + 2] - mean) * rstd, g1.z, xs.z)));
                    r1.w = fmaf((float)c23[1], gt.w, fmaf(p.res_scale, b1n.w, fmaf((acc[4 * g + 3] - mean) * rstd, g1.w, xs.w)));
                    *(float4*)(xrow + 32 * g) = r1;              // fp32 r1 in place of x (this lane's own 16 bytes), for the output epilogue
                    uint2 o;
                    o.x = pack_f16(r1.x, r1.y);
                    o.y = pack_f16(r1.z, r1.w);
                    *(uint2*)(r1row + 16 * g) = o;
                    TR_SB();   // (bounds the live range of the per-channel vectors: the register budget is 256 - 192)
                }
            }
            TR_TIME(t3);
            TR_ADD(2, t3 - t2);
            tr_barrier();                                  // B2: r1 tile complete
            {
                // ---- P2: fc1 + GELU -> hidden tile; then fc2, in two halves around B3b: the first on the hidden channels 0..191 the
                // channel waves produced themselves, while the fc1-only waves' channels get their GELU (see fc1_pre) ----
                TR_LANE();
                fc1_tile(Wt + 12, wave, j, half);
                TR_TIME(t4);
                TR_ADD(3, t4 - t3);
                tr_barrier();                              // B3a: hidden channels 0..191 complete
                tr_bias(acc, vec + 3 * TR_CP, ch0);
                const char* hb = smem + TR_OFF_H + j * TR_HROW + 16 * half;
                tr_gemm<TR_KS2 / 2>(Wt + 24, hb, acc);
                gelu_slice(wave / 3, 48 + 16 * (wave % 3) + 8 * half, 2, j, buf, false);   // this wave's share of the fc1-only waves' GELUs
                TR_TIME(t4b);
                TR_ADD(7, t4b - t4);
                tr_barrier();                              // B3b: hidden tile complete
                tr_gemm<TR_KS2 / 2>(Wt + 36, hb + 32 * (TR_KS2 / 2), acc);
                if (wave < 5) tr_ln_local<false>(acc, 0, inv_nw, st2 + wave * TR_T, j, half);
                else tr_ln_local<true>(acc, realmask_of(ch0), inv_nw, st2 + wave * TR_T, j, half);
            }
            TR_TIME(t5);
            tr_barrier();                                  // B4
            TR_TIME(t6);
            TR_ADD(5, t6 - t5);
            {
                TR_LANE();
                float mean, rstd;
                tr_ln_combine(st2, j, n5, inv_n, p.ln_eps, mean, rstd);
                rstd *= p.res_scale;
                float* orow = p.out + (m0 + j) * p.ldo + ch0;
                const char* xrow = smem + TR_OFF_X + buf * TR_XBUF + j * TR_XROW + ch0 * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 g2 = *(const float4*)(vec + 4 * TR_CP + ch0 + 8 * g), b2n = *(const float4*)(vec + 5 * TR_CP + ch0 + 8 * g);
                    const float4 r1 = *(const float4*)(xrow + 32 * g);
                    float4 o;   // r1 + res_scale ((v - mean) rstd g + b)
                    o.x = fmaf(p.res_scale, b2n.x, fmaf((acc[4 * g] - mean) * rstd, g2.x, r1.x));
                    o.y = fmaf(p.res_scale, b2n.y, fmaf((acc[4 * g + 1] - mean) * rstd, g2.y, r1.y));
                    o.z = fmaf(p.res_scale, b2n.z, fmaf((acc[4 * g + 2] - mean) * rstd, g2.z, r1.z));
                    o.w = fmaf(p.res_scale, b2n.w, fmaf((acc[4 * g + 3] - mean) * rstd, g2.w, r1.w));
                    *(float4*)(orow + 8 * g) = o;
                    TR_SB();
                }
            }
            TR_TIME(t7);
            TR_ADD(6, t7 - t6);
        }
    } else {
        {
            const char* src = (const char*)p.rblob + ((size_t)wave * TR_FRAGS) * 1024 + tr_lane() * 16;
#pragma unroll
            for (int f = 0; f < 36; ++f) Wt[f] = *(const f16x8*)(src + (size_t)f * 1024);
        }
        int it = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
            TR_TIME(t0);
            const int buf = it & 1;
            stage_gate((int)(((int64_t)tile * TR_T) / p.rows_per_image));
            if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tr_barrier();                                  // B0
            TR_TIME(t1);
            TR_ADD(0, t1 - t0);
            const int next = tile + (int)gridDim.x;
            if (next < ntiles) load_next(next, 0);
            TR_TIME(t2);
            TR_ADD(1, t2 - t1);
            tr_barrier();                                  // B1
            if (next < ntiles) { store_next(buf ^ 1, 0); load_next(next, 1); }
            TR_TIME(t3a);
            TR_ADD(2, t3a - t2);
            tr_barrier();                                  // B2
            if (next < ntiles) store_next(buf ^ 1, 1);
            TR_TIME(t3);
            TR_LANE();
            char* region = smem + (wave == 6 ? TR_OFF_ATT : TR_OFF_CAB) + buf * TR_ABUF;   // dead since the first norm of this tile
            fc1_pre(Wt, 0, region, j, half);
            TR_SB();
            fc1_pre(Wt + 12, 1, region, j, half);
            TR_SB();
            fc1_pre(Wt + 24, 2, region, j, half);
            TR_TIME(t4);
            TR_ADD(3, t4 - t3);
            tr_barrier();                                  // B3a: hidden channels 0..191 and all pre-activations complete
            gelu_slice(wave - 6, 24 * half, 6, j, buf, true);
            TR_TIME(t4b);
            TR_ADD(7, t4b - t4);
            tr_barrier();                                  // B3b
            TR_TIME(t5);
            tr_barrier();                                  // B4
            TR_TIME(t6);
            TR_ADD(5, t6 - t5);
        }
    }
#ifdef TR_DEBUG
    if (tr_lane() == 0) for (int i = 0; i < 8; ++i) atomicAdd(&tr_dbg[8 * wave + i], tr_acc[i]);
#endif
}

}  // namespace

#ifdef TR_DEBUG
extern "C" int grl_tr_debug(unsigned long long* out64, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out64, HIP_SYMBOL(tr_dbg), sizeof(unsigned long long) * 64);
    if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(tr_dbg), z, sizeof(z)); }
    return 0;
}
#endif

extern "C" int64_t grl_tail_regs_blob_bytes(void) { return (int64_t)TR_W * TR_FRAGS * 1024 + TR_HP * 4; }

// register-resident path of grl_block_tail_fwd; GRL_ERR_UNSUPPORTED for every other shape (the caller falls back to the streaming kernel)
int grl_tail_regs_launch(const GrlTailArgs& a, hipStream_t st) {
    if (a.rblob == nullptr || a.Cpad != TR_CP || a.Hpad != TR_HP || (a.M % TR_T) || a.M <= 0 || (a.rows_per_image % TR_T) || a.n_real <= 160 ||
        a.n_real > TR_CP || (a.ldatt % 8) || (a.ldcab % 8) || (a.ldx % 4) || (a.ldo % 4) || ((uintptr_t)a.rblob & 15))
        return GRL_ERR_UNSUPPORTED;
    const int64_t rowb = a.ldx * 4 > a.ldatt * 2 ? (a.ldx * 4 > a.ldcab * 2 ? a.ldx * 4 : a.ldcab * 2) : (a.ldatt * 2 > a.ldcab * 2 ? a.ldatt * 2 : a.ldcab * 2);
    if ((int64_t)a.M * rowb >= (1ll << 32)) return GRL_ERR_UNSUPPORTED;   // the tile DMA addresses rows by 32-bit buffer offsets
    const int ntiles = a.M / TR_T;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipError_t e = hipFuncSetAttribute((const void*)tail_regs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TR_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tail_regs_kernel, dim3(grid), dim3(TR_THREADS), TR_LDS, st, a);
    GRL_CHECK_LAUNCH();
    return 0;
}
