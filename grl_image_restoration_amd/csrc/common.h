// Shared device helpers for the GRL gfx950 kernels.
// gfx950 / CDNA4 only: wave64, MFMA bf16 32x32x16 + 16x16x32, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grl_hip_internal.h"

typedef __bf16 bf16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

#define GRL_WAVE 64
#define LOG2E_F 1.4426950408889634f

// C/D fragment row of register r for the 32x32 MFMA shapes (cdna guide section 3):
//   col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int mfma32_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float bf16_to_f32(bf16 x) { return (float)x; }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2 v;
    v[0] = (bf16)lo;
    v[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// Operand type of every MFMA contraction (attention since round 2, token-wise linears, 3x3 convolutions): fp16, fp32
// accumulation.  With bf16 weights + activations the network output is 2.2e-3 away from the fp32 reference (weights alone
// 1.0e-3), with fp16 2e-4 (precision); the attention weights stay inside the fp16 range through the running
// softmax offset (csrc/attention_rows.hip).
typedef _Float16 f16;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(4 * sizeof(_Float16)))) _Float16 f16x4;
typedef f16 gemm_t;
typedef f16x8 gemm_x8;
typedef f16x4 gemm_x4;
__device__ __forceinline__ f32x4 mfma16_gemm(gemm_x8 a, gemm_x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
typedef __attribute__((__vector_size__(2 * sizeof(_Float16)))) _Float16 f16x2;
// fp32 -> fp16 with saturation: a residual-stream value beyond +-65504 becomes the largest finite half instead of inf
// (one v_med3_f32; NaN propagates).  Every fp32 -> fp16 staging / store point of the kernels goes through this.
__device__ __forceinline__ float sat16(float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
__device__ __forceinline__ f16 to_f16(float x) { return (f16)sat16(x); }
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    f16x2 v;
    v[0] = to_f16(lo);
    v[1] = to_f16(hi);
    return __builtin_bit_cast(uint32_t, v);
}
// ... without the saturation: for values bounded by construction (softmax weights <= 2^14)
__device__ __forceinline__ uint32_t pack_f16_raw(float lo, float hi) {
    f16x2 v;
    v[0] = (f16)lo;
    v[1] = (f16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ f32x16 mfma32_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// 16-bit storage kinds of the C ABI: GRL_DT_BF16 / GRL_DT_F16 (include/grl_hip.h)
__device__ __forceinline__ uint32_t pack16(float lo, float hi, int kind) {
    return kind == GRL_DT_BF16 ? pack_bf16(lo, hi) : pack_f16(lo, hi);
}

// ---- lane exchanges across the 16- and 32-lane boundaries (gfx950 v_permlane16_swap / v_permlane32_swap: one VALU instruction
// instead of a ds_bpermute round trip through the LDS pipe, which is what __shfl_xor compiles to).  Written as inline asm:
// with this toolchain's builtins the two results of a swap can come back as the same value (permlane32: tools/probes/
// permlane32_swap_codegen.hip, round 2; permlane16: r[0] + r[1] compiled as 2 * r[0], tools/ubench/permlane.hip).  The s_nops
// are the wait states the compiler places around the builtin form (VALU write -> swap, swap -> VALU / DPP read).
// swap32(a, b): lanes 32..63 of a <-> lanes 0..31 of b;  swap16(a, b): 16-lane rows 1, 3 of a <-> rows 0, 2 of b
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// value of the lane 32 apart
__device__ __forceinline__ float xhalf(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    swap32(a, b);   // a = {lo, lo}, b = {hi, hi}
    const bool lower = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 32u) == 0;
    return __builtin_bit_cast(float, lower ? b : a);
}
// v + (v of lane ^ 32), v + (v of lane ^ 16)
__device__ __forceinline__ float sum_halves(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    swap32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float sum_rows16(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    swap16(a, b);   // {r0, r0, r2, r2}, {r1, r1, r3, r3}
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// exact-GELU x*Phi(x) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. fp32
// round-off level; the reference uses torch's erf-GELU, swin_v1_block.py:23,38).  ~14 VALU ops with two
// quarter-rate transcendentals instead of libm erff's ~40: the fc1 epilogue is VALU-bound otherwise.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * LOG2E_F);
    const float erf_abs = 1.0f - poly * t * e;
    const float erf_x = copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf_x);
}

// d/dx of the exact GELU: Phi(x) + x phi(x), same erf approximation as gelu_erf (the backward of Mlp's activation, swin_v1_block.py:38)
__device__ __forceinline__ float gelu_grad(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * LOG2E_F);      // exp(-x^2 / 2)
    const float erf_x = copysignf(1.0f - poly * t * e, x);
    return fmaf(x * e, 0.39894228040143267794f, 0.5f * (1.0f + erf_x));
}

// Two GELUs at once on packed fp32 math (v_pk_mul/fma/add_f32: one issue slot for two elements -- the fused kernels are bound by
// instruction issue, not by the VALU pipes): same formula and accuracy as gelu_erf, 19 instructions per pair instead of 28.
typedef __attribute__((ext_vector_type(2))) float f32x2v;
__device__ __forceinline__ f32x2v gelu_erf2(f32x2v x) {
    const f32x2v ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2v z = ax * 0.70710678118654752440f;
    const f32x2v d = z * 0.3275911f + 1.0f;
    const f32x2v t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2v poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    const f32x2v a = z * z * (-LOG2E_F);
    const f32x2v e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2v erf_abs = 1.0f - poly * t * e;
    const f32x2v erf_x = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
    return (x * 0.5f) * (erf_x + 1.0f);
}

// Cross-lane moves inside a 16-lane row as DPP modifiers (VALU only; __shfl_xor goes through ds_bpermute, i.e. the LDS pipe)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_ROR4 = 0x124;   // row_ror:4
constexpr int DPP_ROW_ROR8 = 0x128;   // row_ror:8

// sum over the 16 lanes that share (lane >> 4); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<DPP_QUAD_XOR1>(v);
    v += dpp_move<DPP_QUAD_XOR2>(v);   // quad totals
    v += dpp_move<DPP_ROW_ROR4>(v);    // + the next quad
    v += dpp_move<DPP_ROW_ROR8>(v);    // + the other pair of quads
    return v;
}

#define GRL_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)
