// Micro-benchmarks of the instruction mix the attention kernel depends on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[8192 + 64];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = -1.0f - 0.001f * i;
    __syncthreads();
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (threadIdx.x + i);
    f32x16 acc = {0}, acc2 = {0}, acc3 = {0}, acc4 = {0};
    bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // 16 independent v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.0f;
        } else if constexpr (MODE == 1) {  // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 0.999f, -0.5f);
        } else if constexpr (MODE == 2) {  // 4 MFMA 32x32x16 (one accumulator chain)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if constexpr (MODE == 3) {  // attention-like: 4 MFMA + 16 exp + 8 cvt
            f32x16 s = {0};
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);
            bf16x8 p0, p1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)__builtin_amdgcn_exp2f(s[i] + x[i]); p1[i] = (__bf16)__builtin_amdgcn_exp2f(s[8 + i] + x[8 + i]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1, acc, 0, 0, 0);
        } else if constexpr (MODE == 5) {  // co-issue, no data dependency: 4 MFMA || 16 exp + 16 sub
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.0f;
        } else if constexpr (MODE == 6) {  // co-issue, no dependency: 4 MFMA || 32 fma
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(__builtin_fmaf(x[i], 0.999f, -0.5f), 1.001f, 0.25f);
        } else if constexpr (MODE == 7) {  // dependent like attention but 2 independent tiles interleaved in source
            f32x16 s0 = {0}, s1 = {0};
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
            bf16x8 p0, p1, q0, q1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)__builtin_amdgcn_exp2f(s0[i]); p1[i] = (__bf16)__builtin_amdgcn_exp2f(s0[8 + i]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { q0[i] = (__bf16)__builtin_amdgcn_exp2f(s1[i]); q1[i] = (__bf16)__builtin_amdgcn_exp2f(s1[8 + i]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q0, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q1, acc2, 0, 0, 0);
        } else if constexpr (MODE == 11 || MODE == 12 || MODE == 13) {  // mode 9 with the 32 bias values fetched as 16 x b64 / 8 x b128 / 32 x b32 descending
            f32x16 s0, s1;
            if constexpr (MODE == 11) {
                const float2* tp0 = (const float2*)(lds + (((threadIdx.x & 63) * 2 + it * 32) & 4094));
                const float2* tp1 = (const float2*)(lds + (((threadIdx.x & 63) * 2 + it * 32 + 96) & 4094));
#pragma unroll
                for (int r = 0; r < 8; ++r) { float2 a0 = tp0[64 * r], a1 = tp1[64 * r]; s0[2 * r] = a0.x; s0[2 * r + 1] = a0.y; s1[2 * r] = a1.x; s1[2 * r + 1] = a1.y; }
            } else if constexpr (MODE == 12) {
                const float4* tp0 = (const float4*)(lds + (((threadIdx.x & 63) * 4 + it * 32) & 4092));
                const float4* tp1 = (const float4*)(lds + (((threadIdx.x & 63) * 4 + it * 32 + 96) & 4092));
#pragma unroll
                for (int r = 0; r < 4; ++r) { float4 a0 = tp0[64 * r], a1 = tp1[64 * r];
                    s0[4 * r] = a0.x; s0[4 * r + 1] = a0.y; s0[4 * r + 2] = a0.z; s0[4 * r + 3] = a0.w;
                    s1[4 * r] = a1.x; s1[4 * r + 1] = a1.y; s1[4 * r + 2] = a1.z; s1[4 * r + 3] = a1.w; }
            } else {
                const int ln = threadIdx.x & 63;
                const float* tp0 = lds + 2048 + ((it * 32) & 1023) - (ln & 31) + 4 * (ln >> 5);
                const float* tp1 = tp0 + 95;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] = tp0[(r & 3) + 8 * (r >> 2)]; s1[r] = tp1[(r & 3) + 8 * (r >> 2)]; }
            }
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
            bf16x8 p0, p1, q0, q1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)__builtin_amdgcn_exp2f(s0[i]); p1[i] = (__bf16)__builtin_amdgcn_exp2f(s0[8 + i]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { q0[i] = (__bf16)__builtin_amdgcn_exp2f(s1[i]); q1[i] = (__bf16)__builtin_amdgcn_exp2f(s1[8 + i]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q0, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q1, acc2, 0, 0, 0);
        } else if constexpr (MODE == 14 || MODE == 15 || MODE == 16) {
            // 14: C operand from VALU-computed registers (no LDS)   15: LDS bias added AFTER a C=0 MFMA chain
            // 16: LDS bias reads issued one iteration AHEAD (software prefetch into a second register set)
            f32x16 s0, s1;
            if constexpr (MODE == 14) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] = x[r] + (float)it; s1[r] = x[r] - (float)it; }
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
            } else if constexpr (MODE == 15) {
                const float* tp0 = lds + ((threadIdx.x * 5 + it * 32) & 4095);
                const float* tp1 = lds + ((threadIdx.x * 5 + it * 32 + 95) & 4095);
                f32x16 b0, b1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { b0[r] = tp0[(r & 3) + 8 * (r >> 2)]; b1[r] = tp1[(r & 3) + 8 * (r >> 2)]; }
                s0 = f32x16{0}; s1 = f32x16{0};
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] += b0[r]; s1[r] += b1[r]; }
            } else {
                static_assert(MODE == 16, "");
                const float* tp0 = lds + ((threadIdx.x * 5 + (it + 1) * 32) & 4095);
                const float* tp1 = lds + ((threadIdx.x * 5 + (it + 1) * 32 + 95) & 4095);
                f32x16 n0, n1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { n0[r] = tp0[(r & 3) + 8 * (r >> 2)]; n1[r] = tp1[(r & 3) + 8 * (r >> 2)]; }
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc4, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
                acc3 = n0; acc4 = n1;
            }
            bf16x8 p0, p1, q0, q1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)__builtin_amdgcn_exp2f(s0[i]); p1[i] = (__bf16)__builtin_amdgcn_exp2f(s0[8 + i]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { q0[i] = (__bf16)__builtin_amdgcn_exp2f(s1[i]); q1[i] = (__bf16)__builtin_amdgcn_exp2f(s1[8 + i]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q0, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q1, acc2, 0, 0, 0);
        } else if constexpr (MODE == 9 || MODE == 10) {  // mode 7 + the LDS traffic of the real kernel (bias 32 x b32, K/V frags)
            f32x16 s0, s1;
            const float* tp0 = lds + ((threadIdx.x * 5 + it * 32) & 4095);
            const float* tp1 = lds + ((threadIdx.x * 5 + it * 32 + 95) & 4095);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = tp0[(r & 3) + 8 * (r >> 2)]; s1[r] = tp1[(r & 3) + 8 * (r >> 2)]; }
            bf16x8 ka = a, kb = b;
            if constexpr (MODE == 10) {
                ka = *(const bf16x8*)(lds + 4096 + ((threadIdx.x * 16 + it * 64) & 4095));
                kb = *(const bf16x8*)(lds + 4096 + ((threadIdx.x * 16 + it * 64 + 8) & 4095));
            }
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, a, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, a, s1, 0, 0, 0);
            bf16x8 p0, p1, q0, q1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)__builtin_amdgcn_exp2f(s0[i]); p1[i] = (__bf16)__builtin_amdgcn_exp2f(s0[8 + i]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { q0[i] = (__bf16)__builtin_amdgcn_exp2f(s1[i]); q1[i] = (__bf16)__builtin_amdgcn_exp2f(s1[8 + i]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, p0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, q0, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, p1, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, q1, acc2, 0, 0, 0);
        } else if constexpr (MODE == 8) {  // like 3 without exp: fma instead
            f32x16 s = {0};
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);
            bf16x8 p0, p1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)(s[i] * 0.5f + x[i]); p1[i] = (__bf16)(s[8 + i] * 0.5f + x[8 + i]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1, acc, 0, 0, 0);
        } else if constexpr (MODE == 4) {  // 8 cvt_pk
            bf16x8 p0, p1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p0[i] = (__bf16)x[i]; p1[i] = (__bf16)x[8 + i]; }
            asm volatile("" ::"v"(p0), "v"(p1));
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] += 1.0f;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i] + acc[i] + acc2[i] + acc3[i] + acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, double ops_per_iter) {
    float* out;
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves x iters x ops wave-instructions
    const double instr_per_simd = (double)waves_per_simd * iters * ops_per_iter;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}

int main() {
    for (int w : {2, 4}) {
        run<0>("v_exp_f32 (+v_sub)", w, 32);
        run<1>("v_fma_f32", w, 16);
        run<2>("mfma_32x32x16_bf16", w, 4);
        run<4>("cvt_pk_bf16 (8)+16 add", w, 24);
        run<3>("attn mix 4mfma+16exp+8cvt+16add", w, 1);
        run<5>("indep 4mfma || 16exp+16sub", w, 1);
        run<6>("indep 4mfma || 32fma", w, 1);
        run<7>("attn 2 tiles: 8mfma+32exp+16cvt", w, 1);
        run<8>("attn-like fma: 4mfma+16fma+8cvt", w, 1);
        run<9>("attn 2 tiles + 32 bias LDS reads", w, 1);
        run<10>("attn 2 tiles + bias + frag LDS reads", w, 1);
        run<13>("attn 2 tiles + 32 b32 (kernel pattern)", w, 1);
        run<14>("attn 2 tiles, C from VALU regs (no LDS)", w, 1);
        run<15>("attn 2 tiles, LDS bias added after MFMA", w, 1);
        run<16>("attn 2 tiles, LDS bias prefetched 1 iter", w, 1);
        run<11>("attn 2 tiles + bias as 16 x b64", w, 1);
        run<12>("attn 2 tiles + bias as 8 x b128", w, 1);
    }
    return 0;
}
