// Issue / pipe model probes for the attention inner loop (gfx950): are v_exp_f32 (transcendental), plain VALU and MFMA
// overlappable, and at what rates, for 1 / 2 / 4 waves per SIMD.  Bodies are single asm blocks so that the instruction
// stream is exactly what is written.  Cycles from s_memtime (shader clock), wave 0 of block 0.
// Build: hipcc --offload-arch=gfx950 -O3 pipes.hip -o pipes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define E4(a,b,c,d) "v_exp_f32 %" #a ", %" #a "\n v_exp_f32 %" #b ", %" #b "\n v_exp_f32 %" #c ", %" #c "\n v_exp_f32 %" #d ", %" #d "\n"
#define F1(a) "v_fma_f32 %" #a ", %" #a ", %36, %37\n"
#define EF(a,f) "v_exp_f32 %" #a ", %" #a "\n v_fma_f32 %" #f ", %" #f ", %36, %37\n"
#define EFF(a,f,g) "v_exp_f32 %" #a ", %" #a "\n v_fma_f32 %" #f ", %" #f ", %36, %37\n v_fma_f32 %" #g ", %" #g ", %36, %37\n"
#define MF(acc) "v_mfma_f32_32x32x16_f16 %" #acc ", %34, %35, %" #acc "\n"

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = -0.001f * (threadIdx.x + i); y[i] = 0.5f + 0.001f * i; }
    f32x16 A0 = {0}, A1 = {0};
    f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    float k1 = 0.999f, k2 = 0.0001f * threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), \
              "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]), "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]), \
              "+v"(A0), "+v"(A1) : "v"(a), "v"(b), "v"(k1), "v"(k2)
        if constexpr (MODE == 0) {   // 16 exp
            asm volatile(E4(0,1,2,3) E4(4,5,6,7) E4(8,9,10,11) E4(12,13,14,15) OPS);
        } else if constexpr (MODE == 1) {  // 16 fma
            asm volatile(F1(16) F1(17) F1(18) F1(19) F1(20) F1(21) F1(22) F1(23) F1(24) F1(25) F1(26) F1(27) F1(28) F1(29) F1(30) F1(31) OPS);
        } else if constexpr (MODE == 2) {  // 16 exp + 16 fma interleaved
            asm volatile(EF(0,16) EF(1,17) EF(2,18) EF(3,19) EF(4,20) EF(5,21) EF(6,22) EF(7,23) EF(8,24) EF(9,25) EF(10,26) EF(11,27) EF(12,28) EF(13,29) EF(14,30) EF(15,31) OPS);
        } else if constexpr (MODE == 3) {  // 16 exp + 32 fma
            asm volatile(EFF(0,16,17) EFF(1,18,19) EFF(2,20,21) EFF(3,22,23) EFF(4,24,25) EFF(5,26,27) EFF(6,28,29) EFF(7,30,31) EFF(8,16,17) EFF(9,18,19) EFF(10,20,21) EFF(11,22,23) EFF(12,24,25) EFF(13,26,27) EFF(14,28,29) EFF(15,30,31) OPS);
        } else if constexpr (MODE == 4) {  // 4 MFMA, two accumulators alternating
            asm volatile(MF(32) MF(33) MF(32) MF(33) OPS);
        } else if constexpr (MODE == 5) {  // 4 MFMA + 16 exp (4 behind each MFMA)
            asm volatile(MF(32) E4(0,1,2,3) MF(33) E4(4,5,6,7) MF(32) E4(8,9,10,11) MF(33) E4(12,13,14,15) OPS);
        } else if constexpr (MODE == 6) {  // 4 MFMA + 16 exp + 16 fma
            asm volatile(MF(32) EF(0,16) EF(1,17) EF(2,18) EF(3,19) MF(33) EF(4,20) EF(5,21) EF(6,22) EF(7,23) MF(32) EF(8,24) EF(9,25) EF(10,26) EF(11,27) MF(33) EF(12,28) EF(13,29) EF(14,30) EF(15,31) OPS);
        } else if constexpr (MODE == 7) {  // 4 MFMA + 16 exp + 32 fma
            asm volatile(MF(32) EFF(0,16,17) EFF(1,18,19) EFF(2,20,21) EFF(3,22,23) MF(33) EFF(4,24,25) EFF(5,26,27) EFF(6,28,29) EFF(7,30,31) MF(32) EFF(8,16,17) EFF(9,18,19) EFF(10,20,21) EFF(11,22,23) MF(33) EFF(12,24,25) EFF(13,26,27) EFF(14,28,29) EFF(15,30,31) OPS);
        } else if constexpr (MODE == 8) {  // 4 MFMA + 32 fma (8 behind each)
            asm volatile(MF(32) F1(16) F1(17) F1(18) F1(19) F1(20) F1(21) F1(22) F1(23) MF(33) F1(24) F1(25) F1(26) F1(27) F1(28) F1(29) F1(30) F1(31) MF(32) F1(16) F1(17) F1(18) F1(19) F1(20) F1(21) F1(22) F1(23) MF(33) F1(24) F1(25) F1(26) F1(27) F1(28) F1(29) F1(30) F1(31) OPS);
        } else if constexpr (MODE == 9) {  // 4 MFMA + 8 exp
            asm volatile(MF(32) "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" MF(33) "v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" MF(32) "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n" MF(33) "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" OPS);
        } else if constexpr (MODE == 10) {  // 4 MFMA + 12 exp
            asm volatile(MF(32) "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %8, %8\n" MF(33) "v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %9, %9\n" MF(32) "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %10, %10\n" MF(33) "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %11, %11\n" OPS);
        } else if constexpr (MODE == 11) {  // 4 MFMA + 16 exp + 8 cvt_pk + 8 pk_max: the attention tile's VALU mix
#define CV(d,a,b) "v_cvt_pk_f16_f32 %" #d ", %" #a ", %" #b "\n"
#define PM(d,a,b) "v_pk_max_u16 %" #d ", %" #a ", %" #b "\n"
            asm volatile(MF(32) E4(0,1,2,3) CV(16,0,1) CV(17,2,3) PM(24,16,17) PM(25,24,17) MF(33) E4(4,5,6,7) CV(18,4,5) CV(19,6,7) PM(26,18,19) PM(27,26,19)
                         MF(32) E4(8,9,10,11) CV(20,8,9) CV(21,10,11) PM(28,20,21) PM(29,28,21) MF(33) E4(12,13,14,15) CV(22,12,13) CV(23,14,15) PM(30,22,23) PM(31,30,23) OPS);
        } else if constexpr (MODE == 12) {  // 16 exp + 8 cvt + 8 pk_max, no MFMA
            asm volatile(E4(0,1,2,3) CV(16,0,1) CV(17,2,3) PM(24,16,17) PM(25,24,17) E4(4,5,6,7) CV(18,4,5) CV(19,6,7) PM(26,18,19) PM(27,26,19)
                         E4(8,9,10,11) CV(20,8,9) CV(21,10,11) PM(28,20,21) PM(29,28,21) E4(12,13,14,15) CV(22,12,13) CV(23,14,15) PM(30,22,23) PM(31,30,23) OPS);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i] + y[i] + A0[i] + A1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int wps) {
    float* out; long long* cyc;
    const int blocks = 256, nw = 4 * wps;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, 8 * blocks * nw);
    const int iters = 20000;
    k<MODE><<<blocks, 256 * wps>>>(out, cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256 * wps>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long cs[256 * 16]; hipMemcpy(cs, cyc, 8 * blocks * nw, hipMemcpyDeviceToHost);
    long long c = 0, cmax = 0; for (int i = 0; i < blocks * nw; ++i) { c += cs[i]; cmax = cs[i] > cmax ? cs[i] : cmax; } c /= blocks * nw;
    // s_memtime counts at a fixed 100 MHz on some parts; report both
    printf("%-34s w/SIMD=%d  wall %.3f ms -> %.1f ns per body per SIMD (= %.1f cyc @2.4GHz)   memtime avg %.1f max %.1f ticks per body per wave -> %.1f ticks per body per SIMD\n", name, wps, ms,
           ms * 1e6 / iters / wps, ms * 1e6 / iters / wps * 2.4, (double)c / iters, (double)cmax / iters, (double)c / iters / wps);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("16 exp", w);
        run<1>("16 fma", w);
        run<2>("16 exp + 16 fma", w);
        run<3>("16 exp + 32 fma", w);
        run<4>("4 mfma", w);
        run<9>("4 mfma + 8 exp", w);
        run<10>("4 mfma + 12 exp", w);
        run<5>("4 mfma + 16 exp", w);
        run<6>("4 mfma + 16 exp + 16 fma", w);
        run<7>("4 mfma + 16 exp + 32 fma", w);
        run<8>("4 mfma + 32 fma", w);
        run<12>("16 exp + 8 cvt + 8 pkmax", w);
        run<11>("4 mfma + 16 exp + 8 cvt + 8 pkmax", w);
    }
    return 0;
}
