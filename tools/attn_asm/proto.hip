// Prototype of the hand-scheduled attention key-row loop (tools/attn_asm/gen_attn_loop.py): cycles per 32x32 S tile at
// 2 / 3 / 4 waves per SIMD, synthetic LDS contents, no DMA.  Build: python3 gen_attn_loop.py --out attn_loop.inc &&
// hipcc --offload-arch=gfx950 -O3 proto.hip -o proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../grl_image_restoration_amd/csrc/attn_rows_asm.inc"
typedef _Float16 f16;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int WAVES, int MINW, bool QLDS>
__global__ __launch_bounds__(WAVES * 64, MINW) void proto(float* out, long long* cyc, int chunks, int barrier, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = (float*)smem;                 // 4096 floats
    char* Ks = smem + 16384;                   // 8 KB
    char* Vs = Ks + 8192;                      // 8 KB
    char* Qs = Vs + 8192;                      // WAVES * 4 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 4096; i += WAVES * 64) tab[i] = 0.001f * (i & 1023);
    for (int i = tid; i < 8192 / 2; i += WAVES * 64) { ((f16*)Ks)[i] = (f16)(0.05f * ((i * 7) % 13 - 6)); ((f16*)Vs)[i] = (f16)(0.1f * ((i * 5) % 11 - 5)); }
    for (int i = tid; i < WAVES * 2048; i += WAVES * 64) ((f16*)Qs)[i] = (f16)(0.1f * ((i * 3) % 17 - 8));
    if (rnd) {   // random operands (the clock follows the data: zero / regular fills run the chip faster)
        unsigned st = 1234567u + 747796405u * (tid + 1024 * blockIdx.x);
        auto nxt = [&] { st = st * 1664525u + 1013904223u; return (float)(st >> 8) * (1.0f / 16777216.0f) - 0.5f; };
        for (int i = tid; i < 4096; i += WAVES * 64) tab[i] = 8.0f * nxt();
        for (int i = tid; i < 8192 / 2; i += WAVES * 64) { ((f16*)Ks)[i] = (f16)(0.6f * nxt()); ((f16*)Vs)[i] = (f16)(2.0f * nxt()); }
        for (int i = tid; i < WAVES * 2048; i += WAVES * 64) ((f16*)Qs)[i] = (f16)(6.0f * nxt());
    }
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int sw = (l31 >> 2) & 3;
    const uint32_t ka0 = lds0 + 16384 + l31 * 64 + (((0 + half) ^ sw) << 4), ka1 = lds0 + 16384 + l31 * 64 + (((2 + half) ^ sw) << 4);
    const uint32_t va = lds0 + 16384 + 8192 + (4 * half + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const uint32_t qa = lds0 + 16384 + 16384 + wave * 4096 + lane * 16;
    const int D = 63;
    const uint32_t ba0 = lds0 + 4 * (200 + 4 * half - l31 + 31);
    f16x8 q00 = *(const f16x8*)(Qs + wave * 4096 + lane * 16), q01 = *(const f16x8*)(Qs + wave * 4096 + 1024 + lane * 16);
    f16x8 q10 = *(const f16x8*)(Qs + wave * 4096 + 2048 + lane * 16), q11 = *(const f16x8*)(Qs + wave * 4096 + 3072 + lane * 16);
    f32x16 O0 = {0}, O1 = {0};
    int tripped = 0;
    const uint32_t ka0b = ka0 - 0, vab = va, bl = lds0 + 4 * (4 * half - l31 + 31);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        uint32_t sb = 4 * 200 + 4 * D * 4 * (c & 7);
        int done;
        if (barrier) __builtin_amdgcn_s_barrier();
        asm volatile(ATTN_ROWS4_MASK0 : [o0] "+v"(O0), [o1] "+v"(O1), [sb] "+s"(sb), [done] "=s"(done)
                     : [ka0] "v"(ka0b), [va] "v"(vab), [bl] "v"(bl), [q00] "v"(q00), [q01] "v"(q01), [q10] "v"(q10), [q11] "v"(q11),
                       [d4] "s"(4 * D), [rs] "s"(0), [par] "s"(0), [nochk] "s"(nochk) : ATTN_ROWS_CLOBBER);
        tripped += done != 4;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)tripped;
    for (int i = 0; i < 16; ++i) s += O0[i] + O1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int WAVES, int MINW, bool QLDS>
void run(const char* name, int blocks_per_cu, size_t lds, int barrier, int rnd = 0) {
    const int blocks = 256 * blocks_per_cu, chunks = 4000;
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4); hipMalloc(&cyc, 8 * blocks * WAVES);
    auto kfn = proto<WAVES, MINW, QLDS>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kfn<<<blocks, WAVES * 64, lds>>>(out, cyc, 50, barrier, rnd);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kfn<<<blocks, WAVES * 64, lds>>>(out, cyc, chunks, barrier, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * WAVES;
    long long* cs = (long long*)malloc(8 * nw); hipMemcpy(cs, cyc, 8 * nw, hipMemcpyDeviceToHost);
    double avg = 0, mx = 0; for (int i = 0; i < nw; ++i) { avg += cs[i]; mx = cs[i] > mx ? cs[i] : mx; } avg /= nw;
    float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
    const double tiles_per_wave = chunks * 8.0, waves_per_simd = blocks_per_cu * WAVES / 4.0;
    // wall: tiles per SIMD = tiles_per_wave * waves_per_simd
    printf("%-30s barrier=%d rnd=%d  wall %.3f ms = %.1f ns per tile per SIMD (%.0f cyc @2.4GHz; %.0f TF/s MFMA-equiv)   memtime per tile per SIMD: avg %.1f  max %.1f   [chk %g]\n",
           name, barrier, rnd, ms, ms * 1e6 / (tiles_per_wave * waves_per_simd), ms * 1e6 / (tiles_per_wave * waves_per_simd) * 2.4,
           4.0 * 32768 * 1024 / (ms * 1e6 / (tiles_per_wave * waves_per_simd)) * 1e-3, avg / tiles_per_wave / waves_per_simd, mx / tiles_per_wave / waves_per_simd, h);
    hipFree(out); hipFree(cyc); free(cs);
}

template <int WAVES, int MINW>
void run_short(int blocks, int chunks, size_t lds, int launches) {
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4); hipMalloc(&cyc, 8 * blocks * WAVES);
    auto kfn = proto<WAVES, MINW, false>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) kfn<<<blocks, WAVES * 64, lds>>>(out, cyc, chunks, 1, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) kfn<<<blocks, WAVES * 64, lds>>>(out, cyc, chunks, 1, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)blocks * WAVES * chunks * 8 / 1024.0;
    printf("short WGs: %d blocks x %d waves, %d chunks, lds %zu: %.1f us per launch = %.1f ns per tile per SIMD\n", blocks, WAVES, chunks, lds, ms * 1e3 / launches,
           ms * 1e6 / launches / tiles_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    run_short<4, 4>(3072, 8, 40 * 1024, 20);
    run_short<4, 4>(1024, 24, 40 * 1024, 20);
    run_short<4, 4>(768, 32, 40 * 1024, 20);
    run_short<4, 4>(3072, 8, 40 * 1024, 20);
    run_short<4, 4>(12288, 2, 40 * 1024, 20);
    for (int r = 0; r < 2; ++r) run<4, 4, false>("w=4: 4 waves x 4 WG", 4, 40 * 1024, 1, r);
    for (int r = 0; r < 2; ++r) run<4, 4, false>("w=4: 4 waves x 4 WG", 4, 40 * 1024, 1, r);
    for (int b = 0; b < 0; ++b) {
        run<4, 2, false>("w=2: 4 waves x 2 WG", 2, 80 * 1024, b);
        run<4, 4, false>("w=3: 4 waves x 3 WG (52 KB)", 3, 52 * 1024, b);
        run<4, 4, false>("w=4: 4 waves x 4 WG", 4, 40 * 1024, b);
        run<8, 4, false>("w=4: 8 waves x 2 WG", 2, 80 * 1024, b);
    }
    return 0;
}
