// LDS read throughput for the bias-gather address pattern of the attention kernel (gfx950).
// lane address (dwords) = base - (lane & 31) + 4 * (lane >> 5) + offset(r): descending with the query column.
// Build: hipcc --offload-arch=gfx950 -O3 lds_read.hip -o lds_read
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int misalign) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int base = 4096 + ((it * 64) & 2047);
        if constexpr (MODE == 0) {          // 16 x ds_read_b32, real pattern
            const float* tp = lds + base - (lane & 31) + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += tp[(r & 3) + 8 * (r >> 2)];
        } else if constexpr (MODE == 8) {   // kernel offsets (read2 pairs), ascending lanes, no half offset
            const float* tp = lds + base + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += tp[(r & 3) + 8 * (r >> 2)];
        } else if constexpr (MODE == 9) {   // descending lanes + half offset, offsets 64*r (no read2 pairing)
            const float* tp = lds + base - (lane & 31) + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += tp[64 * r];
        } else if constexpr (MODE == 10) {  // descending lanes, NO half offset, kernel offsets
            const float* tp = lds + base - (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += tp[(r & 3) + 8 * (r >> 2)];
        } else if constexpr (MODE == 11) {  // descending lanes, half offset 32 + 4 (second table copy), kernel offsets
            const float* tp = lds + base - (lane & 31) + 36 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += tp[(r & 3) + 8 * (r >> 2)];
        } else if constexpr (MODE == 1) {   // 16 x ds_read_b32, ascending lanes (textbook conflict-free)
            const float* tp = lds + base + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += tp[64 * r];
        } else if constexpr (MODE == 2) {   // 8 x ds_read_b64, lane-consecutive pairs, optionally 4-B misaligned
            const uint32_t a = (uint32_t)(size_t)(lds + base + 2 * lane + misalign) ;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float2 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(0));
                acc[2 * r] += v.x; acc[2 * r + 1] += v.y;
            }
        } else if constexpr (MODE == 3) {   // 8 x ds_read_b64 in the real (descending) pattern: lane reads 2 consecutive dwords
            const uint32_t a = (uint32_t)(size_t)(lds + base - (lane & 31) + 4 * (lane >> 5) + misalign);
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc[2 * r] += v[r].x; acc[2 * r + 1] += v[r].y; }
        } else if constexpr (MODE == 4) {   // 4 x ds_read_b128, descending pattern, 4-B aligned only
            const uint32_t a = (uint32_t)(size_t)(lds + base - (lane & 31) + 4 * (lane >> 5) + misalign);
            float4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[4 * r] += v[r].x; acc[4 * r + 1] += v[r].y; acc[4 * r + 2] += v[r].z; acc[4 * r + 3] += v[r].w; }
        } else if constexpr (MODE == 5) {   // 8 x ds_read_b64, 8-B aligned (two-copy table): lane address forced even
            const uint32_t a = (uint32_t)(size_t)(lds + ((base - (lane & 31) + 4 * (lane >> 5)) & ~1));
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc[2 * r] += v[r].x; acc[2 * r + 1] += v[r].y; }
        } else if constexpr (MODE == 6) {   // 4 x ds_read_b128, 16-B aligned, lane-consecutive (fragment reads of K)
            const uint32_t a = (uint32_t)(size_t)(lds + ((base + 4 * lane) & ~3));
            float4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[4 * r] += v[r].x; acc[4 * r + 1] += v[r].y; acc[4 * r + 2] += v[r].z; acc[4 * r + 3] += v[r].w; }
        } else if constexpr (MODE == 7) {   // 4 x ds_read_b128, descending pattern rounded to 16 B (four-copy table)
            const uint32_t a = (uint32_t)(size_t)(lds + ((base - (lane & 31) + 4 * (lane >> 5)) & ~3));
            float4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[4 * r] += v[r].x; acc[4 * r + 1] += v[r].y; acc[4 * r + 2] += v[r].z; acc[4 * r + 3] += v[r].w; }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wps, int reads, int bytes_per_lane, int misalign) {
    float* out;
    const int blocks = 256 * wps;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(out, 100, misalign);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, misalign);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_reads_per_cu = 4.0 * wps * iters * reads;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-46s misalign=%d waves/SIMD=%d  %.3f ms  %.2f clk per wave-read per CU  %.1f B/clk/CU (at 2.4 GHz)\n", name, misalign, wps, ms,
           cyc / wave_reads_per_cu, wave_reads_per_cu * 64 * bytes_per_lane / cyc);
    hipFree(out);
}

int main() {
    for (int w : {4}) {
        run<0>("b32 x16 descending+half (kernel pattern)", w, 16, 4, 0);
        run<1>("b32 x16 ascending lanes", w, 16, 4, 0);
        run<8>("b32 x16 ascending, kernel offsets (read2)", w, 16, 4, 0);
        run<9>("b32 x16 descending+half, offsets 64r", w, 16, 4, 0);
        run<10>("b32 x16 descending, no half offset", w, 16, 4, 0);
        run<11>("b32 x16 descending, half offset 36", w, 16, 4, 0);
        run<2>("b64 x8 lane-consecutive, serial waits", w, 8, 8, 0);
        run<3>("b64 x8 descending pattern", w, 8, 8, 0);
        run<3>("b64 x8 descending pattern", w, 8, 8, 1);
        run<5>("b64 x8 descending, 8-B aligned", w, 8, 8, 0);
        run<4>("b128 x4 descending pattern (4-B aligned)", w, 4, 16, 0);
        run<4>("b128 x4 descending pattern (4-B aligned)", w, 4, 16, 1);
        run<7>("b128 x4 descending, 16-B aligned", w, 4, 16, 0);
        run<6>("b128 x4 lane-consecutive 16-B aligned", w, 4, 16, 0);
    }
    return 0;
}
