// Probe: global_load_lds_dwordx4 with an SGPR base + 32-bit VGPR offset (saddr form) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const char* src, int* out, int mode) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x;
    const uint32_t m0v = (uint32_t)(uintptr_t)(lds_ptr_t)smem + 1024;
    const uint32_t voff = 1024 + lane * 16;
    if (mode == 0) {
        const char* g = src + voff;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "m0", "memory");
    } else {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(src) : "m0", "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = ((const int*)(smem + 1024))[lane * 4 + i];
}
int main() {
    int h[512]; for (int i = 0; i < 512; ++i) h[i] = i;
    char* d; int* o; hipMalloc(&d, 2048); hipMalloc(&o, 1024); hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(o, 0, 1024);
        k<<<1, 64, 4096>>>(d, o, mode);
        hipError_t e = hipDeviceSynchronize();
        int r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) bad += r[i] != 256 + i;
        printf("mode %d: %s, %d mismatches (first %d %d %d)\n", mode, hipGetErrorString(e), bad, r[0], r[1], r[255]);
    }
    return 0;
}
