// semantic probe of ds_read_b64_tr_b16: prints which source element each (lane, j) receives
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__global__ void k(short* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int idx;
    if (mode == 0) idx = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + 4 * (l & 3);      // [4][16] block per 16-lane group, lane -> (row l/4, cols 4*(l%4))
    else idx = (l >> 4) * 256 + ((l & 15) >> 2) * 32 + 4 * (l & 3);                // rows 32 elements (64 B) apart
    __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(lds + idx);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; l += (l < 20 ? 1 : 8)) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    return 0;
}
