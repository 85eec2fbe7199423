// Probe: which lane does wave_rol:1 / wave_shl:1 read on gfx950?  (hipcc --offload-arch=gfx950 dpp_probe.hip -o dpp_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x134, 0xf, 0xf, true);        // wave_rol:1
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xf, 0xf, true);   // wave_shl:1
    out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x13C, 0xf, 0xf, true);  // wave_ror:1
}
int main() {
    int* d; hipMalloc(&d, 192 * 4);
    k<<<1, 64>>>(d);
    int h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[3] = {"wave_rol:1", "wave_shl:1", "wave_ror:1"};
    for (int s = 0; s < 3; ++s) {
        printf("%s:", names[s]);
        for (int i = 0; i < 64; ++i) if (i < 3 || (i > 29 && i < 35) || i > 60) printf(" [%d]<-%d", i, h[64 * s + i]);
        printf("\n");
    }
    return 0;
}
