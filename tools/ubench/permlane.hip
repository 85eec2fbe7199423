// Semantics probe of v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) and of the helpers of csrc/common.h built on them.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include permlane.hip -o permlane
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../grl_image_restoration_amd/csrc/common.h"
__global__ void k(unsigned* o, float* f) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    auto s = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1]; o[128 + l] = s[0]; o[192 + l] = s[1];
    float t = (float)(l * l);
    f[l] = sum_rows16(t) - (t + __shfl_xor(t, 16, 64));
    f[64 + l] = sum_halves(t) - (t + __shfl_xor(t, 32, 64));
    f[128 + l] = xhalf(t) - __shfl_xor(t, 32, 64);
    float u = t;
    u += dpp_move<DPP_QUAD_XOR1>(u);
    u = sum_rows16(u);
    float w = t;
    w += __shfl_xor(w, 1, 64);
    w += __shfl_xor(w, 16, 64);
    f[192 + l] = u - w;
}
int main() {
    unsigned* d; float* df;
    (void)hipMalloc(&d, 256 * 4); (void)hipMalloc(&df, 256 * 4);
    k<<<1, 64>>>(d, df);
    unsigned h[256]; float hf[256];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(hf, df, sizeof(hf), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int i = 0; i < 4; ++i) { printf("%s:", names[i]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * i + l]); printf("\n"); }
    const char* fn[4] = {"sum_rows16 - ref", "sum_halves - ref", "xhalf - ref", "dpp+rows16 - ref"};
    for (int i = 0; i < 4; ++i) { float mx = 0; for (int l = 0; l < 64; ++l) mx = fmaxf(mx, fabsf(hf[64 * i + l])); printf("%s: max |diff| = %g\n", fn[i], mx); }
    return 0;
}
