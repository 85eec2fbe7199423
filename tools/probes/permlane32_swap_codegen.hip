// Codegen probe (no GPU needed):  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only permlane32_swap_codegen.hip -o - | grep -A8 permlane
// ROCm 7.2.0's clang selects v_permlane32_swap_b32 for __builtin_amdgcn_permlane32_swap but then uses the FIRST result
// register for both elements of the returned pair: the two stores below write the same register (v1, v1) instead of the two
// operands of the swap (v1, v2).  csrc/attention_bwd.hip therefore issues the instruction through inline asm.
#include <hip/hip_runtime.h>
__global__ void k(float* out, const float* in) {
    const float a = in[threadIdx.x];
    const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), 0u, false, false);
    out[threadIdx.x] = __builtin_bit_cast(float, s[0]);        // lanes 0-31 of a over zeros
    out[64 + threadIdx.x] = __builtin_bit_cast(float, s[1]);   // lanes 32-63 of a, moved to lanes 0-31
}
