// gfx950 probe: (1) does v_cvt_scalef32_pk_fp8_f16 divide or multiply by its scale operand, and does it saturate?
// (2) lane <-> k layout of v_mfma_f32_32x32x16_fp8_fp8 (expected: lane = row/column (lane & 31), k = 8 * (lane >> 5) + byte).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/fp8_probe.hip -o tools/probes/fp8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(2 * sizeof(_Float16)))) _Float16 f16x2;
typedef __attribute__((__vector_size__(2 * sizeof(short)))) short s16x2;

__global__ void cvt_probe(const float* in, uint32_t* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    f16x2 h = {(_Float16)in[i], (_Float16)in[i]};
    s16x2 z = {0, 0};
    s16x2 a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, h, 1.0f, false);
    s16x2 b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, h, 16.0f, false);
    s16x2 c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, h, 16.0f, true);
    out[3 * i] = __builtin_bit_cast(uint32_t, a);
    out[3 * i + 1] = __builtin_bit_cast(uint32_t, b);
    out[3 * i + 2] = __builtin_bit_cast(uint32_t, c);
}

// A[i][k] = fp8 code table entry, B[k][n]: D = A . B with the assumed layout; the host checks against a plain loop
__global__ void mfma_probe(const uint8_t* A, const uint8_t* B, float* D) {   // A: [32][16], B: [32 (n)][16 (k)] bytes
    const int lane = threadIdx.x, j = lane & 31, half = lane >> 5;
    long a = *(const long*)(A + j * 16 + 8 * half), b = *(const long*)(B + j * 16 + 8 * half);
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + j] = c[r];   // row (A row), column j (B row n)
}

static float e4m3(uint8_t v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? m / 8.0f * 0.015625f : (1.0f + m / 8.0f) * __builtin_ldexpf(1.0f, e - 7);
    if (e == 15 && m == 7) f = __builtin_nanf("");
    return s ? -f : f;
}

int main() {
    const float vals[] = {1.0f, 16.0f, 100.0f, 448.0f, 500.0f, 1000.0f, 8000.0f, 60000.0f, -3.3f, 0.01f, 0.001f, 0.3f};
    const int n = sizeof(vals) / sizeof(float);
    float* din; uint32_t* dout;
    hipMalloc(&din, sizeof(vals)); hipMalloc(&dout, 3 * n * 4);
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(din, dout, n);
    uint32_t o[3 * 16];
    hipMemcpy(o, dout, 3 * n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("x=%10.4f  scale1: %08x (%g)   scale16 lo-half: %08x (%g)   scale16 hi-half: %08x\n", vals[i], o[3 * i], e4m3(o[3 * i] & 255),
               o[3 * i + 1], e4m3(o[3 * i + 1] & 255), o[3 * i + 2]);
    uint8_t A[32 * 16], B[32 * 16];
    uint32_t seed = 12345;
    for (int i = 0; i < 512; ++i) { seed = seed * 1664525u + 1013904223u; A[i] = (seed >> 24) & 0xBF; if ((A[i] & 0x7F) == 0x7F) A[i] = 0x30; }
    for (int i = 0; i < 512; ++i) { seed = seed * 1664525u + 1013904223u; B[i] = (seed >> 24) & 0xBF; if ((B[i] & 0x7F) == 0x7F) B[i] = 0x30; }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dA, dB, dD);
    float D[1024];
    hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 32; ++i)
        for (int nn = 0; nn < 32; ++nn) {
            double r = 0;
            for (int k = 0; k < 16; ++k) r += (double)e4m3(A[i * 16 + k]) * e4m3(B[nn * 16 + k]);
            const double d = fabs(r - D[i * 32 + nn]);
            if (d > worst) worst = d;
        }
    printf("fp8 mfma 32x32x16 vs host loop with k = 8 * half + byte: max |diff| = %g  (%s)\n", worst, worst < 1e-3 ? "layout OK" : "LAYOUT MISMATCH");
    return 0;
}
