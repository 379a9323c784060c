// does v_mfma_f32_16x16x32_bf16 on concatenated operands equal two v_mfma_f32_16x16x16_bf16 ?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const s16x4* a0, const s16x4* a1, const s16x4* b0, const s16x4* b1, f32x4* o16, f32x4* o32) {
    const int l = threadIdx.x;
    f32x4 z = {0, 0, 0, 0};
    f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0[l], b0[l], z, 0, 0, 0);
    r16 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1[l], b1[l], r16, 0, 0, 0);
    const s16x8 a = __builtin_shufflevector(a0[l], a1[l], 0, 1, 2, 3, 4, 5, 6, 7);
    const s16x8 b = __builtin_shufflevector(b0[l], b1[l], 0, 1, 2, 3, 4, 5, 6, 7);
    f32x4 r32 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), z, 0, 0, 0);
    o16[l] = r16; o32[l] = r32;
}
int main() {
    s16x4 h[4][64]; 
    for (int t = 0; t < 4; ++t) for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        float f = (float)((rand() % 2001) - 1000) / 500.f; unsigned u; memcpy(&u, &f, 4); h[t][l][i] = (short)(u >> 16);
    }
    s16x4* d[4]; f32x4 *o16, *o32;
    for (int t = 0; t < 4; ++t) { hipMalloc(&d[t], sizeof(h[t])); hipMemcpy(d[t], h[t], sizeof(h[t]), hipMemcpyHostToDevice); }
    hipMalloc(&o16, 64 * 16); hipMalloc(&o32, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d[0], d[1], d[2], d[3], o16, o32);
    f32x4 r16[64], r32[64];
    hipMemcpy(r16, o16, sizeof(r16), hipMemcpyDeviceToHost); hipMemcpy(r32, o32, sizeof(r32), hipMemcpyDeviceToHost);
    double worst = 0, mag = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) { double e = fabs(r16[l][i] - r32[l][i]); if (e > worst) worst = e; if (fabs(r16[l][i]) > mag) mag = fabs(r16[l][i]); }
    printf("max |two x16 - one x32| = %g (max |value| %g)\n", worst, mag);
    return 0;
}
