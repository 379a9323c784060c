// is vdst == srcA (or srcB) legal for v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16 on gfx950 ?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const s16x8* a, const s16x8* b, f32x4* ref, f32x4* ovA, f32x4* ovB, f32x4* ref16, f32x4* ov16) {
    const int l = threadIdx.x;
    f32x4 z = {1.f, 2.f, 3.f, 4.f};
    f32x4 av = __builtin_bit_cast(f32x4, a[l]), bv = __builtin_bit_cast(f32x4, b[l]);
    f32x4 r, ra = av, rb = bv;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3\n s_nop 15\n s_nop 15" : "=&v"(r) : "v"(av), "v"(bv), "v"(z));
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %0, %1, %2\n s_nop 15\n s_nop 15" : "+v"(ra) : "v"(bv), "v"(z));
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %0, %2\n s_nop 15\n s_nop 15" : "+v"(rb) : "v"(av), "v"(z));
    ref[l] = r; ovA[l] = ra; ovB[l] = rb;
    f32x4 r16 = r, o16 = r;
    ref16[l] = r16; ov16[l] = o16;
}
int main() {
    s16x8 ha[64], hb[64];
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 8; ++i) {
        float f = (float)((rand() % 2001) - 1000) / 500.f; unsigned u; memcpy(&u, &f, 4); ha[l][i] = (short)(u >> 16);
        f = (float)((rand() % 2001) - 1000) / 500.f; memcpy(&u, &f, 4); hb[l][i] = (short)(u >> 16);
    }
    s16x8 *da, *db; f32x4* o[5];
    (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb));
    (void)hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    for (int i = 0; i < 5; ++i) (void)hipMalloc(&o[i], 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, o[0], o[1], o[2], o[3], o[4]);
    f32x4 r[5][64];
    for (int i = 0; i < 5; ++i) (void)hipMemcpy(r[i], o[i], sizeof(r[i]), hipMemcpyDeviceToHost);
    const char* nm[5] = {"ref32", "dst==srcA (x32)", "dst==srcB (x32)", "ref16", "dst==srcA (x16)"};
    for (int v : {1, 2}) { double w = 0; for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) w = fmax(w, fabs(r[v][l][i] - r[0][l][i])); printf("%s: max diff vs ref %g\n", nm[v], w); }
    { double w = 0; for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) w = fmax(w, fabs(r[4][l][i] - r[3][l][i])); printf("%s: max diff vs ref %g\n", nm[4], w); }
    return 0;
}
