// v_mfma_f32_16x16x32_bf16 on gfx950: is a destination that PARTIALLY overlaps srcC legal?  (hipcc allocates it:
// k_attn_bwd_lb<2> carries `v_mfma_f32_16x16x32_bf16 v[56:59], v[78:81], v[50:53], v[58:61]`.)  Round 4 measured
// vdst == srcA wrong, vdst == srcB fine (profiles/exp_r04/mfma32b.hip); vdst == srcC is the ordinary accumulate.
//   hipcc --offload-arch=gfx950 -O2 profiles/exp_r06_mfma_partial_c.hip -o /tmp/mfma_pc && /tmp/mfma_pc
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const s16x8* a, const s16x8* b, const f32x4* c, f32x4* ref, f32x4* lo, f32x4* hi, f32x4* ova) {
    const int l = threadIdx.x;
    const f32x4 av = __builtin_bit_cast(f32x4, a[l]), bv = __builtin_bit_cast(f32x4, b[l]), cv = c[l];
    f32x4 r, rl, rh, ra;
    // reference: disjoint registers
    asm volatile("v_mov_b32 v12, %3\n v_mov_b32 v13, %4\n v_mov_b32 v14, %5\n v_mov_b32 v15, %6\n s_nop 4\n"
                 "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[12:15]\n s_nop 15\n s_nop 15\n"
                 "v_mov_b32 %0, v20" : "=&v"(r[0]) : "v"(av), "v"(bv), "v"(cv[0]), "v"(cv[1]), "v"(cv[2]), "v"(cv[3])
                 : "v12", "v13", "v14", "v15", "v20", "v21", "v22", "v23");
    asm volatile("v_mov_b32 %0, v21\n v_mov_b32 %1, v22\n v_mov_b32 %2, v23" : "=v"(r[1]), "=v"(r[2]), "=v"(r[3]));
    // destination two registers BELOW srcC: v[10:13] over v[12:15]
    asm volatile("v_mov_b32 v12, %3\n v_mov_b32 v13, %4\n v_mov_b32 v14, %5\n v_mov_b32 v15, %6\n s_nop 4\n"
                 "v_mfma_f32_16x16x32_bf16 v[10:13], %1, %2, v[12:15]\n s_nop 15\n s_nop 15\n"
                 "v_mov_b32 %0, v10" : "=&v"(rl[0]) : "v"(av), "v"(bv), "v"(cv[0]), "v"(cv[1]), "v"(cv[2]), "v"(cv[3])
                 : "v10", "v11", "v12", "v13", "v14", "v15");
    asm volatile("v_mov_b32 %0, v11\n v_mov_b32 %1, v12\n v_mov_b32 %2, v13" : "=v"(rl[1]), "=v"(rl[2]), "=v"(rl[3]));
    // destination two registers ABOVE srcC: v[14:17] over v[12:15]
    asm volatile("v_mov_b32 v12, %3\n v_mov_b32 v13, %4\n v_mov_b32 v14, %5\n v_mov_b32 v15, %6\n s_nop 4\n"
                 "v_mfma_f32_16x16x32_bf16 v[14:17], %1, %2, v[12:15]\n s_nop 15\n s_nop 15\n"
                 "v_mov_b32 %0, v14" : "=&v"(rh[0]) : "v"(av), "v"(bv), "v"(cv[0]), "v"(cv[1]), "v"(cv[2]), "v"(cv[3])
                 : "v12", "v13", "v14", "v15", "v16", "v17");
    asm volatile("v_mov_b32 %0, v15\n v_mov_b32 %1, v16\n v_mov_b32 %2, v17" : "=v"(rh[1]), "=v"(rh[2]), "=v"(rh[3]));
    // the round-4 case again: destination == srcA
    ra = av;
#ifdef NO_PRE_NOP      // (round 4's form: the copy ra = av - four v_mov - sits directly in front of the instruction)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %0, %1, %2\n s_nop 15\n s_nop 15" : "+v"(ra) : "v"(bv), "v"(cv));
#else
    asm volatile("s_nop 4\n v_mfma_f32_16x16x32_bf16 %0, %0, %1, %2\n s_nop 15\n s_nop 15" : "+v"(ra) : "v"(bv), "v"(cv));
#endif
    ref[l] = r; lo[l] = rl; hi[l] = rh; ova[l] = ra;
}
int main() {
    s16x8 ha[64], hb[64];
    f32x4 hc[64];
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 8; ++i) {
            float f = (float)((rand() % 2001) - 1000) / 500.f; unsigned u; memcpy(&u, &f, 4); ha[l][i] = (short)(u >> 16);
            f = (float)((rand() % 2001) - 1000) / 500.f; memcpy(&u, &f, 4); hb[l][i] = (short)(u >> 16);
        }
        for (int i = 0; i < 4; ++i) hc[l][i] = (float)((rand() % 2001) - 1000) / 100.f;
    }
    s16x8 *da, *db; f32x4 *dc, *o[4];
    (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb)); (void)hipMalloc(&dc, sizeof(hc));
    (void)hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice);
    for (int i = 0; i < 4; ++i) (void)hipMalloc(&o[i], 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, o[0], o[1], o[2], o[3]);
    f32x4 r[4][64];
    for (int i = 0; i < 4; ++i) (void)hipMemcpy(r[i], o[i], sizeof(r[i]), hipMemcpyDeviceToHost);
    const char* nm[4] = {"ref", "vdst two registers below srcC", "vdst two registers above srcC", "vdst == srcA"};
    double mag = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) mag = fmax(mag, fabs(r[0][l][i]));
    for (int v = 1; v < 4; ++v) {
        double w = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) w = fmax(w, fabs(r[v][l][i] - r[0][l][i]));
        printf("%s: max |diff| vs disjoint registers %g (largest reference value %g)\n", nm[v], w, mag);
    }
    return 0;
}
