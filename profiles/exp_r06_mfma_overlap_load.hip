// v_mfma_f32_16x16x32_bf16 with vdst == srcA UNDER LOAD: eight waves per workgroup (two per SIMD, sharing the matrix pipe), many
// workgroups, the A operand arriving from LDS right in front of the instruction, srcC written by a VALU instruction just before -
// the neighbourhood the instruction has in k_edge_bwd<bf16, dropout>, where builds carrying it were run-to-run nondeterministic
// (HISTORY.md, round 6).  Every iteration computes the product twice - destination over A, destination on fresh registers - and
// counts lanes whose results differ.   hipcc --offload-arch=gfx950 -O2 profiles/exp_r06_mfma_overlap_load.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const s16x8* a, const s16x8* b, int iters, int filler, unsigned* mismatches, float* sink) {
    __shared__ f32x4 la[512], lb[512];
    const int t = threadIdx.x;
    la[t] = __builtin_bit_cast(f32x4, a[(blockIdx.x * 512 + t) % 4096]);
    lb[t] = __builtin_bit_cast(f32x4, b[(blockIdx.x * 512 + t) % 4096]);
    __syncthreads();
    unsigned bad = 0;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const int i = (t + 64 * it) & 511;
        f32x4 bv = lb[(i + 7 * it) & 511];
        f32x4 cv = {(float)(it & 7), 1.f, -2.f, 0.5f * (float)(t & 3)};
        f32x4 r, ra;
        // destination over A: A comes out of LDS straight into the registers the instruction overwrites
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n"
                     "v_mfma_f32_16x16x32_bf16 %0, %0, %2, %3\n"
                     : "=&v"(ra) : "v"((unsigned)(size_t)&la[i]), "v"(bv), "v"(cv) : "memory");
        // filler MFMAs of the same wave right behind it (independent): the matrix pipe stays busy while the first completes
        f32x4 f0 = cv, f1 = cv;
        for (int q = 0; q < filler; ++q)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %3, %2, %1" : "+v"(f0), "+v"(f1) : "v"(bv), "v"(cv));
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
        const f32x4 av = la[i];
        asm volatile("s_nop 4\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3\n s_nop 15\n s_nop 15" : "=&v"(r) : "v"(av), "v"(bv), "v"(cv));
        asm volatile("s_nop 15" ::: "memory");
        for (int e = 0; e < 4; ++e) bad += __builtin_bit_cast(unsigned, r[e]) != __builtin_bit_cast(unsigned, ra[e]);
        keep += f0 + f1;
    }
    if (bad) atomicAdd(mismatches, bad);
    if (t == 0) sink[blockIdx.x] = keep[0];
}
int main() {
    s16x8 *ha = (s16x8*)malloc(4096 * 16), *hb = (s16x8*)malloc(4096 * 16);
    for (int l = 0; l < 4096; ++l) for (int i = 0; i < 8; ++i) {
        float f = (float)((rand() % 2001) - 1000) / 500.f; unsigned u; memcpy(&u, &f, 4); ha[l][i] = (short)(u >> 16);
        f = (float)((rand() % 2001) - 1000) / 500.f; memcpy(&u, &f, 4); hb[l][i] = (short)(u >> 16);
    }
    s16x8 *da, *db; unsigned* dm; float* ds;
    (void)hipMalloc(&da, 4096 * 16); (void)hipMalloc(&db, 4096 * 16); (void)hipMalloc(&dm, 4); (void)hipMalloc(&ds, 4096 * 4);
    (void)hipMemcpy(da, ha, 4096 * 16, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 4096 * 16, hipMemcpyHostToDevice);
    for (int filler = 0; filler <= 4; filler += 2)
        for (int blocks : {1, 256, 1024}) {
            (void)hipMemset(dm, 0, 4);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, da, db, 400, filler, dm, ds);
            unsigned m = 0;
            (void)hipMemcpy(&m, dm, 4, hipMemcpyDeviceToHost);
            printf("filler MFMA pairs %d, workgroups %4d x 8 waves, 400 iterations: %u of %llu result words differ between vdst == srcA and fresh registers\n",
                   filler, blocks, m, (unsigned long long)blocks * 512ull * 400ull * 4ull);
        }
    return 0;
}
