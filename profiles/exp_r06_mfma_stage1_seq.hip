// The instruction neighbourhood of the ONE v_mfma_f32_16x16x32_bf16 whose operand guard alone restores run-to-run determinism of
// k_edge_bwd<bf16, dropout> (profiles/r06_zzm_stage1_asm.txt), replayed with fixed registers under load (eight waves per
// workgroup, many workgroups):
//   A out of LDS -> srcC written by VALU selects -> x32 MFMA with vdst == srcA (!= srcC) -> an LDS read INTO srcC's registers
//   four instructions later -> a dozen VALU selects -> two more x32 MFMAs (one consuming that LDS read) -> a 16x16x16 MFMA
//   whose srcC is the first MFMA's result.
// Each iteration runs the sequence twice - as compiled (H) and with the first destination on fresh registers and long waits
// everywhere (R) - and counts result words that differ.
//   hipcc --offload-arch=gfx950 -O2 profiles/exp_r06_mfma_stage1_seq.hip -o /tmp/seq && /tmp/seq
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const s16x8* a, const s16x8* b, int iters, unsigned* mismatches) {
    __shared__ f32x4 la[512], lc[512];
    const int t = threadIdx.x;
    la[t] = __builtin_bit_cast(f32x4, a[(blockIdx.x * 512 + t) % 4096]);
    lc[t] = __builtin_bit_cast(f32x4, b[(blockIdx.x * 317 + t) % 4096]);
    __syncthreads();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int i = (t + 64 * it) & 511, j = (t + 192 * it + 5) & 511;
        const f32x4 bv = la[(i + 7 * it + 64) & 511];
        const f32x4 cv = {(float)(it & 7), 1.f, -2.f, 0.5f * (float)(t & 3)};
        const unsigned pa = (unsigned)(size_t)&la[i], pc = (unsigned)(size_t)&lc[j];
        f32x4 h, r;
        asm volatile(
            "v_cmp_ne_u32 vcc, 0, %6\n"
            "ds_read_b64 v[100:101], %2\n ds_read_b64 v[102:103], %2 offset:8\n"
            "v_mov_b32 v110, %4\n v_mov_b32 v111, %4\n v_mov_b32 v112, %4\n v_mov_b32 v113, %4\n"
            "v_mov_b32 v114, %5\n v_mov_b32 v115, %5\n v_mov_b32 v116, %5\n v_mov_b32 v117, %5\n"
            "v_cndmask_b32 v104, 0, %4, vcc\n v_cndmask_b32 v105, 0, %5, vcc\n v_cndmask_b32 v106, 0, %4, vcc\n v_cndmask_b32 v107, 0, %5, vcc\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_add_f32 v120, %4, %5\n v_add_f32 v121, %5, %4\n"
            "v_mfma_f32_16x16x32_bf16 v[100:103], v[100:103], %1, v[104:107]\n"          // M1: vdst == srcA
            "v_add_f32 v120, v120, v121\n v_add_f32 v121, v121, v120\n v_cndmask_b32 v122, 0, v120, vcc\n"
            "ds_read_b128 v[104:107], %3\n"                                              // LDS read into M1's srcC
            "v_cndmask_b32 v123, 0, v121, vcc\n v_cndmask_b32 v122, 0, v122, vcc\n v_cndmask_b32 v123, 0, v123, vcc\n v_cndmask_b32 v122, 0, v122, vcc\n"
            "v_cndmask_b32 v123, 0, v123, vcc\n v_cndmask_b32 v122, 0, v122, vcc\n v_cndmask_b32 v123, 0, v123, vcc\n v_cndmask_b32 v122, 0, v122, vcc\n"
            "v_cndmask_b32 v123, 0, v123, vcc\n v_cndmask_b32 v122, 0, v122, vcc\n v_cndmask_b32 v123, 0, v123, vcc\n v_cndmask_b32 v122, 0, v122, vcc\n"
            "v_mfma_f32_16x16x32_bf16 v[110:113], %1, %1, v[110:113]\n"                  // M2
            "s_waitcnt lgkmcnt(0)\n"
            "v_mfma_f32_16x16x32_bf16 v[114:117], v[104:107], %1, v[114:117]\n"          // M3: A = the LDS read
            "s_nop 7\n"
            "v_mfma_f32_16x16x16_bf16 v[124:127], v[110:111], v[114:115], v[100:103]\n"  // srcC = M1's result, vdst elsewhere
            "s_nop 15\n s_nop 15\n"
            "v_mov_b32 %0, v124\n v_add_f32 %0, %0, v125\n v_add_f32 %0, %0, v126\n v_add_f32 %0, %0, v127\n"
            : "=&v"(h[0]) : "v"(bv), "v"(pa), "v"(pc), "v"(cv[0]), "v"(cv[3]), "v"(1u)
            : "memory", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115",
              "v116", "v117", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        asm volatile(
            "v_cmp_ne_u32 vcc, 0, %6\n"
            "ds_read_b64 v[100:101], %2\n ds_read_b64 v[102:103], %2 offset:8\n"
            "v_mov_b32 v110, %4\n v_mov_b32 v111, %4\n v_mov_b32 v112, %4\n v_mov_b32 v113, %4\n"
            "v_mov_b32 v114, %5\n v_mov_b32 v115, %5\n v_mov_b32 v116, %5\n v_mov_b32 v117, %5\n"
            "v_cndmask_b32 v104, 0, %4, vcc\n v_cndmask_b32 v105, 0, %5, vcc\n v_cndmask_b32 v106, 0, %4, vcc\n v_cndmask_b32 v107, 0, %5, vcc\n"
            "s_waitcnt lgkmcnt(0)\n s_nop 15\n"
            "v_mfma_f32_16x16x32_bf16 v[130:133], v[100:103], %1, v[104:107]\n"          // fresh destination
            "s_nop 15\n s_nop 15\n"
            "ds_read_b128 v[104:107], %3\n"
            "v_mfma_f32_16x16x32_bf16 v[110:113], %1, %1, v[110:113]\n"
            "s_waitcnt lgkmcnt(0)\n s_nop 15\n"
            "v_mfma_f32_16x16x32_bf16 v[114:117], v[104:107], %1, v[114:117]\n"
            "s_nop 15\n s_nop 15\n"
            "v_mfma_f32_16x16x16_bf16 v[124:127], v[110:111], v[114:115], v[130:133]\n"
            "s_nop 15\n s_nop 15\n"
            "v_mov_b32 %0, v124\n v_add_f32 %0, %0, v125\n v_add_f32 %0, %0, v126\n v_add_f32 %0, %0, v127\n"
            : "=&v"(r[0]) : "v"(bv), "v"(pa), "v"(pc), "v"(cv[0]), "v"(cv[3]), "v"(1u)
            : "memory", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115",
              "v116", "v117", "v124", "v125", "v126", "v127", "v130", "v131", "v132", "v133");
        bad += __builtin_bit_cast(unsigned, h[0]) != __builtin_bit_cast(unsigned, r[0]);
    }
    if (bad) atomicAdd(mismatches, bad);
}
int main() {
    s16x8 *ha = (s16x8*)malloc(4096 * 16), *hb = (s16x8*)malloc(4096 * 16);
    for (int l = 0; l < 4096; ++l) for (int i = 0; i < 8; ++i) {
        float f = (float)((rand() % 2001) - 1000) / 500.f; unsigned u; memcpy(&u, &f, 4); ha[l][i] = (short)(u >> 16);
        f = (float)((rand() % 2001) - 1000) / 500.f; memcpy(&u, &f, 4); hb[l][i] = (short)(u >> 16);
    }
    s16x8 *da, *db; unsigned* dm;
    (void)hipMalloc(&da, 4096 * 16); (void)hipMalloc(&db, 4096 * 16); (void)hipMalloc(&dm, 4);
    (void)hipMemcpy(da, ha, 4096 * 16, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 4096 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {1, 256, 2048}) {
            (void)hipMemset(dm, 0, 4);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, da, db, 300, dm);
            unsigned m = 0;
            (void)hipMemcpy(&m, dm, 4, hipMemcpyDeviceToHost);
            printf("workgroups %4d x 8 waves, 300 iterations: %u of %llu lane results differ between the compiled sequence and the spaced one\n",
                   blocks, m, (unsigned long long)blocks * 512ull * 300ull);
        }
    return 0;
}
