// lane ^ m exchanges without the LDS pipe: DPP (m = 1, 2, 4, 8) and v_permlane{16,32}_swap (m = 16, 32) against __shfl_xor
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ __forceinline__ int dppc(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x, v = 1000 + l;
    out[0 * 64 + l] = dppc<0xB1>(v);                      // quad_perm [1,0,3,2]  -> lane ^ 1
    out[1 * 64 + l] = dppc<0x4E>(v);                      // quad_perm [2,3,0,1]  -> lane ^ 2
    out[2 * 64 + l] = dppc<0x1B>(dppc<0x141>(v));         // row_half_mirror then quad reverse -> lane ^ 4
    out[3 * 64 + l] = dppc<0x128>(v);                     // row_ror:8 -> lane ^ 8
    {   // v_permlane16_swap: swaps odd rows of vdst with even rows of src
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        u2 r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        out[4 * 64 + l] = (int)r[0]; out[5 * 64 + l] = (int)r[1];
        u2 q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        out[6 * 64 + l] = (int)q[0]; out[7 * 64 + l] = (int)q[1];
    }
}
int main() {
    int* d; int h[8 * 64];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[8] = {"xor1", "xor2", "xor4", "xor8", "pl16swap[0]", "pl16swap[1]", "pl32swap[0]", "pl32swap[1]"};
    const int m[4] = {1, 2, 4, 8};
    for (int t = 0; t < 4; ++t) { int bad = 0; for (int l = 0; l < 64; ++l) bad += h[t * 64 + l] != 1000 + (l ^ m[t]); printf("%s: %d mismatches\n", nm[t], bad); }
    for (int t = 4; t < 8; ++t) { printf("%s:", nm[t]); for (int l = 0; l < 64; ++l) printf(" %d", h[t * 64 + l] - 1000); printf("\n"); }
    return 0;
}
