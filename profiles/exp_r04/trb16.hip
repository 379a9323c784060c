// what ds_read_b64_tr_b16 returns: LDS holds lds[i] = i (16-bit); lane l reads at short-index addr[l]; dump the 4 values per lane
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
    *(s16x4*)&out[4 * threadIdx.x] = v;
}
int main() {
    int h[64]; short o[256]; int* d; short* od;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&od, sizeof(o));
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h[l] = 4 * l;                         // lane l: shorts 4l .. 4l+3 (linear)
            if (pat == 1) h[l] = 100 * (l & 15) + 4 * (l >> 4);  // row-major tile: row = l15 (stride 100), columns 4g..4g+3
            if (pat == 2) h[l] = 1000 * (l >> 4) + 64 * ((l & 15) >> 2) + 4 * (l & 3);   // rows = (l15>>2) (stride 64), col group l&3
        }
        (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
        (void)hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %5d %5d %5d %5d\n", l, h[l], o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    }
    return 0;
}
