// Experiment: does HIP on gfx950 pass kernel arguments beyond 4 KB?  (k_atb's unit table: 128 x 112 B = 14 KB, so that
// the weight-gradient GEMMs of a whole backward pass are one launch.)  Result on MI355X / ROCm 7.0.2: yes -
// "err=0 (no error) h[1]=3039.000000 h[63]=5519.000000" with this 12 KB argument.
// build: hipcc --offload-arch=gfx950 -O2 profiles/exp_kernarg.hip -o profiles/_exp/kernarg_12k ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float v[3000]; };   // 12 KB
__global__ void k(Big b, float* out) { out[threadIdx.x] = b.v[threadIdx.x * 40] + b.v[2999]; }
int main() {
    Big b;
    for (int i = 0; i < 3000; ++i) b.v[i] = (float)i;
    float* d;
    if (hipMalloc(&d, 256) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipDeviceSynchronize();
    float h[64];
    if (hipMemcpy(h, d, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("err=%d (%s) h[1]=%f h[63]=%f\n", (int)e, hipGetErrorString(e), h[1], h[63]);
    return 0;
}
