// Floor of one "row-kernel step" on gfx950 with ONE wave per SIMD (4 waves per workgroup, 200 workgroups):
//   barrier; 5 b128 LDS stores per thread; barrier; 5 dwordx4 global loads (next step); 8 b128 LDS reads; 16 MFMA (2 chains)
// prints shader clocks per step for variants, to compare with the ~4 900 clocks measured inside k_rowchain.
// build: hipcc --offload-arch=gfx950 -O3 profiles/exp_step_floor.hip -o profiles/_exp/step_floor ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LS 84
typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));
// MODE 0: k-contiguous weights, b128 stores, b128 operand reads.  MODE 1: transposed scalar weight stores (8-way bank
// conflict).  MODE 2: m-contiguous weights with a RUNTIME odd leading dimension (4-byte aligned 16-byte loads), stored as
// Wl[k][m] with b128, A operand = 4 scalar reads per k-group - what a static backward chain would do.
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, const float* __restrict__ X, float* out, long long* clk, int steps,
                                          int ldw) {
    __shared__ __attribute__((aligned(16))) float Xl[16 * LS];
    __shared__ __attribute__((aligned(16))) float Wl[80 * LS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4, tr = t >> 4, tc = t & 15;
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    f32x4 rx, rw[4];
    const float* wp = W + (size_t)tr * (MODE == 2 ? ldw : 64) + 4 * tc;
    const float* xp = X + ((size_t)blockIdx.x * 16 + tr) * 64 + 4 * tc;
    rx = *(const f32x4*)xp;
    for (int j = 0; j < 4; ++j) rw[j] = MODE == 2 ? (f32x4)*(const f4v*)(wp + (size_t)j * 16 * ldw) : *(const f32x4*)(wp + j * 16 * 64);
    const long long c0 = clock64();
    for (int s = 0; s < steps; ++s) {
        __syncthreads();
        *(f32x4*)&Xl[tr * LS + 4 * tc] = rx;
        if (MODE == 1) {       // transposed weight store (16 scalar stores)
            for (int j = 0; j < 4; ++j)
                for (int i = 0; i < 4; ++i) Wl[(4 * tc + i) * LS + tr + 16 * j] = rw[j][i];
        } else {
            for (int j = 0; j < 4; ++j) *(f32x4*)&Wl[(tr + 16 * j) * LS + 4 * tc] = rw[j];
        }
        __syncthreads();
        const float* wq = wp + (size_t)((s + 1) & 7) * (MODE == 2 ? 64 : 4096);
        rx = *(const f32x4*)(xp + ((s + 1) & 3) * 16);
        for (int j = 0; j < 4; ++j)
            rw[j] = MODE == 2 ? (f32x4)*(const f4v*)(wq + (size_t)j * 16 * ldw) : *(const f32x4*)(wq + j * 16 * 64);
        f32x4 a[4], b[4];
        for (int q = 0; q < 4; ++q) {
            if (MODE == 2) {
                for (int j = 0; j < 4; ++j) a[q][j] = Wl[(16 * q + 4 * g + j) * LS + 16 * wave + l15];
            } else {
                a[q] = *(const f32x4*)&Wl[(16 * wave + l15) * LS + 16 * q + 4 * g];
            }
            b[q] = *(const f32x4*)&Xl[l15 * LS + 16 * q + 4 * g];
        }
        for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][0], b[q][0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][1], b[q][1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][2], b[q][2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][3], b[q][3], acc2, 0, 0, 0);
        }
    }
    const long long c1 = clock64();
    out[(size_t)blockIdx.x * 256 + t] = acc[0] + acc2[1] + acc[2] + acc2[3];
    if (blockIdx.x == 0 && t == 0) clk[MODE] = c1 - c0;
}
int main() {
    float *W, *X, *out;
    long long* clk;
    hipMalloc(&W, 64 * 600 * 4 + 4096); hipMalloc(&X, 200 * 16 * 64 * 4 * 2); hipMalloc(&out, 200 * 256 * 4); hipMalloc(&clk, 64);
    hipMemset(W, 0, 64 * 600 * 4 + 4096); hipMemset(X, 0, 200 * 16 * 64 * 4 * 2);
    const int steps = 64;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(200), dim3(256), 0, 0, W, X, out, clk, steps, 64);
        hipLaunchKernelGGL(k<1>, dim3(200), dim3(256), 0, 0, W, X, out, clk, steps, 64);
        hipLaunchKernelGGL(k<2>, dim3(200), dim3(256), 0, 0, W, X, out, clk, steps, 261);
    }
    hipDeviceSynchronize();
    long long h[3];
    hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost);
    printf("step floor (shader clocks per step, 64 steps): b128 weight store %lld, transposed scalar weight store %lld, "
           "m-contiguous weights (ld 261) as Wl[k][m] + scalar operand reads %lld\n", h[0] / steps, h[1] / steps, h[2] / steps);
    return 0;
}
