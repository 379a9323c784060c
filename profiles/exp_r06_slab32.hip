// round 6: the weight-gradient slab GEMM of k_edge_bwd (fp32) on v_mfma_f32_32x32x2_f32 instead of
// v_mfma_f32_16x16x4_f32 - priced in isolation before touching the kernel.
//   slabs: X, Y = [64 features][128 edges] fp32 in LDS, row stride UST = 132 (k_edge_bwd's layout); out = X Y^T (64 x 64)
//   form A (ships): wave w owns the 16 x 32 block (mb = w >> 1, nb0 = 2 (w & 1)) over all 128 edges:
//                   per 16 edges 1 + 2 b128 fragment reads and 8 MFMAs (16x16x4: 8 passes each)      -> 24 reads, 64 MFMAs
//   form B:         wave w owns the 32 x 32 block (mb = w & 1, nb = (w >> 1) & 1) over HALF the edges (w >> 2):
//                   per 8 edges 1 + 1 b128 fragment reads and 4 MFMAs (32x32x2: 16 passes each)      -> 16 reads, 32 MFMAs
// Same MFMA pipe time (2 048 clocks per wave); B issues 2/3 of the LDS reads and half the MFMA instructions, and needs 16
// accumulator registers per matrix instead of 8 (+ a final add of the two edge halves).
// build: hipcc --offload-arch=gfx950 -O3 profiles/exp_r06_slab32.hip -o profiles/_exp/slab32 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define UST 132
#define REPS 64

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int FORM>
__global__ __launch_bounds__(512) void k(const float* __restrict__ Xg, const float* __restrict__ Yg, float* __restrict__ out,
                                         long long* __restrict__ clk) {
    __shared__ __attribute__((aligned(16))) float X[64 * UST], Y[64 * UST];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    for (int i = t; i < 64 * 128; i += 512) {
        X[(i >> 7) * UST + (i & 127)] = Xg[i];
        Y[(i >> 7) * UST + (i & 127)] = Yg[i];
    }
    __syncthreads();
    long long c0 = 0, c1 = 0;
    if constexpr (FORM == 0) {
        const int mb = wave >> 1, nb0 = 2 * (wave & 1);
        f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        c0 = clock64();
        for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll 2
            for (int kc = 0; kc < 8; ++kc) {
                const f32x4 a = *(const f32x4*)&X[(16 * mb + l15) * UST + 16 * kc + 4 * g];
                f32x4 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *(const f32x4*)&Y[(16 * (nb0 + j) + l15) * UST + 16 * kc + 4 * g];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = mfma4(a[u], b[j][u], acc[j]);
            }
        }
        c1 = clock64();
        // D layout of 16x16x4: lane (l15, g) holds rows 4 g + r (of the A operand's M axis), column l15
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * mb + 4 * g + r) * 64 + 16 * (nb0 + j) + l15] = acc[j][r];
    } else {
        const int mb = wave & 1, nb = (wave >> 1) & 1, kh = wave >> 2;
        const int i31 = lane & 31, h = lane >> 5;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        c0 = clock64();
        for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll 2
            for (int kc = 0; kc < 8; ++kc) {
                const f32x4 a = *(const f32x4*)&X[(32 * mb + i31) * UST + 64 * kh + 8 * kc + 4 * h];
                const f32x4 b = *(const f32x4*)&Y[(32 * nb + i31) * UST + 64 * kh + 8 * kc + 4 * h];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            }
        }
        c1 = clock64();
        // D layout of 32x32x2: register 4 q + r of lane (i31, h) holds row 8 q + 4 h + r, column i31; the two edge halves
        // (kh) are added through LDS: kh = 1 parks its block, kh = 0 adds and stores
        __syncthreads();
        float* park = X;      // (the slabs are dead)
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) park[((wave & 3) * 16 + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(32 * mb + 8 * q + 4 * h + r) * 64 + 32 * nb + i31] = acc[4 * q + r] + park[((wave & 3) * 16 + 4 * q + r) * 64 + lane];
        }
    }
    if (lane == 0) clk[blockIdx.x * 8 + wave] = c1 - c0;
}

int main() {
    const int n = 64 * 128;
    float *hx = (float*)malloc(n * 4), *hy = (float*)malloc(n * 4), *ho = (float*)malloc(4096 * 4);
    srand(3);
    for (int i = 0; i < n; ++i) { hx[i] = (rand() % 2001 - 1000) * 1e-3f; hy[i] = (rand() % 2001 - 1000) * 1e-3f; }
    float *dx, *dy, *dout; long long* dclk;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, 4096 * 4); hipMalloc(&dclk, 8 * 8 * 256);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, hy, n * 4, hipMemcpyHostToDevice);
    for (int form = 0; form < 2; ++form) {
        for (int blocks : {1, 256}) {
            hipMemset(dout, 0, 4096 * 4);
            for (int it = 0; it < 3; ++it) {
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, dx, dy, dout, dclk);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, dx, dy, dout, dclk);
            }
            hipDeviceSynchronize();
            long long hc[8];
            hipMemcpy(hc, dclk, 64, hipMemcpyDeviceToHost);
            hipMemcpy(ho, dout, 4096 * 4, hipMemcpyDeviceToHost);
            double worst = 0;      // out = REPS * X Y^T
            for (int i = 0; i < 64; ++i)
                for (int j = 0; j < 64; ++j) {
                    double s = 0;
                    for (int e = 0; e < 128; ++e) s += (double)hx[i * 128 + e] * hy[j * 128 + e];
                    worst = fmax(worst, fabs(ho[i * 64 + j] - REPS * s) / (fabs(REPS * s) + 1.0));
                }
            long long mx = 0;
            for (int w = 0; w < 8; ++w) mx = hc[w] > mx ? hc[w] : mx;
            printf("form %s, %3d workgroups: %.0f shader clocks per slab GEMM per wave (slowest wave of workgroup 0; MFMA pipe minimum 2048), "
                   "max rel err vs fp64 %.1e\n", form == 0 ? "A 16x16x4 (16x32 per wave, 128 edges)" : "B 32x32x2 (32x32 per wave,  64 edges)",
                   blocks, (double)mx / REPS, worst);
        }
    }
    return 0;
}
