// micro-benchmark: how fast can every CU pull the SAME weight block (L2-resident) - in lockstep order vs rotated order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ w, int n4, int mode, int per_wg_stride4, float* out, int reps) {
    // n4 float4 per block; mode 0: same order; 1: rotated start per block; 2: private copy per block
    const float4* base = w + (mode == 2 ? (size_t)blockIdx.x * per_wg_stride4 : 0);
    const int chunk = 256 * INFLIGHT;
    const int nch = n4 / chunk;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        int start = 0;
        if (mode == 1) start = (int)((blockIdx.x * 7u + r) % (unsigned)nch);
        if (mode == 3) start = (int)(((blockIdx.x >> 3) * 5u + r) % (unsigned)nch);
        for (int c = 0; c < nch; ++c) {
            int cc = c + start; if (cc >= nch) cc -= nch;
            float4 v[INFLIGHT];
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i) v[i] = base[(size_t)cc * chunk + i * 256 + threadIdx.x];
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int INF>
void run(const float4* w, int n4, int stride4, float* out, int wgs, const char* tag) {
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        hipLaunchKernelGGL(k_stream<INF>, dim3(wgs), dim3(256), 0, 0, w, n4, mode, stride4, out, 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stream<INF>, dim3(wgs), dim3(256), 0, 0, w, n4, mode, stride4, out, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double bytes = (double)n4 * 16 * reps;       // per WG
        double us_per_pass = ms * 1e3 / reps;
        printf("%s inflight %d wgs %d mode %d: %.2f us per %.0f KB pass = %.1f B/clk/CU (2.4 GHz), aggregate %.2f TB/s\n", tag, INF, wgs, mode,
               us_per_pass, n4 * 16 / 1024.0, bytes / reps / (us_per_pass * 1e-6) / 2.4e9, bytes * wgs / (ms * 1e-3) / 1e12);
    }
}
int main() {
    const int n4 = 168 * 1024 / 16;      // 168 KB
    const int n4r = n4 / (256 * 8) * (256 * 8);
    const int wgs_max = 1024;
    float4* w; float* out;
    CK(hipMalloc(&w, (size_t)n4r * 16 * wgs_max)); CK(hipMalloc(&out, 4));
    CK(hipMemset(w, 0, (size_t)n4r * 16 * wgs_max));
    run<4>(w, n4r, n4r, out, 256, "168KB");
    run<8>(w, n4r, n4r, out, 256, "168KB");
    run<8>(w, n4r, n4r, out, 200, "168KB");
    run<8>(w, n4r, n4r, out, 512, "168KB");
    run<16>(w, n4r / 2 * 2, n4r, out, 256, "168KB");
    return 0;
}
