// micro-benchmark 2: the SAME 160 KB weight block read in order by every workgroup of a launch, from a COLD L2 (every launch
// starts cold; a 256 MB sweep in between makes sure) - plain lockstep stream vs a cooperative warm-up in which workgroup
// b first touches slice (b / 8) % 32 of the block (so that the XCD's L2 has the whole block in flight at once)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int INFLIGHT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_stream(const float4* __restrict__ w, int n4, int mode, float* out, long long* clk) {
    const long long t0 = clock64();
    const int chunk = THREADS * INFLIGHT;
    const int nch = n4 / chunk;
    float acc = 0.f;
    float4 pf[4];
    if (mode >= 1) {        // warm-up: this workgroup's slice of the block (n4 / 32 float4 = 5 KB), results used at the end
        const int slice = (blockIdx.x >> 3) & 31;
        const int per = n4 / 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = slice * per + i * THREADS + threadIdx.x;
            pf[i] = (i * THREADS + (int)threadIdx.x < per) ? w[idx] : make_float4(0, 0, 0, 0);
        }
    }
    if (mode == 2) {        // ... and wait for the warm-up before the in-order stream starts
        acc += pf[0].x + pf[1].x + pf[2].x + pf[3].x;
        __syncthreads();
    }
    for (int c = 0; c < nch; ++c) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) v[i] = w[(size_t)c * chunk + i * THREADS + threadIdx.x];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (mode == 1) acc += pf[0].x + pf[1].x + pf[2].x + pf[3].x;
    if (acc == 12345.678f) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) clk[blockIdx.x] = clock64() - t0;
}
__global__ void k_sweep(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i]; v.x += 1.f; p[i] = v;
    }
}
template <int INF, int THREADS>
void run(float4* w, int n4, float* out, long long* clk, int wgs, float4* big, size_t nbig, bool sweep) {
    std::vector<long long> h(wgs);
    for (int mode = 0; mode < 3; ++mode) {
        double mean = 0, mx = 0, ev = 0; const int reps = 10;
        for (int r = 0; r < reps + 1; ++r) {
            if (sweep) hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, 0, big, nbig);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_stream<INF, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, w, n4, mode, out, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), clk, wgs * 8, hipMemcpyDeviceToHost));
            if (r == 0) continue;
            double s = 0, m = 0; for (auto v : h) { s += v; m = std::max<double>(m, v); }
            mean += s / wgs / reps; mx += m / reps; ev += ms * 1e3 / reps;
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("threads %d inflight %d wgs %d %s mode %d: per-WG clock64 ticks mean %.0f max %.0f (160 KB => %.1f B/tick at the mean), event %.1f us\n",
               THREADS, INF, wgs, sweep ? "cold+sweep" : "back-to-back", mode, mean, mx, n4 * 16.0 / mean, ev);
    }
}
int main() {
    const int n4 = 160 * 1024 / 16;
    float4 *w, *big; float* out; long long* clk;
    const size_t nbig = (size_t)512 * 1024 * 1024 / 16;
    CK(hipMalloc(&w, (size_t)n4 * 16)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&clk, 8 * 4096)); CK(hipMalloc(&big, nbig * 16));
    CK(hipMemset(w, 0, (size_t)n4 * 16)); CK(hipMemset(big, 0, nbig * 16));
    run<4, 256>(w, n4, out, clk, 200, big, nbig, false);
    run<4, 256>(w, n4, out, clk, 200, big, nbig, true);
    run<8, 256>(w, n4, out, clk, 200, big, nbig, true);
    run<2, 512>(w, n4, out, clk, 250, big, nbig, true);
    run<4, 512>(w, n4, out, clk, 250, big, nbig, true);
    return 0;
}
