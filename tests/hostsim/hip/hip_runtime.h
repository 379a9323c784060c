// TEST INFRASTRUCTURE ONLY -- host simulator of the small HIP subset the kernels in
// equidock_public_amd/csrc use, so that the very same kernel sources can be compiled for x86
// (clang++ -x c++ -I tests/hostsim) and their index arithmetic checked against the oracle in a
// container without a GPU.  The product never builds or loads this; see tests/hostsim/README.md.
//
// Execution model: one OS thread; every GPU thread of a workgroup is a fiber; cross-lane
// operations (__shfl*, MFMA, __syncthreads) are rendezvous points between the fibers of a
// wave (64 consecutive threads) or of the workgroup.  MFMA follows the gfx950 register layout
// of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md section 3):
//   A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
//   D[row = 4 * (lane >> 4) + reg][col = lane & 15].
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hostsim_uint3 { unsigned x, y, z; };
extern hostsim_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n); return hipSuccess;
}

#define EQD_NUM_CUS_FIXED 256   /* the simulated device: MI355X */
#define EQD_HOSTSIM 1
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

namespace hostsim {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
float shfl_f(float v, int src_lane);
int shfl_i(int v, int src_lane);
void mfma16x16x4(float a, float b, const float* c_in, float* d_out);
void mfma16x16x16bf16(const unsigned short* a, const unsigned short* b, const float* c_in, float* d_out);
int lane_id();
void wave_barrier();
}  // namespace hostsim

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hostsim::launch(dim3(grid), dim3(block), [&]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { hostsim::sync_block(); }
static inline void __threadfence() {}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
#define EQD_KERNARG_PTR(first_param) ((const void*)&(first_param))   /* host: the by-value argument itself */
#define EQD_GAS   /* no address spaces on the host */
#define EQD_NATIVE_EXP(x) expf(x)   /* the device build uses __expf (v_exp_f32) */
static inline float __shfl_xor(float v, int m) { return hostsim::shfl_f(v, hostsim::lane_id() ^ m); }
static inline int __shfl_xor(int v, int m) { return hostsim::shfl_i(v, hostsim::lane_id() ^ m); }
static inline float __shfl(float v, int l) { return hostsim::shfl_f(v, l); }
static inline int __shfl(int v, int l) { return hostsim::shfl_i(v, l); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }

typedef float hostsim_f32x4 __attribute__((ext_vector_type(4)));
static inline hostsim_f32x4 hostsim_mfma(float a, float b, hostsim_f32x4 c, int, int, int) {
    float ci[4] = {c[0], c[1], c[2], c[3]}, d[4];
    hostsim::mfma16x16x4(a, b, ci, d);
    hostsim_f32x4 r = {d[0], d[1], d[2], d[3]};
    return r;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hostsim_mfma
typedef short hostsim_s16x4 __attribute__((ext_vector_type(4)));
static inline hostsim_f32x4 hostsim_mfma_bf16(hostsim_s16x4 a, hostsim_s16x4 b, hostsim_f32x4 c, int, int, int) {
    unsigned short ai[4], bi[4];
    for (int i = 0; i < 4; ++i) {
        ai[i] = (unsigned short)a[i];
        bi[i] = (unsigned short)b[i];
    }
    float ci[4] = {c[0], c[1], c[2], c[3]}, d[4];
    hostsim::mfma16x16x16bf16(ai, bi, ci, d);
    hostsim_f32x4 r = {d[0], d[1], d[2], d[3]};
    return r;
}
#define __builtin_amdgcn_mfma_f32_16x16x16bf16_1k hostsim_mfma_bf16
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() hostsim::wave_barrier()
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)   /* only applied to wave-uniform values */
#define __builtin_amdgcn_readlane(v, l) hostsim::shfl_i((v), (l))
