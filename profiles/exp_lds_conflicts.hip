// Which of the LDS access patterns used by the kernels really are conflict-free on gfx950?  One wave per workgroup issues
// the same access 256 times per loop trip; the printout is shader clocks per access, to compare with the conflict-free
// reference pattern of the same width (consecutive lanes -> consecutive words / 16-byte units).
// build: hipcc --offload-arch=gfx950 -O3 profiles/exp_lds_conflicts.hip -o profiles/_exp/lds_conflicts ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NF 16384      /* floats of LDS (64 KB) */

// pattern id -> float index of this lane (before the moving offset)
__device__ __forceinline__ int addr_of(int pat, int S, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    switch (pat) {
        case 0: return 4 * lane;                       // b128 reference: consecutive 16-byte units
        case 1: return l15 * S + 4 * g;                // b128: row l15, vector g (MFMA B/A operand reads: X tiles, weights, slabs)
        case 2: return (lane >> 4) * S + 4 * (lane & 15);   // b128: 4 rows x 16 consecutive vectors (tile / weight stores)
        case 3: return lane;                           // b32 reference
        case 4: return l15 * S + g;                    // b32: row l15, column g (attention K/Q operand reads)
        case 5: return (4 * g) * S + l15;              // b32: row 4 g, column l15 (V reads, transposed weight reads, slab stores)
        case 6: return l15 * S + 16 * g;               // b32: row l15, column 16 g (F-layout epilogue / tile element reads)
        case 7: return (lane >> 2) * S + 4 * (lane & 3);    // b128: 16 rows x 4 vectors (extra-column stores)
        default: return lane;
    }
}
template <int WIDTH, bool WRITE>
__global__ __launch_bounds__(64) void k(int pat, int S, long long* clk, float* sink) {
    __shared__ __attribute__((aligned(16))) float L[NF];
    const int lane = threadIdx.x;
    for (int i = lane; i < NF; i += 64) L[i] = (float)i;
    __syncthreads();
    int a = addr_of(pat, S, lane) & ~(WIDTH - 1);
    float acc = 0.f;
    f32x4 acc4 = {0, 0, 0, 0};
    const long long c0 = clock64();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int o = (a + 64 * u + 4 * it) & (NF - 4);
            if (WRITE) {
                if (WIDTH == 4) *(f32x4*)&L[o] = acc4; else L[o] = acc;
            } else {
                if (WIDTH == 4) acc4 += *(const f32x4*)&L[o]; else acc += L[o];
            }
        }
        if (WRITE) { acc += 1.f; acc4 += 1.f; }
    }
    __syncthreads();
    const long long c1 = clock64();
    if (lane == 0) clk[blockIdx.x] = c1 - c0;
    sink[blockIdx.x * 64 + lane] = acc + acc4[0] + acc4[1] + acc4[2] + acc4[3] + L[lane];
}
int main() {
    long long* clk; float* sink;
    hipMalloc(&clk, 8 * 16); hipMalloc(&sink, 4 * 64 * 16);
    struct T { const char* name; int pat, width; bool write; int strides[8]; };
    const T tests[] = {
        {"b128 read  reference (consecutive)", 0, 4, false, {0}},
        {"b128 read  row l15, vector g", 1, 4, false, {64, 68, 72, 80, 84, 132, 136, 0}},
        {"b128 write reference (consecutive)", 0, 4, true, {0}},
        {"b128 write 4 rows x 16 vectors", 2, 4, true, {64, 68, 80, 84, 132, 0}},
        {"b128 write 16 rows x 4 vectors", 7, 4, true, {68, 84, 132, 0}},
        {"b32  read  reference (consecutive)", 3, 1, false, {0}},
        {"b32  read  row l15, column g", 4, 1, false, {64, 68, 84, 0}},
        {"b32  read  row 4g, column l15", 5, 1, false, {45, 68, 84, 132, 136, 0}},
        {"b32  read  row l15, column 16 g", 6, 1, false, {45, 68, 84, 0}},
        {"b32  write reference (consecutive)", 3, 1, true, {0}},
        {"b32  write row 4g, column l15", 5, 1, true, {68, 132, 136, 0}},
    };
    for (const T& t : tests) {
        printf("%-40s", t.name);
        for (int si = 0; si < 8; ++si) {
            const int S = t.strides[si];
            if (si > 0 && S == 0) break;
            for (int rep = 0; rep < 2; ++rep) {
                if (t.width == 4 && !t.write) hipLaunchKernelGGL((k<4, false>), dim3(1), dim3(64), 0, 0, t.pat, S, clk, sink);
                if (t.width == 4 && t.write) hipLaunchKernelGGL((k<4, true>), dim3(1), dim3(64), 0, 0, t.pat, S, clk, sink);
                if (t.width == 1 && !t.write) hipLaunchKernelGGL((k<1, false>), dim3(1), dim3(64), 0, 0, t.pat, S, clk, sink);
                if (t.width == 1 && t.write) hipLaunchKernelGGL((k<1, true>), dim3(1), dim3(64), 0, 0, t.pat, S, clk, sink);
            }
            long long h = 0;
            hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
            printf("  S=%-3d %6.1f", S, (double)h / (64 * 16));
        }
        printf("   clocks/access\n");
    }
    return 0;
}
