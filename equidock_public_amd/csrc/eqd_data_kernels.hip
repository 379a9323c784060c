// Data-side kernels around the hot path: the per-item random rigid augmentation of
// the ligand (src/utils/db5_data.py:195-204), for the whole batch on the device.
#include "eqd_common.h"

// One workgroup per pair: mean of the pair's ligand coordinates (fixed-order tree), then
//   new_x_i = R_p (x_i - mean) + t_p          (src/utils/db5_data.py:197-202: (rot_T @ (x - mean).T).T + rot_b)
// and the same map for the pair's pocket coordinates when given (:201).
__global__ __launch_bounds__(EQD_BLOCK) void k_rigid_augment(const int32_t* __restrict__ seg_off,
                                                             const float* __restrict__ x, const float* __restrict__ R,
                                                             const float* __restrict__ t, float* __restrict__ new_x,
                                                             const int32_t* __restrict__ pocket_off,
                                                             const float* __restrict__ pocket_in,
                                                             float* __restrict__ pocket_out) {
    __shared__ float red[3][EQD_WAVES];
    __shared__ float mean[3];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n0 = seg_off[p], n1 = seg_off[p + 1];
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = n0 + tid; i < n1; i += EQD_BLOCK)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] += x[(size_t)i * 3 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float w = wave_sum(s[c]);
        if ((tid & 63) == 0) red[c][tid >> 6] = w;
    }
    __syncthreads();
    if (tid < 3) mean[tid] = n1 > n0 ? ((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3])) / (float)(n1 - n0) : 0.f;
    __syncthreads();
    float r[9], tt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = R[(size_t)p * 9 + k];
#pragma unroll
    for (int c = 0; c < 3; ++c) tt[c] = t[(size_t)p * 3 + c];
    const float m0 = mean[0], m1 = mean[1], m2 = mean[2];
    for (int i = n0 + tid; i < n1; i += EQD_BLOCK) {
        const float a = x[(size_t)i * 3] - m0, b = x[(size_t)i * 3 + 1] - m1, c = x[(size_t)i * 3 + 2] - m2;
        new_x[(size_t)i * 3] = (r[0] * a + r[1] * b) + r[2] * c + tt[0];
        new_x[(size_t)i * 3 + 1] = (r[3] * a + r[4] * b) + r[5] * c + tt[1];
        new_x[(size_t)i * 3 + 2] = (r[6] * a + r[7] * b) + r[8] * c + tt[2];
    }
    if (pocket_off) {
        const int q0 = pocket_off[p], q1 = pocket_off[p + 1];
        for (int i = q0 + tid; i < q1; i += EQD_BLOCK) {
            const float a = pocket_in[(size_t)i * 3] - m0, b = pocket_in[(size_t)i * 3 + 1] - m1,
                        c = pocket_in[(size_t)i * 3 + 2] - m2;
            pocket_out[(size_t)i * 3] = (r[0] * a + r[1] * b) + r[2] * c + tt[0];
            pocket_out[(size_t)i * 3 + 1] = (r[3] * a + r[4] * b) + r[5] * c + tt[1];
            pocket_out[(size_t)i * 3 + 2] = (r[6] * a + r[7] * b) + r[8] * c + tt[2];
        }
    }
}

extern "C" int eqd_rigid_augment(const EqdGraph* g, const float* x_lig, const float* R, const float* t, float* new_x,
                                 const int32_t* pocket_off, const float* pocket_in, float* pocket_out, void* stream) {
    if (!g || !x_lig || !R || !t || !new_x || (pocket_off && (!pocket_in || !pocket_out))) {
        eqd_set_error("eqd_rigid_augment: NULL argument");
        return EQD_ERR_NULL;
    }
    if (g->n_pairs <= 0) return EQD_OK;
    hipLaunchKernelGGL(k_rigid_augment, dim3(g->n_pairs), dim3(EQD_BLOCK), 0, (hipStream_t)stream, g->seg_off, x_lig, R, t,
                       new_x, pocket_off, pocket_in, pocket_out);
    return eqd_check_launch("k_rigid_augment");
}
