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

// ---------------------------------------------------------------------------------------------------------------
// Graph construction / featurisation: compute_dig_kNN_graph of the reference
// (src/utils/protein_utils.py:311-397), which is an O(N^2) Python double loop over scipy cdist calls plus per-node and
// per-edge Python loops.  fp64 like the reference's numpy code, so that the neighbour sets and their order - the graph's
// int32 indexing - come out identical (distances agree to 1e-16 relative: only exact ties could order differently).
//   1. eqd_protein_graph_distances: D[i][j] = mean over atoms a of residue i, b of residue j of |a - b|  (:322-329)
//   2. eqd_protein_graph_select   : per residue the sources j with D[i][j] < cutoff in index order, or - when more than
//      max_neighbor qualify - the max_neighbor smallest distances in ascending order (np.argsort, :339-343), plus the
//      surface feature mu_r_norm (:351-359)
//   3. eqd_protein_graph_edges    : destination-major edge list, 15 distance RBFs (:71-86) and the 12 orientation
//      features p, q, k, t in the destination's local frame (:370-387)
// ---------------------------------------------------------------------------------------------------------------
#define PG_MAXATOMS 64
__global__ __launch_bounds__(EQD_BLOCK) void k_pg_distances(int n, const float* __restrict__ atoms,
                                                            const int32_t* __restrict__ atom_off, double* __restrict__ D) {
    __shared__ float ai[PG_MAXATOMS][3];
    const int i = blockIdx.x;
    const int a0 = atom_off[i], a1 = atom_off[i + 1];
    const int na = a1 - a0;
    const bool in_lds = na <= PG_MAXATOMS;
    if (in_lds)
        for (int k = threadIdx.x; k < 3 * na; k += EQD_BLOCK) ai[k / 3][k % 3] = atoms[(size_t)a0 * 3 + k];
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.y == 0) D[(size_t)i * n + i] = __builtin_inf();      // np.full(..., np.inf), :320
    const int j = blockIdx.y * EQD_BLOCK + threadIdx.x;
    if (j <= i || j >= n) return;
    const int b0 = atom_off[j], b1 = atom_off[j + 1];
    double s = 0.0;
    for (int a = 0; a < na; ++a) {
        const double ax = in_lds ? (double)ai[a][0] : (double)atoms[(size_t)(a0 + a) * 3];
        const double ay = in_lds ? (double)ai[a][1] : (double)atoms[(size_t)(a0 + a) * 3 + 1];
        const double az = in_lds ? (double)ai[a][2] : (double)atoms[(size_t)(a0 + a) * 3 + 2];
        for (int b = b0; b < b1; ++b) {
            const double dx = ax - (double)atoms[(size_t)b * 3], dy = ay - (double)atoms[(size_t)b * 3 + 1],
                         dz = az - (double)atoms[(size_t)b * 3 + 2];
            s += sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    const double m = s / (double)((long long)na * (long long)(b1 - b0));
    D[(size_t)i * n + j] = m;
    D[(size_t)j * n + i] = m;
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ unsigned long long wave_ballot(bool p) {
    // lanes with p set, as a 64-bit mask (built from shuffles so that the x86 simulator runs the same code)
    unsigned lo = 0, hi = 0;
    const int lane = threadIdx.x & 63;
    int mine_lo = (p && lane < 32) ? (1 << lane) : 0, mine_hi = (p && lane >= 32) ? (1 << (lane - 32)) : 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        mine_lo |= __shfl_xor(mine_lo, m);
        mine_hi |= __shfl_xor(mine_hi, m);
    }
    lo = (unsigned)mine_lo;
    hi = (unsigned)mine_hi;
    return ((unsigned long long)hi << 32) | lo;
}

// one wave per residue
__global__ __launch_bounds__(EQD_BLOCK) void k_pg_select(int n, int K, double cutoff, const double* __restrict__ D,
                                                         const double* __restrict__ x, int32_t* __restrict__ nbr,
                                                         double* __restrict__ nbd, int32_t* __restrict__ deg,
                                                         float* __restrict__ mu) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * EQD_WAVES + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* __restrict__ row = D + (size_t)i * n;
    int count = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool v = j < n && row[j] < cutoff;
        count += __popcll(wave_ballot(v));
    }
    int dg;
    if (count <= K) {                 // np.where order (:339)
        int base = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const bool v = j < n && row[j] < cutoff;
            const unsigned long long mk = wave_ballot(v);
            if (v) {
                const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
                nbr[(size_t)i * K + pos] = j;
                nbd[(size_t)i * K + pos] = row[j];
            }
            base += __popcll(mk);
        }
        dg = count;
    } else {                          // the K smallest distances, ascending (np.argsort(row)[0:K], :342-343)
        double dprev = -1.0;
        int jprev = -1;
        for (int r = 0; r < K; ++r) {
            double best = __builtin_inf();
            int bj = 0x7fffffff;
            for (int j = lane; j < n; j += 64) {
                const double d = row[j];
                const bool after = d > dprev || (d == dprev && j > jprev);
                if (after && (d < best || (d == best && j < bj))) {
                    best = d;
                    bj = j;
                }
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const double od = shfl_xor_d(best, m);
                const int oj = __shfl_xor(bj, m);
                if (od < best || (od == best && oj < bj)) {
                    best = od;
                    bj = oj;
                }
            }
            if (lane == 0) {
                nbr[(size_t)i * K + r] = bj;
                nbd[(size_t)i * K + r] = best;
            }
            dprev = best;
            jprev = bj;
        }
        dg = K;
    }
    if (lane == 0) deg[i] = dg;
    // surface feature (:351-359): for sigma in {1, 2, 5, 10, 30}: w = softmax_k(-d_k^2 / sigma),
    // mu = | sum_k w_k (x_i - x_k) | / sum_k w_k |x_i - x_k|          (lanes 0..4, one sigma each)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 5) {
        const double sg[5] = {1., 2., 5., 10., 30.};
        const double sigma = sg[lane];
        double mx = -__builtin_inf();
        for (int k = 0; k < dg; ++k) {
            const double d = ((volatile double*)nbd)[(size_t)i * K + k];
            mx = fmax(mx, -(d * d) / sigma);
        }
        double se = 0.0, m0 = 0.0, m1 = 0.0, m2 = 0.0, den = 0.0;
        for (int k = 0; k < dg; ++k) {
            const double d = ((volatile double*)nbd)[(size_t)i * K + k];
            const int j = ((volatile int32_t*)nbr)[(size_t)i * K + k];
            const double w = exp(-(d * d) / sigma - mx);
            const double vx = x[(size_t)i * 3] - x[(size_t)j * 3], vy = x[(size_t)i * 3 + 1] - x[(size_t)j * 3 + 1],
                         vz = x[(size_t)i * 3 + 2] - x[(size_t)j * 3 + 2];
            se += w;
            m0 += w * vx; m1 += w * vy; m2 += w * vz;
            den += w * sqrt(vx * vx + vy * vy + vz * vz);
        }
        m0 /= se; m1 /= se; m2 /= se; den /= se;
        mu[(size_t)i * 5 + lane] = (float)(sqrt(m0 * m0 + m1 * m1 + m2 * m2) / den);
    }
}

__global__ __launch_bounds__(EQD_BLOCK) void k_pg_edges(int n, int K, const int32_t* __restrict__ eoff,
                                                        const int32_t* __restrict__ nbr, const double* __restrict__ nbd,
                                                        const double* __restrict__ x, const double* __restrict__ fn,
                                                        const double* __restrict__ fu, const double* __restrict__ fv,
                                                        int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                        float* __restrict__ he) {
    const int idx = blockIdx.x * EQD_BLOCK + threadIdx.x;
    const int i = idx / K, k = idx - i * K;
    if (i >= n) return;
    const int e0 = eoff[i];
    if (k >= eoff[i + 1] - e0) return;
    const int e = e0 + k, j = nbr[(size_t)i * K + k];
    src[e] = j;
    dst[e] = i;
    const double d = nbd[(size_t)i * K + k];
    float* __restrict__ o = he + (size_t)e * 27;
    double ls = 1.0;
    for (int c = 0; c < 15; ++c) {        // distance_list_featurizer (:71-86): exp(-(d - 0)^2 / 1.5^c)
        o[c] = (float)exp(-(d * d) / ls);
        ls *= 1.5;
    }
    const double* B[3] = {fn + (size_t)i * 3, fu + (size_t)i * 3, fv + (size_t)i * 3};   // basis rows n_i, u_i, v_i of dst
    const double vec[4][3] = {{x[(size_t)j * 3] - x[(size_t)i * 3], x[(size_t)j * 3 + 1] - x[(size_t)i * 3 + 1],
                               x[(size_t)j * 3 + 2] - x[(size_t)i * 3 + 2]},
                              {fn[(size_t)j * 3], fn[(size_t)j * 3 + 1], fn[(size_t)j * 3 + 2]},
                              {fu[(size_t)j * 3], fu[(size_t)j * 3 + 1], fu[(size_t)j * 3 + 2]},
                              {fv[(size_t)j * 3], fv[(size_t)j * 3 + 1], fv[(size_t)j * 3 + 2]}};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            o[15 + 3 * q + r] = (float)((B[r][0] * vec[q][0] + B[r][1] * vec[q][1]) + B[r][2] * vec[q][2]);
}

extern "C" int eqd_protein_graph_distances(int n, const float* atoms, const int32_t* atom_off, double* D, void* stream) {
    if (!atoms || !atom_off || !D) {
        eqd_set_error("eqd_protein_graph_distances: NULL argument");
        return EQD_ERR_NULL;
    }
    if (n <= 0) return EQD_OK;
    hipLaunchKernelGGL(k_pg_distances, dim3(n, (n + EQD_BLOCK - 1) / EQD_BLOCK), dim3(EQD_BLOCK), 0, (hipStream_t)stream, n,
                       atoms, atom_off, D);
    return eqd_check_launch("k_pg_distances");
}
extern "C" int eqd_protein_graph_select(int n, int max_neighbor, double cutoff, const double* D, const double* x,
                                        int32_t* nbr, double* nbr_dist, int32_t* deg, float* mu_r_norm, void* stream) {
    if (!D || !x || !nbr || !nbr_dist || !deg || !mu_r_norm) {
        eqd_set_error("eqd_protein_graph_select: NULL argument");
        return EQD_ERR_NULL;
    }
    if (max_neighbor < 1 || max_neighbor > 64) {
        eqd_set_error("eqd_protein_graph_select: max_neighbor %d outside 1..64", max_neighbor);
        return EQD_ERR_UNSUPPORTED;
    }
    if (n <= 0) return EQD_OK;
    hipLaunchKernelGGL(k_pg_select, dim3((n + EQD_WAVES - 1) / EQD_WAVES), dim3(EQD_BLOCK), 0, (hipStream_t)stream, n,
                       max_neighbor, cutoff, D, x, nbr, nbr_dist, deg, mu_r_norm);
    return eqd_check_launch("k_pg_select");
}
extern "C" int eqd_protein_graph_edges(int n, int max_neighbor, const int32_t* edge_off, const int32_t* nbr,
                                       const double* nbr_dist, const double* x, const double* n_i, const double* u_i,
                                       const double* v_i, int32_t* src, int32_t* dst, float* he, void* stream) {
    if (!edge_off || !nbr || !nbr_dist || !x || !n_i || !u_i || !v_i || !src || !dst || !he) {
        eqd_set_error("eqd_protein_graph_edges: NULL argument");
        return EQD_ERR_NULL;
    }
    if (n <= 0) return EQD_OK;
    const long long total = (long long)n * max_neighbor;
    hipLaunchKernelGGL(k_pg_edges, dim3((unsigned)((total + EQD_BLOCK - 1) / EQD_BLOCK)), dim3(EQD_BLOCK), 0,
                       (hipStream_t)stream, n, max_neighbor, edge_off, nbr, nbr_dist, x, n_i, u_i, v_i, src, dst, he);
    return eqd_check_launch("k_pg_edges");
}
