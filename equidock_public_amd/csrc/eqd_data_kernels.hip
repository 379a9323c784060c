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
// nn.Dropout keep masks of the edge MLPs (EqdDropout.edge_z1 / edge_ch): the caller draws them with torch's dropout on
// [E_ll][64] and [E_rr][64] tensors of ones (the reference's shapes and consumption order, rigid_docking_model.py:236-237,
// 263-265); this kernel turns the two factor tensors (0 | 1 / (1 - p)) into the packed words the edge kernels read, in the
// library's edge order.  16 lanes per edge (one float4 each), 4 edges per wave; a torch formulation of the same packing
// moved ~1 GB per mask pair and layer at 64 x (300, 300) (int64 intermediates), this one reads the factors once.
__global__ __launch_bounds__(EQD_BLOCK) void k_dropout_pack(int n_edges, int e_ll, const float* __restrict__ f_ll,
                                                            const float* __restrict__ f_rr,
                                                            const int64_t* __restrict__ perm,
                                                            uint32_t* __restrict__ words) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, l15 = lane & 15;
    const int e = (int)((blockIdx.x * (size_t)EQD_BLOCK + threadIdx.x) >> 6) * 4 + sub;
    const bool live = e < n_edges;
    const int raw = perm ? (int)perm[live ? e : 0] : (live ? e : 0);
    const float* row = raw < e_ll ? f_ll + (size_t)raw * 64 : f_rr + (size_t)(raw - e_ll) * 64;
    const float4 v = *(const float4*)&row[4 * l15];
    int bits = ((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0)) << (4 * (l15 & 7));
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) bits |= __shfl_xor(bits, m);
    if (live && (l15 & 7) == 0) words[(size_t)e * 2 + (l15 >> 3)] = (uint32_t)bits;
}

extern "C" int eqd_dropout_pack_edges(int n_edges, int e_ll, const float* factors_ll, const float* factors_rr,
                                      const int64_t* perm, uint32_t* words, void* stream) {
    if (n_edges < 0 || e_ll < 0 || e_ll > n_edges) {
        eqd_set_error("eqd_dropout_pack_edges: bad edge counts (%d edges, %d ligand)", n_edges, e_ll);
        return EQD_ERR_SHAPE;
    }
    if (n_edges == 0) return EQD_OK;
    if (!words || (e_ll > 0 && !factors_ll) || (e_ll < n_edges && !factors_rr)) {
        eqd_set_error("eqd_dropout_pack_edges: NULL argument");
        return EQD_ERR_NULL;
    }
    if (((uintptr_t)factors_ll | (uintptr_t)factors_rr) & 15) {
        eqd_set_error("eqd_dropout_pack_edges: factor tensors must be 16-byte aligned");
        return EQD_ERR_UNSUPPORTED;
    }
    const int per_block = 4 * (EQD_BLOCK / 64);
    hipLaunchKernelGGL(k_dropout_pack, dim3((n_edges + per_block - 1) / per_block), dim3(EQD_BLOCK), 0, (hipStream_t)stream,
                       n_edges, e_ll, factors_ll, factors_rr, perm, words);
    return eqd_check_launch("k_dropout_pack");
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

// ---------------------------------------------------------------------------------------------------------------
// Inference post-processing: the clash-removal loop of src/inference_rigid.py:207-234 -
// gradient descent on 3 Euler angles + a translation of the docked ligand (ALL atoms) against the receptor (all atoms)
// under compute_body_intersection_loss(sigma = 8, surface_ct = 8), up to 2000 iterations, which the reference runs as
// torch autograd over an (n_lig x n_rec) matrix per iteration on the host.  Here: four small launches per iteration
// with the whole state on the device (angles, translation, iteration counter, stop flag); once the stop rule
// (loss <= loss_stop or it >= max_it) fires, the remaining launches of a chunk return immediately, so the host only
// looks at the flag every few dozen iterations.  float32 like the reference.
//   ligand_th_i = R(euler) p_i + t,   R = RZ(yaw) RY(pitch) RX(roll),  euler = (roll, yaw, pitch)        (:46-73, :213)
//   loss = mean_i max(0, ct - G_r(a_i)) + mean_k max(0, ct - G_l(b_k)),  G(x) = -sigma log(1e-3 + sum exp(-|x - c|^2 / sigma))
//   eta = 1e-3; 1e-4 if loss < 2; 1e-2 if it > 1500                                                      (:219-224)
// ---------------------------------------------------------------------------------------------------------------
#define CL_TILE 1024
struct EqdClashWs {
    float* lig_th;     // [n_lig][3] current ligand positions
    float* s_lig;      // [n_lig]
    float* w_rec;      // [n_rec]  [ct - G_l(b_k) >= 0] / (n_rec (1e-3 + S_k))
    float* part1;      // [nb_lig] term-1 partial sums
    float* part2;      // [nb_rec] term-2 partial sums
    float* gpart;      // [nb_lig][6] partial (d trans, d euler)
};
__device__ __forceinline__ void euler_rot(const float* e, float R[9]) {
    const float cr = cosf(e[0]), sr = sinf(e[0]), cy = cosf(e[1]), sy = sinf(e[1]), cp = cosf(e[2]), sp = sinf(e[2]);
    // RZ(yaw) RY(pitch) RX(roll)
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// pass 1: positions + Gaussian sums of every ligand atom against the receptor
__global__ __launch_bounds__(EQD_BLOCK) void k_clash_lig(int n_lig, int n_rec, const float* __restrict__ lig0,
                                                         const float* __restrict__ rec, float sigma, float ct,
                                                         const EqdClashState* __restrict__ st, EqdClashWs W) {
    __shared__ float pts[CL_TILE][3];
    __shared__ float red[EQD_WAVES];
    if (st->done) return;
    float R[9];
    const float e[3] = {st->euler[0], st->euler[1], st->euler[2]};
    euler_rot(e, R);
    const int i = blockIdx.x * EQD_BLOCK + threadIdx.x;
    const int ic = i < n_lig ? i : n_lig - 1;
    const float px = lig0[(size_t)ic * 3], py = lig0[(size_t)ic * 3 + 1], pz = lig0[(size_t)ic * 3 + 2];
    const float ax = (R[0] * px + R[1] * py) + R[2] * pz + st->trans[0];
    const float ay = (R[3] * px + R[4] * py) + R[5] * pz + st->trans[1];
    const float az = (R[6] * px + R[7] * py) + R[8] * pz + st->trans[2];
    const float inv = 1.f / sigma;
    float S = 0.f;
    for (int c0 = 0; c0 < n_rec; c0 += CL_TILE) {
        const int nc = n_rec - c0 < CL_TILE ? n_rec - c0 : CL_TILE;
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * nc; k += EQD_BLOCK) pts[k / 3][k % 3] = rec[(size_t)c0 * 3 + k];
        __syncthreads();
        for (int k = 0; k < nc; ++k) {
            const float dx = pts[k][0] - ax, dy = pts[k][1] - ay, dz = pts[k][2] - az;
            S += expf(-((dx * dx + dy * dy) + dz * dz) * inv);
        }
    }
    float term = 0.f;
    if (i < n_lig) {
        W.lig_th[(size_t)i * 3] = ax; W.lig_th[(size_t)i * 3 + 1] = ay; W.lig_th[(size_t)i * 3 + 2] = az;
        W.s_lig[i] = S;
        const float G = -sigma * logf(1e-3f + S);
        term = ct - G > 0.f ? ct - G : 0.f;
    }
    const float tot = block_sum256(term, red);
    if (threadIdx.x == 0) W.part1[blockIdx.x] = tot;
}
// pass 2: Gaussian sums of every receptor atom against the moved ligand -> weights of the backward, term-2 partials
__global__ __launch_bounds__(EQD_BLOCK) void k_clash_rec(int n_lig, int n_rec, const float* __restrict__ rec, float sigma,
                                                         float ct, const EqdClashState* __restrict__ st, EqdClashWs W) {
    __shared__ float pts[CL_TILE][3];
    __shared__ float red[EQD_WAVES];
    if (st->done) return;
    const int k = blockIdx.x * EQD_BLOCK + threadIdx.x;
    const int kc = k < n_rec ? k : n_rec - 1;
    const float bx = rec[(size_t)kc * 3], by = rec[(size_t)kc * 3 + 1], bz = rec[(size_t)kc * 3 + 2];
    const float inv = 1.f / sigma;
    float S = 0.f;
    for (int c0 = 0; c0 < n_lig; c0 += CL_TILE) {
        const int nc = n_lig - c0 < CL_TILE ? n_lig - c0 : CL_TILE;
        __syncthreads();
        for (int q = threadIdx.x; q < 3 * nc; q += EQD_BLOCK) pts[q / 3][q % 3] = W.lig_th[(size_t)c0 * 3 + q];
        __syncthreads();
        for (int q = 0; q < nc; ++q) {
            const float dx = pts[q][0] - bx, dy = pts[q][1] - by, dz = pts[q][2] - bz;
            S += expf(-((dx * dx + dy * dy) + dz * dz) * inv);
        }
    }
    float term = 0.f;
    if (k < n_rec) {
        const float G = -sigma * logf(1e-3f + S);
        term = ct - G > 0.f ? ct - G : 0.f;
        W.w_rec[k] = ct - G >= 0.f ? 1.f / ((float)n_rec * (1e-3f + S)) : 0.f;
    }
    const float tot = block_sum256(term, red);
    if (threadIdx.x == 0) W.part2[blockIdx.x] = tot;
}
__device__ __forceinline__ float clash_loss(int n_lig, int n_rec, const EqdClashWs& W) {
    const int nb1 = (n_lig + EQD_BLOCK - 1) / EQD_BLOCK, nb2 = (n_rec + EQD_BLOCK - 1) / EQD_BLOCK;
    float a = 0.f, b = 0.f;
    for (int q = 0; q < nb1; ++q) a += W.part1[q];
    for (int q = 0; q < nb2; ++q) b += W.part2[q];
    return a / (float)n_lig + b / (float)n_rec;
}
// pass 3: gradient w.r.t. every ligand atom, chained to (translation, Euler angles); per-block partial sums
__global__ __launch_bounds__(EQD_BLOCK) void k_clash_grad(int n_lig, int n_rec, const float* __restrict__ lig0,
                                                          const float* __restrict__ rec, float sigma, float ct,
                                                          float loss_stop, int max_it,
                                                          const EqdClashState* __restrict__ st, EqdClashWs W) {
    __shared__ float pts[CL_TILE][4];
    __shared__ float red[EQD_WAVES];
    if (st->done || st->it >= max_it) return;               // (k_clash_step raises the flag at it == max_it)
    // the iteration whose loss comes out <= loss_stop still steps (the reference's loop), so its gradient is needed too
    const float e[3] = {st->euler[0], st->euler[1], st->euler[2]};
    const int i = blockIdx.x * EQD_BLOCK + threadIdx.x;
    const int ic = i < n_lig ? i : n_lig - 1;
    const float ax = W.lig_th[(size_t)ic * 3], ay = W.lig_th[(size_t)ic * 3 + 1], az = W.lig_th[(size_t)ic * 3 + 2];
    const float Si = W.s_lig[ic];
    const float Gi = -sigma * logf(1e-3f + Si);
    const float wi = (ct - Gi >= 0.f) ? 1.f / ((float)n_lig * (1e-3f + Si)) : 0.f;
    const float inv = 1.f / sigma;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int c0 = 0; c0 < n_rec; c0 += CL_TILE) {
        const int nc = n_rec - c0 < CL_TILE ? n_rec - c0 : CL_TILE;
        __syncthreads();
        for (int k = threadIdx.x; k < nc; k += EQD_BLOCK) {
            pts[k][0] = rec[(size_t)(c0 + k) * 3]; pts[k][1] = rec[(size_t)(c0 + k) * 3 + 1];
            pts[k][2] = rec[(size_t)(c0 + k) * 3 + 2]; pts[k][3] = W.w_rec[c0 + k];
        }
        __syncthreads();
        for (int k = 0; k < nc; ++k) {
            const float dx = ax - pts[k][0], dy = ay - pts[k][1], dz = az - pts[k][2];
            const float w = expf(-((dx * dx + dy * dy) + dz * dz) * inv) * (wi + pts[k][3]);
            gx += w * dx; gy += w * dy; gz += w * dz;
        }
    }
    // d loss / d a_i = -2 (gx, gy, gz); a_i = R(euler) p_i + t
    float g[3] = {i < n_lig ? -2.f * gx : 0.f, i < n_lig ? -2.f * gy : 0.f, i < n_lig ? -2.f * gz : 0.f};
    const float px = lig0[(size_t)ic * 3], py = lig0[(size_t)ic * 3 + 1], pz = lig0[(size_t)ic * 3 + 2];
    const float cr = cosf(e[0]), sr = sinf(e[0]), cy = cosf(e[1]), sy = sinf(e[1]), cp = cosf(e[2]), sp = sinf(e[2]);
    // dR/droll, dR/dyaw, dR/dpitch of R = RZ(yaw) RY(pitch) RX(roll)
    const float dRr[9] = {0.f, cy * sp * cr + sy * sr, -cy * sp * sr + sy * cr,
                          0.f, sy * sp * cr - cy * sr, -sy * sp * sr - cy * cr,
                          0.f, cp * cr, -cp * sr};
    const float dRy[9] = {-sy * cp, -sy * sp * sr - cy * cr, -sy * sp * cr + cy * sr,
                          cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                          0.f, 0.f, 0.f};
    const float dRp[9] = {-cy * sp, cy * cp * sr, cy * cp * cr,
                          -sy * sp, sy * cp * sr, sy * cp * cr,
                          -cp, -sp * sr, -sp * cr};
    float out[6];
    out[0] = g[0]; out[1] = g[1]; out[2] = g[2];
    const float* dRs[3] = {dRr, dRy, dRp};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float* d = dRs[j];
        out[3 + j] = g[0] * ((d[0] * px + d[1] * py) + d[2] * pz) + g[1] * ((d[3] * px + d[4] * py) + d[5] * pz) +
                     g[2] * ((d[6] * px + d[7] * py) + d[8] * pz);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float tot = block_sum256(out[j], red);
        if (threadIdx.x == 0) W.gpart[(size_t)blockIdx.x * 6 + j] = tot;
    }
}
// pass 4 (one wave): stop rule, step size, update
__global__ void k_clash_step(int n_lig, int n_rec, float loss_stop, int max_it, EqdClashState* __restrict__ st,
                             EqdClashWs W) {
    if (st->done) return;
    if (threadIdx.x != 0) return;
    // The reference's loop (src/inference_rigid.py:213-232) tests `loss > 0.5 and it < 2000` with the loss of the PREVIOUS
    // evaluation, then evaluates, steps and increments unconditionally: the iteration whose loss comes out <= loss_stop
    // still applies its gradient step (the returned parameters are one step past the converged evaluation, `it` counts
    // that step), and at it == max_it nothing is evaluated any more (the reported loss stays the last one).
    if (st->it >= max_it) {
        st->done = 1;
        return;
    }
    const float loss = clash_loss(n_lig, n_rec, W);
    st->loss = loss;
    if (!(loss > loss_stop)) st->done = 1;
    float eta = 1e-3f;
    if (loss < 2.f) eta = 1e-4f;
    if (st->it > 1500) eta = 1e-2f;
    const int nb = (n_lig + EQD_BLOCK - 1) / EQD_BLOCK;
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nb; ++q)
        for (int j = 0; j < 6; ++j) g[j] += W.gpart[(size_t)q * 6 + j];
    for (int j = 0; j < 3; ++j) {
        st->trans[j] -= eta * g[j];
        st->euler[j] -= eta * g[3 + j];
    }
    st->it += 1;
}

static size_t clash_carve(int n_lig, int n_rec, EqdArena& A, EqdClashWs* W) {
    const size_t nb1 = (size_t)(n_lig + EQD_BLOCK - 1) / EQD_BLOCK, nb2 = (size_t)(n_rec + EQD_BLOCK - 1) / EQD_BLOCK;
    EqdClashWs w;
    w.lig_th = A.take<float>((size_t)n_lig * 3);
    w.s_lig = A.take<float>((size_t)n_lig);
    w.w_rec = A.take<float>((size_t)n_rec);
    w.part1 = A.take<float>(nb1);
    w.part2 = A.take<float>(nb2);
    w.gpart = A.take<float>(nb1 * 6);
    if (W) *W = w;
    return A.off;
}
extern "C" size_t eqd_clash_workspace_bytes(int n_lig, int n_rec) {
    if (n_lig < 1 || n_rec < 1) return 0;
    EqdArena A(nullptr, 0);
    return clash_carve(n_lig, n_rec, A, nullptr) + 256;
}
extern "C" int eqd_clash_iterations(int n_iter, int n_lig, int n_rec, const float* lig0, const float* rec, float sigma,
                                    float surface_ct, float loss_stop, int max_it, EqdClashState* state, void* workspace,
                                    size_t ws_bytes, void* stream) {
    if (!lig0 || !rec || !state || !workspace) {
        eqd_set_error("eqd_clash_iterations: NULL argument");
        return EQD_ERR_NULL;
    }
    if (n_lig < 1 || n_rec < 1 || !(sigma > 0.f) || n_iter < 0) {
        eqd_set_error("eqd_clash_iterations: n_lig = %d, n_rec = %d, sigma = %g", n_lig, n_rec, sigma);
        return EQD_ERR_SHAPE;
    }
    EqdArena A(workspace, ws_bytes);
    EqdClashWs W;
    clash_carve(n_lig, n_rec, A, &W);
    if (!A.ok) {
        eqd_set_error("eqd_clash_iterations: workspace too small (%zu needed)", A.off);
        return EQD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nb1 = (n_lig + EQD_BLOCK - 1) / EQD_BLOCK, nb2 = (n_rec + EQD_BLOCK - 1) / EQD_BLOCK;
    for (int it = 0; it < n_iter; ++it) {
        hipLaunchKernelGGL(k_clash_lig, dim3(nb1), dim3(EQD_BLOCK), 0, st, n_lig, n_rec, lig0, rec, sigma, surface_ct, state, W);
        if (int rc = eqd_check_launch("k_clash_lig")) return rc;
        hipLaunchKernelGGL(k_clash_rec, dim3(nb2), dim3(EQD_BLOCK), 0, st, n_lig, n_rec, rec, sigma, surface_ct, state, W);
        if (int rc = eqd_check_launch("k_clash_rec")) return rc;
        hipLaunchKernelGGL(k_clash_grad, dim3(nb1), dim3(EQD_BLOCK), 0, st, n_lig, n_rec, lig0, rec, sigma, surface_ct,
                           loss_stop, max_it, state, W);
        if (int rc = eqd_check_launch("k_clash_grad")) return rc;
        hipLaunchKernelGGL(k_clash_step, dim3(1), dim3(64), 0, st, n_lig, n_rec, loss_stop, max_it, state, W);
        if (int rc = eqd_check_launch("k_clash_step")) return rc;
    }
    return EQD_OK;
}
/* lig_th of the last evaluated iteration lives at the start of the workspace ([n_lig][3] floats) */
