// Keypoint head of the IEGMN stack: K-head attention pooling, batched 3x3 Kabsch/SVD with
// closed-form backward, rigid apply.  These are small (6 kFLOP/node, 50 points/pair):
// plain VALU kernels, one launch for all pairs instead of the reference's Python loop.
//
// Reference arithmetic replaced (src/model/rigid_docking_model.py):
//   :524-529  q_side = mean_nodes(LeakyReLU(W_m h + b_m))         (the Linear runs in k_linear)
//   :542-560  att = softmax_nodes((W_K h)_k . (W_Q q_partner)_k / sqrt(d));  Y = att^T Z
//             -- computed with collapsed heads: u_k = W_K^(k)T (W_Q^(k) q) / sqrt(d), score = h . u_k
//                (64x fewer FLOPs, equal to 1.3e-7)
//   :563-589  Kabsch: A = (Yr - mean)^T (Yl - mean), SVD, guard loop, T = U diag(1,1,sign det A) V^T,
//             b = mean_r - T mean_l
//   :665      lig' = (T x^T)^T + b
#include "eqd_common.h"

// ---------------------------------------------------------------------------------------------
// segment helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const int32_t* __restrict__ seg_off, int nseg, int node) {
    int lo = 0, hi = nseg;  // seg_off[lo] <= node < seg_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_off[mid] <= node) lo = mid; else hi = mid;
    }
    return lo;
}

// qmean[s][c] = mean over the nodes of segment s of hm[i][c]   (64 columns)
// (1 024 threads: 16 row groups, four rows in flight per thread - with 4 groups and one dependent add per row a
// 200-node segment was 50 serial load round trips, 13.6 us for 50 KB)
__global__ __launch_bounds__(1024) void k_seg_mean(const int32_t* __restrict__ seg_off,
                                                   const float* __restrict__ hm, float* __restrict__ qmean) {
    __shared__ float red[16][64];
    const int s = blockIdx.x;
    const int n0 = seg_off[s], n1 = seg_off[s + 1];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    float acc = 0.f;
    for (int i = n0 + rg; i < n1; i += 64) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = i + 16 * u;
            v[u] = hm[(size_t)(r < n1 ? r : n0) * 64 + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += i + 16 * u < n1 ? v[u] : 0.f;
    }
    red[rg][c] = acc;
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][c];
        qmean[(size_t)s * 64 + c] = n1 > n0 ? t / (float)(n1 - n0) : 0.f;
    }
}
int eqd_launch_seg_mean(const EqdGraph* g, const float* hm, float* qmean, hipStream_t st) {
    if (g->n_pairs == 0) return EQD_OK;
    hipLaunchKernelGGL(k_seg_mean, dim3(2 * g->n_pairs), dim3(1024), 0, st, g->seg_off, hm, qmean);
    return eqd_check_launch("k_seg_mean");
}

// per (segment s, head k): qp = W_Q^(k) qmean[partner(s)];  u = W_K^(k)T qp / 8
__global__ void k_head_u(int B, int K, const float* __restrict__ Wk, const float* __restrict__ Wq,
                         const float* __restrict__ qmean, float* __restrict__ qp, float* __restrict__ u) {
    __shared__ float sq[64], sp[64];
    const int s = blockIdx.x, k = blockIdx.y, t = threadIdx.x;  // 64 threads
    const int partner = s < B ? s + B : s - B;
    sq[t] = qmean[(size_t)partner * 64 + t];
    __syncthreads();
    const float* wq = Wq + ((size_t)k * 64 + t) * 64;
    float a = 0.f;
    for (int c = 0; c < 64; ++c) a += wq[c] * sq[c];
    sp[t] = a;
    qp[((size_t)s * K + k) * 64 + t] = a;
    __syncthreads();
    float b = 0.f;
    for (int j = 0; j < 64; ++j) b += Wk[((size_t)k * 64 + j) * 64 + t] * sp[j];
    u[((size_t)s * K + k) * 64 + t] = b * 0.125f;
}

#include "eqd_keypoint_mm_inl.h"
#include "eqd_headu_mm_inl.h"
// EQD_KEYPOINT_MM: 0 = the first kernels (one workgroup per (segment, head)), 1 = the matrix-product forms everywhere,
// unset = the product forward everywhere, the product backward where it has enough workgroups (keypoint_bwd_chunks)
static int keypoint_mm_mode() {
    const char* f = eqd_tunable("EQD_KEYPOINT_MM");
    if (f && (f[0] == '0' || f[0] == '1') && f[1] == 0) return f[0] - '0';
    return 2;
}
// row chunks per segment of the product backward (0: use the first kernels).  A workgroup spends ~2 000 clocks per 16-row
// tile: enough chunks to fill the chip, of at least four tiles, as long as the partial du blocks fit the dscores workspace
// ([N][K] floats, unused by this form: S NC K 64 <= N K) - EQD_KEYPOINT_NC forces a count (tests, experiments)
static int keypoint_bwd_chunks(const EqdGraph* g, int K) {
    const int mode = keypoint_mm_mode();
    if (mode == 0 || K > 64) return 0;
    const int S = 2 * g->n_pairs;
    const int cap = g->n_nodes / (64 * S);      // partial blocks that fit
    const int nt = (g->max_seg + 15) / 16;
    int nc = (eqd_num_cus() + S - 1) / S;
    if (nc > (nt + 3) / 4) nc = (nt + 3) / 4;
    const char* f = eqd_tunable("EQD_KEYPOINT_NC");
    if (f && f[0]) nc = atoi(f);
    if (nc > cap) nc = cap;
    if (nc < 1) nc = 1;
    if (mode == 2 && S * nc < 32) return 0;      // too few workgroups: the (segment, head) kernels fill the chip better
    return nc;
}

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = fmaxf(v, lane_xor<1>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<4>(v));
    v = fmaxf(v, lane_xor<8>(v));
    v = group_max(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// per (segment, head): scores over the segment's nodes, softmax, Y = att^T Z   (the first form; k_keypoint_mm replaces it,
// EQD_KEYPOINT_MM=0 selects it)
__global__ __launch_bounds__(EQD_BLOCK) void k_keypoint(const int32_t* __restrict__ seg_off, int K,
                                                        const float* __restrict__ u, const float* __restrict__ H,
                                                        const float* __restrict__ Z, float* __restrict__ Y,
                                                        float* __restrict__ Yl_out, float* __restrict__ Yr_out,
                                                        int B, float* __restrict__ scores, float* __restrict__ lse,
                                                        float* __restrict__ Yc) {
    __shared__ float su[64];
    __shared__ float red[4];
    const int s = blockIdx.x, k = blockIdx.y, t = threadIdx.x;
    const int n0 = seg_off[s], n1 = seg_off[s + 1];
    if (t < 64) su[t] = u[((size_t)s * K + k) * 64 + t];
    __syncthreads();
    float mx = EQD_NEG_BIG;
    for (int i = n0 + t; i < n1; i += EQD_BLOCK) {
        const float4* h = (const float4*)&H[(size_t)i * 64];
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 hv = h[c];
            a += hv.x * su[4 * c] + hv.y * su[4 * c + 1] + hv.z * su[4 * c + 2] + hv.w * su[4 * c + 3];
        }
        scores[(size_t)i * K + k] = a;
        mx = fmaxf(mx, a);
    }
    mx = block_reduce_max(mx, red);
    // (sums over z - Z[n0], see k_keypoint_mm)
    const float c0 = Z[(size_t)n0 * 3], c1 = Z[(size_t)n0 * 3 + 1], c2 = Z[(size_t)n0 * 3 + 2];
    float se = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f;
    for (int i = n0 + t; i < n1; i += EQD_BLOCK) {
        const float p = expf(scores[(size_t)i * K + k] - mx);
        se += p;
        y0 += p * (Z[(size_t)i * 3 + 0] - c0);
        y1 += p * (Z[(size_t)i * 3 + 1] - c1);
        y2 += p * (Z[(size_t)i * 3 + 2] - c2);
    }
    se = block_reduce_sum(se, red);
    y0 = block_reduce_sum(y0, red);
    y1 = block_reduce_sum(y1, red);
    y2 = block_reduce_sum(y2, red);
    if (t == 0) {
        const float inv = se > 0.f ? 1.f / se : 0.f;
        const float a0 = se > 0.f ? c0 + y0 * inv : 0.f, a1 = se > 0.f ? c1 + y1 * inv : 0.f, a2 = se > 0.f ? c2 + y2 * inv : 0.f;
        float* y = Y + ((size_t)s * K + k) * 3;
        y[0] = a0; y[1] = a1; y[2] = a2;
        if (Yc) {
            float* yc = Yc + ((size_t)s * K + k) * 3;
            yc[0] = y0 * inv; yc[1] = y1 * inv; yc[2] = y2 * inv;
        }
        float* yo = s < B ? (Yl_out ? Yl_out + ((size_t)s * K + k) * 3 : nullptr)
                          : (Yr_out ? Yr_out + ((size_t)(s - B) * K + k) * 3 : nullptr);
        if (yo) {
            yo[0] = a0; yo[1] = a1; yo[2] = a2;
        }
        lse[(size_t)s * K + k] = se > 0.f ? mx + logf(se) : 0.f;
    }
}

extern "C" int eqd_keypoint_pool_fwd(const EqdGraph* g, int n_heads, const float* Wk, const float* Wq,
                                     const float* qmean, const float* H, const float* Z, float* Y, float* scores,
                                     float* lse, float* qp, float* u, void* stream) {
    return eqd_keypoint_pool_fwd_impl(g, n_heads, Wk, Wq, qmean, H, Z, Y, nullptr, nullptr, scores, lse, qp, u,
                                      (hipStream_t)stream, nullptr);
}
// Yc (optional): [2 B][K][3], the keypoints relative to their segment's first node (what the backward's softmax needs,
// k_keypoint_bwd_mm)
int eqd_keypoint_pool_fwd_impl(const EqdGraph* g, int n_heads, const float* Wk, const float* Wq, const float* qmean,
                               const float* H, const float* Z, float* Y, float* Y_lig_out, float* Y_rec_out,
                               float* scores, float* lse, float* qp, float* u, hipStream_t st, float* Yc) {
    if (!g || !Wk || !Wq || !qmean || !H || !Z || !Y || !scores || !lse || !qp || !u) {
        eqd_set_error("eqd_keypoint_pool_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (g->n_pairs == 0) return EQD_OK;
    if (keypoint_mm_mode() != 0)
        hipLaunchKernelGGL(k_head_u_mm, dim3(n_heads, (2 * g->n_pairs + 15) / 16), dim3(EQD_BLOCK), 0, st, g->n_pairs, n_heads, Wk,
                           Wq, qmean, qp, u, 16);
    else
        hipLaunchKernelGGL(k_head_u, dim3(2 * g->n_pairs, n_heads), dim3(64), 0, st, g->n_pairs, n_heads, Wk, Wq, qmean, qp,
                           u);
    int rc = eqd_check_launch("k_head_u");
    if (rc) return rc;
    if (keypoint_mm_mode() != 0) {
        const dim3 grid(2 * g->n_pairs, (n_heads + 15) / 16);
        if ((g->max_seg + 15) / 16 >= 64 && (int)(grid.x * grid.y) < eqd_num_cus())      // few long segments: 16 waves per workgroup
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_keypoint_mm<16>), grid, dim3(1024), 0, st, g->seg_off, n_heads, u, H, Z, Y,
                               Y_lig_out, Y_rec_out, g->n_pairs, scores, lse, Yc);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_keypoint_mm<4>), grid, dim3(EQD_BLOCK), 0, st, g->seg_off, n_heads, u, H, Z, Y,
                               Y_lig_out, Y_rec_out, g->n_pairs, scores, lse, Yc);
    } else
        hipLaunchKernelGGL(k_keypoint, dim3(2 * g->n_pairs, n_heads), dim3(EQD_BLOCK), 0, st, g->seg_off, n_heads, u, H, Z,
                           Y, Y_lig_out, Y_rec_out, g->n_pairs, scores, lse, Yc);
    return eqd_check_launch("k_keypoint");
}

// backward of k_keypoint, part a: per (segment, head): dscores (written), du (reduced over nodes)
__global__ __launch_bounds__(EQD_BLOCK) void k_keypoint_bwd_a(const int32_t* __restrict__ seg_off, int K,
                                                              const float* __restrict__ H,
                                                              const float* __restrict__ Z,
                                                              const float* __restrict__ scores,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ dY,
                                                              float* __restrict__ dscores, float* __restrict__ du) {
    __shared__ float red[4];
    __shared__ float racc[4][64];
    const int s = blockIdx.x, k = blockIdx.y, t = threadIdx.x;
    const int n0 = seg_off[s], n1 = seg_off[s + 1];
    const float* dy = dY + ((size_t)s * K + k) * 3;
    const float d0 = dy[0], d1 = dy[1], d2 = dy[2];
    const float L = lse[(size_t)s * K + k];
    // dscore = att * (dY . (z - Y)), Y = sum att z: the coordinates are differenced first (see k_keypoint_bwd_mm)
    const float c0 = Z[(size_t)n0 * 3], c1 = Z[(size_t)n0 * 3 + 1], c2 = Z[(size_t)n0 * 3 + 2];
    float y0 = 0.f, y1 = 0.f, y2 = 0.f;
    for (int i = n0 + t; i < n1; i += EQD_BLOCK) {
        const float a = expf(scores[(size_t)i * K + k] - L);
        y0 += a * (Z[(size_t)i * 3] - c0);
        y1 += a * (Z[(size_t)i * 3 + 1] - c1);
        y2 += a * (Z[(size_t)i * 3 + 2] - c2);
    }
    y0 = block_reduce_sum(y0, red);
    y1 = block_reduce_sum(y1, red);
    y2 = block_reduce_sum(y2, red);
    for (int i = n0 + t; i < n1; i += EQD_BLOCK) {
        const float a = expf(scores[(size_t)i * K + k] - L);
        dscores[(size_t)i * K + k] = a * (d0 * ((Z[(size_t)i * 3] - c0) - y0) + d1 * ((Z[(size_t)i * 3 + 1] - c1) - y1) +
                                          d2 * ((Z[(size_t)i * 3 + 2] - c2) - y2));
    }
    __syncthreads();   // dscores of this (segment, head) are re-read below by other threads of the block
    const int c = t & 63, rg = t >> 6;
    float acc = 0.f;
    for (int i = n0 + rg; i < n1; i += 32) {       // 8 rows in flight per thread (clamped, unpredicated loads)
        float ds[8], hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = i + 4 * u < n1 ? i + 4 * u : n0;
            ds[u] = dscores[(size_t)r * K + k];
            hv[u] = H[(size_t)r * 64 + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += i + 4 * u < n1 ? ds[u] * hv[u] : 0.f;
    }
    racc[rg][c] = acc;
    __syncthreads();
    if (rg == 0) du[((size_t)s * K + k) * 64 + c] = racc[0][c] + racc[1][c] + racc[2][c] + racc[3][c];
}

// part b: per node: dH[i] = sum_k dscores[i][k] u[seg][k];  dZ[i] = sum_k att[i][k] dY[seg][k]
__global__ void k_keypoint_bwd_b(const int32_t* __restrict__ seg_off, int nseg, int n, int K,
                                 const float* __restrict__ scores, const float* __restrict__ lse,
                                 const float* __restrict__ u, const float* __restrict__ dY,
                                 const float* __restrict__ dscores, float* __restrict__ dH, float* __restrict__ dZ) {
    __shared__ float sds[128], sal[128];
    const int i = blockIdx.x, t = threadIdx.x;  // 64 threads
    if (i >= n) return;
    const int s = find_segment(seg_off, nseg, i);
    for (int k = t; k < K; k += 64) {
        sds[k] = dscores[(size_t)i * K + k];
        sal[k] = expf(scores[(size_t)i * K + k] - lse[(size_t)s * K + k]);
    }
    __syncthreads();
    float a = 0.f;
    for (int k = 0; k < K; ++k) a += sds[k] * u[((size_t)s * K + k) * 64 + t];
    dH[(size_t)i * 64 + t] = a;
    if (t < 3) {
        float z = 0.f;
        for (int k = 0; k < K; ++k) z += sal[k] * dY[((size_t)s * K + k) * 3 + t];
        dZ[(size_t)i * 3 + t] = z;
    }
}

int eqd_launch_keypoint_bwd(const EqdGraph* g, int K, const float* H, const float* Z, const float* scores,
                            const float* lse, const float* u, const float* dY, float* dscores, float* du, float* dH,
                            float* dZ, hipStream_t st, const float* Yc, int* du_chunks) {
    // Yc: the forward's keypoints relative to their segment's first node (eqd_keypoint_pool_fwd_impl), or NULL
    // du_chunks (optional): the caller's k_head_u_bwd sums the chunks' partial du blocks (they stay in `dscores`,
    // *du_chunks = their count per segment); without it the segments are not split
    if (du_chunks) *du_chunks = 1;
    if (g->n_pairs == 0 || g->n_nodes == 0) return EQD_OK;
    if (K > 128) {
        eqd_set_error("num_att_heads %d > 128 unsupported", K);
        return EQD_ERR_UNSUPPORTED;
    }
    if (int nc = keypoint_bwd_chunks(g, K)) {
        if (!du_chunks) nc = 1;
        // (dscores: the workspace of the first kernels holds the chunks' partial du blocks here)
        hipLaunchKernelGGL(k_keypoint_bwd_mm, dim3(2 * g->n_pairs, nc), dim3(EQD_BLOCK), 0, st, g->seg_off, K, H, Z, scores, lse,
                           u, dY, Yc, nc == 1 ? du : dscores, dH, dZ);
        if (du_chunks) *du_chunks = nc;
        return eqd_check_launch("k_keypoint_bwd");
    }
    hipLaunchKernelGGL(k_keypoint_bwd_a, dim3(2 * g->n_pairs, K), dim3(EQD_BLOCK), 0, st, g->seg_off, K, H, Z, scores,
                       lse, dY, dscores, du);
    int rc = eqd_check_launch("k_keypoint_bwd_a");
    if (rc) return rc;
    hipLaunchKernelGGL(k_keypoint_bwd_b, dim3(g->n_nodes), dim3(64), 0, st, g->seg_off, 2 * g->n_pairs, g->n_nodes, K,
                       scores, lse, u, dY, dscores, dH, dZ);
    return eqd_check_launch("k_keypoint_bwd_b");
}

// backward of k_head_u: one block per (head, group of HU_GROUP segments), sequential over its segments (deterministic):
// with one group the block adds its sums to dWk / dWq itself; with several (large batches: 128 segments at 64 pairs
// were 93 us on 50 of the 256 CUs) every block writes a partial [group][Wk | Wq][head][64 x 64] that the pass's
// fixed-order reduction adds up (EqdRedSeg).
//   dWk^(k) += qp (x) du / 8,  dqp = W_K^(k) du / 8,  dWq^(k) += dqp (x) qmean[partner],
//   dqm_part[s][k] = W_Q^(k)T dqp   (gradient wrt qmean[partner(s)], reduced over k later)
__global__ __launch_bounds__(EQD_BLOCK) void k_head_u_bwd(int B, int K, const float* __restrict__ Wk,
                                                          const float* __restrict__ Wq,
                                                          const float* __restrict__ qmean,
                                                          const float* __restrict__ qp, const float* __restrict__ du,
                                                          int du_chunks, float* __restrict__ dWk, float* __restrict__ dWq,
                                                          float* __restrict__ dqm_part, float* __restrict__ part,
                                                          int segs_per_group) {
    // The head's two 64 x 64 matrices and 16 segments' vectors at a time live in LDS (one coalesced fetch
    // each); all products then run out of LDS.  Sums over segments stay sequential (deterministic).
    // LDS traffic is the kernel's time (15 us for 16 segments with 4-byte reads): every operand is laid out so that the
    // contraction index is contiguous and read 16 bytes at a time - wk[j][c] rows, Wq transposed (wqT[c][j]), the
    // segments' vectors - and a thread's 16 outputs of the outer products are CONSECUTIVE j (4 broadcast b128 reads per
    // segment).  Every sum runs in the order it always did: same bits.
    // du_chunks > 1: du is the product backward's partial blocks [(s NC + chunk) K + k][64] (k_keypoint_bwd_mm), summed
    // here in chunk order instead of by a launch of their own.
    constexpr int WS = 68;      // row stride: 16-byte aligned, 4 c mod 64 banks - conflict-free b128 reads for 16 lanes
    __shared__ __attribute__((aligned(16))) float wk[64 * WS], wqT[64 * WS];
    __shared__ __attribute__((aligned(16))) float sqp[16 * 64], sdu[16 * 64], sqm[16 * 64], sdq[16 * 64];
    const int k = blockIdx.x, t = threadIdx.x;
    const int c = t & 63, j0 = t >> 6;  // thread owns elements (j, c) for j = 16 j0 .. 16 j0 + 15
    const int Sbeg0 = (int)blockIdx.y * segs_per_group;
    const int Send = 2 * B < Sbeg0 + segs_per_group ? 2 * B : Sbeg0 + segs_per_group;
    float pv[4][3];      // [i]: qp, du / 8, qmean[partner] of element t + 256 i of a 16-segment slab
    auto fetch_vectors = [&](int s0, float (&v)[4][3]) {
        const int ns = Send - s0 < 16 ? Send - s0 : 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i, sl = idx >> 6, cc = idx & 63;
            const int s = s0 + sl;
            const bool ok = sl < ns;
            const int partner = s < B ? s + B : s - B;
            v[i][0] = ok ? qp[((size_t)s * K + k) * 64 + cc] : 0.f;
            float d = 0.f;
            if (ok) {
                if (du_chunks <= 1) {
                    d = du[((size_t)s * K + k) * 64 + cc];
                } else {
                    for (int ch = 0; ch < du_chunks; ++ch) d += du[(((size_t)s * du_chunks + ch) * K + k) * 64 + cc];
                }
            }
            v[i][1] = d * 0.125f;
            v[i][2] = ok ? qmean[(size_t)partner * 64 + cc] : 0.f;
        }
    };
    float oldK[16], oldQ[16];
    {
        float4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = ((const float4*)(Wk + (size_t)k * 4096))[t + 256 * i];
            b[i] = ((const float4*)(Wq + (size_t)k * 4096))[t + 256 * i];
        }
        // the first 16 segments' vectors and (single group) the gradients this block adds to are requested behind the
        // weights, not after them: one memory round trip in front of the arithmetic instead of three
        fetch_vectors(Sbeg0, pv);
        if (!part) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oldK[i] = dWk[((size_t)k * 64 + 16 * j0 + i) * 64 + c];
                oldQ[i] = dWq[((size_t)k * 64 + 16 * j0 + i) * 64 + c];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i, r = idx >> 4, c4 = (idx & 15) * 4;
            *(float4*)&wk[r * WS + c4] = a[i];
            wqT[c4 * WS + r] = b[i].x; wqT[(c4 + 1) * WS + r] = b[i].y; wqT[(c4 + 2) * WS + r] = b[i].z; wqT[(c4 + 3) * WS + r] = b[i].w;
        }
    }
    float accK[16], accQ[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) accK[i] = accQ[i] = 0.f;
    const int Sbeg = Sbeg0, S2 = Send;
    for (int s0 = Sbeg; s0 < S2; s0 += 16) {
        const int ns = S2 - s0 < 16 ? S2 - s0 : 16;
        if (s0 != Sbeg) fetch_vectors(s0, pv);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i;
            sqp[idx] = pv[i][0];
            sdu[idx] = pv[i][1];
            sqm[idx] = pv[i][2];
        }
        __syncthreads();
        {          // dqp[s][j] = sum_c Wk[j][c] du[s][c]   (this thread: j = c, segments j0 + 4 i)
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int q = 0; q < 16; ++q) {
                const float4 w = *(const float4*)&wk[c * WS + 4 * q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 d = *(const float4*)&sdu[(j0 + 4 * i) * 64 + 4 * q];
                    a[i] += w.x * d.x; a[i] += w.y * d.y; a[i] += w.z * d.z; a[i] += w.w * d.w;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sdq[(j0 + 4 * i) * 64 + c] = a[i];
        }
        __syncthreads();
        {          // d qmean[partner(s)] part = Wq^T dqp[s]
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int q = 0; q < 16; ++q) {
                const float4 w = *(const float4*)&wqT[c * WS + 4 * q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 d = *(const float4*)&sdq[(j0 + 4 * i) * 64 + 4 * q];
                    a[i] += w.x * d.x; a[i] += w.y * d.y; a[i] += w.z * d.z; a[i] += w.w * d.w;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sl = j0 + 4 * i;
                if (sl < ns) dqm_part[((size_t)(s0 + sl) * K + k) * 64 + c] = a[i];
            }
        }
        for (int sl = 0; sl < ns; ++sl) {
            const float dc = sdu[sl * 64 + c], mc = sqm[sl * 64 + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 pq = *(const float4*)&sqp[sl * 64 + 16 * j0 + 4 * q];
                const float4 dq = *(const float4*)&sdq[sl * 64 + 16 * j0 + 4 * q];
                accK[4 * q] += pq.x * dc; accK[4 * q + 1] += pq.y * dc; accK[4 * q + 2] += pq.z * dc; accK[4 * q + 3] += pq.w * dc;
                accQ[4 * q] += dq.x * mc; accQ[4 * q + 1] += dq.y * mc; accQ[4 * q + 2] += dq.z * mc; accQ[4 * q + 3] += dq.w * mc;
            }
        }
    }
    if (!part) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = 16 * j0 + i;
            dWk[((size_t)k * 64 + j) * 64 + c] = oldK[i] + accK[i];
            dWq[((size_t)k * 64 + j) * 64 + c] = oldQ[i] + accQ[i];
        }
    } else {
        float* pk = part + ((size_t)blockIdx.y * 2 * K + k) * 4096;
        float* pq = pk + (size_t)K * 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = 16 * j0 + i;
            pk[j * 64 + c] = accK[i];
            pq[j * 64 + c] = accQ[i];
        }
    }
}
#define HU_GROUP 32      /* segments per block of k_head_u_bwd when the batch is split */
size_t eqd_head_u_bwd_partial_floats(int n_pairs, int K) {
    const int groups = (2 * n_pairs + HU_GROUP - 1) / HU_GROUP;
    return groups > 1 ? (size_t)groups * 2 * K * 4096 : 0;
}
int eqd_launch_head_u_bwd(const EqdGraph* g, int K, const float* Wk, const float* Wq, const float* qmean,
                          const float* qp, const float* du, float* dWk, float* dWq, float* dqm_part, hipStream_t st,
                          float* part, EqdRedList* defer, int du_chunks) {
    if (g->n_pairs == 0) return EQD_OK;
    const int groups = (2 * g->n_pairs + HU_GROUP - 1) / HU_GROUP;
    const bool mm = keypoint_mm_mode() != 0;      // the matrix-product form (eqd_headu_mm_inl.h) or the first kernel
    if (groups <= 1 || !part || !defer || defer->n + 2 > 512) {      // small batch (or no partial buffer): one block per head
        if (mm)
            hipLaunchKernelGGL(k_head_u_bwd_mm, dim3(K), dim3(EQD_BLOCK), 0, st, g->n_pairs, K, Wk, Wq, qmean, qp, du, du_chunks,
                               dWk, dWq, dqm_part, (float*)nullptr, 2 * g->n_pairs);
        else
            hipLaunchKernelGGL(k_head_u_bwd, dim3(K), dim3(EQD_BLOCK), 0, st, g->n_pairs, K, Wk, Wq, qmean, qp, du, du_chunks, dWk,
                               dWq, dqm_part, (float*)nullptr, 2 * g->n_pairs);
        return eqd_check_launch("k_head_u_bwd");
    }
    if (mm)
        hipLaunchKernelGGL(k_head_u_bwd_mm, dim3(K, groups), dim3(EQD_BLOCK), 0, st, g->n_pairs, K, Wk, Wq, qmean, qp, du,
                           du_chunks, dWk, dWq, dqm_part, part, HU_GROUP);
    else
        hipLaunchKernelGGL(k_head_u_bwd, dim3(K, groups), dim3(EQD_BLOCK), 0, st, g->n_pairs, K, Wk, Wq, qmean, qp, du, du_chunks,
                           dWk, dWq, dqm_part, part, HU_GROUP);
    const int n = K * 4096, stride = 2 * K * 4096;
    defer->seg[defer->n++] = EqdRedSeg{part, groups, stride, n, dWk, 0, 0, 0};
    defer->seg[defer->n++] = EqdRedSeg{part + n, groups, stride, n, dWq, 0, 0, 0};
    return eqd_check_launch("k_head_u_bwd");
}

// dqmean[p] = sum_k dqm_part[partner(p)][k]; dhm[i] = dqmean[seg(i)] / n_seg for the nodes of segment p
__global__ void k_qmean_bwd(const int32_t* __restrict__ seg_off, int B, int K, const float* __restrict__ dqm_part,
                            float* __restrict__ dhm) {
    // 1 024 threads: every 64-lane group sums the K parts of its column (same order everywhere, 10 loads in flight) and
    // then writes the rows rg, rg + 16, .. of the segment (one group writing all rows one after the other was 17 us)
    const int p = blockIdx.x, t = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int s = p < B ? p + B : p - B;        // the segment whose keypoints used qmean[p]
    float a = 0.f;
#pragma unroll 10
    for (int k = 0; k < K; ++k) a += dqm_part[((size_t)s * K + k) * 64 + t];
    const int n0 = seg_off[p], n1 = seg_off[p + 1];
    const float v = n1 > n0 ? a / (float)(n1 - n0) : 0.f;
    for (int i = n0 + rg; i < n1; i += 16) dhm[(size_t)i * 64 + t] = v;
}
int eqd_launch_qmean_bwd(const EqdGraph* g, int K, const float* dqm_part, float* dhm, hipStream_t st) {
    if (g->n_pairs == 0) return EQD_OK;
    hipLaunchKernelGGL(k_qmean_bwd, dim3(2 * g->n_pairs), dim3(1024), 0, st, g->seg_off, g->n_pairs, K, dqm_part, dhm);
    return eqd_check_launch("k_qmean_bwd");
}

// Operator-level backward of eqd_keypoint_pool_fwd (the three launches the model's backward issues for this stage).
extern "C" size_t eqd_keypoint_pool_bwd_workspace_bytes(const EqdGraph* g, int n_heads) {
    if (!g || n_heads < 1) return 0;
    const size_t N = (size_t)g->n_nodes, S = (size_t)2 * g->n_pairs, K = (size_t)n_heads;
    return eqd_align_up(N * K * sizeof(float)) + 2 * eqd_align_up(S * K * 64 * sizeof(float)) + 256;
}
extern "C" int eqd_keypoint_pool_bwd(const EqdGraph* g, int n_heads, const float* Wk, const float* Wq,
                                     const float* qmean, const float* qp, const float* u, const float* H, const float* Z,
                                     const float* scores, const float* lse, const float* dY, float* dH, float* dZ,
                                     float* dWk, float* dWq, float* d_hm, void* workspace, size_t ws_bytes,
                                     void* stream) {
    if (!g || !Wk || !Wq || !qmean || !qp || !u || !H || !Z || !scores || !lse || !dY || !dH || !dZ || !dWk || !dWq ||
        !d_hm || !workspace) {
        eqd_set_error("eqd_keypoint_pool_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    EqdArena A(workspace, ws_bytes);
    const size_t N = (size_t)g->n_nodes, S = (size_t)2 * g->n_pairs, K = (size_t)n_heads;
    float* dscores = A.take<float>(N * K);
    float* du = A.take<float>(S * K * 64);
    float* dqm_part = A.take<float>(S * K * 64);
    if (!A.ok) {
        eqd_set_error("eqd_keypoint_pool_bwd: workspace too small (%zu needed)", A.off);
        return EQD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int nch = 1;
    int rc = eqd_launch_keypoint_bwd(g, n_heads, H, Z, scores, lse, u, dY, dscores, du, dH, dZ, st, nullptr, &nch);
    if (rc) return rc;
    rc = eqd_launch_head_u_bwd(g, n_heads, Wk, Wq, qmean, qp, nch > 1 ? dscores : du, dWk, dWq, dqm_part, st, nullptr, nullptr, nch);
    if (rc) return rc;
    return eqd_launch_qmean_bwd(g, n_heads, dqm_part, d_hm, st);
}

// ---------------------------------------------------------------------------------------------
// Kabsch (rigid_docking_model.py:563-589), one WAVE per pair:
//   * lane k holds keypoint k of both proteins (two per lane when K > 64); the 6 mean and the 9 covariance sums are
//     wave reductions (DPP steps inside a row of 16 lanes, v_permlane swaps across rows; fixed order, no LDS);
//   * the 3x3 SVD is a one-sided (Hestenes) Jacobi iteration in fp32 with the matrix held ACROSS lanes: lane r (r = lane
//     & 3 < 3) of a quad holds row r of B = A V and row r of V; a column dot product is two DPP steps inside the quad,
//     the rotation parameters are computed redundantly, every lane rotates its own row (every quad does the same work);
//   * fp64 only where a sign or a threshold is decided: det A and the reference's stability guard (:574).
// Rounding-level differences to an fp64 SVD (the previous form, one lane per pair in fp64): T to ~3e-7.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float quad_sum(float v) {
    v += lane_xor<1>(v);
    v += lane_xor<2>(v);
    return v;
}
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {      // lane J of the caller's quad
#if defined(EQD_HOSTSIM) || defined(EQD_NO_DPP)
    return __shfl(v, (int)((threadIdx.x & 63) & ~3u) + J);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
#endif
}
__device__ __forceinline__ float kab_rcp(float x) {
#ifdef EQD_HOSTSIM
    return 1.f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float kab_rsqrt(float x) {      // one Newton step on v_rsq_f32: c^2 + s^2 = 1 to half an ulp
#ifdef EQD_HOSTSIM
    return 1.f / sqrtf(x);
#else
    const float r = __builtin_amdgcn_rsqf(x);
    return r * (1.5f - 0.5f * x * r * r);
#endif
}
// in: A (every lane, fp32), row = lane & 3.  out: this lane's row of U and of V (zeros on row 3), S descending (every lane)
__device__ __forceinline__ void svd3_lanes(const float (&A)[3][3], int row, float (&u)[3], float (&v)[3], float (&S)[3]) {
    float b[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        b[j] = row == 0 ? A[0][j] : row == 1 ? A[1][j] : row == 2 ? A[2][j] : 0.f;
        v[j] = row == j ? 1.f : 0.f;
    }
    const float TOL2 = 9e-14f;      // (3e-7)^2: columns orthogonal to fp32 rounding
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;      // (0,1), (0,2), (1,2): cyclic by rows
            const float al = quad_sum(b[p] * b[p]), be = quad_sum(b[q] * b[q]), ga = quad_sum(b[p] * b[q]);
            if (ga * ga <= TOL2 * (al * be)) continue;               // (uniform over the wave: every quad holds the same matrix)
            rotated = true;
            const float zeta = (be - al) * kab_rcp(2.f * ga);
            const float tt = copysignf(1.f, zeta) * kab_rcp(fabsf(zeta) + sqrtf(1.f + zeta * zeta));
            const float cs = kab_rsqrt(1.f + tt * tt), sn = cs * tt;
            const float bp = b[p], bq = b[q], vp = v[p], vq = v[q];
            b[p] = cs * bp - sn * bq;
            b[q] = sn * bp + cs * bq;
            v[p] = cs * vp - sn * vq;
            v[q] = sn * vp + cs * vq;
        }
        if (!rotated) break;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) S[j] = sqrtf(quad_sum(b[j] * b[j]));
    auto swap_cols = [&](int a, int c) {
        float t0 = S[a]; S[a] = S[c]; S[c] = t0;
        t0 = b[a]; b[a] = b[c]; b[c] = t0;
        t0 = v[a]; v[a] = v[c]; v[c] = t0;
    };
    // descending (selection sort on columns, decisions uniform)
    if (S[1] > S[0] && S[1] >= S[2]) swap_cols(0, 1);
    else if (S[2] > S[0]) swap_cols(0, 2);
    if (S[2] > S[1]) swap_cols(1, 2);
#pragma unroll
    for (int j = 0; j < 3; ++j) u[j] = S[j] > 0.f ? b[j] / S[j] : 0.f;
    if (!(S[2] > 0.f)) {   // rank deficient: complete the basis so that U stays orthonormal, U[:,2] = U[:,0] x U[:,1]
        const int base = (int)((threadIdx.x & 63) & ~3u), r1 = (row + 1) % 3, r2 = (row + 2) % 3;
        const float a0 = __shfl(u[0], base + r1), a1 = __shfl(u[1], base + r1);
        const float c0 = __shfl(u[0], base + r2), c1 = __shfl(u[1], base + r2);
        u[2] = row < 3 ? a0 * c1 - c0 * a1 : 0.f;
    }
}
__device__ __forceinline__ double det3f(const float (&A)[3][3]) {
    const double a00 = A[0][0], a01 = A[0][1], a02 = A[0][2], a10 = A[1][0], a11 = A[1][1], a12 = A[1][2], a20 = A[2][0],
                 a21 = A[2][1], a22 = A[2][2];
    return a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
}
__device__ __forceinline__ bool svd_unstable(const float (&Sf)[3]) {
    // rigid_docking_model.py:574 (the "+ eye" keeps the diagonal out of the min); thresholds evaluated in fp64
    const double S0 = Sf[0], S1 = Sf[1], S2 = Sf[2];
    const double mn = fmin(S0, fmin(S1, S2));
    if (mn < 1e-3) return true;
    const double s0 = S0 * S0, s1 = S1 * S1, s2 = S2 * S2;
    const double gap = fmin(fabs(s0 - s1), fmin(fabs(s0 - s2), fabs(s1 - s2)));
    return gap < 1e-2;
}
__device__ __forceinline__ float uniform_draw(unsigned seed, unsigned pair, unsigned it, unsigned c) {
    unsigned h = seed * 0x9E3779B9u + pair * 0x85EBCA6Bu + it * 0xC2B2AE35u + c * 0x27D4EB2Fu + 0x165667B1u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

#define KAB_MAXK 128
#define KAB_KPL (KAB_MAXK / 64)      /* keypoints per lane */
// this lane's keypoints of both proteins (zeros beyond K), their means and the covariance A = (Yr - mr)^T (Yl - ml)
__device__ __forceinline__ void kab_load(const float* __restrict__ Yl, const float* __restrict__ Yr, int K, int t,
                                         float (&yl)[KAB_KPL][3], float (&yr)[KAB_KPL][3]) {
#pragma unroll
    for (int kk = 0; kk < KAB_KPL; ++kk) {
        const int k = t + 64 * kk;
        const bool ok = k < K;
        const size_t o = (size_t)(ok ? k : 0) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = Yl[o + c], b = Yr[o + c];
            yl[kk][c] = ok ? a : 0.f;
            yr[kk][c] = ok ? b : 0.f;
        }
    }
}
__global__ __launch_bounds__(64) void k_kabsch_fwd(int B, int K, const float* __restrict__ Y,
                                                   const float* __restrict__ draws, int seed, float* __restrict__ T,
                                                   float* __restrict__ T2, float* __restrict__ bvec,
                                                   float* __restrict__ A_out, int32_t* __restrict__ status,
                                                   const int32_t* __restrict__ seg_off, const float* __restrict__ x0,
                                                   float* __restrict__ lig_out, double* __restrict__ usv) {
    // lig_out != NULL: the rigid apply of the pair's ligand nodes (k_apply_fwd's arithmetic) runs here as well
    const int p = blockIdx.x, t = threadIdx.x, row = t & 3;
    float yl[KAB_KPL][3], yr[KAB_KPL][3];
    kab_load(Y + (size_t)p * K * 3, Y + (size_t)(B + p) * K * 3, K, t, yl, yr);
    // (the rigid apply's first rows are requested now, beside the keypoints: one memory round trip instead of two)
    int n0 = 0, n1 = 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (lig_out) {
        n0 = seg_off[p]; n1 = seg_off[p + 1];      // ligand segments are the first B entries
        if (n0 + t < n1) {
            px = x0[(size_t)(n0 + t) * 3]; py = x0[(size_t)(n0 + t) * 3 + 1]; pz = x0[(size_t)(n0 + t) * 3 + 2];
        }
    }
    const float invK = 1.f / (float)K;
    float ml[3], mr[3], Af[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sl = 0.f, sr = 0.f;
#pragma unroll
        for (int kk = 0; kk < KAB_KPL; ++kk) {
            sl += yl[kk][c];
            sr += yr[kk][c];
        }
        ml[c] = wave_sum(sl) * invK;
        mr[c] = wave_sum(sr) * invK;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = 0.f;
#pragma unroll
            for (int kk = 0; kk < KAB_KPL; ++kk)
                a += (t + 64 * kk < K) ? (yr[kk][i] - mr[i]) * (yl[kk][j] - ml[j]) : 0.f;
            Af[i][j] = wave_sum(a);
        }
    float u[3], v[3], S[3];
    svd3_lanes(Af, row, u, v, S);
    int it = 0;
    while (svd_unstable(S)) {
        if (it >= 10) {   // reference: sys.exit(1) (:582-584); here: status 11, keep going
            it = 11;
            break;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)      // A = A + rand(3,3) * eye(3)  (:578), in fp32 like the reference
            Af[c][c] += draws ? draws[((size_t)p * 10 + it) * 3 + c] : uniform_draw((unsigned)seed, p, it, c);
        svd3_lanes(Af, row, u, v, S);
        ++it;
    }
    const float sd = det3f(Af) < 0.0 ? -1.f : 1.f;
    // T = U diag(1, 1, sd) V^T: lane `row` computes row `row` (the rows of V come from the quad's lanes 0..2)
    float Trow[3];
    {
        const float v00 = quad_bcast<0>(v[0]), v01 = quad_bcast<0>(v[1]), v02 = quad_bcast<0>(v[2]);
        const float v10 = quad_bcast<1>(v[0]), v11 = quad_bcast<1>(v[1]), v12 = quad_bcast<1>(v[2]);
        const float v20 = quad_bcast<2>(v[0]), v21 = quad_bcast<2>(v[1]), v22 = quad_bcast<2>(v[2]);
        Trow[0] = u[0] * v00 + u[1] * v01 + sd * u[2] * v02;
        Trow[1] = u[0] * v10 + u[1] * v11 + sd * u[2] * v12;
        Trow[2] = u[0] * v20 + u[1] * v21 + sd * u[2] * v22;
    }
    const float mrr = row == 0 ? mr[0] : row == 1 ? mr[1] : mr[2];
    const float brow = mrr - (Trow[0] * ml[0] + Trow[1] * ml[1] + Trow[2] * ml[2]);
    if (t < 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T[(size_t)p * 9 + t * 3 + j] = Trow[j];
            if (T2) T2[(size_t)p * 9 + t * 3 + j] = Trow[j];
            A_out[(size_t)p * 9 + t * 3 + j] = t == 0 ? Af[0][j] : t == 1 ? Af[1][j] : Af[2][j];
        }
        bvec[(size_t)p * 3 + t] = brow;
        if (usv) {       // U, S, V of the (guarded) A for the backward
            double* o = usv + (size_t)p * 21;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                o[t * 3 + j] = (double)u[j];
                o[12 + t * 3 + j] = (double)v[j];
            }
            o[9 + t] = (double)(t == 0 ? S[0] : t == 1 ? S[1] : S[2]);
        }
    }
    if (t == 0) status[p] = it;
    if (!lig_out) return;
    float Tm[3][3], bb[3];
    Tm[0][0] = quad_bcast<0>(Trow[0]); Tm[0][1] = quad_bcast<0>(Trow[1]); Tm[0][2] = quad_bcast<0>(Trow[2]);
    Tm[1][0] = quad_bcast<1>(Trow[0]); Tm[1][1] = quad_bcast<1>(Trow[1]); Tm[1][2] = quad_bcast<1>(Trow[2]);
    Tm[2][0] = quad_bcast<2>(Trow[0]); Tm[2][1] = quad_bcast<2>(Trow[1]); Tm[2][2] = quad_bcast<2>(Trow[2]);
    bb[0] = quad_bcast<0>(brow); bb[1] = quad_bcast<1>(brow); bb[2] = quad_bcast<2>(brow);
    for (int i = n0 + t; i < n1; i += 64) {
        float x = px, y = py, z = pz;
        if (i != n0 + t) {
            x = x0[(size_t)i * 3]; y = x0[(size_t)i * 3 + 1]; z = x0[(size_t)i * 3 + 2];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) lig_out[(size_t)i * 3 + r] = Tm[r][0] * x + Tm[r][1] * y + Tm[r][2] * z + bb[r];
    }
}

extern "C" int eqd_kabsch_fwd(int n_pairs, int n_heads, const float* Y, const float* svd_draws, int svd_seed, float* T,
                              float* b, float* A_out, int32_t* status, void* stream) {
    return eqd_kabsch_fwd_impl(n_pairs, n_heads, Y, svd_draws, svd_seed, T, nullptr, b, A_out, status, (hipStream_t)stream,
                               nullptr, nullptr, nullptr);
}
// g + lig_out: also lig_out = T x0 + b for every ligand node (eqd_rigid_apply_fwd fused in: one launch less)
int eqd_kabsch_fwd_impl(int n_pairs, int n_heads, const float* Y, const float* svd_draws, int svd_seed, float* T,
                        float* T2, float* b, float* A_out, int32_t* status, hipStream_t stream, const EqdGraph* g,
                        float* lig_out, double* usv) {
    if (!Y || !T || !b || !A_out || !status) {
        eqd_set_error("eqd_kabsch_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (n_pairs <= 0) return EQD_OK;
    if (n_heads > KAB_MAXK) {
        eqd_set_error("eqd_kabsch_fwd: %d keypoints > %d unsupported", n_heads, KAB_MAXK);
        return EQD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_kabsch_fwd, dim3(n_pairs), dim3(64), 0, (hipStream_t)stream, n_pairs, n_heads, Y, svd_draws,
                       svd_seed, T, T2, b, A_out, status, g ? g->seg_off : (const int32_t*)nullptr,
                       g ? g->x0 : (const float*)nullptr, g ? lig_out : (float*)nullptr, usv);
    return eqd_check_launch("k_kabsch_fwd");
}

// Closed-form backward: with G = dL/dT (including the b = mean_r - T mean_l
// path), M = U^T G V, c = (1, 1, sign det A):
//   dP_ij = (c_j M_ij - c_i M_ji) / (s_j + c_i c_j s_i)  (i != j),  dA = U dP V^T
// One wave per pair like the forward: lane k holds keypoint k of both proteins, sums over keypoints / ligand nodes are
// wave reductions, the 3x3 algebra (fp64, ~150 multiply-adds) runs redundantly on every lane - no LDS, no barrier.
__global__ __launch_bounds__(64) void k_kabsch_bwd(int B, int K, const float* __restrict__ Y,
                                                   const float* __restrict__ A_in, const float* __restrict__ T,
                                                   const float* __restrict__ dT, const float* __restrict__ db,
                                                   const float* __restrict__ dYl_ext,
                                                   const float* __restrict__ dYr_ext, int use_ext,
                                                   float* __restrict__ dY, const int32_t* __restrict__ seg_off,
                                                   const float* __restrict__ x0, const float* __restrict__ d_lig,
                                                   const double* __restrict__ usv) {
    const int p = blockIdx.x, t = threadIdx.x;
    // seg_off != NULL: the backward of the rigid apply (k_apply_bwd: dT += d_lig^T x0, db += colsum d_lig over the pair's
    // ligand nodes) is taken here, on top of the external dT / db
    float yl[KAB_KPL][3], yr[KAB_KPL][3];
    kab_load(Y + (size_t)p * K * 3, Y + (size_t)(B + p) * K * 3, K, t, yl, yr);
    float app[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) app[i] = 0.f;
    if (seg_off) {
        float acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0.f;
        const int n0 = seg_off[p], n1 = seg_off[p + 1];
        for (int i = n0 + t; i < n1 && d_lig; i += 64) {
            const float g0 = d_lig[(size_t)i * 3], g1 = d_lig[(size_t)i * 3 + 1], g2 = d_lig[(size_t)i * 3 + 2];
            const float x = x0[(size_t)i * 3], y = x0[(size_t)i * 3 + 1], z = x0[(size_t)i * 3 + 2];
            acc[0] += g0 * x; acc[1] += g0 * y; acc[2] += g0 * z;
            acc[3] += g1 * x; acc[4] += g1 * y; acc[5] += g1 * z;
            acc[6] += g2 * x; acc[7] += g2 * y; acc[8] += g2 * z;
            acc[9] += g0; acc[10] += g1; acc[11] += g2;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) app[i] = wave_sum(acc[i]);
    }
    float* dYl = dY + (size_t)p * K * 3;
    float* dYr = dY + (size_t)(B + p) * K * 3;
    const float invK = 1.f / (float)K;
    float ml[3], mr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sl = 0.f, sr = 0.f;
#pragma unroll
        for (int kk = 0; kk < KAB_KPL; ++kk) {
            sl += yl[kk][c];
            sr += yr[kk][c];
        }
        ml[c] = wave_sum(sl) * invK;
        mr[c] = wave_sum(sr) * invK;
    }
    double U[3][3], S[3], V[3][3], G[3][3], Tm[3][3], dbv[3], dA[3][3];
    float Af[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dbv[i] = (db ? (double)db[(size_t)p * 3 + i] : 0.0) + (double)app[9 + i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Af[i][j] = A_in[(size_t)p * 9 + i * 3 + j];
            Tm[i][j] = (double)T[(size_t)p * 9 + i * 3 + j];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            G[i][j] = (dT ? (double)dT[(size_t)p * 9 + i * 3 + j] : 0.0) + (double)app[i * 3 + j] - dbv[i] * (double)ml[j];
    if (usv) {       // saved by the forward
        const double* o = usv + (size_t)p * 21;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            S[i] = o[9 + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                U[i][j] = o[i * 3 + j];
                V[i][j] = o[12 + i * 3 + j];
            }
        }
    } else {         // the operator on its own: the forward's decomposition again (same arithmetic, same bits)
        float u[3], v[3], Sf[3];
        svd3_lanes(Af, t & 3, u, v, Sf);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            S[j] = (double)Sf[j];
            U[0][j] = (double)quad_bcast<0>(u[j]); U[1][j] = (double)quad_bcast<1>(u[j]); U[2][j] = (double)quad_bcast<2>(u[j]);
            V[0][j] = (double)quad_bcast<0>(v[j]); V[1][j] = (double)quad_bcast<1>(v[j]); V[2][j] = (double)quad_bcast<2>(v[j]);
        }
    }
    {
        const double c[3] = {1.0, 1.0, det3f(Af) < 0.0 ? -1.0 : 1.0};
        double M[3][3], dP[3][3], GV[3][3], UdP[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) GV[a][j] = G[a][0] * V[0][j] + G[a][1] * V[1][j] + G[a][2] * V[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[i][j] = U[0][i] * GV[0][j] + U[1][i] * GV[1][j] + U[2][i] * GV[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (i == j) {
                    dP[i][j] = 0.0;
                    continue;
                }
                double den = S[j] + c[i] * c[j] * S[i];
                if (fabs(den) < 1e-12) den = den < 0 ? -1e-12 : 1e-12;
                dP[i][j] = (c[j] * M[i][j] - c[i] * M[j][i]) / den;
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) UdP[i][b2] = U[i][0] * dP[0][b2] + U[i][1] * dP[1][b2] + U[i][2] * dP[2][b2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) dA[i][j] = UdP[i][0] * V[j][0] + UdP[i][1] * V[j][1] + UdP[i][2] * V[j][2];
    }
    // means: d mean_r = db ; d mean_l = -T^T db
    double dml[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) dml[j] = -(Tm[0][j] * dbv[0] + Tm[1][j] * dbv[1] + Tm[2][j] * dbv[2]);
    // centred-point gradients, then un-centre (the mean of the centred gradients is removed)
    float gr[KAB_KPL][3], gl[KAB_KPL][3], gmr[3], gml[3];
#pragma unroll
    for (int kk = 0; kk < KAB_KPL; ++kk) {
        const bool ok = t + 64 * kk < K;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = 0, b2 = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                a += dA[i][j] * (double)(yl[kk][j] - ml[j]);
                b2 += dA[j][i] * (double)(yr[kk][j] - mr[j]);
            }
            gr[kk][i] = ok ? (float)a : 0.f;
            gl[kk][i] = ok ? (float)b2 : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float sr = 0.f, sl = 0.f;
#pragma unroll
        for (int kk = 0; kk < KAB_KPL; ++kk) {
            sr += gr[kk][i];
            sl += gl[kk][i];
        }
        gmr[i] = wave_sum(sr) * invK;
        gml[i] = wave_sum(sl) * invK;
    }
#pragma unroll
    for (int kk = 0; kk < KAB_KPL; ++kk) {
        const int k = t + 64 * kk;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float vr = gr[kk][i] - gmr[i] + (float)(dbv[i] * (double)invK);
            const float vl = gl[kk][i] - gml[i] + (float)(dml[i] * (double)invK);
            if (use_ext) {   // dY = external gradient (or 0) + Kabsch path; no pre-initialised buffer needed
                dYr[k * 3 + i] = vr + (dYr_ext ? dYr_ext[(size_t)p * K * 3 + k * 3 + i] : 0.f);
                dYl[k * 3 + i] = vl + (dYl_ext ? dYl_ext[(size_t)p * K * 3 + k * 3 + i] : 0.f);
            } else {
                dYr[k * 3 + i] += vr;
                dYl[k * 3 + i] += vl;
            }
        }
    }
}

extern "C" int eqd_kabsch_bwd(int n_pairs, int n_heads, const float* Y, const float* A, const float* T,
                              const float* dT, const float* db, float* dY, void* stream) {
    if (!Y || !A || !T || !dY) {
        eqd_set_error("eqd_kabsch_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (n_pairs <= 0) return EQD_OK;
    if (n_heads > KAB_MAXK) {
        eqd_set_error("eqd_kabsch_bwd: %d keypoints > %d unsupported", n_heads, KAB_MAXK);
        return EQD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_kabsch_bwd, dim3(n_pairs), dim3(64), 0, (hipStream_t)stream, n_pairs, n_heads, Y,
                       A, T, dT, db, (const float*)nullptr, (const float*)nullptr, 0, dY, (const int32_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const double*)nullptr);
    return eqd_check_launch("k_kabsch_bwd");
}
// g != NULL: dT / db are the EXTERNAL gradients (may be NULL) and the rigid apply's backward from d_lig is added inside
int eqd_kabsch_bwd_impl(int n_pairs, int n_heads, const float* Y, const float* A, const float* T, const float* dT,
                        const float* db, const float* dYl_ext, const float* dYr_ext, float* dY, hipStream_t stream,
                        const EqdGraph* g, const float* d_lig, const double* usv) {
    if (!Y || !A || !T || !dY) {
        eqd_set_error("eqd_kabsch_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (n_pairs <= 0) return EQD_OK;
    if (n_heads > KAB_MAXK) {
        eqd_set_error("eqd_kabsch_bwd: %d keypoints > %d unsupported", n_heads, KAB_MAXK);
        return EQD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_kabsch_bwd, dim3(n_pairs), dim3(64), 0, (hipStream_t)stream, n_pairs, n_heads, Y,
                       A, T, dT, db, dYl_ext, dYr_ext, 1, dY, g ? g->seg_off : (const int32_t*)nullptr,
                       g ? g->x0 : (const float*)nullptr, g ? d_lig : (const float*)nullptr, usv);
    return eqd_check_launch("k_kabsch_bwd");
}

// ---------------------------------------------------------------------------------------------
// rigid apply
// ---------------------------------------------------------------------------------------------
__global__ void k_apply_fwd(const int32_t* __restrict__ seg_off, int B, int n_lig, const float* __restrict__ x0,
                            const float* __restrict__ T, const float* __restrict__ b, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lig) return;
    const int p = find_segment(seg_off, B, i);   // ligand segments are the first B entries
    const float* t = T + (size_t)p * 9;
    const float x = x0[(size_t)i * 3], y = x0[(size_t)i * 3 + 1], z = x0[(size_t)i * 3 + 2];
    for (int r = 0; r < 3; ++r) out[(size_t)i * 3 + r] = t[r * 3] * x + t[r * 3 + 1] * y + t[r * 3 + 2] * z + b[(size_t)p * 3 + r];
}
extern "C" int eqd_rigid_apply_fwd(const EqdGraph* g, const float* T, const float* b, float* lig_out, void* stream) {
    if (!g || !T || !b || !lig_out) {
        eqd_set_error("eqd_rigid_apply_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (g->n_lig <= 0) return EQD_OK;
    hipLaunchKernelGGL(k_apply_fwd, dim3((g->n_lig + 255) / 256), dim3(256), 0, (hipStream_t)stream, g->seg_off,
                       g->n_pairs, g->n_lig, g->x0, T, b, lig_out);
    return eqd_check_launch("k_apply_fwd");
}

__global__ __launch_bounds__(EQD_BLOCK) void k_apply_bwd(const int32_t* __restrict__ seg_off,
                                                         const float* __restrict__ x0,
                                                         const float* __restrict__ d_lig,
                                                         const float* __restrict__ dT_ext,
                                                         const float* __restrict__ db_ext, int use_ext,
                                                         float* __restrict__ dT, float* __restrict__ db) {
    __shared__ float red[4];
    const int p = blockIdx.x, t = threadIdx.x;
    const int n0 = seg_off[p], n1 = seg_off[p + 1];
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int i = n0 + t; i < n1 && d_lig; i += EQD_BLOCK) {
        const float g0 = d_lig[(size_t)i * 3], g1 = d_lig[(size_t)i * 3 + 1], g2 = d_lig[(size_t)i * 3 + 2];
        const float x = x0[(size_t)i * 3], y = x0[(size_t)i * 3 + 1], z = x0[(size_t)i * 3 + 2];
        acc[0] += g0 * x; acc[1] += g0 * y; acc[2] += g0 * z;
        acc[3] += g1 * x; acc[4] += g1 * y; acc[5] += g1 * z;
        acc[6] += g2 * x; acc[7] += g2 * y; acc[8] += g2 * z;
        acc[9] += g0; acc[10] += g1; acc[11] += g2;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float s = block_reduce_sum(acc[i], red);
        if (t == 0) {
            if (use_ext) {
                if (i < 9) dT[(size_t)p * 9 + i] = s + (dT_ext ? dT_ext[(size_t)p * 9 + i] : 0.f);
                else db[(size_t)p * 3 + (i - 9)] = s + (db_ext ? db_ext[(size_t)p * 3 + (i - 9)] : 0.f);
            } else {
                if (i < 9) dT[(size_t)p * 9 + i] += s; else db[(size_t)p * 3 + (i - 9)] += s;
            }
        }
    }
}
extern "C" int eqd_rigid_apply_bwd(const EqdGraph* g, const float* d_lig, float* dT, float* db, void* stream) {
    if (!g || !d_lig || !dT || !db) {
        eqd_set_error("eqd_rigid_apply_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (g->n_pairs <= 0) return EQD_OK;
    hipLaunchKernelGGL(k_apply_bwd, dim3(g->n_pairs), dim3(EQD_BLOCK), 0, (hipStream_t)stream, g->seg_off, g->x0, d_lig,
                       (const float*)nullptr, (const float*)nullptr, 0, dT, db);
    return eqd_check_launch("k_apply_bwd");
}
int eqd_rigid_apply_bwd_impl(const EqdGraph* g, const float* d_lig, const float* dT_ext, const float* db_ext, float* dT,
                             float* db, hipStream_t st) {
    if (!g || !dT || !db) {
        eqd_set_error("eqd_rigid_apply_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (g->n_pairs <= 0) return EQD_OK;
    hipLaunchKernelGGL(k_apply_bwd, dim3(g->n_pairs), dim3(EQD_BLOCK), 0, st, g->seg_off, g->x0, d_lig, dT_ext, db_ext, 1,
                       dT, db);
    return eqd_check_launch("k_apply_bwd");
}
