// Block-diagonal cross-graph attention (ligand <-> receptor of the same pair), forward and backward.
//
// Reference arithmetic replaced (src/model/rigid_docking_model.py):
//   :68-78   get_mask        dense 0/1 (sum N_l x sum N_r) mask rebuilt every layer   -> never built
//   :46-64   compute_cross_attention: a = mask * (Q K^T) - 1000 (1 - mask); softmax; a V
//            (single head, NO 1/sqrt(d) scale), called for both directions (:247-256)
// The reference's softmax runs over the whole batch with a -1000 fill; exp(-1000 - max) underflows
// to exactly 0 in fp32 unless every in-pair logit is < -900, so the per-pair softmax computed here
// is bit-identical in practice (0.0 difference on the golden vectors).
//
// Flash-style: a wave owns 32 nodes of one protein ("block") and streams over the partner
// protein ("other") in tiles, with an online softmax; score tiles and the P.V / dS.K products
// chain through registers in the transposed MFMA formulation (eqd_common.h), so neither the
// (N_l x N_r) scores nor any transposed copy ever exists in memory.
//   forward / backward pass 1: block = queries (N axis), other = keys   (M axis)  -> out / dq
//   backward pass 2          : block = keys    (N axis), other = queries(M axis)  -> dk, dv
// One work list (EqdGraph.att_items) serves all three.
// A WORKGROUP owns one work item; its 4 waves split the partner protein's tiles round-robin and
// merge their partial results through LDS (flash-decoding style split along the streamed axis):
// 4x the waves and 4x shorter dependent load->MFMA chains on DB5-sized proteins (200 residues =
// 4 tiles), where a single wave per item left the chip latency-bound.
#include "eqd_common.h"

template <int DB>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_fwd(EqdGraph G, int d, const float* __restrict__ q,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        float* __restrict__ out, float* __restrict__ lse) {
    constexpr int KS = DB * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    __shared__ float red[EQD_WAVES][DB * 2 * 4 * 64];
    __shared__ float sm_m[EQD_WAVES][32], sm_l[EQD_WAVES][32];
    const int item = blockIdx.x;
    const int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    int rowq[2] = {b0 + l15, b0 + 16 + l15};
    bool qv[2] = {rowq[0] < b1, rowq[1] < b1};
    float qf[2][KS];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + g;
            qf[nb][ks] = (qv[nb] && kk < d) ? q[(size_t)rowq[nb] * d + kk] : 0.f;
        }
    f32x4 O[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        O[db][0] = f4zero();
        O[db][1] = f4zero();
    }
    float mrun[2] = {EQD_NEG_BIG, EQD_NEG_BIG}, lrun[2] = {0.f, 0.f};
    for (int kt = o0 + 64 * wave; kt < o1; kt += 64 * EQD_WAVES) {
        f32x4 S[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            S[mb][0] = f4zero();
            S[mb][1] = f4zero();
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + g;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int key = kt + 16 * mb + l15;
                const float a = (key < o1 && kk < d) ? k[(size_t)key * d + kk] : 0.f;
                S[mb][0] = mfma4(a, qf[0][ks], S[mb][0]);
                S[mb][1] = mfma4(a, qf[1][ks], S[mb][1]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float mx = EQD_NEG_BIG;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float s = key < o1 ? S[mb][nb][r] : EQD_NEG_BIG;
                    S[mb][nb][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = group_max(mx);
            const float mnew = fmaxf(mrun[nb], mx);
            const float alpha = expf(mrun[nb] - mnew);
            float ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float p = key < o1 ? expf(S[mb][nb][r] - mnew) : 0.f;
                    S[mb][nb][r] = p;
                    ps += p;
                }
            lrun[nb] = lrun[nb] * alpha + group_sum(ps);
            mrun[nb] = mnew;
#pragma unroll
            for (int db = 0; db < DB; ++db) O[db][nb] *= alpha;
        }
#pragma unroll
        for (int mbk = 0; mbk < 4; ++mbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * mbk + 4 * g + r;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int f = 16 * db + l15;
                    const float a = (key < o1 && f < d) ? v[(size_t)key * d + f] : 0.f;
                    O[db][0] = mfma4(a, S[mbk][0][r], O[db][0]);
                    O[db][1] = mfma4(a, S[mbk][1][r], O[db][1]);
                }
            }
    }
    // ---- merge the 4 waves' partial softmax states -----------------------------------------------
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][((db * 2 + nb) * 4 + r) * 64 + lane] = O[db][nb][r];
    if (g == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            sm_m[wave][16 * nb + l15] = mrun[nb];
            sm_l[wave][16 * nb + l15] = lrun[nb];
        }
    }
    __syncthreads();
    float sc[2][EQD_WAVES], inv[2], mtot[2], ltot[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        float mm = EQD_NEG_BIG;
#pragma unroll
        for (int w = 0; w < EQD_WAVES; ++w) mm = fmaxf(mm, sm_m[w][16 * nb + l15]);
        float ll = 0.f;
#pragma unroll
        for (int w = 0; w < EQD_WAVES; ++w) {
            sc[nb][w] = expf(sm_m[w][16 * nb + l15] - mm);
            ll += sm_l[w][16 * nb + l15] * sc[nb][w];
        }
        mtot[nb] = mm;
        ltot[nb] = ll;
        inv[nb] = ll > 0.f ? 1.f / ll : 0.f;
    }
    for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = 0.f;
#pragma unroll
                for (int w = 0; w < EQD_WAVES; ++w) o += red[w][((db * 2 + nb) * 4 + r) * 64 + lane] * sc[nb][w];
                const int f = 16 * db + 4 * g + r;
                if (f < d) out[(size_t)rowq[nb] * d + f] = o * inv[nb];
            }
        }
    if (wave == 0 && g == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            if (qv[nb]) lse[rowq[nb]] = ltot[nb] > 0.f ? mtot[nb] + logf(ltot[nb]) : 0.f;
    }
}

// backward pass 1: dq for the block's queries; also writes delta[q] = sum_f dO[q][f] O[q][f]
template <int DB>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_bwd_q(EqdGraph G, int d, const float* __restrict__ q,
                                                          const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ out,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ d_out, float* __restrict__ dq,
                                                          float* __restrict__ delta) {
    constexpr int KS = DB * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    __shared__ float red[EQD_WAVES][DB * 2 * 4 * 64];
    const int item = blockIdx.x;
    const int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    int rowq[2] = {b0 + l15, b0 + 16 + l15};
    bool qv[2] = {rowq[0] < b1, rowq[1] < b1};
    float qf[2][KS], dof[2][KS], dl[2], lq[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + g;
            const bool ok = qv[nb] && kk < d;
            const size_t o = (size_t)rowq[nb] * d + kk;
            qf[nb][ks] = ok ? q[o] : 0.f;
            dof[nb][ks] = ok ? d_out[o] : 0.f;
            s += ok ? dof[nb][ks] * out[o] : 0.f;
        }
        dl[nb] = group_sum(s);
        lq[nb] = qv[nb] ? lse[rowq[nb]] : 0.f;
        if (wave == 0 && g == 0 && qv[nb]) delta[rowq[nb]] = dl[nb];
    }
    f32x4 dQ[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        dQ[db][0] = f4zero();
        dQ[db][1] = f4zero();
    }
    for (int kt = o0 + 32 * wave; kt < o1; kt += 32 * EQD_WAVES) {
        f32x4 S[2][2], dP[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            S[mb][0] = S[mb][1] = f4zero();
            dP[mb][0] = dP[mb][1] = f4zero();
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + g;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int key = kt + 16 * mb + l15;
                const bool ok = key < o1 && kk < d;
                const float a = ok ? k[(size_t)key * d + kk] : 0.f;
                const float b = ok ? v[(size_t)key * d + kk] : 0.f;
                S[mb][0] = mfma4(a, qf[0][ks], S[mb][0]);
                S[mb][1] = mfma4(a, qf[1][ks], S[mb][1]);
                dP[mb][0] = mfma4(b, dof[0][ks], dP[mb][0]);
                dP[mb][1] = mfma4(b, dof[1][ks], dP[mb][1]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float p = key < o1 ? expf(S[mb][nb][r] - lq[nb]) : 0.f;
                    S[mb][nb][r] = p * (dP[mb][nb][r] - dl[nb]);
                }
#pragma unroll
        for (int mbk = 0; mbk < 2; ++mbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * mbk + 4 * g + r;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int f = 16 * db + l15;
                    const float a = (key < o1 && f < d) ? k[(size_t)key * d + f] : 0.f;
                    dQ[db][0] = mfma4(a, S[mbk][0][r], dQ[db][0]);
                    dQ[db][1] = mfma4(a, S[mbk][1][r], dQ[db][1]);
                }
            }
    }
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][((db * 2 + nb) * 4 + r) * 64 + lane] = dQ[db][nb][r];
    __syncthreads();
    for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                const int f = 16 * db + 4 * g + r;
                if (f < d) dq[(size_t)rowq[nb] * d + f] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
            }
        }
}

// backward pass 2: dk, dv for the block's keys (queries = the partner protein)
template <int DB>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_bwd_kv(EqdGraph G, int d, const float* __restrict__ q,
                                                           const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ d_out,
                                                           const float* __restrict__ delta, float* __restrict__ dk,
                                                           float* __restrict__ dv) {
    constexpr int KS = DB * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    __shared__ float red[EQD_WAVES][DB * 2 * 4 * 64];
    const int item = blockIdx.x;
    const int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    int rowk[2] = {b0 + l15, b0 + 16 + l15};
    bool kvd[2] = {rowk[0] < b1, rowk[1] < b1};
    float kf[2][KS], vf[2][KS];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + g;
            const bool ok = kvd[nb] && kk < d;
            const size_t o = (size_t)rowk[nb] * d + kk;
            kf[nb][ks] = ok ? k[o] : 0.f;
            vf[nb][ks] = ok ? v[o] : 0.f;
        }
    f32x4 dK[DB][2], dV[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        dK[db][0] = dK[db][1] = f4zero();
        dV[db][0] = dV[db][1] = f4zero();
    }
    for (int qt = o0 + 32 * wave; qt < o1; qt += 32 * EQD_WAVES) {
        f32x4 S[2][2], dP[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            S[mb][0] = S[mb][1] = f4zero();
            dP[mb][0] = dP[mb][1] = f4zero();
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + g;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int qr = qt + 16 * mb + l15;
                const bool ok = qr < o1 && kk < d;
                const float a = ok ? q[(size_t)qr * d + kk] : 0.f;
                const float b = ok ? d_out[(size_t)qr * d + kk] : 0.f;
                S[mb][0] = mfma4(a, kf[0][ks], S[mb][0]);
                S[mb][1] = mfma4(a, kf[1][ks], S[mb][1]);
                dP[mb][0] = mfma4(b, vf[0][ks], dP[mb][0]);
                dP[mb][1] = mfma4(b, vf[1][ks], dP[mb][1]);
            }
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qt + 16 * mb + 4 * g + r;
                const bool ok = qr < o1;
                const float lq = ok ? lse[qr] : 0.f;
                const float dl = ok ? delta[qr] : 0.f;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const float p = ok ? expf(S[mb][nb][r] - lq) : 0.f;
                    S[mb][nb][r] = p;
                    dP[mb][nb][r] = p * (dP[mb][nb][r] - dl);
                }
            }
#pragma unroll
        for (int mbq = 0; mbq < 2; ++mbq)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qt + 16 * mbq + 4 * g + r;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int f = 16 * db + l15;
                    const bool ok = qr < o1 && f < d;
                    const float a = ok ? d_out[(size_t)qr * d + f] : 0.f;
                    const float b = ok ? q[(size_t)qr * d + f] : 0.f;
                    dV[db][0] = mfma4(a, S[mbq][0][r], dV[db][0]);
                    dV[db][1] = mfma4(a, S[mbq][1][r], dV[db][1]);
                    dK[db][0] = mfma4(b, dP[mbq][0][r], dK[db][0]);
                    dK[db][1] = mfma4(b, dP[mbq][1][r], dK[db][1]);
                }
            }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {     // dK then dV through the same LDS buffer
        if (pass) __syncthreads();
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[wave][((db * 2 + nb) * 4 + r) * 64 + lane] = pass ? dV[db][nb][r] : dK[db][nb][r];
        __syncthreads();
        float* __restrict__ dst = pass ? dv : dk;
        for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (!kvd[nb]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                    const int f = 16 * db + 4 * g + r;
                    if (f < d) dst[(size_t)rowk[nb] * d + f] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
                }
            }
    }
}

extern "C" int eqd_cross_attention_fwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                       float* out, float* lse, void* stream) {
    if (!g || !q || !k || !v || !out || !lse) {
        eqd_set_error("eqd_cross_attention_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (d <= 0 || d > 80) {
        eqd_set_error("eqd_cross_attention_fwd: feature width %d outside 1..80", d);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    dim3 grid(g->n_att_items);
    if (d <= 64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd<4>), grid, dim3(EQD_BLOCK), 0, (hipStream_t)stream, *g, d, q, k, v,
                           out, lse);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd<5>), grid, dim3(EQD_BLOCK), 0, (hipStream_t)stream, *g, d, q, k, v,
                           out, lse);
    return eqd_check_launch("k_attn_fwd");
}

extern "C" int eqd_cross_attention_bwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                       const float* out, const float* lse, const float* d_out, float* dq, float* dk,
                                       float* dv, float* delta, void* stream) {
    if (!g || !q || !k || !v || !out || !lse || !d_out || !dq || !dk || !dv || !delta) {
        eqd_set_error("eqd_cross_attention_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (d <= 0 || d > 80) {
        eqd_set_error("eqd_cross_attention_bwd: feature width %d outside 1..80", d);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(g->n_att_items);
    if (d <= 64) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_q<4>), grid, dim3(EQD_BLOCK), 0, st, *g, d, q, k, v, out, lse,
                           d_out, dq, delta);
        int rc = eqd_check_launch("k_attn_bwd_q");
        if (rc) return rc;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kv<4>), grid, dim3(EQD_BLOCK), 0, st, *g, d, q, k, v, lse, d_out,
                           delta, dk, dv);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_q<5>), grid, dim3(EQD_BLOCK), 0, st, *g, d, q, k, v, out, lse,
                           d_out, dq, delta);
        int rc = eqd_check_launch("k_attn_bwd_q");
        if (rc) return rc;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kv<5>), grid, dim3(EQD_BLOCK), 0, st, *g, d, q, k, v, lse, d_out,
                           delta, dk, dv);
    }
    return eqd_check_launch("k_attn_bwd_kv");
}
