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
// Flash-style: a WORKGROUP owns 32 nodes of one protein ("block"); its 4 waves split the partner
// protein ("other") into 32-row tiles round-robin, each wave running an online softmax over its
// tiles, and the partial states are merged through LDS (flash-decoding style).  Score tiles and
// the P.V / dS.K products chain through registers in the transposed MFMA formulation
// (eqd_common.h), so neither the (N_l x N_r) scores nor any transposed copy ever exists in memory.
//   forward / backward pass 1: block = queries (N axis), other = keys    (M axis)  -> out / dq
//   backward pass 2          : block = keys    (N axis), other = queries (M axis)  -> dk, dv
// One work list (EqdGraph.att_items) serves all three.
//
// Memory: every tile is a CONTIGUOUS [32][d] slab of a row-major matrix, so a wave fetches it with
// fully coalesced loads (float4 when d == 64) that are all in flight at once, parks them in
// registers while it multiplies the previous tile, then writes them to its private LDS tile
// (row stride 16*DB + 4 floats); MFMA operands are read from LDS.  One HBM/L2 round trip per tile.
#include "eqd_common.h"
#include "eqd_attn_fwd_inl.h"
#include "eqd_gather_inl.h"
#include "eqd_attn_lb_inl.h"

#include <stdlib.h>
#include <string.h>

// backward pass 1: dq for the block's queries; also writes delta[q] = sum_f dO[q][f] O[q][f]
// LDS of the backward passes: 2 block tiles + 2 x EQD_WAVES streamed tiles + the merge buffer + 32 floats per wave
// ALIAS (the merged float4 launch): the two block tiles are dead once their fragments sit in registers, so they live in
// the streamed V / dO tiles of waves 2 and 3 (one barrier before the tile loop), and each wave's part of the merge buffer
// is its own first streamed tile (written after its last tile, as in the forward): 70 KB instead of 120 KB at d = 64,
// i.e. TWO workgroups per CU - with one, every LDS-read -> MFMA dependency of the single wave per SIMD is exposed.
template <int DB, bool ALIAS>
struct AttnBwdSmem {
    typedef AttnCfg<DB> C;
    enum { ALIASED = ALIAS ? 1 : 0 };
    float str_[2][EQD_WAVES][C::TILE];
    float blk_[2][ALIAS ? 4 : C::TILE];
    float red_[EQD_WAVES][ALIAS ? 4 : C::RED];
    float dls[EQD_WAVES][32];
    static_assert(C::RED <= C::TILE, "merge buffer must fit a streamed tile");
    __device__ __forceinline__ float* blk(int i) { return ALIAS ? str_[1][2 + i] : blk_[i]; }
    __device__ __forceinline__ float (*str(int i))[C::TILE] { return str_[i]; }
    __device__ __forceinline__ float* red(int w) { return ALIAS ? str_[0][w] : red_[w]; }
};

template <int DB, bool FAST, int NB, class SM, bool BF = false>
__device__ __forceinline__ void attn_bwd_q_body(SM& sm, const EqdGraph& G, int item, int d,
                                                const float* __restrict__ q, const float* __restrict__ k,
                                                const float* __restrict__ v, const float* __restrict__ out,
                                                const float* __restrict__ lse, const float* __restrict__ d_out,
                                                float* __restrict__ dq, float* __restrict__ delta, int half = 0,
                                                float qk_slope = 1.f) {
    // qk_slope: q is LeakyReLU(z) (att_mlp_Q, rigid_docking_model.py:130-133) - the stored gradient is multiplied by
    // LeakyReLU'(q) (1 | qk_slope), i.e. it is the gradient w.r.t. the pre-activation.  Both consumers (the dh job and
    // the weight-gradient GEMM) applied that mask themselves until round 3, each from a second stream of rows (the q rows:
    // +1.0 % / +1.4 % of the fp32 / bf16 step at 64 x (300, 300)); 1 = off (the operator-level entry points)
    typedef AttnCfg<DB> C;
    constexpr int DS = C::DS, KS = C::KS;
    float* Qt = sm.blk(0);
    float* Gt = sm.blk(1);   // dO rows of the block
    float (*Kt)[C::TILE] = sm.str(0);
    float (*Vt)[C::TILE] = sm.str(1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {          // half blocks: rows b0 + 16 half .. of the item (see k_attn_fwd)
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;   // short half block, or a padding item of the XCD-interleaved work list
    int rowq[NB];
    bool qv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowq[nb] = b0 + 16 * nb + l15;
        qv[nb] = rowq[nb] < b1;
    }

    TileRegs<DB, FAST> rk, rv;
    int kt = o0 + 32 * wave;
    tile_load<DB, FAST>(rk, k, d, kt, o1, lane);
    tile_load<DB, FAST>(rv, v, d, kt, o1, lane);
    float dl[NB], lq[NB];
    if (FAST) {
        // delta partial of the lane: columns 16 q + 4 g .. + 3 of its two rows (unpredicated, clamped rows)
        float4 a[NB][DB], b[NB][DB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const size_t ro = (size_t)(qv[nb] ? rowq[nb] : b1 - 1) * (16 * DB);
#pragma unroll
            for (int qq = 0; qq < DB; ++qq) {
                a[nb][qq] = *(const float4*)&d_out[ro + 16 * qq + 4 * g];
                b[nb][qq] = *(const float4*)&out[ro + 16 * qq + 4 * g];
            }
            lq[nb] = lse[qv[nb] ? rowq[nb] : b1 - 1];
        }
        block_tile_stage_fast<DB>(q, DS, b0, b1, Qt, t);
        block_tile_stage_fast<DB>(d_out, DS, b0, b1, Gt, t);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float s = 0.f;
#pragma unroll
            for (int qq = 0; qq < DB; ++qq)
                s += a[nb][qq].x * b[nb][qq].x + a[nb][qq].y * b[nb][qq].y + a[nb][qq].z * b[nb][qq].z +
                     a[nb][qq].w * b[nb][qq].w;
            dl[nb] = qv[nb] ? s : 0.f;
            lq[nb] = qv[nb] ? lq[nb] : 0.f;
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kk = 4 * ks + g;
                if (qv[nb] && kk < d) {
                    const size_t o = (size_t)rowq[nb] * d + kk;
                    s += d_out[o] * out[o];
                }
            }
            dl[nb] = s;
            lq[nb] = qv[nb] ? lse[rowq[nb]] : 0.f;
        }
        zero_fill(Qt, C::TILE, t, EQD_BLOCK);
        zero_fill(Gt, C::TILE, t, EQD_BLOCK);
        zero_fill(Kt[wave], C::TILE, lane, 64);
        zero_fill(Vt[wave], C::TILE, lane, 64);
        __syncthreads();
        block_tile_stage(q, d, DS, b0, b1, Qt, t);
        block_tile_stage(d_out, d, DS, b0, b1, Gt, t);
    }
    __syncthreads();
    KFrag<BF, KS> qf[NB], dof[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        kfrag_load<BF, KS>(qf[nb], Qt, 16 * nb + l15, DS, g);
        kfrag_load<BF, KS>(dof[nb], Gt, 16 * nb + l15, DS, g);
        dl[nb] = group_sum(dl[nb]);
        if (wave == 0 && g == 0 && qv[nb]) delta[rowq[nb]] = dl[nb];
    }
    if (SM::ALIASED) __syncthreads();      // every wave has its fragments: the block tiles may be overwritten
    f32x4 dQ[DB][NB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dQ[db][nb] = f4zero();
    const float* __restrict__ Kw = Kt[wave];
    const float* __restrict__ Vw = Vt[wave];
    for (; kt < o1; kt += 32 * EQD_WAVES) {
        wave_lds_fence();
        tile_store<DB, FAST>(rk, Kt[wave], d, lane);
        tile_store<DB, FAST>(rv, Vt[wave], d, lane);
        wave_lds_fence();
        EQD_TR(30);
        tile_load<DB, FAST>(rk, k, d, kt + 32 * EQD_WAVES, o1, lane);
        tile_load<DB, FAST>(rv, v, d, kt + 32 * EQD_WAVES, o1, lane);
        EQD_TR(31);
        f32x4 S[2][NB], dP[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) S[mb][nb] = dP[mb][nb] = f4zero();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            mma_k<BF, KS, NB>(S[mb], Kw, 16 * mb + l15, DS, g, qf);
            mma_k<BF, KS, NB>(dP[mb], Vw, 16 * mb + l15, DS, g, dof);
        }
        EQD_TR(32);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float p = key < o1 ? bwd_exp(S[mb][nb][r] - lq[nb]) : 0.f;
                    S[mb][nb][r] = p * (dP[mb][nb][r] - dl[nb]);
                }
        EQD_TR(33);
        mma_r2<BF, DB, NB>(dQ, Kw, g, DS, l15, S);
        EQD_TR(34);
    }
    wave_lds_fence();       // (aliased layout) this wave's reads of its K tile are done before it is overwritten
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm.red(wave)[((db * 2 + nb) * 4 + r) * 64 + lane] = dQ[db][nb][r];
    __syncthreads();
    for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                const int f = 16 * db + 4 * g + r;
                if (f < d)
                    dq[(size_t)rowq[nb] * d + f] = (sm.red(0)[o] + sm.red(1)[o] + sm.red(2)[o] + sm.red(3)[o]) *
                                                   lrelu_grad(q[(size_t)rowq[nb] * d + f], qk_slope);
            }
        }
}

// backward pass 2: dk, dv for the block's keys (queries = the partner protein).
// OWN_DELTA (float4 path only): delta = rowsum(dO * O) of each streamed query tile is recomputed here from the
// O tile instead of being read from pass 1's output, so that both passes can run in ONE launch.
// WDS (the dS hand-off, large batches): the pass also WRITES its dS = P o (dP - delta) tiles - element (query, key) at
// ds[query * ds_stride + (key - first node of the key's protein)] - and the dq pass becomes ONE contraction over them
// (attn_bwd_qds_body) instead of recomputing S and dP: 5 executed GEMM units per (query tile, key tile) instead of 7 (4 as
// written, rigid_docking_model.py:46-64 backward).  seg_start[node] = first node of the node's protein.
template <int DB, bool FAST, bool OWN_DELTA, int NB, class SM, bool BF = false, bool WDS = false>
__device__ __forceinline__ void attn_bwd_kv_body(SM& sm, const EqdGraph& G, int item, int d,
                                                 const float* __restrict__ q, const float* __restrict__ k,
                                                 const float* __restrict__ v, const float* __restrict__ out,
                                                 const float* __restrict__ lse, const float* __restrict__ d_out,
                                                 const float* __restrict__ delta, float* __restrict__ dk,
                                                 float* __restrict__ dv, int half = 0, float qk_slope = 1.f,
                                                 float* __restrict__ ds = nullptr, int ds_stride = 0,
                                                 const int32_t* __restrict__ seg_start = nullptr) {
    static_assert(FAST || !OWN_DELTA, "OWN_DELTA needs the float4 tile layout");
    typedef AttnCfg<DB> C;
    constexpr int DS = C::DS, KS = C::KS;
    float* Kb = sm.blk(0);   // the block's own key / value rows
    float* Vb = sm.blk(1);
    float (*Qt)[C::TILE] = sm.str(0);   // streamed query / dO tiles
    float (*Gt)[C::TILE] = sm.str(1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;
    int rowk[NB];
    bool kvd[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowk[nb] = b0 + 16 * nb + l15;
        kvd[nb] = rowk[nb] < b1;
    }

    int kcol[NB];      // WDS: this lane's key columns inside a dS row
    if constexpr (WDS) {
        const int y0 = seg_start[b0];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) kcol[nb] = rowk[nb] - y0;
    }
    TileRegs<DB, FAST> rq, rg;
    TileRegs<DB, FAST && OWN_DELTA> ro;
    int qt = o0 + 32 * wave;
    tile_load<DB, FAST>(rq, q, d, qt, o1, lane);
    tile_load<DB, FAST>(rg, d_out, d, qt, o1, lane);
    if (OWN_DELTA) tile_load<DB, FAST && OWN_DELTA>(ro, out, d, qt, o1, lane);
    float lr[2][4], dr[2][4];   // lse / delta of the tile's query rows 16 mb + 4 g + r
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = qt + 16 * mb + 4 * g + r;
            const int qc = o1 > o0 ? (qr < o1 ? qr : o1 - 1) : 0;       // unpredicated, clamped
            const float lv = lse[qc], dv_ = OWN_DELTA ? 0.f : delta[qc];
            lr[mb][r] = qr < o1 ? lv : 0.f;
            dr[mb][r] = qr < o1 ? dv_ : 0.f;
        }
    if (FAST) {
        block_tile_stage_fast<DB>(k, DS, b0, b1, Kb, t);
        block_tile_stage_fast<DB>(v, DS, b0, b1, Vb, t);
    } else {
        zero_fill(Kb, C::TILE, t, EQD_BLOCK);
        zero_fill(Vb, C::TILE, t, EQD_BLOCK);
        zero_fill(Qt[wave], C::TILE, lane, 64);
        zero_fill(Gt[wave], C::TILE, lane, 64);
        __syncthreads();
        block_tile_stage(k, d, DS, b0, b1, Kb, t);
        block_tile_stage(v, d, DS, b0, b1, Vb, t);
    }
    __syncthreads();
    KFrag<BF, KS> kf[NB], vf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        kfrag_load<BF, KS>(kf[nb], Kb, 16 * nb + l15, DS, g);
        kfrag_load<BF, KS>(vf[nb], Vb, 16 * nb + l15, DS, g);
    }
    if (SM::ALIASED) __syncthreads();      // every wave has its fragments: the block tiles may be overwritten
    f32x4 dK[DB][NB], dV[DB][NB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dK[db][nb] = dV[db][nb] = f4zero();
    const float* __restrict__ Qw = Qt[wave];
    const float* __restrict__ Gw = Gt[wave];
    for (; qt < o1; qt += 32 * EQD_WAVES) {
        wave_lds_fence();
        tile_store<DB, FAST>(rq, Qt[wave], d, lane);
        tile_store<DB, FAST>(rg, Gt[wave], d, lane);
        if (OWN_DELTA) {
            // float4 j of a lane holds row (lane >> 4) + 4 j, columns 4 (lane & 15) .. + 3 of the tile
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 a = rg.q[FAST ? j : 0], b = ro.q[FAST && OWN_DELTA ? j : 0];
                const float p = l16_sum(a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w);
                if (l15 == 0) sm.dls[wave][g + 4 * j] = p;
            }
            if constexpr (DB > 4 && FAST && OWN_DELTA) {      // columns 64 .. 79: 4 lanes per row
                wave_lds_fence();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 a = rg.qx[j], b = ro.qx[j];
                    float p = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
                    p += lane_xor<1>(p);
                    p += lane_xor<2>(p);
                    if ((lane & 3) == 0) sm.dls[wave][(lane + 64 * j) >> 2] += p;
                }
            }
        }
        float lc[2][4], dc[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lc[mb][r] = lr[mb][r];
                dc[mb][r] = dr[mb][r];
            }
        wave_lds_fence();
        if (OWN_DELTA) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dc[mb][r] = sm.dls[wave][16 * mb + 4 * g + r];
        }
        const int qn = qt + 32 * EQD_WAVES;
        tile_load<DB, FAST>(rq, q, d, qn, o1, lane);
        tile_load<DB, FAST>(rg, d_out, d, qn, o1, lane);
        if (OWN_DELTA) tile_load<DB, FAST && OWN_DELTA>(ro, out, d, qn, o1, lane);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qn + 16 * mb + 4 * g + r;
                const int qc = o1 > o0 ? (qr < o1 ? qr : o1 - 1) : 0;
                const float lv = lse[qc], dv_ = OWN_DELTA ? 0.f : delta[qc];
                lr[mb][r] = qr < o1 ? lv : 0.f;
                dr[mb][r] = qr < o1 ? dv_ : 0.f;
            }
        f32x4 S[2][NB], dP[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) S[mb][nb] = dP[mb][nb] = f4zero();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            mma_k<BF, KS, NB>(S[mb], Qw, 16 * mb + l15, DS, g, kf);
            mma_k<BF, KS, NB>(dP[mb], Gw, 16 * mb + l15, DS, g, vf);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = qt + 16 * mb + 4 * g + r < o1;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float p = ok ? bwd_exp(S[mb][nb][r] - lc[mb][r]) : 0.f;
                    S[mb][nb][r] = p;
                    dP[mb][nb][r] = p * (dP[mb][nb][r] - dc[mb][r]);
                }
            }
        if constexpr (WDS) {      // rows = the tile's queries, 16 consecutive keys per lane group: 64-byte segments
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qr = qt + 16 * mb + 4 * g + r;
                    if (qr < o1) {
                        float* __restrict__ row = ds + (size_t)qr * ds_stride;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            if (kvd[nb]) row[kcol[nb]] = dP[mb][nb][r];
                    }
                }
        }
        mma_r2<BF, DB, NB>(dV, Gw, g, DS, l15, S);
        mma_r2<BF, DB, NB>(dK, Qw, g, DS, l15, dP);
    }
    wave_lds_fence();       // (aliased layout) this wave's reads of its query tile are done before it is overwritten
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {     // dK then dV through the same LDS buffer
        if (pass) __syncthreads();
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sm.red(wave)[((db * 2 + nb) * 4 + r) * 64 + lane] = pass ? dV[db][nb][r] : dK[db][nb][r];
        __syncthreads();
        float* __restrict__ dst = pass ? dv : dk;
        for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (!kvd[nb]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                    const int f = 16 * db + 4 * g + r;
                    if (f < d) {      // dK carries LeakyReLU'(k) (see attn_bwd_q_body), dV does not (att_mlp_V is linear)
                        const float s = sm.red(0)[o] + sm.red(1)[o] + sm.red(2)[o] + sm.red(3)[o];
                        dst[(size_t)rowk[nb] * d + f] = pass ? s : s * lrelu_grad(k[(size_t)rowk[nb] * d + f], qk_slope);
                    }
                }
            }
    }
}

// backward pass 1 over the dS tiles the key / value pass left (d = 64, 16-byte aligned rows): dq[query] = sum over the
// partner's keys of dS[query][key] K[key], times LeakyReLU'(q).  The workgroup's 4 waves split the partner's 32-key tiles as
// in attn_bwd_q_body and merge in the same order; a lane's B operand - 4 consecutive keys of its query row - is ONE
// 16-byte load from the dS row (rows are ds_stride floats, a multiple of 32, so a tile never leaves its row; keys beyond the
// partner are whatever the buffer holds and are selected away).
template <int DB>
struct AttnQdsSmem {
    typedef AttnCfg<DB> C;
    float str_[EQD_WAVES][C::TILE];
    __device__ __forceinline__ float* red(int w) { return str_[w]; }      // merge buffer = each wave's own tile, as in the alias layout
    static_assert(C::RED <= C::TILE, "merge buffer must fit a streamed tile");
};
template <int DB, int NB, bool BF = false>
__device__ __forceinline__ void attn_bwd_qds_body(AttnQdsSmem<DB>& sm, const EqdGraph& G, int item, const float* __restrict__ q,
                                                  const float* __restrict__ k, const float* __restrict__ ds, int ds_stride,
                                                  float* __restrict__ dq, int half, float qk_slope) {
    typedef AttnCfg<DB> C;
    constexpr int DS = C::DS, d = 16 * DB;      // (80 = the 69-wide first layer, zero-padded by the caller)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;
    int rowq[NB];
    bool qv[NB];
    const float* __restrict__ dsr[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowq[nb] = b0 + 16 * nb + l15;
        qv[nb] = rowq[nb] < b1;
        dsr[nb] = ds + (size_t)(qv[nb] ? rowq[nb] : b1 - 1) * ds_stride;      // + (key - o0)
    }
    TileRegs<DB, true> rk;
    int kt = o0 + 32 * wave;
    tile_load<DB, true>(rk, k, d, kt, o1, lane);
    f32x4 sv[2][NB];
    auto ds_load = [&](int kt_) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = kt_ < o1 ? kt_ - o0 + 16 * mb + 4 * g : 0;      // (a tile beyond the partner: any valid address)
                sv[mb][nb] = *(const f32x4*)(dsr[nb] + col);
            }
    };
    ds_load(kt);
    f32x4 dQ[DB][NB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dQ[db][nb] = f4zero();
    float* __restrict__ Kw = sm.str_[wave];
    for (; kt < o1; kt += 32 * EQD_WAVES) {
        wave_lds_fence();
        tile_store<DB, true>(rk, Kw, d, lane);
        wave_lds_fence();
        f32x4 S[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[mb][nb][r] = kt + 16 * mb + 4 * g + r < o1 ? sv[mb][nb][r] : 0.f;
        tile_load<DB, true>(rk, k, d, kt + 32 * EQD_WAVES, o1, lane);
        ds_load(kt + 32 * EQD_WAVES);
        mma_r2<BF, DB, NB>(dQ, Kw, g, DS, l15, S);      // (BF: K tile and dS rounded to bf16 as the operands are formed)
    }
    wave_lds_fence();
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm.red(wave)[((db * 2 + nb) * 4 + r) * 64 + lane] = dQ[db][nb][r];
    __syncthreads();
    for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                const int f = 16 * db + 4 * g + r;
                dq[(size_t)rowq[nb] * d + f] = (sm.red(0)[o] + sm.red(1)[o] + sm.red(2)[o] + sm.red(3)[o]) *
                                               lrelu_grad(q[(size_t)rowq[nb] * d + f], qk_slope);
            }
        }
}
// (half blocks like the kernels around it: workgroup b -> item 8 (b / 16) + b % 8, half (b / 8) % 2, so a direction's blocks
// and the dS rows they read stay on the XCD whose L2 the key / value pass left them in)
// (launch bound 4 for the 64-wide fp32 instance: 130 registers unasked, 127 when asked - four workgroups per CU instead of three)
template <int DB, int NB, bool BF = false>
__global__ __launch_bounds__(EQD_BLOCK, (DB == 4 && !BF) ? 4 : 2) void k_attn_bwd_qds(EqdGraph G, const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ ds, int ds_stride,
                                                               float* __restrict__ dq, float qk_slope) {
    __shared__ __attribute__((aligned(16))) AttnQdsSmem<DB> sm;
    const int item = NB == 1 ? att_half_item((int)blockIdx.x) : (int)blockIdx.x;
    attn_bwd_qds_body<DB, NB, BF>(sm, G, item, q, k, ds, ds_stride, dq, NB == 1 ? att_half_of((int)blockIdx.x) : 0, qk_slope);
}
// the 80-wide (zero-padded 69) first layer: 32-row blocks, one workgroup per item, no gather riding (its gather is launched
// on its own, as with the recompute form of that layer)
template <bool BF, bool BF_DZ>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_bwd_kvds80(EqdGraph G, const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ out,
                                                               const float* __restrict__ lse, const float* __restrict__ d_out,
                                                               float* __restrict__ dk, float* __restrict__ dv, float qk_slope,
                                                               float* __restrict__ ds, int ds_stride,
                                                               const int32_t* __restrict__ seg_start, int n_attn, int nred,
                                                               EqdGatherArgs GA, EqdRedArg RA) {
    if ((int)blockIdx.x >= n_attn) {      // the layer's node gather + pending reductions (k_attn_bwd_gather's arrangement)
        const int b = (int)blockIdx.x - n_attn;
        if (b < GA.ngather) {
            node_gather_body<BF_DZ>(GA, b);
        } else if (b < GA.ngather + nred) {
            __shared__ __attribute__((aligned(16))) float red[16][68];
            __shared__ float red2[4][64];
            reduce_block<16>(RA, b - GA.ngather, red, red2);
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) AttnBwdSmem<5, true> sm;
    attn_bwd_kv_body<5, true, true, 2, AttnBwdSmem<5, true>, BF, true>(sm, G, (int)blockIdx.x, 80, q, k, v, out, lse, d_out, nullptr,
                                                                          dk, dv, 0, qk_slope, ds, ds_stride, seg_start);
}
// the key / value pass with the dS hand-off (half blocks), optionally with a layer's node gather + pending reductions as
// trailing workgroups (k_attn_bwd_gather's arrangement)
template <bool BF_DZ>
__global__ __launch_bounds__(EQD_BLOCK, 2) void k_attn_bwd_kvds(EqdGraph G, const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ out,
                                                                const float* __restrict__ lse, const float* __restrict__ d_out,
                                                                float* __restrict__ dk, float* __restrict__ dv, float qk_slope,
                                                                float* __restrict__ ds, int ds_stride,
                                                                const int32_t* __restrict__ seg_start, int n_attn, int nred,
                                                                EqdGatherArgs GA, EqdRedArg RA) {
    static_assert(EQD_BLOCK == 256, "the gather body is written for 256-thread workgroups");
    if ((int)blockIdx.x >= n_attn) {
        const int b = (int)blockIdx.x - n_attn;
        if (b < GA.ngather) {
            node_gather_body<BF_DZ>(GA, b);
        } else if (b < GA.ngather + nred) {
            __shared__ __attribute__((aligned(16))) float red[16][68];
            __shared__ float red2[4][64];
            reduce_block<16>(RA, b - GA.ngather, red, red2);
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) AttnBwdSmem<4, true> sm;
    const int b = (int)blockIdx.x;
    attn_bwd_kv_body<4, true, true, 1, AttnBwdSmem<4, true>, false, true>(sm, G, att_half_item(b), 64, q, k, v, out, lse, d_out,
                                                                          nullptr, dk, dv, att_half_of(b), qk_slope, ds, ds_stride,
                                                                          seg_start);
}
// bf16 mode (bf16 tiles in LDS), d = 64: the same two launches
template <bool BF_DZ, bool QB = false>
__global__ __launch_bounds__(EQD_BLOCK, 2) void k_attn_bwd_kvds_lb(EqdGraph G, const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ out,
                                                                   const float* __restrict__ lse, const float* __restrict__ d_out,
                                                                   float* __restrict__ dk, float* __restrict__ dv, float qk_slope,
                                                                   float* __restrict__ ds, int ds_stride,
                                                                   const int32_t* __restrict__ seg_start, int n_attn, int nred,
                                                                   EqdGatherArgs GA, EqdRedArg RA) {
    if ((int)blockIdx.x >= n_attn) {
        const int b = (int)blockIdx.x - n_attn;
        if (b < GA.ngather) {
            node_gather_body<BF_DZ>(GA, b);
        } else if (b < GA.ngather + nred) {
            __shared__ __attribute__((aligned(16))) float red[16][68];
            __shared__ float red2[4][64];
            reduce_block<16>(RA, b - GA.ngather, red, red2);
        }
        return;
    }
    __shared__ AttnBwdSmemLb sm;
    const int b = (int)blockIdx.x;
    // (the dS workspace of this pair of kernels holds bf16 - same bits as the dq pass would form, half the round trip)
    attn_bwd_kv_body_lb<1, true, QB, true>(sm, G, att_half_item(b), q, k, v, out, lse, d_out, dk, dv, att_half_of(b), qk_slope, ds,
                                           ds_stride, seg_start);
}
template <int NB, bool QB = false>
__global__ __launch_bounds__(EQD_BLOCK, 2) void k_attn_bwd_qds_lb(EqdGraph G, const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ ds, int ds_stride,
                                                                  float* __restrict__ dq, float qk_slope) {
    __shared__ AttnQdsSmemLb sm;
    const int item = NB == 1 ? att_half_item((int)blockIdx.x) : (int)blockIdx.x;
    attn_bwd_qds_body_lb<NB, QB, true>(sm, G, item, q, k, ds, ds_stride, dq, NB == 1 ? att_half_of((int)blockIdx.x) : 0, qk_slope);
}
// seg_start[node] = first node of the node's protein (EqdGraph.seg_off: ligand segments, then receptor segments)
__global__ void k_seg_start(const int32_t* __restrict__ seg_off, int nseg, int n, int32_t* __restrict__ out) {
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
        int lo = 0, hi = nseg;      // last segment with seg_off[s] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_off[mid] <= i) lo = mid; else hi = mid;
        }
        out[i] = seg_off[lo];
    }
}

template <int DB, bool FAST, int NB>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_bwd_q(EqdGraph G, int d, const float* __restrict__ q,
                                                          const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ out,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ d_out, float* __restrict__ dq,
                                                          float* __restrict__ delta, float qk_slope) {
    __shared__ __attribute__((aligned(16))) AttnBwdSmem<DB, false> sm;
    attn_bwd_q_body<DB, FAST, NB>(sm, G, blockIdx.x, d, q, k, v, out, lse, d_out, dq, delta, 0, qk_slope);
}
template <int DB, bool FAST, int NB>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_bwd_kv(EqdGraph G, int d, const float* __restrict__ q,
                                                           const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ d_out,
                                                           const float* __restrict__ delta, float* __restrict__ dk,
                                                           float* __restrict__ dv, float qk_slope) {
    __shared__ __attribute__((aligned(16))) AttnBwdSmem<DB, false> sm;
    attn_bwd_kv_body<DB, FAST, false, NB>(sm, G, blockIdx.x, d, q, k, v, nullptr, lse, d_out, delta, dk, dv, 0, qk_slope);
}
// both passes in one launch (float4 path): workgroups [0, n_items) run pass 1, [n_items, 2 n_items) pass 2
// NB = 1: two workgroups per item and pass (16-row half blocks); with half the accumulators the kernel fits 256
// registers, i.e. with its 70 KB of LDS two workgroups share a CU (NB = 2 needs 371 registers: one wave per SIMD)
template <int DB, int NB, bool BF = false>
__global__ __launch_bounds__(EQD_BLOCK, NB == 1 ? 2 : 1) void k_attn_bwd(EqdGraph G, int d, const float* __restrict__ q,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        const float* __restrict__ out, const float* __restrict__ lse,
                                                        const float* __restrict__ d_out, float* __restrict__ dq,
                                                        float* __restrict__ dk, float* __restrict__ dv,
                                                        float* __restrict__ delta, float qk_slope) {
    __shared__ __attribute__((aligned(16))) AttnBwdSmem<DB, true> sm;
    const int per = NB == 1 ? 2 * G.n_att_items : G.n_att_items;      // workgroups per pass
    const bool kv = (int)blockIdx.x >= per;
    const int idx = kv ? (int)blockIdx.x - per : (int)blockIdx.x;
    const int item = NB == 1 ? att_half_item(idx) : idx, half = NB == 1 ? att_half_of(idx) : 0;
    if (!kv)
        attn_bwd_q_body<DB, true, NB, AttnBwdSmem<DB, true>, BF>(sm, G, item, d, q, k, v, out, lse, d_out, dq, delta, half,
                                                                 qk_slope);
    else
        attn_bwd_kv_body<DB, true, true, NB, AttnBwdSmem<DB, true>, BF>(sm, G, item, d, q, k, v, out, lse, d_out, nullptr, dk,
                                                                        dv, half, qk_slope);
}

// k_attn_bwd<4, 1> with the layer's node gather (and the partial reductions that ride with it) as the LAST workgroups of the
// same launch: the two are independent - the gather reads what the edge backward wrote, the attention backward what the row
// chain wrote - and at DB5.5 sizes neither fills the chip: the 448 attention workgroups take 448 of the 512 slots (two per
// CU: the gather's 5 KB of LDS ride on the attention's 70 KB) for ~20 us, the ~240 short gather / reduction workgroups flow
// through the other 64 meanwhile (as the FIRST workgroups they ran before the attention instead of beside it: 27.4 us per
// launch against 20.4 + 9.6 apart).
// LB: the bf16 attention with bf16 tiles in LDS (k_attn_bwd_lb<1>: 74 KB + the gather's 5 KB, still two per CU)
template <bool BF_DZ, bool LB>
__global__ __launch_bounds__(EQD_BLOCK, 2) void k_attn_bwd_gather(EqdGraph G, int d, const float* __restrict__ q,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        const float* __restrict__ out, const float* __restrict__ lse,
                                                        const float* __restrict__ d_out, float* __restrict__ dq,
                                                        float* __restrict__ dk, float* __restrict__ dv,
                                                        float* __restrict__ delta, float qk_slope, int n_attn, int nred,
                                                        EqdGatherArgs GA, EqdRedArg RA) {
    static_assert(EQD_BLOCK == 256, "the gather body is written for 256-thread workgroups");
    if ((int)blockIdx.x >= n_attn) {
        const int b = (int)blockIdx.x - n_attn;
        if (b < GA.ngather) {
            node_gather_body<BF_DZ>(GA, b);
        } else if (b < GA.ngather + nred) {
            __shared__ __attribute__((aligned(16))) float red[16][68];
            __shared__ float red2[4][64];
            reduce_block<16>(RA, b - GA.ngather, red, red2);
        }
        return;
    }
    const int b = (int)blockIdx.x;
    const int per = 2 * G.n_att_items;      // workgroups per pass
    const bool kv = b >= per;
    const int idx = kv ? b - per : b;
    const int item = att_half_item(idx), half = att_half_of(idx);
    if constexpr (LB) {
        __shared__ AttnBwdSmemLb sm;
        if (!kv)
            attn_bwd_q_body_lb<1>(sm, G, item, q, k, v, out, lse, d_out, dq, delta, half, qk_slope);
        else
            attn_bwd_kv_body_lb<1>(sm, G, item, q, k, v, out, lse, d_out, dk, dv, half, qk_slope);
    } else {
        __shared__ __attribute__((aligned(16))) AttnBwdSmem<4, true> sm;
        if (!kv)
            attn_bwd_q_body<4, true, 1, AttnBwdSmem<4, true>, false>(sm, G, item, d, q, k, v, out, lse, d_out, dq, delta, half,
                                                                     qk_slope);
        else
            attn_bwd_kv_body<4, true, true, 1, AttnBwdSmem<4, true>, false>(sm, G, item, d, q, k, v, out, lse, d_out, nullptr,
                                                                            dk, dv, half, qk_slope);
    }
}

// the 80-wide (zero-padded 69) first layer's merged backward (k_attn_bwd<5, 2>: one workgroup per CU, 224 of them for a
// DB5.5-sized batch) with that layer's node gather and pending reductions as trailing workgroups: they run on the CUs the
// attention leaves free instead of as a launch of their own behind it
template <bool BF, bool BF_DZ>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_bwd80_gather(EqdGraph G, const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ out,
                                                                 const float* __restrict__ lse, const float* __restrict__ d_out,
                                                                 float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                                 float* __restrict__ delta, float qk_slope, int n_attn, int nred,
                                                                 EqdGatherArgs GA, EqdRedArg RA) {
    if ((int)blockIdx.x >= n_attn) {
        const int b = (int)blockIdx.x - n_attn;
        if (b < GA.ngather) {
            node_gather_body<BF_DZ>(GA, b);
        } else if (b < GA.ngather + nred) {
            __shared__ __attribute__((aligned(16))) float red[16][68];
            __shared__ float red2[4][64];
            reduce_block<16>(RA, b - GA.ngather, red, red2);
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) AttnBwdSmem<5, true> sm;
    const int per = G.n_att_items;
    const bool kv = (int)blockIdx.x >= per;
    const int item = kv ? (int)blockIdx.x - per : (int)blockIdx.x;
    if (!kv)
        attn_bwd_q_body<5, true, 2, AttnBwdSmem<5, true>, BF>(sm, G, item, 80, q, k, v, out, lse, d_out, dq, delta, 0, qk_slope);
    else
        attn_bwd_kv_body<5, true, true, 2, AttnBwdSmem<5, true>, BF>(sm, G, item, 80, q, k, v, out, lse, d_out, nullptr, dk, dv, 0,
                                                                     qk_slope);
}

// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// The tile loaders address their tensors through buffer descriptors with 32-bit byte counts and lane offsets (tile_load:
// 64 * DB <= 320 B per row; lb_load: 256 B per row): a tensor must stay below 2^31 bytes, i.e. 6.7 million rows.  Beyond
// that the offsets would wrap and rows come back as zeros without an error - refused here instead (the 288 GB of one
// MI355X hold ~400 million residues of saved state at 8 layers, so this is a guard, not a limit anyone meets).
#define EQD_ATT_MAX_ROWS 6000000
static int attn_rows_ok(const EqdGraph* g, const char* who) {
    if (g->n_nodes > EQD_ATT_MAX_ROWS) {
        eqd_set_error("%s: %d nodes in one batch; the attention kernels' 32-bit tile offsets allow %d", who, g->n_nodes,
                      EQD_ATT_MAX_ROWS);
        return EQD_ERR_UNSUPPORTED;
    }
    return EQD_OK;
}

template <int DB, bool FAST, int NB, bool BF = false>
static int attn_launch_fwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v, float* out,
                           float* lse, hipStream_t st) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd<DB, FAST, NB, BF>), dim3(NB == 1 ? 2 * g->n_att_items : g->n_att_items),
                       dim3(EQD_BLOCK), 0, st, *g, d, q, k, v, out, lse);
    return eqd_check_launch("k_attn_fwd");
}
template <int DB, int NB>
static int attn_launch_bwd_bf(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                              const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                              hipStream_t st, float qk_slope) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd<DB, NB, true>), dim3((NB == 1 ? 4 : 2) * g->n_att_items), dim3(EQD_BLOCK), 0,
                       st, *g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, qk_slope);
    return eqd_check_launch("k_attn_bwd");
}
template <int DB, bool FAST, int NB>
static int attn_launch_bwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                           const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                           hipStream_t st, float qk_slope) {
    if constexpr (FAST) if (aligned16(out)) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd<DB, NB>), dim3((NB == 1 ? 4 : 2) * g->n_att_items), dim3(EQD_BLOCK), 0,
                           st, *g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, qk_slope);
        return eqd_check_launch("k_attn_bwd");
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_q<DB, FAST, NB>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, d,
                       q, k, v, out, lse, d_out, dq, delta, qk_slope);
    int rc = eqd_check_launch("k_attn_bwd_q");
    if (rc) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kv<DB, FAST, NB>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, d,
                       q, k, v, lse, d_out, delta, dk, dv, qk_slope);
    return eqd_check_launch("k_attn_bwd_kv");
}
// forward, float4 path: half blocks while that still fits one round of workgroups (two of them share a CU's LDS);
// EQD_ATT_SPLIT=0|1 forces either (tests)
static bool att_half_blocks(const EqdGraph* g) {
    if (g->n_att_items % 8) return false;   // the half-block workgroup map needs the 8-way interleaved list (header)
    const char* f = eqd_tunable("EQD_ATT_SPLIT");
    if (f && (f[0] == '0' || f[0] == '1') && f[1] == 0) return f[0] == '1';
    return g->n_att_items <= eqd_num_cus();
}
extern "C" int eqd_cross_attention_fwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                       float* out, float* lse, void* stream) {
    if (!g || !q || !k || !v || !out || !lse) {
        eqd_set_error("eqd_cross_attention_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (int rc = attn_rows_ok(g, "eqd_cross_attention_fwd")) return rc;
    if (d <= 0 || d > 80) {
        eqd_set_error("eqd_cross_attention_fwd: feature width %d outside 1..80", d);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool al = aligned16(q) && aligned16(k) && aligned16(v);
    const bool half = att_half_blocks(g);
    if (d == 64 && al)
        return half ? attn_launch_fwd<4, true, 1>(g, d, q, k, v, out, lse, st) : attn_launch_fwd<4, true, 2>(g, d, q, k, v, out, lse, st);
    if (d == 80 && al)
        return half ? attn_launch_fwd<5, true, 1>(g, d, q, k, v, out, lse, st) : attn_launch_fwd<5, true, 2>(g, d, q, k, v, out, lse, st);
    if (d <= 64) return attn_launch_fwd<4, false, 2>(g, d, q, k, v, out, lse, st);
    return attn_launch_fwd<5, false, 2>(g, d, q, k, v, out, lse, st);
}

// The backward with the LeakyReLU derivative of att_mlp_Q / att_mlp_K folded in (qk_slope = the activation's negative slope;
// 1 = the plain operator): dq, dk are then gradients w.r.t. the PRE-activations - what eqd_model_backward's dh job and
// weight-gradient GEMMs consume.
static int attention_bwd_f32(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                             const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                             float qk_slope, hipStream_t stream) {
    if (!g || !q || !k || !v || !out || !lse || !d_out || !dq || !dk || !dv || !delta) {
        eqd_set_error("eqd_cross_attention_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (int rc = attn_rows_ok(g, "eqd_cross_attention_bwd")) return rc;
    if (d <= 0 || d > 80) {
        eqd_set_error("eqd_cross_attention_bwd: feature width %d outside 1..80", d);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    hipStream_t st = stream;
    const bool al = aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_out) && aligned16(out);
    // d = 64: half blocks (two workgroups per CU, see k_attn_bwd) - config C +2.3 %, E +0.9 %, B unchanged;
    // EQD_ATT_BWD_SPLIT=0 keeps 32-row blocks (tests)
    const char* hb = eqd_tunable("EQD_ATT_BWD_SPLIT");
    const bool half = !(hb && hb[0] == '0' && hb[1] == 0) && g->n_att_items % 8 == 0;
    if (d == 64 && al && half) return attn_launch_bwd<4, true, 1>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
    if (d == 64 && al) return attn_launch_bwd<4, true, 2>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
    if (d == 80 && al) return attn_launch_bwd<5, true, 2>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
    if (d <= 64) return attn_launch_bwd<4, false, 2>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
    return attn_launch_bwd<5, false, 2>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
}

// bf16 mode, d = 64: the streamed tiles live in LDS as bf16 (eqd_attn_lb_inl.h) unless EQD_ATT_LB=0 (the first bf16 version:
// fp32 tiles in LDS, rounded per MFMA operand; kept for the 80-wide first layer, tests and A/B measurements)
static bool att_lds_bf16() {
    const char* f = eqd_tunable("EQD_ATT_LB");
    return !(f && f[0] == '0' && f[1] == 0);
}

// ---- bf16 mode: the four contractions of the forward (Q K^T, P V) and the ten of the backward run on
// v_mfma_f32_16x16x16_bf16 - inputs rounded to bf16 when the MFMA operands are formed, fp32 accumulate; logits, softmax
// statistics, exponentials, delta and all outputs stay fp32.  Float4 tile path only (d = 64, or 80 = the zero-padded
// first layer, 16-byte aligned operands), which is what the model always presents; anything else is an error here.
extern "C" int eqd_cross_attention_fwd_bf16(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                            float* out, float* lse, void* stream) {
    return eqd_attention_fwd_bf16_impl(g, d, q, k, v, out, lse, (hipStream_t)stream, false);
}
// qkv_bf16: q, k, v are bf16 rows ([n_nodes][64]; the saved form of the bf16 storage mode) - only the LDS-bf16 kernels of the
// 64-wide layers read them
int eqd_attention_fwd_bf16_impl(const EqdGraph* g, int d, const float* q, const float* k, const float* v, float* out,
                                float* lse, hipStream_t stream, bool qkv_bf16) {
    if (!g || !q || !k || !v || !out || !lse) {
        eqd_set_error("eqd_cross_attention_fwd_bf16: NULL argument");
        return EQD_ERR_NULL;
    }
    if (int rc = attn_rows_ok(g, "eqd_cross_attention_fwd_bf16")) return rc;
    if ((d != 64 && d != 80) || !(aligned16(q) && aligned16(k) && aligned16(v))) {
        eqd_set_error("eqd_cross_attention_fwd_bf16: needs d = 64 or 80 and 16-byte aligned operands (d = %d)", d);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    hipStream_t st = stream;
    const bool half = att_half_blocks(g);
    if (qkv_bf16 && !(d == 64 && att_lds_bf16())) {
        eqd_set_error("attention forward: bf16 q / k / v are only read by the LDS-bf16 kernels of the 64-wide layers");
        return EQD_ERR_UNSUPPORTED;
    }
    if (d == 64 && att_lds_bf16()) {      // streamed tiles held in LDS as bf16 (eqd_attn_lb_inl.h)
        if (qkv_bf16) {
            if (half)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd_lb<1, true>), dim3(2 * g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd_lb<2, true>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse);
        } else if (half)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd_lb<1>), dim3(2 * g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_fwd_lb<2>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse);
        return eqd_check_launch("k_attn_fwd");
    }
    if (d == 64)
        return half ? attn_launch_fwd<4, true, 1, true>(g, d, q, k, v, out, lse, st)
                    : attn_launch_fwd<4, true, 2, true>(g, d, q, k, v, out, lse, st);
    return half ? attn_launch_fwd<5, true, 1, true>(g, d, q, k, v, out, lse, st)
                : attn_launch_fwd<5, true, 2, true>(g, d, q, k, v, out, lse, st);
}
static int attention_bwd_bf16(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                              const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                              float qk_slope, hipStream_t stream) {
    if (!g || !q || !k || !v || !out || !lse || !d_out || !dq || !dk || !dv || !delta) {
        eqd_set_error("eqd_cross_attention_bwd_bf16: NULL argument");
        return EQD_ERR_NULL;
    }
    if (int rc = attn_rows_ok(g, "eqd_cross_attention_bwd_bf16")) return rc;
    if ((d != 64 && d != 80) || !(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_out) && aligned16(out))) {
        eqd_set_error("eqd_cross_attention_bwd_bf16: needs d = 64 or 80 and 16-byte aligned operands (d = %d)", d);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d == 64 && att_lds_bf16()) {      // streamed tiles held in LDS as bf16 (eqd_attn_lb_inl.h)
        const char* nb2 = eqd_tunable("EQD_ATT_LB_NB");      // experiments: 2 = 32-row blocks in the backward
        if (g->n_att_items % 8 == 0 && !(nb2 && nb2[0] == '2'))
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_lb<1>), dim3(4 * g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k, v, out,
                               lse, d_out, dq, dk, dv, delta, qk_slope);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_lb<2>), dim3(2 * g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k, v, out,
                               lse, d_out, dq, dk, dv, delta, qk_slope);
        return eqd_check_launch("k_attn_bwd");
    }
    if (d == 64 && g->n_att_items % 8 == 0)
        return attn_launch_bwd_bf<4, 1>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
    if (d == 64) return attn_launch_bwd_bf<4, 2>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
    return attn_launch_bwd_bf<5, 2>(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, st, qk_slope);
}

extern "C" int eqd_cross_attention_bwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                       const float* out, const float* lse, const float* d_out, float* dq, float* dk,
                                       float* dv, float* delta, void* stream) {
    return attention_bwd_f32(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, 1.f, (hipStream_t)stream);
}
extern "C" int eqd_cross_attention_bwd_bf16(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                            const float* out, const float* lse, const float* d_out, float* dq, float* dk,
                                            float* dv, float* delta, void* stream) {
    return attention_bwd_bf16(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, 1.f, (hipStream_t)stream);
}
int eqd_launch_attention_bwd_act(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                 const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                                 float qk_slope, bool bf16, hipStream_t st) {
    return bf16 ? attention_bwd_bf16(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, qk_slope, st)
                : attention_bwd_f32(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, qk_slope, st);
}
// 1 if eqd_launch_attention_bwd_gather will take the fused launch (the attention of a 64-wide layer on the half-block path,
// fp32 or with bf16 tiles in LDS - what every layer but the first runs at every size); otherwise it issues the two launches
// one after the other
int eqd_attention_bwd_gather_fused(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                   const float* d_out, bool bf16) {
    const char* f = eqd_tunable("EQD_FUSE_GATHER");
    if (f && f[0] == '0' && f[1] == 0) return 0;
    const char* hb = eqd_tunable("EQD_ATT_BWD_SPLIT");
    if (hb && hb[0] == '0' && hb[1] == 0) return 0;
    if (d == 80 && g->n_att_items > 0)      // the first layer's merged backward (one workgroup per item and pass)
        return aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_out) && aligned16(out);
    if (d != 64 || g->n_att_items <= 0 || g->n_att_items % 8 != 0) return 0;
    if (bf16) {
        const char* nb2 = eqd_tunable("EQD_ATT_LB_NB");
        if (!att_lds_bf16() || (nb2 && nb2[0] == '2')) return 0;
    }
    return aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_out) && aligned16(out);
}
// ---- the dS hand-off form of the backward (fp32, d = 64, half-block work list) ------------------------------------------
// Workspace: dS [n_nodes][stride] floats (stride = the longest protein rounded up to whole 32-key tiles: element (query, key)
// at query * stride + key's index inside its protein) followed by seg_start [n_nodes] int32.
static bool att_lds_bf16();
int eqd_attention_ds_stride(const EqdGraph* g) { return (g->max_seg + 31) / 32 * 32; }
size_t eqd_attention_ds_bytes(const EqdGraph* g) {
    return eqd_align_up((size_t)g->n_nodes * eqd_attention_ds_stride(g) * sizeof(float)) +
           eqd_align_up((size_t)g->n_nodes * sizeof(int32_t));
}
// Taken when the batch gives every CU more than one attention item (64 x (300, 300): 1 280; 4 x (2000, 2000): 504): there
// the backward is arithmetic-bound and 5 / 7 of the MFMA work wins; a DB5.5-sized batch (112 items) is one latency-bound
// round of workgroups either way and keeps the single launch.  EQD_ATT_DS=0|1 forces either (tests, A/B runs).
// max_seg sets the dS row stride, and the kernels index a row by (key - first node of the key's protein) without a bound
// check: a max_seg below the longest protein would write across rows / past the arena.  The host cannot see seg_off (device
// memory), but it can reject values that are impossible for ANY segmentation of the node counts: the longest of n_pairs
// proteins is at least their average and at most their sum.  An implausible max_seg keeps the recompute form (no workspace).
static bool att_max_seg_plausible(const EqdGraph* g) {
    const long long big = g->n_lig > g->n_rec ? g->n_lig : g->n_rec;
    return g->n_pairs > 0 && g->max_seg > 0 && (long long)g->max_seg * g->n_pairs >= big && g->max_seg <= big;
}
int eqd_attention_ds_wanted(const EqdGraph* g, int d, bool bf16) {
    if ((d != 64 && d != 80) || g->n_att_items <= 0 || g->n_att_items % 8 != 0 || !att_max_seg_plausible(g)) return 0;
    if (bf16 && d == 64) {      // bf16 mode: the LDS-bf16 kernels of the 64-wide layers (the 80-wide first layer: fp32 tiles, bf16 operands)
        const char* nb2 = eqd_tunable("EQD_ATT_LB_NB");
        if (!att_lds_bf16() || (nb2 && nb2[0] == '2')) return 0;
    }
    const char* hb = eqd_tunable("EQD_ATT_BWD_SPLIT");
    if (hb && hb[0] == '0' && hb[1] == 0) return 0;
    const char* f = eqd_tunable("EQD_ATT_DS");
    if (f && (f[0] == '0' || f[0] == '1') && f[1] == 0) return f[0] == '1';
    return g->n_att_items > eqd_num_cus();
}
int eqd_launch_seg_start(const EqdGraph* g, int32_t* seg_start, hipStream_t st) {
    int blocks = (g->n_nodes + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(k_seg_start, dim3(blocks), dim3(256), 0, st, g->seg_off, 2 * g->n_pairs, g->n_nodes, seg_start);
    return eqd_check_launch("k_seg_start");
}
// key / value pass (+ the gather `gc` and the pending reductions as trailing workgroups when gc != NULL), then the dq pass
static int attention_bwd_ds(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out, const float* lse,
                            const float* d_out, float* dq, float* dk, float* dv, float qk_slope, float* ds,
                            const int32_t* seg_start, const EqdGatherCall* gc, EqdRedList* pending, hipStream_t st,
                            bool bf16 = false, bool qkv_bf16 = false) {
    if (qkv_bf16 && !(bf16 && d == 64)) {
        eqd_set_error("attention backward: saved bf16 q / k / v are only read by the bf16 kernels of the 64-wide layers");
        return EQD_ERR_UNSUPPORTED;
    }
    if (int rc = attn_rows_ok(g, "attention backward")) return rc;
    if (d == 80) {
        static thread_local EqdRedArg RA8;
        EqdGatherArgs GA8;
        memset(&GA8, 0, sizeof(GA8));
        memset(&RA8, 0, sizeof(RA8));
        int nred8 = 0;
        if (gc) {
            if (int e = eqd_gather_plan(g, gc, pending, &GA8, &RA8, &nred8)) return e;
        }
        const int stride = eqd_attention_ds_stride(g), n_attn = g->n_att_items;
        const dim3 grid(n_attn + GA8.ngather + nred8);
        const bool dzb = gc && gc->dz_bf16;
#define EQD_KV80(BF_, DZ_)                                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kvds80<BF_, DZ_>), grid, dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse, d_out, dk, dv, \
                       qk_slope, ds, stride, seg_start, n_attn, nred8, GA8, RA8)
        if (bf16) { if (dzb) EQD_KV80(true, true); else EQD_KV80(true, false); }
        else { if (dzb) EQD_KV80(false, true); else EQD_KV80(false, false); }
#undef EQD_KV80
        if (int rc = eqd_check_launch("k_attn_bwd_kvds")) return rc;
        if (bf16)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_qds<5, 2, true>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k,
                               (const float*)ds, stride, dq, qk_slope);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_qds<5, 2>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k,
                               (const float*)ds, stride, dq, qk_slope);
        if (int rc = eqd_check_launch("k_attn_bwd_qds")) return rc;
        return gc ? eqd_gather_rest(pending, st) : EQD_OK;
    }
    static thread_local EqdRedArg RA;
    EqdGatherArgs GA;
    memset(&GA, 0, sizeof(GA));
    memset(&RA, 0, sizeof(RA));
    int nred = 0;
    if (gc) {
        if (int e = eqd_gather_plan(g, gc, pending, &GA, &RA, &nred)) return e;
    }
    const int n_attn = 2 * g->n_att_items, stride = eqd_attention_ds_stride(g);
    const dim3 grid(n_attn + GA.ngather + nred);
    if (bf16) {
        // (row stride of the bf16 dS workspace in ELEMENTS: the same count, rows half as long in bytes)
        const bool dzb = gc && gc->dz_bf16;
#define EQD_KVLB(DZ_, QB_)                                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kvds_lb<DZ_, QB_>), grid, dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse, d_out, dk, \
                       dv, qk_slope, ds, stride, seg_start, n_attn, nred, GA, RA)
        if (qkv_bf16) { if (dzb) EQD_KVLB(true, true); else EQD_KVLB(false, true); }
        else { if (dzb) EQD_KVLB(true, false); else EQD_KVLB(false, false); }
#undef EQD_KVLB
        if (int rc = eqd_check_launch("k_attn_bwd_kvds")) return rc;
        if (qkv_bf16)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_qds_lb<2, true>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k,
                               (const float*)ds, stride, dq, qk_slope);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_qds_lb<2, false>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k,
                               (const float*)ds, stride, dq, qk_slope);
        if (int rc = eqd_check_launch("k_attn_bwd_qds")) return rc;
        return gc ? eqd_gather_rest(pending, st) : EQD_OK;
    }
    if (gc && gc->dz_bf16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kvds<true>), grid, dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse, d_out, dk, dv,
                           qk_slope, ds, stride, seg_start, n_attn, nred, GA, RA);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_kvds<false>), grid, dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse, d_out, dk, dv,
                           qk_slope, ds, stride, seg_start, n_attn, nred, GA, RA);
    if (int rc = eqd_check_launch("k_attn_bwd_kvds")) return rc;
    // dq pass on 32-row blocks (one workgroup per item: half the K-tile traffic per dS byte; C + 0.5 %, E + 0.6 %, R equal
    // against half blocks, profiles/r04_k_*); EQD_ATT_QDS_NB=1 keeps half blocks (A/B runs)
    const char* qn = eqd_tunable("EQD_ATT_QDS_NB");
    if (!(qn && qn[0] == '1' && qn[1] == 0))
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_qds<4, 2>), dim3(g->n_att_items), dim3(EQD_BLOCK), 0, st, *g, q, k,
                           (const float*)ds, stride, dq, qk_slope);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_qds<4, 1>), dim3(n_attn), dim3(EQD_BLOCK), 0, st, *g, q, k, (const float*)ds,
                           stride, dq, qk_slope);
    if (int rc = eqd_check_launch("k_attn_bwd_qds")) return rc;
    return gc ? eqd_gather_rest(pending, st) : EQD_OK;
}
extern "C" size_t eqd_cross_attention_bwd_ds_workspace_bytes(const EqdGraph* g) { return g ? eqd_attention_ds_bytes(g) : 0; }
extern "C" int eqd_cross_attention_bwd_ds(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                          const float* lse, const float* d_out, float* dq, float* dk, float* dv, void* ws,
                                          size_t ws_bytes, void* stream) {
    if (!g || !q || !k || !v || !out || !lse || !d_out || !dq || !dk || !dv || !ws) {
        eqd_set_error("eqd_cross_attention_bwd_ds: NULL argument");
        return EQD_ERR_NULL;
    }
    if (!att_max_seg_plausible(g)) {
        eqd_set_error("eqd_cross_attention_bwd_ds: EqdGraph.max_seg = %d cannot be the longest protein of %d pairs with %d + %d "
                      "nodes (it is the dS row stride: see include/equidock_hip.h)", g->max_seg, g->n_pairs, g->n_lig, g->n_rec);
        return EQD_ERR_SHAPE;
    }
    if ((d != 64 && d != 80) || g->n_att_items % 8 != 0 || g->max_seg <= 0 ||
        !(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_out) && aligned16(out) && aligned16(ws))) {
        eqd_set_error("eqd_cross_attention_bwd_ds: needs d = 64 or 80, 16-byte aligned operands and the 8-way interleaved work list");
        return EQD_ERR_UNSUPPORTED;
    }
    if (ws_bytes < eqd_attention_ds_bytes(g)) {
        eqd_set_error("eqd_cross_attention_bwd_ds: workspace too small (%zu needed)", eqd_attention_ds_bytes(g));
        return EQD_ERR_WORKSPACE;
    }
    if (g->n_att_items <= 0) return EQD_OK;
    float* ds = (float*)ws;
    int32_t* seg = (int32_t*)((char*)ws + eqd_align_up((size_t)g->n_nodes * eqd_attention_ds_stride(g) * sizeof(float)));
    if (int rc = eqd_launch_seg_start(g, seg, (hipStream_t)stream)) return rc;
    return attention_bwd_ds(g, d, q, k, v, out, lse, d_out, dq, dk, dv, 1.f, ds, seg, nullptr, nullptr, (hipStream_t)stream);
}

int eqd_launch_attention_bwd_gather(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                    const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                                    float qk_slope, bool bf16, const EqdGatherCall* gc, EqdRedList* pending, hipStream_t st,
                                    float* ds, const int32_t* seg_start, bool qkv_bf16) {
    if (ds && seg_start && eqd_attention_ds_wanted(g, d, bf16) && aligned16(q) && aligned16(k) && aligned16(v) &&
        aligned16(d_out) && aligned16(out) && aligned16(ds))
        return attention_bwd_ds(g, d, q, k, v, out, lse, d_out, dq, dk, dv, qk_slope, ds, seg_start, gc, pending, st, bf16, qkv_bf16);
    if (qkv_bf16) {      // (the driver saves q / k / v as bf16 exactly when this launcher will take the dS form: same predicate)
        eqd_set_error("attention backward: q / k / v were saved as bf16 but the dS hand-off form is not taken");
        return EQD_ERR_UNSUPPORTED;
    }
    if (!eqd_attention_bwd_gather_fused(g, d, q, k, v, out, d_out, bf16)) {
        int rc = eqd_launch_attention_bwd_act(g, d, q, k, v, out, lse, d_out, dq, dk, dv, delta, qk_slope, bf16, st);
        if (rc) return rc;
        return eqd_launch_node_gather(g, gc->dz, gc->dxrel, gc->d_xnew, gc->a, gc->dP, gc->dQ, gc->dx, st, pending,
                                      gc->dz_bf16 != 0);
    }
    if (!q || !k || !v || !out || !lse || !d_out || !dq || !dk || !dv || !delta) {
        eqd_set_error("eqd_launch_attention_bwd_gather: NULL argument");
        return EQD_ERR_NULL;
    }
    static thread_local EqdRedArg RA;
    EqdGatherArgs GA;
    int nred = 0;
    if (int e = eqd_gather_plan(g, gc, pending, &GA, &RA, &nred)) return e;
    if (d == 80) {
        const int n_attn80 = 2 * g->n_att_items;
        const dim3 grid80(n_attn80 + GA.ngather + nred);
#define EQD_AB80(BF_, DZ_)                                                                                                         \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd80_gather<BF_, DZ_>), grid80, dim3(EQD_BLOCK), 0, st, *g, q, k, v, out, lse, d_out, dq, \
                       dk, dv, delta, qk_slope, n_attn80, nred, GA, RA)
        if (bf16) { if (gc->dz_bf16) EQD_AB80(true, true); else EQD_AB80(true, false); }
        else { if (gc->dz_bf16) EQD_AB80(false, true); else EQD_AB80(false, false); }
#undef EQD_AB80
        if (int rc = eqd_check_launch("k_attn_bwd_gather")) return rc;
        return eqd_gather_rest(pending, st);
    }
    const int n_attn = 4 * g->n_att_items;
    const dim3 grid(n_attn + GA.ngather + nred);
#define EQD_ABG_LAUNCH(BFDZ_, LB_)                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attn_bwd_gather<BFDZ_, LB_>), grid, dim3(EQD_BLOCK), 0, st, *g, d, q, k, v, out, lse,     \
                       d_out, dq, dk, dv, delta, qk_slope, n_attn, nred, GA, RA)
    if (bf16) {
        if (gc->dz_bf16) EQD_ABG_LAUNCH(true, true); else EQD_ABG_LAUNCH(false, true);
    } else {
        if (gc->dz_bf16) EQD_ABG_LAUNCH(true, false); else EQD_ABG_LAUNCH(false, false);
    }
#undef EQD_ABG_LAUNCH
    if (int rc = eqd_check_launch("k_attn_bwd_gather")) return rc;
    return eqd_gather_rest(pending, st);
}
