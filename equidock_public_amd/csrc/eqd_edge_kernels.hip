// Edge message + coordinate update of one IEGMN layer, forward and backward, for the ll and rr
// graphs of every pair in ONE launch (both edge types share all weights).
//
// Reference arithmetic replaced (src/model/rigid_docking_model.py):
//   :204-205  x_rel = x[src] - x[dst]                (DGL u_sub_v)
//   :208-214  15 RBFs exp(-|x_rel|^2 / 1.5^k)
//   :226-237  edge_mlp([h_src, h_dst, he, rbf])      = Linear, LeakyReLU, LayerNorm, Linear
//   :263-266  coef = coors_mlp(msg); x_moment = x_rel * coef
//   :274-283  per-destination means of x_moment and msg (DGL copy_edge + mean, zero if no in-edge)
//   :286-292  x' = eta x0 + (1 - eta) x + x_update
// and its autograd backward (src/train.py:154).
//
// Design (see eqd_common.h for the MFMA convention):
//   * the first Linear is split algebraically: W1 [h_s; h_d; he; rbf] + b1 =
//     P[src] + Q[dst] + W1c he + W1d rbf with P = h W1a^T, Q = h W1b^T + b1 computed once per NODE
//     by k_linear -- per-edge MFMA work drops from K=170 to K=42;
//   * FORWARD: a wave owns a node-aligned tile of <= 32 edges (edges are destination-sorted, so the
//     per-destination mean is a within-tile reduction; no atomics); the three 64x64 GEMMs chain through
//     registers; weights (45 KB) are staged once per workgroup in LDS; nothing per-edge goes to HBM;
//   * BACKWARD: a wave owns 16 consecutive edges (no node alignment needed: the per-destination sums
//     dQ[dst], dx[dst] are done by k_node_gather from the per-edge dz1 / dx_rel that are written anyway),
//     which halves the live accumulators -> 4 waves per SIMD instead of 1, and twice the tiles to spread
//     over the chip at DB5 sizes.  It recomputes the tile forward and writes the five per-edge operands
//     of the weight-gradient GEMMs (a1, m, d_chid, dm, dz1) for k_atb.
#include "eqd_common.h"
#include "eqd_attn_fwd_inl.h"

#include <stdlib.h>
#include <string.h>

#define WS1 45   /* row stride of the staged W1[:, 2d_in:] block (42 used + 3 zero) */
#define WS2 68   /* row stride of staged 64x64 weights: 272 B = 17 x 16 B -> conflict-free b128 */
#define TS 68    /* row stride of the per-wave [edges][64+4] message tile (forward) */
#define FS 45    /* row stride of the per-wave [edges][42+3] feature tile */
#define VEC_LNG 0
#define VEC_LNB 64
#define VEC_B2 128
#define VEC_BC1 192
#define VEC_WC2 256
#define VEC_BC2 320
#define VEC_N 324
#define FWD_WAVES 8   /* forward : 32-edge tiles, ~166 VGPRs -> 2 waves per SIMD */
#define BWD_WAVES 8   /* backward: 16-edge tiles, small register footprint -> 2 workgroups per CU */

template <int NW, int TILE_FLOATS>
struct alignas(16) EdgeSmem {
    float w1[64 * WS1];
    float w2[64 * WS2];
    float wc1[64 * WS2];
    float vec[VEC_N];
    float tile[NW][TILE_FLOATS];
};

template <int NW, int TF>
__device__ __forceinline__ void edge_stage_weights(EdgeSmem<NW, TF>& sm, const EqdEdgeParams& P) {
    // ALL global loads of a thread (W1's feature columns, W2, Wc1, the five vectors) are issued before the first LDS
    // store (constant trip counts, fully unrolled, unpredicated): ONE L2 round trip for the whole staging - three
    // batches with their own waits were 4 200 clocks at the head of every workgroup of the backward
    constexpr int NT = 64 * NW;
    constexpr int N1 = (64 * 11 + NT - 1) / NT, N2 = 1024 / NT;
    const int t = threadIdx.x;
    const int koff = 2 * P.d_in;
    f32x4 v[N1];
    float4 a[N2], b[N2];
    float vec5[5];
#pragma unroll
    for (int j = 0; j < N1; ++j) {       // 64 rows x 42 floats = 64 x 11 16-byte segments (the last holds 2)
        const int i = t + j * NT;
        const int r = i / 11, c = 4 * (i - r * 11);
        v[j] = ld4u_raw(P.W1 + (size_t)(i < 64 * 11 ? r : 0) * P.ldw1 + koff + c, i < 64 * 11 ? 42 - c : 0, P.W1);
    }
#pragma unroll
    for (int j = 0; j < N2; ++j) {       // 64 x 64 floats = 1024 float4
        const int i = t + j * NT;
        a[j] = ((const float4*)P.W2)[i];
        b[j] = ((const float4*)P.Wc1)[i];
    }
    {
        const int tc = t & 63;
        vec5[0] = P.ln_g[tc];
        vec5[1] = P.ln_b[tc];
        vec5[2] = P.b2[tc];
        vec5[3] = P.bc1[tc];
        vec5[4] = P.wc2[tc];
    }
    const float bc2 = P.bc2[0];
#pragma unroll
    for (int j = 0; j < N1; ++j) {
        const int i = t + j * NT;
        const int r = i / 11, c = 4 * (i - r * 11);
        if (i < 64 * 11) {
            const float4 f = ld4u_fix(v[j], 42 - c);
            float* o = &sm.w1[r * WS1 + c];
            o[0] = f.x;
            if (c + 1 < WS1) o[1] = f.y;      // c = 40: 42, 43, 44 are the zero padding (ld4u_fix zero-fills)
            if (c + 2 < WS1) o[2] = f.z;
            if (c + 3 < WS1) o[3] = f.w;
        }
    }
    if (t < 64) sm.w1[t * WS1 + 44] = 0.f;
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        const int i = t + j * NT;
        const int r = i >> 4, c = (i & 15) * 4;
        *(float4*)&sm.w2[r * WS2 + c] = a[j];
        *(float4*)&sm.wc1[r * WS2 + c] = b[j];
    }
    if (t < 64) {
        sm.vec[VEC_LNG + t] = vec5[0];
        sm.vec[VEC_LNB + t] = vec5[1];
        sm.vec[VEC_B2 + t] = vec5[2];
        sm.vec[VEC_BC1 + t] = vec5[3];
        sm.vec[VEC_WC2 + t] = vec5[4];
    }
    if (t == 0) sm.vec[VEC_BC2] = bc2;
    __syncthreads();
}

__device__ __forceinline__ float rbf_sigma(int k) {
    // 1.5 ** k as fp32, k = 0..14 (rigid_docking_model.py:116)
    const float t[16] = {1.f,          1.5f,         2.25f,        3.375f,        5.0625f,     7.59375f,
                         11.390625f,   17.0859375f,  25.62890625f, 38.443359375f, 57.6650390625f,
                         86.49755859375f, 129.746337890625f, 194.6195068359375f, 291.92926025390625f, 1.f};
    return t[k & 15];
}

__device__ __forceinline__ float rbf_neg_inv_sigma(int k) {      // -1.f / rbf_sigma(k), the correctly rounded quotients
    const float t[16] = {-1.f / 1.f,          -1.f / 1.5f,         -1.f / 2.25f,        -1.f / 3.375f,        -1.f / 5.0625f,
                         -1.f / 7.59375f,     -1.f / 11.390625f,   -1.f / 17.0859375f,  -1.f / 25.62890625f,  -1.f / 38.443359375f,
                         -1.f / 57.6650390625f, -1.f / 86.49755859375f, -1.f / 129.746337890625f, -1.f / 194.6195068359375f,
                         -1.f / 291.92926025390625f, -1.f};
    return t[k & 15];
}

// y = W x chained through registers: out[mbo][nb] += sum_{mbi,r} W[16 mbo + l15][16 mbi + 4 g + r] * in[mbi][nb][r]
// W staged in LDS with row stride WS2.  If AFF, `in` is first mapped through v * ga[f] + be[f].
template <bool AFF, int NB>
__device__ __forceinline__ void chain64(f32x4 (&out)[4][NB], const f32x4 (&in)[4][NB], const float* __restrict__ W,
                                        const float* __restrict__ ga, const float* __restrict__ be, int l15, int g) {
#pragma unroll
    for (int mbi = 0; mbi < 4; ++mbi) {
        f32x4 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = in[mbi][nb];
        if (AFF) {
            const float4 gg = *(const float4*)&ga[16 * mbi + 4 * g];
            const float4 bb = *(const float4*)&be[16 * mbi + 4 * g];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                b[nb][0] = b[nb][0] * gg.x + bb.x;
                b[nb][1] = b[nb][1] * gg.y + bb.y;
                b[nb][2] = b[nb][2] * gg.z + bb.z;
                b[nb][3] = b[nb][3] * gg.w + bb.w;
            }
        }
        float4 w[4];      // (the four weight fragments of the input block first, see chain64T)
#pragma unroll
        for (int mbo = 0; mbo < 4; ++mbo) w[mbo] = *(const float4*)&W[(16 * mbo + l15) * WS2 + 16 * mbi + 4 * g];
#if !defined(EQD_HOSTSIM) && !defined(EQD_NO_LDS_BATCH) && !defined(EQD_NO_LDS_BATCH_ROWEDGE)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int mbo = 0; mbo < 4; ++mbo) {
            const float wv[4] = {w[mbo].x, w[mbo].y, w[mbo].z, w[mbo].w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) out[mbo][nb] = mfma4(wv[r], b[nb][r], out[mbo][nb]);
        }
    }
}
// y = W^T x: out[mbo][nb] += sum_{mbi,r} W[16 mbi + 4 g + r][16 mbo + l15] * in[mbi][nb][r]
template <int NB>
__device__ __forceinline__ void chain64T(f32x4 (&out)[4][NB], const f32x4 (&in)[4][NB], const float* __restrict__ W,
                                         int l15, int g) {
#pragma unroll
    for (int mbi = 0; mbi < 4; ++mbi) {
        // the 16 weight fragments of the input block first, then its 16 NB MFMAs (left alone the scheduler puts every LDS
        // read right in front of its MFMA: read - s_waitcnt lgkmcnt(0) - MFMA, 64 times per call)
        float w[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mbo = 0; mbo < 4; ++mbo) w[r][mbo] = W[(16 * mbi + 4 * g + r) * WS2 + 16 * mbo + l15];
#if !defined(EQD_HOSTSIM) && !defined(EQD_NO_LDS_BATCH) && !defined(EQD_NO_LDS_BATCH_ROWEDGE)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mbo = 0; mbo < 4; ++mbo) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) out[mbo][nb] = mfma4(w[r][mbo], in[mbi][nb][r], out[mbo][nb]);
            }
    }
}

template <int NB>
struct EdgeTileState {
    int e0, ne, n0, n1;
    int src[NB], dst[NB];
    bool ev[NB];
    float xrel[NB][3];
    float d2[NB];
    float rbf[NB][4];   // PRE (the backward's recompute), bf16 mode: exp(-d2 / sigma_k) of k = 4 g + j in fp32, for d rbf -> d(d^2)
    float mean[NB], rstd[NB];
    float coef[NB];
    unsigned zpos;   // bit (16 nb + 4 mb + r): edge_mlp.0 pre-activation > 0 (exact LeakyReLU mask for the backward)
    unsigned keepz, keepc;   // dropout (training): bit (16 nb + 4 mb + r) = the element of edge_mlp.1 / coors_mlp.1 is kept
};

// Dropout of the two edge MLPs (rigid_docking_model.py:119-125 edge_mlp.1, :153-159 coors_mlp.1; active while training with
// args['dropout'] > 0).  The masks are drawn by the CALLER with torch's generator in the reference's consumption order
// (model.py: _draw_dropout) and arrive bit-packed: two 32-bit words per edge, bit f of the pair = feature f kept.
// Dropout sits between the Linear and the LeakyReLU; LeakyReLU is positively homogeneous, so
// LeakyReLU(keep * s * z) = keep * s * LeakyReLU(z): the kernels apply the factor to the activation, and the backward
// multiplies the LeakyReLU derivative by the same factor.  This lane's 16 bits of edge `e`: features 16 mb + 4 g + r.
__device__ __forceinline__ unsigned drop_bits16(const uint32_t* __restrict__ words, int e, int g) {
    const uint32_t w0 = words[(size_t)e * 2], w1 = words[(size_t)e * 2 + 1];
    unsigned b = 0u;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) b |= (((mb < 2 ? w0 : w1) >> (16 * (mb & 1) + 4 * g)) & 0xfu) << (4 * mb);
    return b;
}
template <int NB>
__device__ __forceinline__ void drop_load(const EqdEdgeParams& P, EdgeTileState<NB>& S, int l15, int g) {
    S.keepz = S.keepc = 0u;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int e = S.e0 + ((16 * nb + l15) < S.ne ? 16 * nb + l15 : 0);
        S.keepz |= drop_bits16(P.drop_z1, e, g) << (16 * nb);
        S.keepc |= drop_bits16(P.drop_ch, e, g) << (16 * nb);
    }
}

// Training forward: leave the per-edge state the backward would otherwise recompute (EqdEdgeParams.xh_save / rstd_save /
// zpos_save).  Called INSIDE the tile forward, as soon as xh is final: the two 64 x 64 GEMM chains, the coefficient and the
// per-node aggregation that follow issue no global load, so the stores drain behind arithmetic.  Placed after the tile
// forward - in front of the next tile's index loads, whose s_waitcnt vmcnt counts the stores too - they cost k_edge_fwd<bf16>
// + 16 us per launch at 64 x (300, 300) (66 -> 83 us; profiles/r06_b_kernels_C_bf16.md).
template <int NB>
__device__ __forceinline__ void edge_state_store(const EqdEdgeParams& P, const EdgeTileState<NB>& S, const f32x4 (&xh)[4][NB],
                                                 int l15, int g) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        if (S.ev[nb]) {
            const size_t er = (size_t)S.e0 + 16 * nb + l15;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                *(float4*)&P.xh_save[er * 64 + 16 * mb + 4 * g] =
                    make_float4(xh[mb][nb][0], xh[mb][nb][1], xh[mb][nb][2], xh[mb][nb][3]);
            P.zpos_save[er * 4 + g] = (uint16_t)((S.zpos >> (16 * nb)) & 0xffffu);
            if (g == 0) P.rstd_save[er] = S.rstd[nb];
        }
}

// Forward of one tile of 16*NB edges up to (and including) the coefficient. On return:
//   xh = LayerNorm-normalised hidden (before the affine), m = msg, ch = coors_mlp hidden pre-activation.
// If rbf_out != nullptr the 15 RBFs of each edge are also written there ([E][16]).
// PRE: S.src / S.dst hold the tile's endpoints already (k_edge_bwd fetches them one tile ahead)
// SAVED (the backward, round 6): the forward left the edge's LayerNorm-normalised hidden row xh (fp32), its rstd and the
// LeakyReLU sign bits of z1 in HBM (EqdEdgeParams.xh_save / rstd_save / zpos_save, 268 B per edge).  The tile then LOADS
// them - one contiguous 256-byte row per edge - instead of gathering P[src] and Q[dst] (512 B per edge from two random
// rows), running the first Linear's feature GEMM, the LeakyReLU and both LayerNorm statistics again.  The saved values
// are what this function computes: same bits downstream (test_edge_saved_state_is_bit_identical).
template <int NB, bool DROP = false, bool PRE = false, bool SAVED = false>
__device__ __forceinline__ void edge_tile_forward(const EqdGraph& G, const EqdEdgeParams& P, const float* __restrict__ w1,
                                                  const float* __restrict__ w2, const float* __restrict__ wc1,
                                                  const float* __restrict__ vec, float* __restrict__ tile,
                                                  const float* __restrict__ Pn, const float* __restrict__ Qn,
                                                  const float* __restrict__ x, int lane, EdgeTileState<NB>& S,
                                                  f32x4 (&xh)[4][NB], f32x4 (&m)[4][NB], f32x4 (&ch)[4][NB],
                                                  float* __restrict__ rbf_out) {
    const int l15 = lane & 15, g = lane >> 4;
    EQD_TR(2);
    // Every global load of the tile is issued up front, unpredicated (lanes beyond the tile read edge e0, always a
    // valid index, and are zeroed afterwards; predicated loads compile to exec-masked branches with a wait behind
    // each): the he rows, src / dst, then - one dependent round trip later - the coordinates and the P / Q rows.
    // ---- he rows: 16 NB x 27 contiguous floats ---------------------------------------------------
    constexpr int NH = (16 * NB * 27 / 4 + 63) / 64;       // 16-byte segments per lane
    const float* __restrict__ he = G.he + (size_t)S.e0 * 27;
    const int nhe = P.use_he ? S.ne * 27 : 0;
    f32x4 hv[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int i = 4 * (lane + 64 * j);
        hv[j] = ld4u_raw(he + (i < nhe ? i : 0), nhe - i, G.he);
    }
    // ---- geometry ---------------------------------------------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int el = 16 * nb + l15;
        S.ev[nb] = el < S.ne;
        if constexpr (!PRE) {
            const int ei = S.e0 + (S.ev[nb] ? el : 0);
            S.src[nb] = G.src[ei];
            S.dst[nb] = G.dst[ei];
        }
    }
    float xs[NB][3], xd[NB][3];
    float4 pv[NB][4], qv[NB][4];      // !SAVED: P[src], Q[dst];  SAVED: pv = the edge's saved xh row (qv unused)
    float rsv[NB];
    unsigned zsv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xs[nb][c] = x[(size_t)S.src[nb] * 3 + c];
            xd[nb][c] = x[(size_t)S.dst[nb] * 3 + c];
        }
        if constexpr (SAVED) {
            const size_t er = (size_t)S.e0 + (S.ev[nb] ? 16 * nb + l15 : 0);      // (lanes beyond the tile: its first edge)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) pv[nb][mb] = *(const float4*)&P.xh_save[er * 64 + 16 * mb + 4 * g];
            rsv[nb] = P.rstd_save[er];
            zsv[nb] = P.zpos_save[er * 4 + g];
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                pv[nb][mb] = *(const float4*)&Pn[(size_t)S.src[nb] * 64 + 16 * mb + 4 * g];
                qv[nb][mb] = *(const float4*)&Qn[(size_t)S.dst[nb] * 64 + 16 * mb + 4 * g];
            }
        }
    }
    // ---- feature tile [16 NB][45]: he (27) | rbf (15) | 0 -----------------------------------------
    for (int i = lane; i < 16 * NB * FS; i += 64) tile[i] = 0.f;
    wave_lds_fence();
    EQD_TR(3);
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int i = 4 * (lane + 64 * j);
        const float4 hf = ld4u_fix(hv[j], nhe - i);
        const float vv[4] = {hf.x, hf.y, hf.z, hf.w};
        int e = i / 27, c = i - e * 27;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u < nhe) tile[e * FS + c] = vv[u];
            if (++c == 27) {
                c = 0;
                ++e;
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = S.ev[nb] ? xs[nb][c] - xd[nb][c] : 0.f;
            S.xrel[nb][c] = v;
            q += v * v;
        }
        S.d2[nb] = q;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int el = 16 * nb + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = PRE ? 4 * g + j : g + 4 * j;      // (which lane computes which k is free: same expression, same bits)
            if (k < 15 && S.ev[nb]) {
                const float v = P.use_dist ? exp_nooverflow(-S.d2[nb] / rbf_sigma(k)) : 0.f;
                tile[el * FS + 27 + k] = v;
                if (rbf_out) rbf_out[(size_t)(S.e0 + el) * 16 + k] = v;
            }
        }
        if (rbf_out && g == 3 && S.ev[nb]) rbf_out[(size_t)(S.e0 + el) * 16 + 15] = 0.f;
    }
    wave_lds_fence();
    EQD_TR(4);
    S.zpos = 0u;
    if constexpr (DROP) drop_load<NB>(P, S, l15, g);
    if constexpr (SAVED) {
        // the saved row IS xh (lanes beyond the tile: 0, what the recompute gives them), with its rstd and sign bits
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 v = pv[nb][mb];
                f32x4 a;
                a[0] = S.ev[nb] ? v.x : 0.f; a[1] = S.ev[nb] ? v.y : 0.f; a[2] = S.ev[nb] ? v.z : 0.f; a[3] = S.ev[nb] ? v.w : 0.f;
                xh[mb][nb] = a;
            }
            S.zpos |= (S.ev[nb] ? zsv[nb] : 0u) << (16 * nb);
            S.mean[nb] = 0.f;
            S.rstd[nb] = rsv[nb];
        }
    } else {
    // ---- stage 1: z1 = P[src] + Q[dst] + W1cd feat ----------------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 p = pv[nb][mb], q = qv[nb][mb];
            f32x4 a;
            a[0] = S.ev[nb] ? p.x + q.x : 0.f; a[1] = S.ev[nb] ? p.y + q.y : 0.f;
            a[2] = S.ev[nb] ? p.z + q.z : 0.f; a[3] = S.ev[nb] ? p.w + q.w : 0.f;
            xh[mb][nb] = a;
        }
    EQD_TR(5);
#pragma unroll
    for (int s = 0; s < 11; ++s) {
        const int k = 4 * s + g;
        float b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = tile[(16 * nb + l15) * FS + k];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float a = w1[(16 * mb + l15) * WS1 + k];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) xh[mb][nb] = mfma4(a, b[nb], xh[mb][nb]);
        }
    }
    EQD_TR(6);
    // ---- LeakyReLU + LayerNorm statistics (two-pass like torch) ------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = xh[mb][nb][r];
                if (z > 0.f) S.zpos |= 1u << (16 * nb + 4 * mb + r);
                float v = lrelu(z, P.slope);
                if constexpr (DROP) v *= ((S.keepz >> (16 * nb + 4 * mb + r)) & 1u) ? P.drop_scale : 0.f;
                xh[mb][nb][r] = v;
                s += v;
            }
        const float mean = group_sum(s) * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = xh[mb][nb][r] - mean;
                q += d * d;
            }
        const float rstd = 1.f / sqrtf(group_sum(q) * (1.f / 64.f) + P.ln_eps);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) xh[mb][nb][r] = (xh[mb][nb][r] - mean) * rstd;
        S.mean[nb] = mean;
        S.rstd[nb] = rstd;
    }
    }      // (!SAVED)
    if constexpr (!SAVED) {
        if (P.xh_save) edge_state_store<NB>(P, S, xh, l15, g);
    }
    EQD_TR(7);
    // ---- stage 2: m = W2 (xh * gamma + beta) + b2 --------------------------------------------------
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const float4 b = *(const float4*)&vec[VEC_B2 + 16 * mb + 4 * g];
        f32x4 v = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) m[mb][nb] = v;
    }
    chain64<true, NB>(m, xh, w2, &vec[VEC_LNG], &vec[VEC_LNB], l15, g);
    EQD_TR(8);
    // ---- stage 3: ch = Wc1 m + bc1; coef = wc2 . LeakyReLU(ch) + bc2 ---------------------------------
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const float4 b = *(const float4*)&vec[VEC_BC1 + 16 * mb + 4 * g];
        f32x4 v = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) ch[mb][nb] = v;
    }
    chain64<false, NB>(ch, m, wc1, nullptr, nullptr, l15, g);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 w = *(const float4*)&vec[VEC_WC2 + 16 * mb + 4 * g];
            if constexpr (DROP) {
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s += lrelu(ch[mb][nb][r], P.slope) *
                         (((S.keepc >> (16 * nb + 4 * mb + r)) & 1u) ? P.drop_scale : 0.f) * wv[r];
            } else {
                s += lrelu(ch[mb][nb][0], P.slope) * w.x + lrelu(ch[mb][nb][1], P.slope) * w.y +
                     lrelu(ch[mb][nb][2], P.slope) * w.z + lrelu(ch[mb][nb][3], P.slope) * w.w;
            }
        }
        S.coef[nb] = group_sum(s) + vec[VEC_BC2];
    }
    EQD_TR(9);
}

// ---------------------------------------------------------------------------------------------
// bf16 mode (EqdEdgeParams.bf16): he is read as bf16 rows of 32 (27 used, 64 B per edge, EqdGraph.he_bf16), the
// staged weights and the GEMM inputs (features, LayerNorm output, messages) are rounded to bf16, every product is
// accumulated in fp32 by v_mfma_f32_16x16x16_bf16 (16x the fp32 MFMA rate).  P/Q, coordinates, RBF arguments,
// LayerNorm statistics, biases, the coordinate head and all outputs stay fp32.
// ---------------------------------------------------------------------------------------------
#define WSB 72   /* bf16 row stride of the staged weights: 36 dwords = 4 mod 32 -> b64 fragment reads at the LDS rate */
#define FSB 72   /* bf16 row stride of the per-wave [edges][48] feature tile */
template <int NW, int TILE_FLOATS>
struct alignas(16) EdgeSmemBf {
    unsigned short w1[64 * WSB];    // W1[:, 2 d_in:] : 42 columns + 6 zeros
    unsigned short w2[64 * WSB];
    unsigned short wc1[64 * WSB];
    float vec[VEC_N];
    float tile[NW][TILE_FLOATS];    // fp32 message tile; the bf16 feature tile aliases its first bytes
};
// backward: additionally the transposed 64 x 64 matrices (data-gradient chains) and the RBF columns of W1 transposed
template <int NW, int TILE_FLOATS>
struct alignas(16) EdgeSmemBfBwd {
    unsigned short w1[64 * WSB];
    unsigned short w2[64 * WSB];
    unsigned short wc1[64 * WSB];
    unsigned short w2T[64 * WSB];
    unsigned short wc1T[64 * WSB];
    unsigned short w1rT[16 * WSB];  // [k = RBF index (15 used)][m]
    float vec[VEC_N];
    float tile[NW][TILE_FLOATS];
};
template <int NW, bool WITH_T, class SM>
__device__ __forceinline__ void edge_stage_weights_bf(SM& sm, const EqdEdgeParams& P) {
    constexpr int NT = 64 * NW;
    constexpr int N1 = (64 * 12 + NT - 1) / NT, N2 = 1024 / NT;
    const int t = threadIdx.x;
    const int koff = 2 * P.d_in;
    if constexpr (WITH_T) {
        if (t < 64) sm.w1rT[15 * WSB + t] = 0;
    }
    {
        f32x4 v[N1];
#pragma unroll
        for (int j = 0; j < N1; ++j) {       // 64 rows x 12 four-column segments (42 valid columns, rest zero)
            const int i = t + j * NT;
            const int r = i / 12, c = 4 * (i - r * 12);
            v[j] = ld4u_raw(P.W1 + (size_t)(i < 64 * 12 ? r : 0) * P.ldw1 + koff + c, i < 64 * 12 ? 42 - c : 0, P.W1);
        }
#pragma unroll
        for (int j = 0; j < N1; ++j) {
            const int i = t + j * NT;
            const int r = i / 12, c = 4 * (i - r * 12);
            if (i < 64 * 12) {
                const float4 f = ld4u_fix(v[j], 42 - c);
                const s16x4 h = pack_bf4(f.x, f.y, f.z, f.w);
                *(s16x4*)&sm.w1[r * WSB + c] = h;
                if constexpr (WITH_T) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c + u >= 27 && c + u < 42) sm.w1rT[(c + u - 27) * WSB + r] = (unsigned short)h[u];
                }
            }
        }
    }
    {
        float4 a[N2], b[N2];
#pragma unroll
        for (int j = 0; j < N2; ++j) {       // 64 x 64 floats = 1024 float4
            const int i = t + j * NT;
            a[j] = ((const float4*)P.W2)[i];
            b[j] = ((const float4*)P.Wc1)[i];
        }
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            const int i = t + j * NT;
            const int r = i >> 4, c = (i & 15) * 4;
            const s16x4 ha = pack_bf4(a[j].x, a[j].y, a[j].z, a[j].w), hb = pack_bf4(b[j].x, b[j].y, b[j].z, b[j].w);
            *(s16x4*)&sm.w2[r * WSB + c] = ha;
            *(s16x4*)&sm.wc1[r * WSB + c] = hb;
            if constexpr (WITH_T) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sm.w2T[(c + u) * WSB + r] = (unsigned short)ha[u];
                    sm.wc1T[(c + u) * WSB + r] = (unsigned short)hb[u];
                }
            }
        }
    }
    if (t < 64) {
        sm.vec[VEC_LNG + t] = P.ln_g[t];
        sm.vec[VEC_LNB + t] = P.ln_b[t];
        sm.vec[VEC_B2 + t] = P.b2[t];
        sm.vec[VEC_BC1 + t] = P.bc1[t];
        sm.vec[VEC_WC2 + t] = P.wc2[t];
    }
    if (t == 0) sm.vec[VEC_BC2] = P.bc2[0];
    __syncthreads();
}

// out[mb][nb] += W[16 mb .., :] in[.., nb] over a 64-wide contraction: `in` is the previous tile (fp32 registers, optionally
// through the LayerNorm affine), packed to bf16 per 16-feature block - it IS the B operand of k-chunk mbi.
template <bool AFF, int NB>
__device__ __forceinline__ void chain64_bf(f32x4 (&out)[4][NB], const f32x4 (&in)[4][NB], const unsigned short* __restrict__ w,
                                           const float* __restrict__ ga, const float* __restrict__ be, int l15, int g) {
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {      // two 16-feature blocks of `in` = one 32-deep k-chunk of v_mfma_f32_16x16x32_bf16
        s16x8 b[NB];
        s16x4 bh[2][NB];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int mbi = 2 * mp + h;
            float gg[4] = {1.f, 1.f, 1.f, 1.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
            if (AFF) {
                const float4 gv = *(const float4*)&ga[16 * mbi + 4 * g];
                const float4 bv = *(const float4*)&be[16 * mbi + 4 * g];
                gg[0] = gv.x; gg[1] = gv.y; gg[2] = gv.z; gg[3] = gv.w;
                bb[0] = bv.x; bb[1] = bv.y; bb[2] = bv.z; bb[3] = bv.w;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bh[h][nb] = pack_bf4(in[mbi][nb][0] * gg[0] + bb[0], in[mbi][nb][1] * gg[1] + bb[1],
                                     in[mbi][nb][2] * gg[2] + bb[2], in[mbi][nb][3] * gg[3] + bb[3]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = cat_bf(bh[0][nb], bh[1][nb]);
        s16x8 a[4];      // (the chunk's four weight fragments first, see chain64T)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            a[mb] = cat_bf(*(const s16x4*)&w[(16 * mb + l15) * WSB + 32 * mp + 4 * g],
                           *(const s16x4*)&w[(16 * mb + l15) * WSB + 32 * mp + 16 + 4 * g]);
#if !defined(EQD_HOSTSIM) && !defined(EQD_NO_LDS_BATCH) && !defined(EQD_NO_LDS_BATCH_ROWEDGE)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) out[mb][nb] = mfma_bf32(a[mb], b[nb], out[mb][nb]);
        }
    }
}

// bf16 counterpart of edge_tile_forward (same outputs, same EdgeTileState)
// PRE: S.src / S.dst hold the tile's endpoints already (k_edge_bwd fetches them one tile ahead)
template <int NB, bool DROP = false, bool PRE = false, bool SAVED = false>
__device__ __forceinline__ void edge_tile_forward_bf(const EqdGraph& G, const EqdEdgeParams& P,
                                                     const unsigned short* __restrict__ w1,
                                                     const unsigned short* __restrict__ w2,
                                                     const unsigned short* __restrict__ wc1,
                                                     const float* __restrict__ vec, float* __restrict__ tile,
                                                     const float* __restrict__ Pn, const float* __restrict__ Qn,
                                                     const float* __restrict__ x, int lane, EdgeTileState<NB>& S,
                                                     f32x4 (&xh)[4][NB], f32x4 (&m)[4][NB], f32x4 (&ch)[4][NB]) {
    const int l15 = lane & 15, g = lane >> 4;
    unsigned short* __restrict__ ft = (unsigned short*)tile;
    EQD_TR(2);
    // ---- he rows: 64 B per edge, four 16-byte parts; lane -> (edge, part), all loads up front, unpredicated ------
    f32x4 hv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i4 = lane + 64 * j, e = i4 >> 2, part = i4 & 3;
        hv[j] = *(const f32x4*)(G.he_bf16 + ((size_t)S.e0 + (e < S.ne ? e : 0)) * 32 + 8 * part);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int el = 16 * nb + l15;
        S.ev[nb] = el < S.ne;
        if constexpr (!PRE) {
            const int ei = S.e0 + (S.ev[nb] ? el : 0);
            S.src[nb] = G.src[ei];
            S.dst[nb] = G.dst[ei];
        }
    }
    float xs[NB][3], xd[NB][3];
    float4 pv[NB][4], qv[NB][4];      // !SAVED: P[src], Q[dst];  SAVED: pv = the edge's saved xh row (qv unused)
    float rsv[NB];
    unsigned zsv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xs[nb][c] = x[(size_t)S.src[nb] * 3 + c];
            xd[nb][c] = x[(size_t)S.dst[nb] * 3 + c];
        }
        if constexpr (SAVED) {
            const size_t er = (size_t)S.e0 + (S.ev[nb] ? 16 * nb + l15 : 0);      // (lanes beyond the tile: its first edge)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) pv[nb][mb] = *(const float4*)&P.xh_save[er * 64 + 16 * mb + 4 * g];
            rsv[nb] = P.rstd_save[er];
            zsv[nb] = P.zpos_save[er * 4 + g];
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                pv[nb][mb] = *(const float4*)&Pn[(size_t)S.src[nb] * 64 + 16 * mb + 4 * g];
                qv[nb][mb] = *(const float4*)&Qn[(size_t)S.dst[nb] * 64 + 16 * mb + 4 * g];
            }
        }
    }
    // ---- feature tile [16 NB][48] bf16: he (27) | rbf (15) | 0 (6) -------------------------------------------------
    EQD_TR(3);
    {
        const f32x4 z = f4zero();
        if (lane < 32 * NB) *(f32x4*)&ft[(lane >> 1) * FSB + 32 + 8 * (lane & 1)] = z;      // columns 32..47
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i4 = lane + 64 * j, e = i4 >> 2, part = i4 & 3;
            *(f32x4*)&ft[e * FSB + 8 * part] = (P.use_he && e < S.ne) ? hv[j] : z;            // columns 0..31 (27.. are 0)
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = S.ev[nb] ? xs[nb][c] - xd[nb][c] : 0.f;
            S.xrel[nb][c] = v;
            q += v * v;
        }
        S.d2[nb] = q;
        const int el = 16 * nb + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = PRE ? 4 * g + j : g + 4 * j;
            float v = 0.f;
            if (k < 15 && S.ev[nb]) {
                // (bf16 mode: times the rounded reciprocal instead of the fp32 division - ten VALU instructions per value; the
                //  feature is rounded to bf16 below, the backward's recompute runs this same code)
                v = P.use_dist ? exp_nooverflow(q * rbf_neg_inv_sigma(k)) : 0.f;
                ft[el * FSB + 27 + k] = f2bf(v);
            }
            if constexpr (PRE) S.rbf[nb][j] = v;      // fp32 copy for the backward's d rbf -> d(d^2) (the tile holds bf16)
        }
    }
    wave_lds_fence();
    EQD_TR(4);
    S.zpos = 0u;
    if constexpr (DROP) drop_load<NB>(P, S, l15, g);
    if constexpr (SAVED) {      // (see edge_tile_forward)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 v = pv[nb][mb];
                f32x4 a;
                a[0] = S.ev[nb] ? v.x : 0.f; a[1] = S.ev[nb] ? v.y : 0.f; a[2] = S.ev[nb] ? v.z : 0.f; a[3] = S.ev[nb] ? v.w : 0.f;
                xh[mb][nb] = a;
            }
            S.zpos |= (S.ev[nb] ? zsv[nb] : 0u) << (16 * nb);
            S.mean[nb] = 0.f;
            S.rstd[nb] = rsv[nb];
        }
    } else {
    // ---- stage 1: z1 = P[src] + Q[dst] (fp32) + W1cd feat (3 k-chunks of 16) ------------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 p = pv[nb][mb], q = qv[nb][mb];
            f32x4 a;
            a[0] = S.ev[nb] ? p.x + q.x : 0.f; a[1] = S.ev[nb] ? p.y + q.y : 0.f;
            a[2] = S.ev[nb] ? p.z + q.z : 0.f; a[3] = S.ev[nb] ? p.w + q.w : 0.f;
            xh[mb][nb] = a;
        }
    EQD_TR(5);
    {      // 48 feature columns = one 32-deep chunk (v_mfma_f32_16x16x32_bf16) + one 16-deep one
        s16x8 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            b[nb] = cat_bf(*(const s16x4*)&ft[(16 * nb + l15) * FSB + 4 * g], *(const s16x4*)&ft[(16 * nb + l15) * FSB + 16 + 4 * g]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const s16x8 a = cat_bf(*(const s16x4*)&w1[(16 * mb + l15) * WSB + 4 * g], *(const s16x4*)&w1[(16 * mb + l15) * WSB + 16 + 4 * g]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) xh[mb][nb] = mfma_bf32(a, b[nb], xh[mb][nb]);
        }
    }
    {
        s16x4 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = *(const s16x4*)&ft[(16 * nb + l15) * FSB + 32 + 4 * g];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const s16x4 a = *(const s16x4*)&w1[(16 * mb + l15) * WSB + 32 + 4 * g];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) xh[mb][nb] = mfma_bf(a, b[nb], xh[mb][nb]);
        }
    }
    EQD_TR(6);
    // ---- LeakyReLU + LayerNorm statistics (fp32, two-pass like torch) ------------------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = xh[mb][nb][r];
                if (z > 0.f) S.zpos |= 1u << (16 * nb + 4 * mb + r);
                float v = lrelu(z, P.slope);
                if constexpr (DROP) v *= ((S.keepz >> (16 * nb + 4 * mb + r)) & 1u) ? P.drop_scale : 0.f;
                xh[mb][nb][r] = v;
                s += v;
            }
        const float mean = group_sum(s) * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = xh[mb][nb][r] - mean;
                q += d * d;
            }
        const float rstd = 1.f / sqrtf(group_sum(q) * (1.f / 64.f) + P.ln_eps);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) xh[mb][nb][r] = (xh[mb][nb][r] - mean) * rstd;
        S.mean[nb] = mean;
        S.rstd[nb] = rstd;
    }
    }      // (!SAVED)
    if constexpr (!SAVED) {
        if (P.xh_save) edge_state_store<NB>(P, S, xh, l15, g);
    }
    EQD_TR(7);
    // ---- stage 2: m = W2 bf16(xh * gamma + beta) + b2;  stage 3: ch = Wc1 bf16(m) + bc1 -----------------------------
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const float4 b = *(const float4*)&vec[VEC_B2 + 16 * mb + 4 * g];
        const float4 c = *(const float4*)&vec[VEC_BC1 + 16 * mb + 4 * g];
        f32x4 v = {b.x, b.y, b.z, b.w}, u = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            m[mb][nb] = v;
            ch[mb][nb] = u;
        }
    }
    chain64_bf<true, NB>(m, xh, w2, &vec[VEC_LNG], &vec[VEC_LNB], l15, g);
    EQD_TR(8);
    chain64_bf<false, NB>(ch, m, wc1, nullptr, nullptr, l15, g);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 w = *(const float4*)&vec[VEC_WC2 + 16 * mb + 4 * g];
            if constexpr (DROP) {
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s += lrelu(ch[mb][nb][r], P.slope) *
                         (((S.keepc >> (16 * nb + 4 * mb + r)) & 1u) ? P.drop_scale : 0.f) * wv[r];
            } else {
                s += lrelu(ch[mb][nb][0], P.slope) * w.x + lrelu(ch[mb][nb][1], P.slope) * w.y +
                     lrelu(ch[mb][nb][2], P.slope) * w.z + lrelu(ch[mb][nb][3], P.slope) * w.w;
            }
        }
        S.coef[nb] = group_sum(s) + vec[VEC_BC2];
    }
    EQD_TR(9);
}

// store an F-layout tile to HBM as [edge][64]
template <int NB>
__device__ __forceinline__ void hbm_store(float* __restrict__ dst, const f32x4 (&v)[4][NB], const EdgeTileState<NB>& S,
                                          int l15, int g) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        if (S.ev[nb]) {
            float* row = dst + (size_t)(S.e0 + 16 * nb + l15) * 64;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                *(float4*)&row[16 * mb + 4 * g] = make_float4(v[mb][nb][0], v[mb][nb][1], v[mb][nb][2], v[mb][nb][3]);
        }
}

template <int NW, int TF, bool BF>
struct EdgeSmemSel {
    typedef EdgeSmem<NW, TF> type;
};
template <int NW, int TF>
struct EdgeSmemSel<NW, TF, true> {
    typedef EdgeSmemBf<NW, TF> type;
};

// ---------------------------------------------------------------------------------------------
// forward: node-aligned 32-edge tiles
// ---------------------------------------------------------------------------------------------
template <int NW, bool BF>
struct EdgeFwdSmem {
    typename EdgeSmemSel<NW, 32 * TS, BF>::type sm;
    float sxw[NW][96];      // per-node mean of x_rel * coef of the wave's tile
};
// forward of the tiles blk * NW + wave, + nblk * NW, ... (blk of nblk workgroups of NW waves)
template <int NW, bool BF, bool DROP = false>
__device__ __forceinline__ void edge_fwd_body(EdgeFwdSmem<NW, BF>& S_, const EqdGraph& G, const EqdEdgeParams& P, int blk,
                                              int nblk, const float* __restrict__ Pn, const float* __restrict__ Qn,
                                              const float* __restrict__ x, float* __restrict__ aggr_msg,
                                              float* __restrict__ x_new) {
    auto& sm = S_.sm;
    float (*sxw)[96] = S_.sxw;
    EQD_TR_WG();
    EQD_TR(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    float* tile = sm.tile[wave];
    // A tile's loads hang off a chain of three index fetches (tile -> nodes -> edge range -> endpoints).  That chain is
    // walked ONE TILE AHEAD, one link per phase of the current tile, so that a tile starts with everything it needs to issue
    // its gathers at once (k_edge_bwd does the same with its one link).  bf16 kernels only: k_edge_fwd 69.2 -> 67.4 us at
    // 64 x (300, 300); the fp32 kernel (12 more live registers on 226) measured 131.9 -> 133.7 us and keeps the plain chain.
    constexpr bool AHEAD = BF;
    int nx_n0 = 0, nx_n1 = 0, nx_e0 = 0, nx_ne = 0, nx_src[2] = {0, 0}, nx_dst[2] = {0, 0};
    auto link_nodes = [&](int t_) {
        nx_n0 = G.tile_node[t_];
        nx_n1 = G.tile_node[t_ + 1];
    };
    auto link_range = [&]() {
        nx_e0 = G.rowptr[nx_n0];
        nx_ne = G.rowptr[nx_n1] - nx_e0;
    };
    auto link_endpoints = [&]() {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int el = 16 * nb + l15;
            const int ei = nx_e0 + (el < nx_ne ? el : 0);
            nx_src[nb] = G.src[ei];
            nx_dst[nb] = G.dst[ei];
        }
    };
    // The FIRST tile's chain is walked before the weights are staged (both forms: at DB5.5 sizes a wave has one tile, and
    // the staging - 2 400 clocks - hides the two or three round trips).
    const int t_first = blk * NW + wave, t_step = nblk * NW;
    if (t_first < G.n_tiles) {
        link_nodes(t_first);
        link_range();
        if constexpr (AHEAD) link_endpoints();
    }
    if constexpr (BF)
        edge_stage_weights_bf<NW, false>(sm, P);
    else
        edge_stage_weights(sm, P);
    EQD_TR(1);
    for (int t = t_first; t < G.n_tiles; t += t_step) {
        EdgeTileState<2> S;
        if constexpr (AHEAD) {
            S.n0 = nx_n0;
            S.n1 = nx_n1;
            S.e0 = nx_e0;
            S.ne = nx_ne;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                S.src[nb] = nx_src[nb];
                S.dst[nb] = nx_dst[nb];
            }
        } else if (t == t_first) {      // (wave-uniform)
            S.n0 = nx_n0;
            S.n1 = nx_n1;
            S.e0 = nx_e0;
            S.ne = nx_ne;
        } else {
            S.n0 = G.tile_node[t];
            S.n1 = G.tile_node[t + 1];
            S.e0 = G.rowptr[S.n0];
            S.ne = G.rowptr[S.n1] - S.e0;
        }
        const bool more = AHEAD && t + t_step < G.n_tiles;      // (wave-uniform)
        if (more) link_nodes(t + t_step);
        f32x4 xh[4][2], m[4][2], ch[4][2];
        // row pointers (one per lane) and coordinates (3 nn <= 96 contiguous floats) of the tile's nodes: fetched
        // now, used by the aggregation at the end
        const int nn_pre = S.n1 - S.n0;
        const int rp = G.rowptr[S.n0 + (lane <= nn_pre ? lane : 0)] - S.e0;
        const size_t xo = (size_t)S.n0 * 3;
        const int c0 = lane, c1 = lane + 64;
        const float x0a = G.x0[xo + (c0 < 3 * nn_pre ? c0 : 0)], xa = x[xo + (c0 < 3 * nn_pre ? c0 : 0)];
        const float x0b = G.x0[xo + (c1 < 3 * nn_pre ? c1 : 0)], xb = x[xo + (c1 < 3 * nn_pre ? c1 : 0)];
        if constexpr (BF)
            edge_tile_forward_bf<2, DROP, true>(G, P, sm.w1, sm.w2, sm.wc1, sm.vec, tile, Pn, Qn, x, lane, S, xh, m, ch);
        else
            edge_tile_forward<2, DROP, false>(G, P, sm.w1, sm.w2, sm.wc1, sm.vec, tile, Pn, Qn, x, lane, S, xh, m, ch, nullptr);
        if (more) link_range();
        wave_lds_fence();   // feature tile is dead: reuse as the message tile [edge][64 + x_moment]
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                *(float4*)&tile[(16 * nb + l15) * TS + 16 * mb + 4 * g] =
                    make_float4(m[mb][nb][0], m[mb][nb][1], m[mb][nb][2], m[mb][nb][3]);
        if (g == 0) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                for (int c = 0; c < 3; ++c) tile[(16 * nb + l15) * TS + 64 + c] = S.xrel[nb][c] * S.coef[nb];
                tile[(16 * nb + l15) * TS + 67] = 0.f;
            }
        }
        wave_lds_fence();
        EQD_TR(10);
        // per-node means as a small MFMA product: out[node][f] = sum_e A[node][e] tile[e][f] with the 0/1 membership
        // matrix A built in registers from the tile's row pointers (node l15 of the block, edge 4 ks + g), then
        // scaled by 1 / degree.  Column block 4 carries the x_rel * coef moments (columns 64..66).
        const int nn = S.n1 - S.n0;   // <= 32 nodes per tile
        const int degl = __shfl(rp, (lane + 1) & 63) - rp;    // lane = node
        const float invl = (lane < nn && degl > 0) ? 1.f / (float)degl : 0.f;
        for (int nb0 = 0; nb0 < nn; nb0 += 16) {
            const int node = nb0 + l15;
            const int lo = __shfl(rp, node), hi = __shfl(rp, node + 1);
            const bool nv = node < nn;
            f32x4 acc[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[j] = f4zero();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int e = 4 * ks + g;
                const float am = (nv && e >= lo && e < hi) ? 1.f : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma4(am, tile[e * TS + 16 * j + l15], acc[j]);
                acc[4] = mfma4(am, tile[e * TS + 64 + (l15 & 3)], acc[4]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nd = nb0 + 4 * g + r;                 // output row of this lane
                const float inv = __shfl(invl, nd);
                if (nd < nn) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float vv = acc[j][r] * inv;
                        aggr_msg[(size_t)(S.n0 + nd) * 64 + 16 * j + l15] = vv;
                        if constexpr (BF) {      // the saved bf16 copy of the bf16 storage mode (EqdEdgeParams.aggr_bf16)
                            if (P.aggr_bf16) P.aggr_bf16[(size_t)(S.n0 + nd) * 64 + 16 * j + l15] = f2bf(vv);
                        }
                    }
                    if (l15 < 3) sxw[wave][3 * nd + l15] = acc[4][r] * inv;
                }
            }
        }
        wave_lds_fence();
        if (more) link_endpoints();
        if (c0 < 3 * nn) x_new[xo + c0] = P.eta * x0a + (1.f - P.eta) * xa + sxw[wave][c0];
        if (c1 < 3 * nn) x_new[xo + c1] = P.eta * x0b + (1.f - P.eta) * xb + sxw[wave][c1];
        wave_lds_fence();
        EQD_TR(11);
    }
    EQD_TR_WG_END();
}
template <int NW, bool BF, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void k_edge_fwd(EqdGraph G, EqdEdgeParams P, const float* __restrict__ Pn,
                                                              const float* __restrict__ Qn,
                                                              const float* __restrict__ x,
                                                              float* __restrict__ aggr_msg,
                                                              float* __restrict__ x_new) {
    __shared__ EdgeFwdSmem<NW, BF> S;
    edge_fwd_body<NW, BF, DROP>(S, G, P, (int)blockIdx.x, (int)gridDim.x, Pn, Qn, x, aggr_msg, x_new);
}
// the three arrays of the saved per-edge state come together or not at all
static int edge_saved_check(const EqdEdgeParams* p, const char* who) {
    const int n = (p->xh_save != nullptr) + (p->rstd_save != nullptr) + (p->zpos_save != nullptr);
    if (n != 0 && n != 3) {
        eqd_set_error("%s: xh_save, rstd_save and zpos_save must all be given or all be NULL", who);
        return EQD_ERR_NULL;
    }
    return EQD_OK;
}
// NULL-ness of the two dropout masks must agree; returns 1 when dropout is active
static int edge_drop_mode(const EqdEdgeParams* p, const char* who) {
    if ((p->drop_z1 == nullptr) != (p->drop_ch == nullptr) || (p->drop_z1 && !(p->drop_scale > 0.f))) {
        eqd_set_error("%s: dropout needs both masks (drop_z1, drop_ch) and drop_scale > 0", who);
        return -1;
    }
    return p->drop_z1 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Small batches (DB5.5-sized: 8 pairs x 200 + 200 residues): the edge-message forward needs 134 workgroups and the
// cross-attention forward 112 - each leaves half of the 256 CUs idle, and the two are independent (both read the
// layer's node projections, neither reads the other's output).  ONE launch runs them side by side: workgroups
// [0, n_edge) are edge-forward workgroups (8 waves, one 32-edge tile per wave), workgroups [n_edge, n_edge + n_items)
// each run one 32-row attention work item as two 4-wave groups (its two 16-row halves: same partner range, hence the
// same number of barriers).  LDS is a union of the two layouts (the attention side is the bigger one: 2 x 79 KB), so
// one workgroup per CU; taken when n_edge + n_items <= CUs, i.e. when everything is resident at once and the launch
// lasts as long as its longer half (23 us at config B instead of 23 + 9.5).  Same arithmetic as the separate launches.
// ---------------------------------------------------------------------------------------------
union alignas(16) EdgeAttnFwdSmem {
    EdgeFwdSmem<FWD_WAVES, false> edge;
    AttnFwdSmem<4> att[2];
    __device__ EdgeAttnFwdSmem() {}
};
template <bool DROP>
__global__ __launch_bounds__(64 * FWD_WAVES) void k_edge_attn_fwd(EqdGraph G, EqdEdgeParams P, int n_edge,
                                                                   const float* __restrict__ Pn,
                                                                   const float* __restrict__ Qn,
                                                                   const float* __restrict__ x,
                                                                   float* __restrict__ aggr_msg,
                                                                   float* __restrict__ x_new,
                                                                   const float* __restrict__ q,
                                                                   const float* __restrict__ k,
                                                                   const float* __restrict__ v,
                                                                   float* __restrict__ att_out,
                                                                   float* __restrict__ lse) {
    __shared__ EdgeAttnFwdSmem S;
    if ((int)blockIdx.x < n_edge) {
        edge_fwd_body<FWD_WAVES, false, DROP>(S.edge, G, P, (int)blockIdx.x, n_edge, Pn, Qn, x, aggr_msg, x_new);
    } else {
        const int half = (int)threadIdx.x >> 8;
        attn_fwd_body<4, true, 1>(S.att[half], G, (int)blockIdx.x - n_edge, half, (int)threadIdx.x & 255, 64, q, k, v,
                                  att_out, lse, true);
    }
}

// The same for the 69-wide FIRST layer (attention rows zero-padded to 80): one 32-row attention item per workgroup on its
// first four waves (the other four exit at once; two 16-row halves of the 80-wide forward would need 2 x 98 KB of LDS), the
// edge tiles as above.  A row's keys are split over the waves and merged in the same order whatever the block height: the
// same bits as the separate launches.  At 8 x (200, 200): 134 + 112 workgroups on 256 CUs, the attention (12 us on its
// own) runs beside the 24 us edge forward instead of in front of it.
union alignas(16) EdgeAttnFwdSmem80 {
    EdgeFwdSmem<FWD_WAVES, false> edge;
    AttnFwdSmem<5> att;
    __device__ EdgeAttnFwdSmem80() {}
};
template <bool DROP>
__global__ __launch_bounds__(64 * FWD_WAVES) void k_edge_attn_fwd80(EqdGraph G, EqdEdgeParams P, int n_edge,
                                                                     const float* __restrict__ Pn,
                                                                     const float* __restrict__ Qn,
                                                                     const float* __restrict__ x,
                                                                     float* __restrict__ aggr_msg,
                                                                     float* __restrict__ x_new,
                                                                     const float* __restrict__ q,
                                                                     const float* __restrict__ k,
                                                                     const float* __restrict__ v,
                                                                     float* __restrict__ att_out,
                                                                     float* __restrict__ lse) {
    __shared__ EdgeAttnFwdSmem80 S;
    if ((int)blockIdx.x < n_edge) {
        edge_fwd_body<FWD_WAVES, false, DROP>(S.edge, G, P, (int)blockIdx.x, n_edge, Pn, Qn, x, aggr_msg, x_new);
    } else {
        if (threadIdx.x >= 256) return;      // (a finished wave does not take part in the others' barriers)
        attn_fwd_body<5, true, 2>(S.att, G, (int)blockIdx.x - n_edge, 0, (int)threadIdx.x, 80, q, k, v, att_out, lse, false);
    }
}

// Smallest grid with the minimal number of tile rounds: with `wg_per_cu` workgroups per CU the makespan is
// ceil(tiles / (CUs * wg_per_cu * waves)) tile times whatever the grid, so use as few CUs as that allows
// and leave the rest to the kernels that run concurrently on the auxiliary streams.
static int edge_grid(int n_tiles, int waves, int wg_per_cu) {
    if (n_tiles <= 0) return 1;
    const int cus = eqd_num_cus();
    const int slots = cus * waves * wg_per_cu;
    const int rounds = (n_tiles + slots - 1) / slots;
    int blocks = (n_tiles + waves * rounds - 1) / (waves * rounds);
    if (blocks > cus * wg_per_cu) blocks = cus * wg_per_cu;
    return blocks < 1 ? 1 : blocks;
}

extern "C" int eqd_edge_message_fwd(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q,
                                    const float* x, float* aggr_msg, float* x_new, void* stream) {
    if (!g || !p || !P || !Q || !x || !aggr_msg || !x_new) {
        eqd_set_error("eqd_edge_message_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (g->n_tiles <= 0) return EQD_OK;
    // Smallest grid with the minimal number of tile rounds (edge_grid): at config B that is 134 of the 256 CUs, which
    // leaves room for the attention kernel that runs beside this one on the auxiliary stream.  (Dealing the 1067
    // tiles over all 256 CUs measured no faster - 4 % of the SIMDs still get two tiles - and serialised the two.)
    const int blocks = edge_grid(g->n_tiles, FWD_WAVES, 1);
    const int drop = edge_drop_mode(p, "eqd_edge_message_fwd");
    if (drop < 0) return EQD_ERR_NULL;
    if (int rc = edge_saved_check(p, "eqd_edge_message_fwd")) return rc;
    if (p->bf16 && p->use_he && !g->he_bf16) {
        eqd_set_error("eqd_edge_message_fwd: bf16 mode needs EqdGraph.he_bf16");
        return EQD_ERR_NULL;
    }
#define EQD_EDGE_FWD_LAUNCH(BF_, DROP_)                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_fwd<FWD_WAVES, BF_, DROP_>), dim3(blocks), dim3(64 * FWD_WAVES), 0,         \
                       (hipStream_t)stream, *g, *p, P, Q, x, aggr_msg, x_new)
    if (p->bf16) {
        if (drop) EQD_EDGE_FWD_LAUNCH(true, true); else EQD_EDGE_FWD_LAUNCH(true, false);
    } else {
        if (drop) EQD_EDGE_FWD_LAUNCH(false, true); else EQD_EDGE_FWD_LAUNCH(false, false);
    }
#undef EQD_EDGE_FWD_LAUNCH
    return eqd_check_launch("k_edge_fwd");
}

// 1 if eqd_edge_attn_fwd will take the fused launch for this graph / width / mode (else it issues the two launches)
int eqd_edge_attn_fused(const EqdGraph* g, const EqdEdgeParams* p, int d_att, const float* q, const float* k, const float* v) {
    const char* f = eqd_tunable("EQD_FUSE_FWD");
    if (f && f[0] == '0' && f[1] == 0) return 0;
    if (p->bf16 || (d_att != 64 && d_att != 80) || g->n_tiles <= 0 || g->n_att_items <= 0) return 0;
    if (d_att == 80) {      // the first layer's form (round 5); EQD_FUSE_FWD80=0 keeps the two launches
        const char* f8 = eqd_tunable("EQD_FUSE_FWD80");
        if (f8 && f8[0] == '0' && f8[1] == 0) return 0;
    }
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) != 0) return 0;
    const int n_edge = edge_grid(g->n_tiles, FWD_WAVES, 1);
    return n_edge + g->n_att_items <= eqd_num_cus();
}
// edge-message forward + cross-attention forward of one layer (independent of each other): one launch when both fit
// the chip at once (eqd_edge_attn_fused), the two separate launches otherwise
int eqd_edge_attn_fwd(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q, const float* x,
                      float* aggr_msg, float* x_new, int d_att, const float* q, const float* k, const float* v,
                      float* att_out, float* lse, hipStream_t st) {
    if (!eqd_edge_attn_fused(g, p, d_att, q, k, v)) {
        int rc = eqd_cross_attention_fwd(g, d_att, q, k, v, att_out, lse, st);
        if (rc) return rc;
        return eqd_edge_message_fwd(g, p, P, Q, x, aggr_msg, x_new, st);
    }
    const int n_edge = edge_grid(g->n_tiles, FWD_WAVES, 1);
    const int drop = edge_drop_mode(p, "eqd_edge_attn_fwd");
    if (drop < 0) return EQD_ERR_NULL;
    if (int rc = edge_saved_check(p, "eqd_edge_attn_fwd")) return rc;
    if (d_att == 80) {
        if (drop)
            hipLaunchKernelGGL(k_edge_attn_fwd80<true>, dim3(n_edge + g->n_att_items), dim3(64 * FWD_WAVES), 0, st, *g, *p, n_edge,
                               P, Q, x, aggr_msg, x_new, q, k, v, att_out, lse);
        else
            hipLaunchKernelGGL(k_edge_attn_fwd80<false>, dim3(n_edge + g->n_att_items), dim3(64 * FWD_WAVES), 0, st, *g, *p, n_edge,
                               P, Q, x, aggr_msg, x_new, q, k, v, att_out, lse);
        return eqd_check_launch("k_edge_attn_fwd");
    }
    if (drop)
        hipLaunchKernelGGL(k_edge_attn_fwd<true>, dim3(n_edge + g->n_att_items), dim3(64 * FWD_WAVES), 0, st, *g, *p, n_edge, P,
                           Q, x, aggr_msg, x_new, q, k, v, att_out, lse);
    else
        hipLaunchKernelGGL(k_edge_attn_fwd<false>, dim3(n_edge + g->n_att_items), dim3(64 * FWD_WAVES), 0, st, *g, *p, n_edge, P,
                           Q, x, aggr_msg, x_new, q, k, v, att_out, lse);
    return eqd_check_launch("k_edge_attn_fwd");
}

// ---------------------------------------------------------------------------------------------
// backward: 16-edge tiles (not node aligned), weight-gradient GEMMs fused
// ---------------------------------------------------------------------------------------------
// Per workgroup iteration the 8 waves own 8 x 16 = 128 consecutive edges.  Each wave recomputes its tile
// forward and runs the data-gradient chain in registers; three times per iteration the waves drop a
// pair of per-edge operands (d_chid | m), (dm | a1), (dz1 | [he rbf]) as [128][.] slabs into two shared
// LDS buffers and every wave accumulates its share of the 16x16 blocks of
//      dWc1 = d_chid^T m,   dW2 = dm^T a1,   dW1[:, 2d:] = dz1^T [he rbf]
// over the 128 edges (K axis of the MFMA = edges).  The per-edge operands therefore never reach HBM
// (the unfused version wrote and re-read 1.36 KB/edge); only dz1 (for the by-source gather dP[src]) and
// dx_rel are written: 272 B/edge.  Weight-gradient blocks are kept in registers across iterations and
// written once per workgroup; bias / LayerNorm / coordinate-MLP vector gradients go through per-wave
// LDS accumulators.  Both partial sets are summed in a fixed order by k_reduce_segments.
#define VP 384         /* floats per wave in the vector partial buffer */
#define V_DLNG 0
#define V_DLNB 64
#define V_DWC2 128
#define V_DBC2 192
#define V_DB2 256
#define V_DBC1 320
#define VA 384         /* floats per wave actually used */
#define WP_W2 0        /* offsets inside a workgroup's weight-gradient partial */
#define WP_WC1 4096
#define WP_W1 8192     /* [64][48], 42 columns used */
#define WP_N 11264

struct EdgeBwdWs {
    float* dz1;    // [E][64] grad wrt edge_mlp.0 output (read by k_node_gather)
    float* dxrel;  // [E][4]
    float* vecp;   // [nblocks * 8][VP] vector-gradient partials per wave
    float* wpart;  // [nblocks][WP_N]  weight-gradient partials per workgroup
};

// The two operand slabs of a weight-gradient GEMM (K axis = the super-tile's 128 edges) are stored TRANSPOSED,
// [feature][edge] with row stride UST: an MFMA operand of lane (l15, g) - 4 consecutive edges of one feature, used as the
// k-values of 4 successive MFMAs (any assignment of edges to k-steps is a valid contraction as long as A and B share
// it) - is then ONE b128 read.  With [edge][feature] slabs and one b32 read per operand and k-step a slab GEMM was 96 LDS
// reads for 64 MFMAs per wave and ran at 46 % of the MFMA rate (8 960 clocks per GEMM, three per super-tile = 37 % of
// the kernel); now it is 24 b128 reads.  UST = 132 floats = 33 x 16 B: the 16 lanes of a b128 read phase (l15 = 0..15)
// hit 16 different 16-byte bank groups, and the scalar stores below (16 g + l15 mod 64) all 64 banks.
#define UST 132
// F-layout tile (16 edges x 64 features) -> columns 16 w .. 16 w + 15 of a shared [64][UST] slab
__device__ __forceinline__ void slab_store(float* __restrict__ slab, int wave, const f32x4 (&v)[4][1], int l15, int g) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(16 * mb + 4 * g + r) * UST + 16 * wave + l15] = v[mb][0][r];
}
// acc[j] += X^T Y over the 128 edges of the slabs for output block (mb, nb0 + j)
template <int NJ>
__device__ __forceinline__ void slab_atb(f32x4 (&acc)[NJ], const float* __restrict__ Xt, const float* __restrict__ Yt,
                                         int mb, int nb0, int l15, int g) {
#pragma unroll 2
    for (int kc = 0; kc < 8; ++kc) {
        const f32x4 a = *(const f32x4*)&Xt[(16 * mb + l15) * UST + 16 * kc + 4 * g];
        f32x4 b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *(const f32x4*)&Yt[(16 * (nb0 + j) + l15) * UST + 16 * kc + 4 * g];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = mfma4(a[u], b[j][u], acc[j]);
    }
}

// acc += X^T F where F = the waves' [16][FS] feature tiles stacked (row = edge within the 128-edge super-tile)
__device__ __forceinline__ void feat_atb(f32x4 (&acc)[1], const float* __restrict__ Xt, const float* __restrict__ tiles,
                                         int mb, int nb, int l15, int g) {
    const int c = 16 * nb + l15;
    const bool ok = c < FS;
    const int cc = ok ? c : 0;
#pragma unroll 2
    for (int kc = 0; kc < 8; ++kc) {
        const f32x4 a = *(const f32x4*)&Xt[(16 * mb + l15) * UST + 16 * kc + 4 * g];
        float b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = tiles[kc * (16 * FS) + (4 * g + u) * FS + cc];     // edge 16 kc + 4 g + u
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[0] = mfma4(a[u], ok ? b[u] : 0.f, acc[0]);
    }
}

// bf16 mode: the slabs are stored ROW-MAJOR ([128 edges][64 features], bf16 - a lane's four consecutive features of its
// edge are ONE 8-byte store) and an MFMA operand - 4 consecutive edges (the K axis) of one feature - comes out of them
// through the transposing LDS read (lds_tr16, eqd_common.h; round 4).  Until then the slabs were stored transposed
// ([feature][128 edges]) with one 2-byte store per element: 111 ds_write_b16 per 16-edge tile, the largest group of LDS
// instructions of the kernel.  Same operand values in the same k-slots: bit-identical results.
#define USB 72         /* bf16 row stride of the row-major slabs: 144-byte rows keep every 4-feature group 8-byte aligned */
__device__ __forceinline__ void slab_store_bf(unsigned short* __restrict__ slab, int wave, const f32x4 (&v)[4][1], int l15,
                                              int g) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
        *(s16x4*)&slab[(16 * wave + l15) * USB + 16 * mb + 4 * g] = pack_bf4(v[mb][0][0], v[mb][0][1], v[mb][0][2], v[mb][0][3]);
}
// acc[j] += X[:, 16 mb ..]^T Y[:, 16 (nb0 + j) ..] over the slabs' 128 edges: lane (l15, g) points at features
// 16 mb + 4 (l15 & 3) .. + 3 of edge 4 g + (l15 >> 2) of a 16-edge chunk and receives edges 4 g .. 4 g + 3 of feature
// 16 mb + l15
// (FULL: all four chunks unrolled - 1.6 % faster per launch at 64 x (300, 300); the dropout instances, which carry the keep
//  masks as well, spill 9 registers to scratch that way and unroll by two instead: no scratch in any instance)
template <int NJ, bool FULL>
__device__ __forceinline__ void slab_atb_bf(f32x4 (&acc)[NJ], const unsigned short* __restrict__ Xr,
                                            const unsigned short* __restrict__ Yr, int mb, int nb0, int l15, int g) {
    const int lo = (4 * g + (l15 >> 2)) * USB + 4 * (l15 & 3);
    const unsigned short* const px = Xr + lo + 16 * mb;
    auto chunk = [&](int kp) {      // 32 edges: one v_mfma_f32_16x16x32_bf16 per output block
        const s16x8 a = cat_bf(lds_tr16(px + 32 * kp * USB), lds_tr16(px + (32 * kp + 16) * USB));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned short* const py = Yr + lo + 16 * (nb0 + j);
            acc[j] = mfma_bf32(a, cat_bf(lds_tr16(py + 32 * kp * USB), lds_tr16(py + (32 * kp + 16) * USB)), acc[j]);
        }
    };
    if constexpr (FULL) {
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) chunk(kp);
    } else {
#pragma unroll 2
        for (int kp = 0; kp < 4; ++kp) chunk(kp);
    }
}

template <bool BF>
struct EdgeBwdSmemSel {
    typedef EdgeSmem<BWD_WAVES, 16 * FS> type;
    typedef float slab_t;
    enum { SLAB = 64 * UST };
};
template <>
struct EdgeBwdSmemSel<true> {
    typedef EdgeSmemBfBwd<BWD_WAVES, 16 * FS> type;
    typedef unsigned short slab_t;
    enum { SLAB = 128 * USB };
};

template <bool BF, bool DROP = false, bool SAVED = false>
__global__ __launch_bounds__(64 * BWD_WAVES) void k_edge_bwd(EqdGraph G, EqdEdgeParams P, const float* __restrict__ Pn,
                                                              const float* __restrict__ Qn,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ d_aggr,
                                                              const float* __restrict__ d_xnew, EdgeBwdWs W) {
    typedef EdgeBwdSmemSel<BF> Sel;
    __shared__ typename Sel::type sm;
    // bf16: TWO slab pairs, used alternately by the three phases of an iteration (and on across iterations).  A phase's
    // stores then only have to wait for the readers of two phases back, and those are behind the barrier of the phase in
    // between - ONE barrier per phase (stores -> reads) instead of two, none at the end of the iteration (the phase-3 B
    // operand is a copy of the feature tiles in the slab): 7 -> 3 barriers per 128 edges (the kernel's waves are parked at
    // barriers / s_waitcnt 48 % of their cycles at 64 x (300, 300), profiles/r03_l_sq_C_bf16_pass1.json).  The fp32 slabs
    // (2 x 33.8 KB) have no room for a second pair.
    constexpr bool DBUF = BF;
    // knock-out build for pricing (-DEQD_EXP_NO_SLABS, profiles/exp_r06_knockouts.sh: results WRONG, timing only): the kernel
    // without its three weight-gradient slab GEMMs, their LDS stores and barriers
#ifdef EQD_EXP_NO_SLABS
    constexpr bool NOSLAB = true;
#else
    constexpr bool NOSLAB = false;
#endif
    __shared__ __attribute__((aligned(16))) typename Sel::slab_t Ubuf[(DBUF ? 2 : 1) * Sel::SLAB];
    __shared__ __attribute__((aligned(16))) typename Sel::slab_t Vbuf[(DBUF ? 2 : 1) * Sel::SLAB];
    int slab_par = 0;
    EQD_TR_WG();
    EQD_TR(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int n_tiles = (G.n_edges + 15) >> 4;
    const int n_super = (n_tiles + BWD_WAVES - 1) / BWD_WAVES;
    // The endpoints of a tile's edges are fetched ONE ITERATION AHEAD: everything else the tile loads (coordinates, P / Q
    // rows, the destination's incoming gradients) hangs off them, so an iteration starts with one round trip to memory
    // instead of two (lanes beyond the tile read the tile's first edge, an empty tile edge 0 - like the tile forward).
    // The first tile's are requested before the weights are staged (they land behind the staging barrier).
    int nx_src = 0, nx_dst = 0;
    auto endpoints = [&](int it_) {
        int e0 = 16 * (it_ * BWD_WAVES + wave), ne = G.n_edges - e0;
        ne = ne < 0 ? 0 : (ne > 16 ? 16 : ne);
        if (ne == 0) e0 = 0;
        const int ei = e0 + (l15 < ne ? l15 : 0);
        nx_src = G.src[ei];
        nx_dst = G.dst[ei];
    };
    if ((int)blockIdx.x < n_super) endpoints((int)blockIdx.x);
    if constexpr (BF)
        edge_stage_weights_bf<BWD_WAVES, true>(sm, P);
    else
        edge_stage_weights(sm, P);
    EQD_TR(1);
    float* tile = sm.tile[wave];
    // vector-gradient sums of this lane's feature fo = 16 (l15 >> 2) + 4 g + (l15 & 3) (see reduce16x16), kept in
    // registers over all tiles of the wave; r_dbc2 is a plain per-lane partial
    float r_dlng = 0.f, r_dlnb = 0.f, r_dwc2 = 0.f, r_dbc1 = 0.f, r_db2 = 0.f, r_dbc2 = 0.f;
    // this wave's blocks of the weight gradients: rows 16*wmb.., columns 16*(2*(wave&1)) + {0,16} of the 64x64
    // matrices; for dW1[:, 2d:] (3 column blocks) block (wmb, wave&1) and, for waves 0..3, block (wave, 2)
    const int wmb = wave >> 1, wnb = 2 * (wave & 1);
    f32x4 gW2[2] = {f4zero(), f4zero()}, gWc1[2] = {f4zero(), f4zero()}, gW1a[1] = {f4zero()}, gW1b[1] = {f4zero()};
    for (int it = blockIdx.x; it < n_super; it += gridDim.x) {
        const int t = it * BWD_WAVES + wave;
        EdgeTileState<1> S;
        S.n0 = S.n1 = 0;
        S.e0 = 16 * t;
        S.ne = G.n_edges - S.e0;
        S.ne = S.ne < 0 ? 0 : (S.ne > 16 ? 16 : S.ne);
        if (S.ne == 0) S.e0 = 0;
        S.src[0] = nx_src;
        S.dst[0] = nx_dst;
        if (it + (int)gridDim.x < n_super) endpoints(it + (int)gridDim.x);
        f32x4 xh[4][1], m[4][1], ch[4][1];
        // incoming gradients of the tile's destination nodes: fetched before the forward recompute (unpredicated;
        // lanes beyond the tile use the clamped edge e0), consumed after it
        float4 dag[4];
        float gxn[3];
        int r0, r1;
        {
            const int d = S.dst[0];
            r0 = G.rowptr[d];
            r1 = G.rowptr[d + 1];
#pragma unroll
            for (int c = 0; c < 3; ++c) gxn[c] = d_xnew[(size_t)d * 3 + c];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) dag[mb] = *(const float4*)&d_aggr[(size_t)d * 64 + 16 * mb + 4 * g];
        }
        if constexpr (BF)
            edge_tile_forward_bf<1, DROP, true, SAVED>(G, P, sm.w1, sm.w2, sm.wc1, sm.vec, tile, Pn, Qn, x, lane, S, xh, m, ch);
        else
            edge_tile_forward<1, DROP, true, SAVED>(G, P, sm.w1, sm.w2, sm.wc1, sm.vec, tile, Pn, Qn, x, lane, S, xh, m, ch, nullptr);
        // ---- coordinate path ---------------------------------------------------------------------
        float invdeg = 0.f, dcoef = 0.f, dxr[3] = {0.f, 0.f, 0.f};
        {
            invdeg = S.ev[0] ? 1.f / (float)(r1 - r0) : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gx = gxn[c] * invdeg;
                dcoef += gx * S.xrel[0][c];
                dxr[c] = gx * S.coef[0];
            }
        }
        // d wc2 / d bc1 partials, then ch := d_chid = wc2 * dcoef * LeakyReLU'(ch)
        {
            float va[16], vb[16];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 w = *(const float4*)&sm.vec[VEC_WC2 + 16 * mb + 4 * g];
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = ch[mb][0][r];
                    float keep = 1.f;       // dropout of coors_mlp.1: factor on the activation and on its derivative
                    if constexpr (DROP) keep = ((S.keepc >> (4 * mb + r)) & 1u) ? P.drop_scale : 0.f;
                    const float dch = wv[r] * dcoef * (lrelu_grad(c, P.slope) * keep);
                    va[4 * mb + r] = (lrelu(c, P.slope) * keep) * dcoef;
                    vb[4 * mb + r] = dch;
                    ch[mb][0][r] = dch;
                }
            }
            r_dwc2 += reduce16x16(va, l15);
            r_dbc1 += reduce16x16(vb, l15);
            r_dbc2 += g == 0 ? dcoef : 0.f;
        }
        EQD_TR(12);
        // ---- phase 1: dWc1 += d_chid^T m -------------------------------------------------------------
        typename Sel::slab_t* U = Ubuf + slab_par * Sel::SLAB;
        typename Sel::slab_t* V = Vbuf + slab_par * Sel::SLAB;
        if constexpr (DBUF) slab_par ^= 1;
        if constexpr (!NOSLAB) {
        if constexpr (BF) {
            slab_store_bf(U, wave, ch, l15, g);
            slab_store_bf(V, wave, m, l15, g);
        } else {
            slab_store(U, wave, ch, l15, g);
            slab_store(V, wave, m, l15, g);
        }
        __syncthreads();
        EQD_TR(13);
        if constexpr (BF)
            slab_atb_bf<2, !DROP>(gWc1, U, V, wmb, wnb, l15, g);
        else
            slab_atb<2>(gWc1, U, V, wmb, wnb, l15, g);
        }
        EQD_TR(14);
        // ---- dm = d_aggr[dst] / deg + Wc1^T d_chid ---------------------------------------------------
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 v = dag[mb];     // invdeg is 0 for lanes beyond the tile
            f32x4 a;
            a[0] = v.x * invdeg; a[1] = v.y * invdeg; a[2] = v.z * invdeg; a[3] = v.w * invdeg;
            m[mb][0] = a;
        }
        if constexpr (BF)
            chain64_bf<false, 1>(m, ch, sm.wc1T, nullptr, nullptr, l15, g);
        else
            chain64T<1>(m, ch, sm.wc1, l15, g);
        {
            float va[16];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) va[4 * mb + r] = m[mb][0][r];
            r_db2 += reduce16x16(va, l15);
        }
        EQD_TR(15);
        // ---- phase 2: dW2 += dm^T a1 ---------------------------------------------------------------------
        if constexpr (!DBUF && !NOSLAB) __syncthreads();      // every wave is done reading the phase-1 slabs
        U = Ubuf + slab_par * Sel::SLAB;
        V = Vbuf + slab_par * Sel::SLAB;
        if constexpr (DBUF) slab_par ^= 1;
        if constexpr (!NOSLAB) {
        if constexpr (BF)
            slab_store_bf(U, wave, m, l15, g);
        else
            slab_store(U, wave, m, l15, g);
        }
        if constexpr (!NOSLAB) {
            f32x4 a1[4][1];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 gg = *(const float4*)&sm.vec[VEC_LNG + 16 * mb + 4 * g];
                const float4 bb = *(const float4*)&sm.vec[VEC_LNB + 16 * mb + 4 * g];
                a1[mb][0][0] = xh[mb][0][0] * gg.x + bb.x;
                a1[mb][0][1] = xh[mb][0][1] * gg.y + bb.y;
                a1[mb][0][2] = xh[mb][0][2] * gg.z + bb.z;
                a1[mb][0][3] = xh[mb][0][3] * gg.w + bb.w;
            }
            if constexpr (BF)
                slab_store_bf(V, wave, a1, l15, g);
            else
                slab_store(V, wave, a1, l15, g);
        }
        if constexpr (!NOSLAB) {
        __syncthreads();
        EQD_TR(16);
        if constexpr (BF)
            slab_atb_bf<2, !DROP>(gW2, U, V, wmb, wnb, l15, g);
        else
            slab_atb<2>(gW2, U, V, wmb, wnb, l15, g);
        }
        EQD_TR(17);
        // ---- da1 = W2^T dm ---------------------------------------------------------------------------
        f32x4 dz[4][1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) dz[mb][0] = f4zero();
        if constexpr (BF)
            chain64_bf<false, 1>(dz, m, sm.w2T, nullptr, nullptr, l15, g);
        else
            chain64T<1>(dz, m, sm.w2, l15, g);
        EQD_TR(18);
        // ---- LayerNorm + LeakyReLU backward --------------------------------------------------------------
        {
            float va[16], vb[16];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    va[4 * mb + r] = dz[mb][0][r] * xh[mb][0][r];
                    vb[4 * mb + r] = dz[mb][0][r];
                }
            r_dlng += reduce16x16(va, l15);
            r_dlnb += reduce16x16(vb, l15);
        }
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 gg = *(const float4*)&sm.vec[VEC_LNG + 16 * mb + 4 * g];
                const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dxh = dz[mb][0][r] * gv[r];
                    dz[mb][0][r] = dxh;
                    s1 += dxh;
                    s2 += dxh * xh[mb][0][r];
                }
            }
            s1 = group_sum(s1) * (1.f / 64.f);
            s2 = group_sum(s2) * (1.f / 64.f);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xhv = xh[mb][0][r];
                    float lg = ((S.zpos >> (4 * mb + r)) & 1u) ? 1.f : P.slope;
                    if constexpr (DROP) lg *= ((S.keepz >> (4 * mb + r)) & 1u) ? P.drop_scale : 0.f;
                    const float v = S.ev[0] ? S.rstd[0] * (dz[mb][0][r] - s1 - xhv * s2) * lg : 0.f;
                    dz[mb][0][r] = v;
                }
        }
        EQD_TR(19);
        if constexpr (BF) {
            // bf16 mode: dz1 goes to HBM as bf16 rows (128 B per edge instead of 256): it is the largest stream of the
            // backward (98 MB per launch at 64 x (300, 300)), written here and gathered twice by k_node_gather
            if (S.ev[0]) {
                unsigned short* row = (unsigned short*)W.dz1 + (size_t)(S.e0 + l15) * 64;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    *(s16x4*)&row[16 * mb + 4 * g] = pack_bf4(dz[mb][0][0], dz[mb][0][1], dz[mb][0][2], dz[mb][0][3]);
            }
        } else {
            hbm_store<1>(W.dz1, dz, S, l15, g);
        }
        // ---- phase 3: dW1[:, 2d:] += dz1^T [he | rbf] ------------------------------------------------------
        if constexpr (!DBUF && !NOSLAB) __syncthreads();      // phase-2 slabs are free
        U = Ubuf + slab_par * Sel::SLAB;
        V = Vbuf + slab_par * Sel::SLAB;
        if constexpr (DBUF) slab_par ^= 1;
        if constexpr (NOSLAB) {
        } else if constexpr (BF) {
            slab_store_bf(U, wave, dz, l15, g);
            // this wave's [16][48] feature tile is rows 16 w .. of V (a copy: the tile is overwritten by the next iteration's
            // recompute while other waves may still be in this phase)
            const unsigned short* __restrict__ ft = (const unsigned short*)tile;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                *(s16x4*)&V[(16 * wave + l15) * USB + 12 * g + 4 * i] = *(const s16x4*)&ft[l15 * FSB + 12 * g + 4 * i];
            __syncthreads();
            slab_atb_bf<1, !DROP>(gW1a, U, V, wmb, wave & 1, l15, g);
            if (wave < 4) slab_atb_bf<1, !DROP>(gW1b, U, V, wave, 2, l15, g);
        } else {
            slab_store(U, wave, dz, l15, g);
            __syncthreads();
            feat_atb(gW1a, U, &sm.tile[0][0], wmb, wave & 1, l15, g);     // the feature tiles are the B operand as they lie
            if (wave < 4) feat_atb(gW1b, U, &sm.tile[0][0], wave, 2, l15, g);
        }
        EQD_TR(20);
        // ---- d rbf = W1d^T dz1 -> d(d^2) -> d x_rel --------------------------------------------------------
        if (P.use_dist) {
            f32x4 dr = f4zero();
            if constexpr (BF) {
#pragma unroll
                for (int mp = 0; mp < 2; ++mp)
                    dr = mfma_bf32(cat_bf(*(const s16x4*)&sm.w1rT[l15 * WSB + 32 * mp + 4 * g],
                                          *(const s16x4*)&sm.w1rT[l15 * WSB + 32 * mp + 16 + 4 * g]),
                                   cat_bf(pack_bf4(dz[2 * mp][0][0], dz[2 * mp][0][1], dz[2 * mp][0][2], dz[2 * mp][0][3]),
                                          pack_bf4(dz[2 * mp + 1][0][0], dz[2 * mp + 1][0][1], dz[2 * mp + 1][0][2],
                                                   dz[2 * mp + 1][0][3])), dr);
            } else {
#pragma unroll
                for (int mbi = 0; mbi < 4; ++mbi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float w = (l15 < 15) ? sm.w1[(16 * mbi + 4 * g + r) * WS1 + 27 + l15] : 0.f;
                        dr = mfma4(w, dz[mbi][0][r], dr);
                    }
            }
            // d rbf_k / d(d^2) = rbf_k * (-1 / sigma_k): the recompute above evaluated exp(-d^2 / sigma_k) for exactly this
            // lane's k = 4 g + r (fp32: still in the feature tile; bf16: S.rbf) - until round 5 both the exponential and two
            // divisions per k were evaluated again here (12 of the kernel's 14 fp32 divisions).  Same values, same products.
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g + r;
                if (k < 15) {
                    float e;
                    if constexpr (BF) e = S.rbf[0][r];
                    else e = S.ev[0] ? tile[l15 * FS + 27 + k] : 1.f;      // (lanes beyond the tile: exp(-0) as before)
                    s += dr[r] * e * rbf_neg_inv_sigma(k);
                }
            }
            const float dd2 = group_sum(s);
#pragma unroll
            for (int c = 0; c < 3; ++c) dxr[c] += 2.f * S.xrel[0][c] * dd2;
        }
        if (g == 0 && S.ev[0]) {
            float* o = W.dxrel + (size_t)(S.e0 + l15) * 4;
            o[0] = dxr[0]; o[1] = dxr[1]; o[2] = dxr[2]; o[3] = 0.f;
        }
        if constexpr (!DBUF && !NOSLAB) __syncthreads();      // phase-3 slabs and the feature tiles are free for the next iteration
        EQD_TR(21);
    }
    // ---- partials: the workgroup's vector sums (its 8 waves' sums added in wave order through the free U slab: one
    //      partial row per workgroup, not per wave, for the later fixed-order reduction) and weight-gradient blocks ----
    {
        if constexpr (DBUF) __syncthreads();    // (no barrier at the end of an iteration in this form)
        float* const U = (float*)Ubuf;
        float* vs = U + wave * VP;              // the slabs are free after the loop's last barrier
        const int fo = 16 * (l15 >> 2) + 4 * g + (l15 & 3);
        vs[V_DLNG + fo] = r_dlng;
        vs[V_DLNB + fo] = r_dlnb;
        vs[V_DWC2 + fo] = r_dwc2;
        vs[V_DB2 + fo] = r_db2;
        vs[V_DBC1 + fo] = r_dbc1;
        const float t = wave_sum(r_dbc2);
        if (lane == 0) vs[V_DBC2] = t;
        __syncthreads();
        float* vp = W.vecp + (size_t)blockIdx.x * VP;
        for (int i = threadIdx.x; i < VP; i += 64 * BWD_WAVES) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < BWD_WAVES; ++w) a += ((const float*)U)[w * VP + i];
            vp[i] = a;
        }
    }
    if (!W.wpart) return;      // (timing experiment only, eqd_edge_message_bwd_kernel_only with EQD_EXP_EDGE_NO_PARTIALS=1)
    float* wp = W.wpart + (size_t)blockIdx.x * WP_N;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wmb + 4 * g + r, col = 16 * (wnb + j) + l15;
            wp[WP_W2 + row * 64 + col] = gW2[j][r];
            wp[WP_WC1 + row * 64 + col] = gWc1[j][r];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wp[WP_W1 + (16 * wmb + 4 * g + r) * 48 + 16 * (wave & 1) + l15] = gW1a[0][r];
        if (wave < 4) wp[WP_W1 + (16 * wave + 4 * g + r) * 48 + 32 + l15] = gW1b[0][r];
    }
    EQD_TR(22);
    EQD_TR_WG_END();
}

// ---------------------------------------------------------------------------------------------
// Test / debug aid (eqd_model_lrelu_signs): which side of its two LeakyReLU kinks every edge took.  Runs the SAME
// per-edge forward recompute as k_edge_bwd (edge_tile_forward<1>: same operands, same MFMA order, hence the same bits)
// and writes, per edge and feature, one byte: 1 = pre-activation > 0 (derivative 1), 0 = derivative `slope`.
// z1_pos: edge_mlp.0's output (EdgeTileState::zpos, the mask the backward applies); ch_pos: coors_mlp.0's output.
// ---------------------------------------------------------------------------------------------
template <bool BF, bool DROP>
__global__ __launch_bounds__(64 * BWD_WAVES) void k_edge_signs(EqdGraph G, EqdEdgeParams P, const float* __restrict__ Pn,
                                                                const float* __restrict__ Qn,
                                                                const float* __restrict__ x,
                                                                unsigned char* __restrict__ z1_pos,
                                                                unsigned char* __restrict__ ch_pos) {
    __shared__ typename EdgeSmemSel<BWD_WAVES, 16 * FS, BF>::type sm;
    if constexpr (BF)
        edge_stage_weights_bf<BWD_WAVES, false>(sm, P);
    else
        edge_stage_weights(sm, P);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    float* tile = sm.tile[wave];
    const int n_tiles = (G.n_edges + 15) >> 4;
    for (int t = blockIdx.x * BWD_WAVES + wave; t < n_tiles; t += gridDim.x * BWD_WAVES) {
        EdgeTileState<1> S;
        S.n0 = S.n1 = 0;
        S.e0 = 16 * t;
        S.ne = G.n_edges - S.e0;
        S.ne = S.ne > 16 ? 16 : S.ne;
        f32x4 xh[4][1], m[4][1], ch[4][1];
        if constexpr (BF)
            edge_tile_forward_bf<1, DROP>(G, P, sm.w1, sm.w2, sm.wc1, sm.vec, tile, Pn, Qn, x, lane, S, xh, m, ch);
        else
            edge_tile_forward<1, DROP>(G, P, sm.w1, sm.w2, sm.wc1, sm.vec, tile, Pn, Qn, x, lane, S, xh, m, ch, nullptr);
        if (S.ev[0]) {
            const size_t row = (size_t)(S.e0 + l15) * 64;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z1_pos[row + 16 * mb + 4 * g + r] = (unsigned char)((S.zpos >> (4 * mb + r)) & 1u);
                    ch_pos[row + 16 * mb + 4 * g + r] = ch[mb][0][r] > 0.f ? 1 : 0;
                }
        }
        wave_lds_fence();
    }
}

int eqd_launch_edge_signs(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q, const float* x,
                          unsigned char* z1_pos, unsigned char* ch_pos, hipStream_t st) {
    if (g->n_edges <= 0) return EQD_OK;
    const int n_tiles = (g->n_edges + 15) / 16;
    int blocks = (n_tiles + BWD_WAVES - 1) / BWD_WAVES;
    if (blocks > eqd_num_cus()) blocks = eqd_num_cus();
#define EQD_EDGE_SIGNS_LAUNCH(BF_, DROP_)                                                                                 \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_signs<BF_, DROP_>), dim3(blocks), dim3(64 * BWD_WAVES), 0, st, *g, *p, P, Q, x, \
                       z1_pos, ch_pos)
    if (p->bf16) {
        if (p->drop_z1) EQD_EDGE_SIGNS_LAUNCH(true, true); else EQD_EDGE_SIGNS_LAUNCH(true, false);
    } else {
        if (p->drop_z1) EQD_EDGE_SIGNS_LAUNCH(false, true); else EQD_EDGE_SIGNS_LAUNCH(false, false);
    }
#undef EQD_EDGE_SIGNS_LAUNCH
    return eqd_check_launch("k_edge_signs");
}

static int edge_bwd_blocks(const EqdGraph* g) {
    const int n_tiles = (g->n_edges + 15) / 16;
    int n_super = (n_tiles + BWD_WAVES - 1) / BWD_WAVES;
    if (n_super < 1) n_super = 1;
    const int cus = eqd_num_cus();
    return n_super < cus ? n_super : cus;      // one workgroup per CU (LDS-bound), persistent over super-tiles
}
size_t eqd_edge_bwd_vecp_floats(const EqdGraph* g) {
    return (size_t)edge_bwd_blocks(g) * (BWD_WAVES * VP + WP_N);
}

static size_t edge_bwd_carve(const EqdGraph* g, EqdArena& A, EdgeBwdWs* W) {
    const size_t E = (size_t)g->n_edges;
    EdgeBwdWs w;
    w.dz1 = A.take<float>(E * 64);
    w.dxrel = A.take<float>(E * 4);
    w.vecp = A.take<float>(eqd_edge_bwd_vecp_floats(g));
    w.wpart = w.vecp ? w.vecp + (size_t)edge_bwd_blocks(g) * BWD_WAVES * VP : nullptr;
    if (W) *W = w;
    return A.off;
}

extern "C" size_t eqd_edge_message_bwd_workspace_bytes(const EqdGraph* g) {
    EqdArena A(nullptr, 0);
    return edge_bwd_carve(g, A, nullptr) + 256;
}

// Profiling aid: ONLY the per-edge backward kernel of eqd_edge_message_bwd (no partial reductions, no
// CSR/CSC gather), so that its duration can be bracketed with HIP events.
extern "C" int eqd_edge_message_bwd_kernel_only(const EqdGraph* g, const EqdEdgeParams* p, const float* P,
                                                const float* Q, const float* x, const float* d_aggr_msg,
                                                const float* d_xnew, float* dQ, float* dx, void* workspace,
                                                size_t ws_bytes, void* stream) {
    (void)dQ;
    (void)dx;
    if (!g || !p || !P || !Q || !x || !d_aggr_msg || !d_xnew) {
        eqd_set_error("eqd_edge_message_bwd_kernel_only: NULL argument");
        return EQD_ERR_NULL;
    }
    EqdArena A(workspace, ws_bytes);
    EdgeBwdWs W;
    edge_bwd_carve(g, A, &W);
    if (!A.ok) {
        eqd_set_error("eqd_edge_message_bwd_kernel_only: workspace too small");
        return EQD_ERR_WORKSPACE;
    }
    if (g->n_edges <= 0) return EQD_OK;
    // knock-out experiment: the launch WITHOUT its 45 KB-per-workgroup weight-gradient partial writes
    // (results incomplete, timing only) prices that write stream
    if (const char* ko = eqd_tunable("EQD_EXP_EDGE_NO_PARTIALS"))
        if (ko[0] == '1' && ko[1] == 0) W.wpart = nullptr;
    if (int rc = edge_saved_check(p, "eqd_edge_message_bwd_kernel_only")) return rc;
    const bool sv = p->xh_save != nullptr;
#define EQD_EDGE_BWD_KO(BF_, SV_)                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_bwd<BF_, false, SV_>), dim3(edge_bwd_blocks(g)), dim3(64 * BWD_WAVES), 0, \
                       (hipStream_t)stream, *g, *p, P, Q, x, d_aggr_msg, d_xnew, W)
    if (p->bf16) {
        if (sv) EQD_EDGE_BWD_KO(true, true); else EQD_EDGE_BWD_KO(true, false);
    } else {
        if (sv) EQD_EDGE_BWD_KO(false, true); else EQD_EDGE_BWD_KO(false, false);
    }
#undef EQD_EDGE_BWD_KO
    return eqd_check_launch("k_edge_bwd");
}

extern "C" int eqd_edge_message_bwd(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q,
                                    const float* x, const float* d_aggr_msg, const float* d_xnew, float* dP,
                                    float* dQ, float* dx, const EqdEdgeGrads* grads, void* workspace,
                                    size_t ws_bytes, void* stream) {
    return eqd_edge_message_bwd_impl(g, p, P, Q, x, d_aggr_msg, d_xnew, dP, dQ, dx, grads, workspace, ws_bytes,
                                     (hipStream_t)stream, nullptr, nullptr, nullptr);
}

int eqd_edge_message_bwd_impl(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q, const float* x,
                              const float* d_aggr_msg, const float* d_xnew, float* dP, float* dQ, float* dx,
                              const EqdEdgeGrads* grads, void* workspace, size_t ws_bytes, hipStream_t st,
                              float* part_override, EqdRedList* defer, EqdGatherCall* hold_gather) {
    if (!g || !p || !P || !Q || !x || !d_aggr_msg || !d_xnew || !dP || !dQ || !dx || !grads) {
        eqd_set_error("eqd_edge_message_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    EqdArena A(workspace, ws_bytes);
    EdgeBwdWs W;
    edge_bwd_carve(g, A, &W);
    if (!A.ok) {
        eqd_set_error("eqd_edge_message_bwd: workspace too small (%zu needed, %zu given)", A.off, ws_bytes);
        return EQD_ERR_WORKSPACE;
    }
    const int blocks = edge_bwd_blocks(g);
    if (part_override) {       // per-layer partial buffers when the reductions are deferred
        W.vecp = part_override;
        W.wpart = part_override + (size_t)blocks * BWD_WAVES * VP;
    }
    if (g->n_edges > 0) {
        const int drop = edge_drop_mode(p, "eqd_edge_message_bwd");
        if (drop < 0) return EQD_ERR_NULL;
        if (p->bf16 && p->use_he && !g->he_bf16) {
            eqd_set_error("eqd_edge_message_bwd: bf16 mode needs EqdGraph.he_bf16");
            return EQD_ERR_NULL;
        }
        if (int rcs = edge_saved_check(p, "eqd_edge_message_bwd")) return rcs;
        const bool sv = p->xh_save != nullptr;      // the forward saved the per-edge state: load it instead of recomputing
#define EQD_EDGE_BWD_LAUNCH(BF_, DROP_, SV_)                                                                               \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_bwd<BF_, DROP_, SV_>), dim3(blocks), dim3(64 * BWD_WAVES), 0, st, *g, *p, P, Q, \
                       x, d_aggr_msg, d_xnew, W)
#define EQD_EDGE_BWD_LAUNCH2(BF_, DROP_) do { if (sv) EQD_EDGE_BWD_LAUNCH(BF_, DROP_, true); else EQD_EDGE_BWD_LAUNCH(BF_, DROP_, false); } while (0)
        if (p->bf16) {
            if (drop) EQD_EDGE_BWD_LAUNCH2(true, true); else EQD_EDGE_BWD_LAUNCH2(true, false);
        } else {
            if (drop) EQD_EDGE_BWD_LAUNCH2(false, true); else EQD_EDGE_BWD_LAUNCH2(false, false);
        }
#undef EQD_EDGE_BWD_LAUNCH2
#undef EQD_EDGE_BWD_LAUNCH
        int rc = eqd_check_launch("k_edge_bwd");
        if (rc) return rc;
        const int nw = blocks;               // one vector partial row per workgroup
        const int koff = 2 * p->d_in;
        EqdRedSeg segs[9] = {
            {W.vecp + V_DLNG, nw, VP, 64, grads->dln_g, 0, 0, 0}, {W.vecp + V_DLNB, nw, VP, 64, grads->dln_b, 0, 0, 0},
            {W.vecp + V_DWC2, nw, VP, 64, grads->dwc2, 0, 0, 0},  {W.vecp + V_DBC2, nw, VP, 1, grads->dbc2, 0, 0, 0},
            {W.vecp + V_DB2, nw, VP, 64, grads->db2, 0, 0, 0},    {W.vecp + V_DBC1, nw, VP, 64, grads->dbc1, 0, 0, 0},
            {W.wpart + WP_W2, blocks, WP_N, 4096, grads->dW2, 0, 0, 0},
            {W.wpart + WP_WC1, blocks, WP_N, 4096, grads->dWc1, 0, 0, 0},
            {W.wpart + WP_W1, blocks, WP_N, 64 * 48, grads->dW1 + koff, 48, 42, grads->ldw1}};
        if (defer && defer->n + 9 <= 512) {
            for (int i = 0; i < 9; ++i) defer->seg[defer->n++] = segs[i];
        } else if ((rc = eqd_launch_reduce_segments(segs, 9, st))) {
            return rc;
        }
    }
    // per-node sums: dP (by source), dQ (by destination), dx = (1 - eta) d_xnew + sum_src dx_rel - sum_dst dx_rel;
    // the pending reductions (this layer's partials and whatever the caller had queued) ride in the same launch
    // (hold_gather: the caller launches it - beside the layer's attention backward, eqd_launch_attention_bwd_gather)
    if (hold_gather) {
        *hold_gather = EqdGatherCall{W.dz1, W.dxrel, d_xnew, 1.f - p->eta, dP, dQ, dx, p->bf16 != 0 ? 1 : 0};
        return EQD_OK;
    }
    return eqd_launch_node_gather(g, W.dz1, W.dxrel, d_xnew, 1.f - p->eta, dP, dQ, dx, st, defer, p->bf16 != 0);
}
