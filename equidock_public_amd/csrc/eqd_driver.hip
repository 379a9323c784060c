// Host-side driver: Rigid_Body_Docking_Net.forward / backward as ONE C call each.
//
// The reference runs this path as hundreds of small PyTorch/DGL kernel launches per layer driven
// from Python, with per-pair Python loops and host syncs in the keypoint/SVD stage
// (src/model/rigid_docking_model.py:483-501, 521-600, 642-692).  Here a single C function enqueues
// the whole forward (or backward) on the caller's stream: no host sync, no allocation, all
// intermediate state in caller-owned workspaces.  The Python drop-in (equidock_public_amd/model.py)
// wraps the two calls in one torch.autograd.Function.
#include "eqd_common.h"

#include <vector>

#include <string.h>

namespace {

enum {
    P_W1 = 0, P_B1, P_LNG, P_LNB, P_W2, P_B2, P_WQ, P_WK, P_WV, P_WN1, P_BN1, P_NLG, P_NLB, P_WN2, P_BN2,
    P_WC1, P_BC1, P_WC2, P_BC2
};
enum { G_EMB = 0, G_WK, G_WQ, G_WM, G_BM };

struct Dims {
    int N, E, B, K, L, d0, dh;
    int d_in(int l) const { return l == 0 ? d0 : dh; }
    // Row stride of the attention operands q / k / v / aggr_cross and of their gradients: the width rounded up to whole
    // 16-feature blocks (69 -> 80) with zero padding, which is exactly what the attention kernels' LDS tiles hold anyway -
    // but 16-byte aligned rows of whole blocks take the float4 path and the merged backward launch (the first layer's
    // attention cost 109 us of a 1 650 us config-B step unpadded, against 34 us for a 64-wide layer).
    int d_att(int l) const { return (d_in(l) + 15) / 16 * 16; }
    int ldw1(int l) const { return 2 * d_in(l) + 27 + 15; }
    int ldwn(int l) const { return d0 + 2 * d_in(l) + dh; }
};

Dims make_dims(const EqdModelDesc* m, const EqdGraph* g) {
    Dims D;
    D.N = g->n_nodes; D.E = g->n_edges; D.B = g->n_pairs; D.K = m->n_heads; D.L = m->n_layers;
    D.d0 = m->d_emb + (m->use_mean_node_features ? 5 : 0);
    D.dh = m->d_hid;
    return D;
}

struct LayerSaved {
    float *P, *Q, *qa, *ka, *va, *aggr_msg, *aggr_cross, *lse, *y_act, *a1n;
    // bf16 storage mode (EqdModelDesc.storage_bf16): the SAVED form of tensors whose every use in the backward rounds them
    // to bf16 anyway (operands of the weight-gradient GEMMs); the fp32 originals the forward itself reads (aggr_msg, h) are
    // transients of the scratch workspace, shared by all layers.  NULL otherwise.
    uint16_t *a1n_b, *aggr_b;
    int ld_a1n;
    // q / k / v of a 64-wide layer whose attention backward will run in the dS hand-off form on the bf16 LDS kernels (large
    // batches): written by the projections as bf16 ([N][64]) and read as bf16 by the attention forward and backward; no fp32
    // form exists then (qa / ka / va NULL).  NULL otherwise.
    uint16_t *qa_b, *ka_b, *va_b;
    // saved per-edge state of the edge-message forward (EqdEdgeParams.xh_save / rstd_save / zpos_save): the backward loads it
    // instead of recomputing the first Linear + LeakyReLU + LayerNorm statistics.  NULL: the backward recomputes.
    float *xh_e, *rstd_e;
    uint16_t* zpos_e;
};
struct Saved {
    float* h[64 + 1];
    uint16_t* hb[64 + 1];      // bf16 storage mode: saved bf16 copy of h[l], 0 < l < L ([N][64]); NULL otherwise
    bool bfs;                  // bf16 storage layout
    float* x[64 + 1];
    LayerSaved lay[64];
    float *hm, *qmean, *qp, *u, *scores, *klse, *Y, *A, *T;
    float* Yc;       // [2 B][K][3] keypoints relative to their segment's first node (the keypoint backward's softmax)
    double* usv;     // [B][21] U, S, V of the Kabsch SVD (fp64), reused by the backward
};

// A: the saved state (forward -> backward); T: transients of the forward (bf16 storage mode only; the same arena as A when
// the forward keeps no state).  bfs = EqdModelDesc.storage_bf16.
void carve_saved(const Dims& D, const EqdGraph* g, EqdArena& A, Saved& S, bool bfs, EqdArena& T, bool qkv_b, bool esave) {
    const size_t N = (size_t)D.N;
    S.bfs = bfs;
    (void)T;
    S.h[0] = A.take<float>(N * D.d0);
    S.hb[0] = nullptr;
    S.x[0] = const_cast<float*>(g->x0);
    float* hp[2] = {nullptr, nullptr};
    float* aggr_t = nullptr;
    if (bfs) {
        hp[0] = T.take<float>(N * D.dh);
        hp[1] = T.take<float>(N * D.dh);
        aggr_t = T.take<float>(N * 64);
    }
    for (int l = 0; l < D.L; ++l) {
        const int d = D.d_in(l);
        LayerSaved& Ls = S.lay[l];
        Ls.P = A.take<float>(N * 64);
        Ls.Q = A.take<float>(N * 64);
        const int da = D.d_att(l);
        Ls.qa_b = Ls.ka_b = Ls.va_b = nullptr;
        if (bfs && qkv_b && d == 64) {
            Ls.qa = Ls.ka = Ls.va = nullptr;      // (no fp32 form at all: the attention kernels of these layers read bf16)
            Ls.qa_b = A.take<uint16_t>(N * 64);
            Ls.ka_b = A.take<uint16_t>(N * 64);
            Ls.va_b = A.take<uint16_t>(N * 64);
        } else {
            Ls.qa = A.take<float>(N * da);
            Ls.ka = A.take<float>(N * da);
            Ls.va = A.take<float>(N * da);
        }
        Ls.aggr_cross = A.take<float>(N * da);
        Ls.lse = A.take<float>(N);
        Ls.y_act = A.take<float>(N * d);
        Ls.a1n_b = Ls.aggr_b = nullptr;
        Ls.ld_a1n = d;
        if (bfs) {
            Ls.aggr_msg = aggr_t;
            Ls.aggr_b = A.take<uint16_t>(N * 64);
            Ls.a1n = nullptr;                     // node_mlp's LayerNorm output only travels through LDS in the forward
            Ls.ld_a1n = (d + 7) / 8 * 8;          // 16-byte rows
            Ls.a1n_b = A.take<uint16_t>(N * Ls.ld_a1n);
            if (l + 1 < D.L) {
                S.h[l + 1] = hp[l & 1];
                S.hb[l + 1] = A.take<uint16_t>(N * D.dh);
            } else {
                S.h[l + 1] = A.take<float>(N * D.dh);      // what the keypoint head consumes: fp32
                S.hb[l + 1] = nullptr;
            }
        } else {
            Ls.aggr_msg = A.take<float>(N * 64);
            Ls.a1n = A.take<float>(N * d);
            S.h[l + 1] = A.take<float>(N * D.dh);
            S.hb[l + 1] = nullptr;
        }
        S.x[l + 1] = A.take<float>(N * 3);
    }
    S.hm = A.take<float>(N * 64);
    S.qmean = A.take<float>((size_t)2 * D.B * 64);
    S.qp = A.take<float>((size_t)2 * D.B * D.K * 64);
    S.u = A.take<float>((size_t)2 * D.B * D.K * 64);
    S.scores = A.take<float>(N * D.K);
    S.klse = A.take<float>((size_t)2 * D.B * D.K);
    S.Y = A.take<float>((size_t)2 * D.B * D.K * 3);
    S.A = A.take<float>((size_t)D.B * 9);
    S.usv = A.take<double>((size_t)D.B * 21);
    S.T = A.take<float>((size_t)D.B * 9);
    S.Yc = A.take<float>((size_t)2 * D.B * D.K * 3);
    // the per-edge state comes LAST: everything above lies where it lay without it
    for (int l = 0; l < D.L; ++l) {
        LayerSaved& Ls = S.lay[l];
        Ls.xh_e = Ls.rstd_e = nullptr;
        Ls.zpos_e = nullptr;
        if (esave) {
            Ls.xh_e = A.take<float>((size_t)D.E * 64);
            Ls.rstd_e = A.take<float>((size_t)D.E);
            Ls.zpos_e = A.take<uint16_t>((size_t)D.E * 4);
        }
    }
}
// the saved part alone (the backward and the test aids: the transients are dead by then and get no memory)
void carve_saved(const Dims& D, const EqdGraph* g, EqdArena& A, Saved& S, bool bfs, bool qkv_b, bool esave) {
    EqdArena none(nullptr, 0);
    carve_saved(D, g, A, S, bfs, none, qkv_b, esave);
}
// bf16 storage mode: are q / k / v of the 64-wide layers saved as bf16?  Exactly when eqd_launch_attention_bwd_gather will
// run their backward in the dS hand-off form on the bf16 LDS kernels - the only kernels that read the bf16 form (same
// predicate as carve_scratch's W.att_ds; the EQD_* switches are snapshotted once per process, so a forward and its
// backward agree).
bool qkv_saved_bf16(const Dims& D, const EqdModelDesc* m, const EqdGraph* g) {
    return m->storage_bf16 && m->cross_msgs && D.dh == 64 && eqd_attention_ds_wanted(g, 64, true);
}
// Does a training forward save the per-edge state of its edge-message kernels (268 B per edge and layer) for the backward?
// EQD_EDGE_SAVE = 1 turns it on (0: off).  DEFAULT OFF - measured on MI355X (profiles/r06_c_edge_save_ab.txt, r06_b_kernels_*):
// the backward gains less than the forward's stores cost.  64 x (300, 300) bf16: k_edge_bwd 124.3 -> 117.2 us per launch but
// k_edge_fwd 66.2 -> 82.6 us (103 MB of stores per launch), step 16 260 -> 16 065 pairs/s; fp32: 8 898 -> 8 948 (+ 0.6 %);
// 4 x (2000, 2000): 735 -> 736.5; 8 x (200, 200): 7 240 -> 7 067 (- 2.4 %) - and the saved state doubles.  Recomputing the
// tile from P[src] + Q[dst] is cheaper than reading it back: the fused design stands.  Kept as a tested option (same bits).
// Like every EQD_* switch it is snapshotted once per process; eqd_model_saved_layout lets a caller check that a forward and
// its backward agree (model.py does).
static bool edge_state_default(const EqdModelDesc* m, const EqdGraph* g) {
    (void)m;
    (void)g;
    return false;
}
bool edge_state_saved(const EqdModelDesc* m, const EqdGraph* g) {
    const char* f = eqd_tunable("EQD_EDGE_SAVE");
    if (f && (f[0] == '0' || f[0] == '1') && f[1] == 0) return f[0] == '1';
    return edge_state_default(m, g);
}

void lin_src(EqdLinJob& J, int i, const float* X, int ldx, int K, const float* W, int w_rs, int w_cs,
             const float* mask = nullptr) {
    J.s[i].X = X; J.s[i].ldx = ldx; J.s[i].K = K; J.s[i].W = W; J.s[i].w_rs = w_rs; J.s[i].w_cs = w_cs;
    J.s[i].mask = mask;
}
thread_local int g_bf16_mode = 0;      // EqdModelDesc.storage_bf16 of the call being enqueued (stamped on every job)
EqdLinJob lin_job(int rows, int M, float* Y, int ldy, float slope, float eps) {
    EqdLinJob J;
    memset(&J, 0, sizeof(J));
    J.rows = rows; J.M = M; J.Y = Y; J.ldy = ldy; J.alpha = 1.f; J.beta = 0.f; J.slope = slope; J.ln_eps = eps;
    J.bf16 = g_bf16_mode;
    return J;
}
EqdAtbJob atb_job(const float* X, int ldx, int M, const float* Y, int ldy, int N, int rows, float* out, int o_rs,
                  float* bias_out, float slope, const float* xmask = nullptr, float scale = 1.f) {
    EqdAtbJob J;
    memset(&J, 0, sizeof(J));
    J.X = X; J.ldx = ldx; J.M = M; J.Y = Y; J.ldy = ldy; J.N = N; J.rows = rows; J.out = out; J.o_rs = o_rs;
    J.o_cs = 1; J.bias_out = bias_out; J.slope = slope; J.xmask = xmask; J.scale = scale;
    J.bf16 = g_bf16_mode;
    return J;
}

// node-level weight-gradient jobs of one layer (also used with NULL pointers for sizing)
int node_atb_jobs(const Dims& D, int l, const EqdModelDesc* m, const Saved* S, const float* dHout, const float* dz,
                  const float* dP, const float* dQ, const float* dq, const float* dk, const float* dv,
                  float* const* gp, EqdAtbJob* jobs) {
    const int d = D.d_in(l), da = D.d_att(l), N = D.N;
    const bool skip = (d == D.dh);
    const float alpha = skip ? m->skip_weight_h : 1.f;
    const LayerSaved* Ls = S ? &S->lay[l] : nullptr;
    const float* h = S ? S->h[l] : nullptr;
    const float* h0 = S ? S->h[0] : nullptr;
    auto G = [&](int i) -> float* { return gp ? gp[i] : nullptr; };
    int n = 0;
    // node_mlp.4: dWn2 = alpha dH^T a1n, dbn2 = alpha colsum(dH)
    const bool bfs = S && S->bfs;      // bf16 storage: a1n, aggr_msg and h[l] (l > 0) are saved as bf16 rows
    auto ybf = [&](EqdAtbJob& J, const uint16_t* Yb, int ldy) {
        J.Y = (const float*)Yb; J.ldy = ldy; J.y_bf16 = 1;
    };
    jobs[n++] = atb_job(dHout, D.dh, D.dh, Ls ? Ls->a1n : nullptr, d, d, N, G(P_WN2), d, G(P_BN2), m->lrelu_slope,
                        nullptr, alpha);
    if (bfs) ybf(jobs[n - 1], Ls->a1n_b, Ls->ld_a1n);
    // (order: the jobs that share dz back to back, then the six that share h - eqd_atb runs units that are neighbours in
    // the list on the same XCD at about the same time, so a shared operand's rows are fetched into that L2 once)
    // node_mlp.0: four column segments [h | aggr_msg | aggr_cross | h0]
    const int ldn = D.ldwn(l);
    jobs[n++] = atb_job(dz, d, d, Ls ? Ls->aggr_msg : nullptr, 64, 64, N, gp ? G(P_WN1) + d : nullptr, ldn, nullptr,
                        m->lrelu_slope);
    if (bfs) ybf(jobs[n - 1], Ls->aggr_b, 64);
    const bool hbf = bfs && S->hb[l] != nullptr;      // (layer 0 reads the fp32 embedding h[0])
    if (m->cross_msgs)
        jobs[n++] = atb_job(dz, d, d, Ls ? Ls->aggr_cross : nullptr, da, d, N, gp ? G(P_WN1) + d + 64 : nullptr, ldn,
                            nullptr, m->lrelu_slope);
    jobs[n++] = atb_job(dz, d, d, h0, D.d0, D.d0, N, gp ? G(P_WN1) + 2 * d + 64 : nullptr, ldn, nullptr,
                        m->lrelu_slope);
    const int n_h0 = n;
    jobs[n++] = atb_job(dz, d, d, h, d, d, N, G(P_WN1), ldn, G(P_BN1), m->lrelu_slope);
    // edge_mlp.0 node part: dW1a = dP^T h, dW1b = dQ^T h, db1 = colsum dQ
    const int ld1 = D.ldw1(l);
    jobs[n++] = atb_job(dP, 64, 64, h, d, d, N, G(P_W1), ld1, nullptr, m->lrelu_slope);
    jobs[n++] = atb_job(dQ, 64, 64, h, d, d, N, gp ? G(P_W1) + d : nullptr, ld1, G(P_B1), m->lrelu_slope);
    if (m->cross_msgs) {
        // dq, dk arrive multiplied by LeakyReLU'(q), LeakyReLU'(k) (eqd_launch_attention_bwd_act)
        jobs[n++] = atb_job(dq, da, d, h, d, d, N, G(P_WQ), d, nullptr, m->lrelu_slope);
        jobs[n++] = atb_job(dk, da, d, h, d, d, N, G(P_WK), d, nullptr, m->lrelu_slope);
        jobs[n++] = atb_job(dv, da, d, h, d, d, N, G(P_WV), d, nullptr, m->lrelu_slope);
    }
    if (hbf)
        for (int i = n_h0; i < n; ++i) ybf(jobs[i], S->hb[l], D.dh);
    return n;
}

struct Scratch {
    float *dXa, *dXb, *d_aggr_msg, *d_aggr_cross, *delta, *dh0acc;
    // kept for every layer: the weight-gradient GEMMs of ALL layers run in a few launches at the end of the pass
    float* dH_all;   // [L + 1][N][80]: entry i = grad wrt h[i]
    float *dz_all, *dq_all, *dk_all, *dv_all;   // [L][N][80]
    float *dP_all, *dQ_all;                     // [L][N][64]
    float *dY, *dT, *db, *dscores, *du, *dHk, *dqm_part, *dhm, *head_part;
    void* edge_ws; size_t edge_ws_bytes;
    float* atb_part; size_t atb_bytes;
    float* ln_part; size_t ln_part_stride;     // per layer (reductions are deferred to the end of the pass)
    float* vecp_all; size_t vecp_stride;
    float* emb_part;
    float* att_ds; int32_t* att_seg_start;      // dS hand-off workspace of the attention backward (large batches), else NULL
};

void carve_scratch(const Dims& D, const EqdModelDesc* m, const EqdGraph* g, EqdArena& A, Scratch& W) {
    const size_t N = (size_t)D.N, L = (size_t)D.L;
    W.dXa = A.take<float>(N * 3);
    W.dXb = A.take<float>(N * 3);
    W.d_aggr_msg = A.take<float>(N * 64);
    W.d_aggr_cross = A.take<float>(N * 80);
    W.dH_all = A.take<float>((L + 1) * N * 80);
    W.dz_all = A.take<float>(L * N * 80);
    W.dq_all = A.take<float>(L * N * 80);
    W.dk_all = A.take<float>(L * N * 80);
    W.dv_all = A.take<float>(L * N * 80);
    W.dP_all = A.take<float>(L * N * 64);
    W.dQ_all = A.take<float>(L * N * 64);
    W.delta = A.take<float>(N);
    W.dh0acc = A.take<float>(N * D.d0);
    W.dY = A.take<float>((size_t)2 * D.B * D.K * 3);
    W.dT = A.take<float>((size_t)D.B * 9);
    W.db = A.take<float>((size_t)D.B * 3);
    W.dscores = A.take<float>(N * D.K);
    W.du = A.take<float>((size_t)2 * D.B * D.K * 64);
    W.dHk = A.take<float>(N * 64);
    W.dqm_part = A.take<float>((size_t)2 * D.B * D.K * 64);
    W.head_part = A.take<float>(eqd_head_u_bwd_partial_floats(D.B, D.K));
    W.dhm = A.take<float>(N * 64);
    W.edge_ws_bytes = eqd_edge_message_bwd_workspace_bytes(g);
    W.edge_ws = A.take<char>(W.edge_ws_bytes);
    W.atb_bytes = eqd_atb_batch_partial_bytes(D.N);
    W.atb_part = (float*)A.take<char>(W.atb_bytes);
    W.ln_part_stride = eqd_align_up(eqd_ln_act_bwd_partial_floats(D.N, 80) * sizeof(float)) / sizeof(float);
    W.ln_part = A.take<float>(W.ln_part_stride * D.L);
    W.vecp_stride = eqd_align_up(eqd_edge_bwd_vecp_floats(g) * sizeof(float)) / sizeof(float);
    W.vecp_all = A.take<float>(W.vecp_stride * D.L);
    W.emb_part = A.take<float>(eqd_embed_bwd_partial_floats(g, m->d_emb));
    W.att_ds = nullptr;
    W.att_seg_start = nullptr;
    if (m->cross_msgs && D.dh == 64 && eqd_attention_ds_wanted(g, 64, m->storage_bf16 != 0)) {
        W.att_ds = A.take<float>((size_t)D.N * eqd_attention_ds_stride(g));
        W.att_seg_start = A.take<int32_t>((size_t)D.N);
    }
}

// dropout factors of node_mlp.1 of layer l inside EqdDropout.node (layer 0 is d0 wide, the others dh)
const float* drop_node(const Dims& D, const EqdDropout* drop, int l) {
    if (!drop) return nullptr;
    return drop->node + (l == 0 ? (size_t)0 : (size_t)D.N * D.d0 + (size_t)(l - 1) * D.N * D.dh);
}
int drop_check(const EqdDropout* drop) {
    if (!drop) return EQD_OK;
    if (!(drop->p > 0.f && drop->p < 1.f) || !drop->edge_z1 || !drop->edge_ch || !drop->node || !drop->head) {
        eqd_set_error("EqdDropout: need 0 < p < 1 and all four mask arrays (p = %g)", drop->p);
        return EQD_ERR_NULL;
    }
    return EQD_OK;
}

// ---- library-drawn dropout masks (eqd_dropout_draw) ------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter-based: element block i of array a under key *seed is
// philox(counter = (lo(i), hi(i), j, a), key = (lo(seed), hi(seed))) - no state, any launch shape gives the same masks.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// work items: [0, n_words) edge_z1 words, [n_words, 2 n_words) edge_ch words (4 Philox calls of 8 elements each), then
// the groups of 8 floats of node and of head (one call each).  A call's four 32-bit words are eight 16-bit draws (low half
// first); an element is KEPT when its draw >= thr = round(p * 2^16) (p = 0.25, the published family's value, is exact).
// (32-bit draws - twice the calls - cost 190 us per step at 64 x (300, 300), 2 % of it: the integer multiplies of the ten
// rounds are quarter rate.)
__global__ __launch_bounds__(EQD_BLOCK) void k_dropout_draw(const unsigned long long* __restrict__ seed, uint32_t thr,
                                                            float scale, size_t n_words, uint32_t* __restrict__ edge_z1,
                                                            uint32_t* __restrict__ edge_ch, size_t n_node,
                                                            float* __restrict__ node, size_t n_head,
                                                            float* __restrict__ head) {
    const unsigned long long s = seed[0];
    const uint32_t k0 = (uint32_t)s, k1 = (uint32_t)(s >> 32);
    const size_t gn = (n_node + 7) >> 3, gh = (n_head + 7) >> 3;
    const size_t total = 2 * n_words + gn + gh;
    for (size_t w = (size_t)blockIdx.x * EQD_BLOCK + threadIdx.x; w < total; w += (size_t)gridDim.x * EQD_BLOCK) {
        uint32_t r[4];
        if (w < 2 * n_words) {
            const uint32_t arr = w < n_words ? 0u : 1u;
            const size_t i = arr ? w - n_words : w;
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                philox4x32_10((uint32_t)i, (uint32_t)((unsigned long long)i >> 32), (uint32_t)j, arr, k0, k1, r);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    bits |= (((r[b] & 0xffffu) >= thr ? 1u : 0u) | ((r[b] >> 16) >= thr ? 2u : 0u)) << (8 * j + 2 * b);
            }
            (arr ? edge_ch : edge_z1)[i] = bits;
        } else {
            const bool hd = w >= 2 * n_words + gn;
            const size_t i = w - 2 * n_words - (hd ? gn : 0), n = hd ? n_head : n_node;
            float* __restrict__ dst = hd ? head : node;
            philox4x32_10((uint32_t)i, (uint32_t)((unsigned long long)i >> 32), 0u, hd ? 3u : 2u, k0, k1, r);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (8 * i + 2 * b < n) dst[8 * i + 2 * b] = (r[b] & 0xffffu) >= thr ? scale : 0.f;
                if (8 * i + 2 * b + 1 < n) dst[8 * i + 2 * b + 1] = (r[b] >> 16) >= thr ? scale : 0.f;
            }
        }
    }
}

extern "C" int eqd_dropout_draw(const EqdModelDesc* m, const EqdGraph* g, float p, const uint64_t* seed, uint32_t* edge_z1,
                                uint32_t* edge_ch, float* node, float* head, void* stream) {
    if (!m || !g || !seed || !edge_z1 || !edge_ch || !node || !head) {
        eqd_set_error("eqd_dropout_draw: NULL argument");
        return EQD_ERR_NULL;
    }
    if (!(p > 0.f && p < 1.f)) {
        eqd_set_error("eqd_dropout_draw: need 0 < p < 1 (p = %g)", p);
        return EQD_ERR_UNSUPPORTED;
    }
    const Dims D = make_dims(m, g);
    const size_t n_words = (size_t)D.L * D.E * 2;
    const size_t n_node = (size_t)D.N * D.d0 + (size_t)(D.L - 1) * D.N * D.dh, n_head = (size_t)D.N * 64;
    const size_t total = 2 * n_words + (n_node + 7) / 8 + (n_head + 7) / 8;
    if (total == 0) return EQD_OK;
    uint32_t thr = (uint32_t)((double)p * 65536.0 + 0.5);
    thr = thr < 1 ? 1 : (thr > 65535 ? 65535 : thr);
    size_t blocks = (total + EQD_BLOCK - 1) / EQD_BLOCK;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_dropout_draw, dim3((unsigned)blocks), dim3(EQD_BLOCK), 0, (hipStream_t)stream,
                       (const unsigned long long*)seed, thr, 1.f / (1.f - p), n_words, edge_z1, edge_ch, n_node, node, n_head,
                       head);
    return eqd_check_launch("k_dropout_draw");
}

__global__ void k_mul_inplace(float* __restrict__ y, const float* __restrict__ m, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] *= m[i];
}

EqdEdgeParams edge_params(const Dims& D, const EqdModelDesc* m, int l, const float* const* p,
                          const EqdDropout* drop = nullptr) {
    EqdEdgeParams e;
    memset(&e, 0, sizeof(e));
    if (drop) {
        e.drop_z1 = drop->edge_z1 + (size_t)l * D.E * 2;
        e.drop_ch = drop->edge_ch + (size_t)l * D.E * 2;
        e.drop_scale = 1.f / (1.f - drop->p);
    }
    e.W1 = p[P_W1]; e.ldw1 = D.ldw1(l); e.d_in = D.d_in(l);
    e.ln_g = p[P_LNG]; e.ln_b = p[P_LNB]; e.W2 = p[P_W2]; e.b2 = p[P_B2];
    e.Wc1 = p[P_WC1]; e.bc1 = p[P_BC1]; e.wc2 = p[P_WC2]; e.bc2 = p[P_BC2];
    e.slope = m->lrelu_slope; e.ln_eps = m->ln_eps; e.eta = m->x_connection_init;
    e.use_dist = m->use_dist_in_layers; e.use_he = m->use_edge_features; e.bf16 = m->storage_bf16;
    return e;
}

}  // namespace

extern "C" int eqd_model_check(const EqdModelDesc* m, const EqdGraph* g) {
    if (!m || !g) {
        eqd_set_error("eqd_model_check: NULL argument");
        return EQD_ERR_NULL;
    }
    if (m->d_hid != 64 || m->d_emb < 1 || m->d_emb > 64 || m->edge_feats != 27) {
        eqd_set_error("unsupported widths: iegmn_lay_hid_dim=%d (need 64), residue_emb_dim=%d (need 1..64), "
                      "input_edge_feats_dim=%d (need 27)", m->d_hid, m->d_emb, m->edge_feats);
        return EQD_ERR_UNSUPPORTED;
    }
    if (m->n_layers < 1 || m->n_layers > 64 || m->n_heads < 1 || m->n_heads > 128) {
        eqd_set_error("unsupported sizes: iegmn_n_lays=%d (1..64), num_att_heads=%d (1..128)", m->n_layers, m->n_heads);
        return EQD_ERR_UNSUPPORTED;
    }
    if (g->n_pairs < 1 || g->n_nodes < 1) {
        eqd_set_error("empty batch");
        return EQD_ERR_SHAPE;
    }
    return EQD_OK;
}

// The layout of the saved-state buffer depends on the process-wide EQD_* switches (which attention-backward form runs,
// whether the per-edge state is saved): bit 0 = bf16 storage, bit 1 = q / k / v of the 64-wide layers saved as bf16, bit 2 =
// per-edge state saved.  A forward and its backward must see the same value (eqd_tunables_reload in between is an error the
// library cannot detect on its own - the buffer is device memory); model.py compares the two and refuses.  -1: bad arguments.
extern "C" int eqd_model_saved_layout(const EqdModelDesc* m, const EqdGraph* g) {
    if (eqd_model_check(m, g)) return -1;
    const Dims D = make_dims(m, g);
    return (m->storage_bf16 ? 1 : 0) | (qkv_saved_bf16(D, m, g) ? 2 : 0) | (edge_state_saved(m, g) ? 4 : 0);
}

extern "C" size_t eqd_model_saved_bytes(const EqdModelDesc* m, const EqdGraph* g) {
    if (eqd_model_check(m, g)) return 0;
    Dims D = make_dims(m, g);
    EqdArena A(nullptr, 0);
    Saved S;
    carve_saved(D, g, A, S, m->storage_bf16 != 0, qkv_saved_bf16(D, m, g), edge_state_saved(m, g));
    return A.off + 256;
}
// transients of a forward in bf16 storage mode (fp32 h ping-pong, aggr_msg): carved from the scratch workspace
static size_t forward_transient_bytes(const EqdModelDesc* m, const EqdGraph* g) {
    if (!m->storage_bf16) return 0;
    Dims D = make_dims(m, g);
    EqdArena A(nullptr, 0), T(nullptr, 0);
    Saved S;
    carve_saved(D, g, A, S, true, T, qkv_saved_bf16(D, m, g), edge_state_saved(m, g));
    return T.off + 256;
}

extern "C" size_t eqd_model_scratch_bytes(const EqdModelDesc* m, const EqdGraph* g) {
    if (eqd_model_check(m, g)) return 0;
    Dims D = make_dims(m, g);
    EqdArena A(nullptr, 0);
    Scratch W;
    carve_scratch(D, m, g, A, W);
    size_t bwd = A.off + 256;
    // a forward without a saved-state buffer (inference) carves state + transients from the scratch workspace; a training
    // forward in bf16 storage mode its transients
    size_t sv = eqd_model_saved_bytes(m, g) + forward_transient_bytes(m, g);
    return bwd > sv ? bwd : sv;
}

// Debug / test aid: where the state after `layer` layers (0 = the embedding) lives inside a forward's `saved` buffer.
extern "C" int eqd_model_layer_state(const EqdModelDesc* m, const EqdGraph* g, const void* saved, size_t saved_bytes,
                                     int layer, const float** h, int* h_width, const float** x) {
    if (int rc = eqd_model_check(m, g)) return rc;
    if (!saved || !h || !h_width || !x) {
        eqd_set_error("eqd_model_layer_state: NULL argument");
        return EQD_ERR_NULL;
    }
    const Dims D = make_dims(m, g);
    if (layer < 0 || layer > D.L) {
        eqd_set_error("eqd_model_layer_state: layer %d outside 0..%d", layer, D.L);
        return EQD_ERR_SHAPE;
    }
    EqdArena A(const_cast<void*>(saved), saved_bytes);
    Saved S;
    carve_saved(D, g, A, S, m->storage_bf16 != 0, qkv_saved_bf16(D, m, g), edge_state_saved(m, g));
    if (!A.ok) {
        eqd_set_error("eqd_model_layer_state: saved buffer too small");
        return EQD_ERR_WORKSPACE;
    }
    if (m->storage_bf16 && layer > 0 && layer < D.L) {
        eqd_set_error("eqd_model_layer_state: in bf16 storage mode the fp32 node features of layers 1 .. n_layers - 1 are "
                      "transients of the forward (their saved form is bf16); layer 0 and n_layers are available");
        return EQD_ERR_UNSUPPORTED;
    }
    *h = S.h[layer];
    *h_width = layer == 0 ? D.d0 : D.dh;
    *x = S.x[layer];
    return EQD_OK;
}

// Test / debug aid: the LeakyReLU branch decisions of a forward whose state is in `saved`, as one byte per element
// (1 = pre-activation > 0).  The backward takes its node-level masks from the saved activations (y_act, qa, ka, hm: the
// sign of LeakyReLU(z) is the sign of z) and the edge-level ones from the per-tile recompute; both are reproduced here
// bit for bit, so that a CPU oracle can be evaluated with exactly the slopes the library used and its gradient compared
// plainly (tests/parity_common.py) - a pre-activation within rounding of 0 otherwise takes either slope depending on
// the summation order, in the reference as much as here.
__global__ void k_sign_rows(const float* __restrict__ src, int ld, int rows, int d, unsigned char* __restrict__ out) {
    const size_t n = (size_t)rows * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / d;
        out[i] = src[r * ld + (i - r * d)] > 0.f ? 1 : 0;
    }
}
__global__ void k_sign_rows_bf(const uint16_t* __restrict__ src, int ld, int rows, int d, unsigned char* __restrict__ out) {
    const size_t n = (size_t)rows * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / d;
        out[i] = bf2f(src[r * ld + (i - r * d)]) > 0.f ? 1 : 0;
    }
}
int eqd_launch_edge_signs(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q, const float* x,
                          unsigned char* z1_pos, unsigned char* ch_pos, hipStream_t st);

extern "C" int eqd_model_lrelu_signs(const EqdModelDesc* m, const EqdGraph* g, const float* const* params,
                                     const EqdDropout* drop, const void* saved, size_t saved_bytes, int layer,
                                     unsigned char* edge_z1,
                                     unsigned char* edge_ch, unsigned char* node, unsigned char* q, unsigned char* k,
                                     void* stream) {
    if (int rc = eqd_model_check(m, g)) return rc;
    if (!params || !saved) {
        eqd_set_error("eqd_model_lrelu_signs: NULL argument");
        return EQD_ERR_NULL;
    }
    const Dims D = make_dims(m, g);
    if (layer < 0 || layer > D.L) {
        eqd_set_error("eqd_model_lrelu_signs: layer %d outside 0..%d", layer, D.L);
        return EQD_ERR_SHAPE;
    }
    EqdArena A(const_cast<void*>(saved), saved_bytes);
    Saved S;
    carve_saved(D, g, A, S, m->storage_bf16 != 0, qkv_saved_bf16(D, m, g), edge_state_saved(m, g));
    if (!A.ok) {
        eqd_set_error("eqd_model_lrelu_signs: saved buffer too small");
        return EQD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    auto rows = [&](const float* src, int ld, int d, unsigned char* out) -> int {
        if (!out) return EQD_OK;
        const size_t n = (size_t)D.N * d;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_sign_rows, dim3(blocks), dim3(256), 0, st, src, ld, D.N, d, out);
        return eqd_check_launch("k_sign_rows");
    };
    auto rows_b = [&](const uint16_t* src, int ld, int d, unsigned char* out) -> int {
        if (!out) return EQD_OK;
        const size_t n = (size_t)D.N * d;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_sign_rows_bf, dim3(blocks), dim3(256), 0, st, src, ld, D.N, d, out);
        return eqd_check_launch("k_sign_rows");
    };
    if (layer == D.L)      // head: mlp_h_mean_ROT's LeakyReLU (rigid_docking_model.py:434-438), [n_nodes][64]
        return rows(S.hm, 64, 64, node);
    const float* const* p = params + (size_t)EQD_PARAMS_PER_LAYER * layer;
    const LayerSaved& Ls = S.lay[layer];
    const int d = D.d_in(layer), da = D.d_att(layer);
    if (int rc = rows(Ls.y_act, d, d, node)) return rc;
    if (m->cross_msgs) {
        if (Ls.qa_b) {      // saved as bf16 (sign of a bf16-rounded value = sign of the value)
            if (int rc = rows_b(Ls.qa_b, 64, d, q)) return rc;
            if (int rc = rows_b(Ls.ka_b, 64, d, k)) return rc;
        } else {
            if (int rc = rows(Ls.qa, da, d, q)) return rc;
            if (int rc = rows(Ls.ka, da, d, k)) return rc;
        }
    }
    if (edge_z1 && edge_ch) {
        if (int rc = drop_check(drop)) return rc;
        EqdEdgeParams ep = edge_params(D, m, layer, p, drop);
        return eqd_launch_edge_signs(g, &ep, Ls.P, Ls.Q, S.x[layer], edge_z1, edge_ch, st);
    }
    return EQD_OK;
}

// ---- execution context: auxiliary streams + events -----------------------------------------------------
struct EqdCtx {
    hipStream_t sa;              // attention branch of the forward (runs beside the edge-message kernel)
    hipEvent_t fork, join_a;
};
extern "C" int eqd_ctx_create(void** ctx) {
    if (!ctx) return EQD_ERR_NULL;
    EqdCtx* c = new EqdCtx();
    bool ok = hipStreamCreateWithFlags(&c->sa, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->join_a, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        delete c;
        eqd_set_error("eqd_ctx_create: could not create streams/events");
        return EQD_ERR_LAUNCH;
    }
    *ctx = c;
    return EQD_OK;
}
extern "C" int eqd_ctx_destroy(void* ctx) {
    if (!ctx) return EQD_OK;
    EqdCtx* c = (EqdCtx*)ctx;
    (void)hipEventDestroy(c->fork);
    (void)hipEventDestroy(c->join_a);
    (void)hipStreamDestroy(c->sa);
    delete c;
    return EQD_OK;
}
#define HIPOK(x)                                    \
    do {                                            \
        if ((x) != hipSuccess) {                    \
            eqd_set_error("HIP runtime call failed: %s", #x); \
            return EQD_ERR_LAUNCH;                  \
        }                                           \
    } while (0)

#define RC(x)                 \
    do {                      \
        int rc_ = (x);        \
        if (rc_) return rc_;  \
    } while (0)

extern "C" int eqd_model_forward(const EqdModelDesc* m, const EqdGraph* g, const float* const* params,
                                 const EqdDropout* drop, const float* svd_draws, float* lig_out, float* Y_lig, float* Y_rec, float* T,
                                 float* b, int32_t* svd_status, void* saved, size_t saved_bytes, void* scratch,
                                 size_t scratch_bytes, void* stream, void* ctx) {
    RC(eqd_model_check(m, g));
    if (!params || !lig_out || !Y_lig || !Y_rec || !T || !b || !svd_status) {
        eqd_set_error("eqd_model_forward: NULL argument");
        return EQD_ERR_NULL;
    }
    RC(drop_check(drop));
    hipStream_t st = (hipStream_t)stream;
    EqdCtx* cx = (EqdCtx*)ctx;
    const Dims D = make_dims(m, g);
    g_bf16_mode = m->storage_bf16 ? 1 : 0;
    // state: `saved` when a backward will follow, else the scratch workspace; bf16 storage mode also needs transients
    // (fp32 h ping-pong, aggr_msg), always from the scratch workspace - behind the state when both live there
    const bool bfs = m->storage_bf16 != 0;
    EqdArena A(saved ? saved : scratch, saved ? saved_bytes : scratch_bytes);
    EqdArena Tr(scratch, scratch_bytes);
    Saved S;
    carve_saved(D, g, A, S, bfs, saved ? Tr : A, qkv_saved_bf16(D, m, g), saved != nullptr && edge_state_saved(m, g));
    if (!A.ok || (bfs && saved && !Tr.ok)) {
        eqd_set_error("eqd_model_forward: workspace too small or missing (state %zu bytes%s)", A.off,
                      bfs && saved ? "; bf16 storage mode also needs the scratch workspace (eqd_model_scratch_bytes) in a "
                                     "forward that saves state" : "");
        return EQD_ERR_WORKSPACE;
    }
    const float slope = m->lrelu_slope, eps = m->ln_eps;
    const int N = D.N;
    const float* const* gp = params + (size_t)EQD_PARAMS_PER_LAYER * D.L;

    RC(eqd_launch_embed_fwd(g, gp[G_EMB], m->d_emb, m->use_mean_node_features, S.h[0], D.d0, st));
    // node projections of layer `l` from h (global) or from the LDS tile `loc` of a row chain:
    // P, Q (split first edge Linear), attention q / k / v
    auto node_pre_jobs = [&](int l, const float* hsrc, int loc, EqdChainJob* cj) -> int {
        const float* const* p = params + (size_t)EQD_PARAMS_PER_LAYER * l;
        const int d = D.d_in(l);
        const LayerSaved& Ls = S.lay[l];
        int nj = 0;
        auto add = [&](float* Y, int M, int ldy, const float* Wp, int w_rs, const float* bias, int act) {
            EqdChainJob& C = cj[nj++];
            memset(&C, 0, sizeof(C));
            C.lin = lin_job(N, M, Y, ldy, slope, eps);
            lin_src(C.lin, 0, hsrc, d, d, Wp, w_rs, 1);
            C.lin.nsrc = 1; C.lin.bias = bias; C.lin.act = act;
            for (int i = 0; i < EQD_MAX_SRC; ++i) C.src_local[i] = -1;
            C.src_local[0] = loc;
            C.out_local = -1;
        };
        add(Ls.P, 64, 64, p[P_W1], D.ldw1(l), nullptr, 0);
        add(Ls.Q, 64, 64, p[P_W1] + d, D.ldw1(l), p[P_B1], 0);
        if (m->cross_msgs) {
            const int da = D.d_att(l);
            add(Ls.qa, d, da, p[P_WQ], d, nullptr, 1);
            add(Ls.ka, d, da, p[P_WK], d, nullptr, 1);
            add(Ls.va, d, da, p[P_WV], d, nullptr, 0);
            if (Ls.qa_b) {      // bf16 storage mode: q / k / v leave the projections as bf16 only (Y = NULL)
                cj[nj - 3].lin.Yb = Ls.qa_b; cj[nj - 2].lin.Yb = Ls.ka_b; cj[nj - 1].lin.Yb = Ls.va_b;
                for (int i = nj - 3; i < nj; ++i) cj[i].lin.ldyb = 64;
            }
            if (da != d)
                for (int i = nj - 3; i < nj; ++i) cj[i].lin.pad_to = da;
        }
        return nj;
    };
    bool proj_in_chain = false;      // layer l's projections were computed by layer l - 1's node-update chain
    for (int l = 0; l < D.L; ++l) {
        const float* const* p = params + (size_t)EQD_PARAMS_PER_LAYER * l;
        const int d = D.d_in(l);
        const LayerSaved& Ls = S.lay[l];
        const float* h = S.h[l];
        const int da = D.d_att(l);
        // (padding columns of q / k / v when da != d - the 69-wide first layer's 80-float attention rows -: written as zeros by
        //  the projection jobs themselves, EqdLinJob.pad_to; until round 3 a fill of the three buffers in front of them)
        // ---- node projections (5 independent jobs, one launch: they run side by side) - unless the previous
        //      layer's node-update chain already carried them (large batches, see below) ---------------------
        if (!proj_in_chain) {
            EqdChainJob cj[8];
            const int nj = node_pre_jobs(l, h, -1, cj);
            EqdLinJob jobs[8];
            for (int i = 0; i < nj; ++i) jobs[i] = cj[i].lin;
            RC(eqd_linear(jobs, nj, st));
        }
        // ---- cross attention (on the auxiliary stream when a context is given) || edge messages ----------
        hipStream_t sat = (cx && m->cross_msgs) ? cx->sa : st;
        if (sat != st) {
            HIPOK(hipEventRecord(cx->fork, st));
            HIPOK(hipStreamWaitEvent(sat, cx->fork, 0));
        }
        EqdEdgeParams ep = edge_params(D, m, l, p, drop);
        ep.aggr_bf16 = Ls.aggr_b;      // bf16 storage mode: the saved copy (the fp32 aggr_msg is a transient)
        ep.xh_save = Ls.xh_e; ep.rstd_save = Ls.rstd_e; ep.zpos_save = Ls.zpos_e;      // (NULL unless this forward saves them)
        if (m->cross_msgs && sat == st && !m->storage_bf16) {
            // the two independent halves of the layer: ONE launch when both fit the chip at once (small batches)
            RC(eqd_edge_attn_fwd(g, &ep, Ls.P, Ls.Q, S.x[l], Ls.aggr_msg, S.x[l + 1], da, Ls.qa, Ls.ka, Ls.va, Ls.aggr_cross,
                                 Ls.lse, st));
        } else {
            if (m->cross_msgs) {
                if (m->storage_bf16 && Ls.qa_b)
                    RC(eqd_attention_fwd_bf16_impl(g, da, (const float*)Ls.qa_b, (const float*)Ls.ka_b, (const float*)Ls.va_b,
                                                   Ls.aggr_cross, Ls.lse, sat, true));
                else if (m->storage_bf16)
                    RC(eqd_cross_attention_fwd_bf16(g, da, Ls.qa, Ls.ka, Ls.va, Ls.aggr_cross, Ls.lse, sat));
                else
                    RC(eqd_cross_attention_fwd(g, da, Ls.qa, Ls.ka, Ls.va, Ls.aggr_cross, Ls.lse, sat));
            } else {
                if (hipMemsetAsync(Ls.aggr_cross, 0, (size_t)N * da * sizeof(float), st) != hipSuccess) return EQD_ERR_LAUNCH;
            }
            if (sat != st) HIPOK(hipEventRecord(cx->join_a, sat));
            RC(eqd_edge_message_fwd(g, &ep, Ls.P, Ls.Q, S.x[l], Ls.aggr_msg, S.x[l + 1], st));
            if (sat != st) HIPOK(hipStreamWaitEvent(st, cx->join_a, 0));
        }
        // ---- node update: node_mlp([h, aggr_msg, aggr_cross, h0]) -> LayerNorm, then node_mlp.4 (+ skip).
        //      (A fused row chain of these + the next layer's projections measured SLOWER: 46 vs 33 us,
        //       because the five projections then run one after the other instead of side by side.) -----------
        const int ldn = D.ldwn(l);
        EqdLinJob j1 = lin_job(N, d, Ls.a1n, d, slope, eps);
        lin_src(j1, 0, h, d, d, p[P_WN1], ldn, 1);
        lin_src(j1, 1, Ls.aggr_msg, 64, 64, p[P_WN1] + d, ldn, 1);
        lin_src(j1, 2, Ls.aggr_cross, da, d, p[P_WN1] + d + 64, ldn, 1);
        lin_src(j1, 3, S.h[0], D.d0, D.d0, p[P_WN1] + 2 * d + 64, ldn, 1);
        j1.nsrc = 4; j1.bias = p[P_BN1]; j1.act = 1; j1.ln_g = p[P_NLG]; j1.ln_b = p[P_NLB];
        j1.pre_ln = Ls.y_act; j1.ld_pre = d;
        j1.Yb = Ls.a1n_b; j1.ldyb = Ls.ld_a1n;               // bf16 storage mode: a1n is saved as bf16 (Y = NULL)
        j1.mul = drop_node(D, drop, l); j1.ld_mul = d;       // node_mlp.1 (Dropout) in training mode
        EqdLinJob j2 = lin_job(N, D.dh, S.h[l + 1], D.dh, slope, eps);
        lin_src(j2, 0, Ls.a1n, d, d, p[P_WN2], d, 1);
        j2.nsrc = 1; j2.bias = p[P_BN2];
        j2.Yb = S.hb[l + 1]; j2.ldyb = D.dh;                 // bf16 storage mode: the saved copy of h[l + 1] (or NULL)
        if (d == D.dh) {
            j2.alpha = m->skip_weight_h; j2.beta = 1.f - m->skip_weight_h; j2.R = h; j2.ldr = d;
        }
        {   // one launch: the LayerNorm output stays in LDS for node_mlp.4
            EqdChainJob cj[EQD_CHAIN_MAXJOBS];
            memset(cj, 0, sizeof(cj));
            for (int k = 0; k < 2; ++k) {
                for (int i = 0; i < EQD_MAX_SRC; ++i) cj[k].src_local[i] = -1;
                cj[k].out_local = -1;
            }
            cj[0].lin = j1; cj[0].out_local = 0;
            cj[1].lin = j2; cj[1].src_local[0] = 0;
            int nj = 2;
            // Large batches (row chains on k_rowres, where a wave runs a chain's jobs one after the other anyway): the
            // next layer's projections ride in this chain, reading h(l+1) from the LDS tile - one launch, and one
            // pass over the h rows, less per layer.  (With the four-wave kernels of small batches this was measured
            // slower: 46 vs 33 us, the five projections then run one after the other instead of side by side; layer 0's
            // 69-wide chain stays on those kernels at every size, so it never carries projections.)
            // (round 5: in bf16 mode the 69-wide first layer's chain runs on k_rowres80 at these sizes and carries layer 1's
            //  projections the same way)
            proj_in_chain = l + 1 < D.L && (d == 64 || (m->storage_bf16 && eqd_rowres80_on())) && D.d_in(l + 1) == 64 &&
                            D.dh == 64 && eqd_rows_resident(N) && 2 + (m->cross_msgs ? 5 : 2) <= EQD_CHAIN_MAXJOBS;
            if (proj_in_chain) {
                cj[1].out_local = 1;
                nj += node_pre_jobs(l + 1, S.h[l + 1], 1, cj + 2);
            }
            RC(eqd_launch_rowchain(cj, nj, N, st));
        }
    }
    {
        EqdLinJob jm = lin_job(N, 64, S.hm, 64, slope, eps);
        lin_src(jm, 0, S.h[D.L], D.dh, D.dh, gp[G_WM], D.dh, 1);
        jm.nsrc = 1; jm.bias = gp[G_BM]; jm.act = 1;
        if (drop) { jm.mul = drop->head; jm.ld_mul = 64; }      // mlp_h_mean_ROT.1 (Dropout) in training mode
        RC(eqd_linear(&jm, 1, st));
    }
    // ---- keypoint head ----------------------------------------------------------------------------------
    const float* H = S.h[D.L];
    const float* Z = S.x[D.L];
    RC(eqd_launch_seg_mean(g, S.hm, S.qmean, st));
    RC(eqd_keypoint_pool_fwd_impl(g, D.K, gp[G_WK], gp[G_WQ], S.qmean, H, Z, S.Y, Y_lig, Y_rec, S.scores, S.klse, S.qp,
                                  S.u, st, S.Yc));
    RC(eqd_kabsch_fwd_impl(D.B, D.K, S.Y, svd_draws, m->svd_seed, S.T, T, b, S.A, svd_status, st, g, lig_out,
                           S.usv));   // + rigid apply
    return EQD_OK;
}

// Backward of the keypoint / Kabsch head (rigid_docking_model.py:521-600, 665): Kabsch + rigid apply -> keypoints -> per-head
// key / query maps -> segment means -> mlp_h_mean_ROT.  Leaves d h_L in dH_L ([N][d_hid]) and d x_L in dX_L ([N][3]), adds
// the head parameters' gradients (att_mlp_key / query_ROT directly, partial sums on `defer`; mlp_h_mean_ROT as a job on
// `wjobs`).  The first stage of eqd_model_backward, and all of eqd_model_head_backward.
static int head_backward(const EqdModelDesc* m, const EqdGraph* g, const Dims& D, const Saved& S, const Scratch& W,
                         const float* const* gpar, float* const* ggrad, const EqdDropout* drop, const float* d_lig,
                         const float* d_Ylig, const float* d_Yrec, const float* d_T, const float* d_b, const float* d_h_last,
                         const float* d_x_last, float* dH_L, float* dX_L, hipStream_t st, EqdRedList* defer,
                         std::vector<EqdAtbJob>& wjobs) {
    const int N = D.N, B = D.B, K = D.K;
    const float slope = m->lrelu_slope, eps = m->ln_eps;
    RC(eqd_kabsch_bwd_impl(B, K, S.Y, S.A, S.T, d_T, d_b, d_Ylig, d_Yrec, W.dY, st, g, d_lig, S.usv));   // + rigid apply backward
    const float* H = S.h[D.L];
    const float* Z = S.x[D.L];
    int du_chunks = 1;      // > 1: the keypoint backward left partial du blocks in W.dscores for k_head_u_bwd to sum
    RC(eqd_launch_keypoint_bwd(g, K, H, Z, S.scores, S.klse, S.u, W.dY, W.dscores, W.du, W.dHk, dX_L, st, S.Yc, &du_chunks));
    if (d_h_last) RC(eqd_launch_axpy(W.dHk, d_h_last, 1.f, (size_t)N * 64, st));      // a loss on the last layer's node data
    if (d_x_last) RC(eqd_launch_axpy(dX_L, d_x_last, 1.f, (size_t)N * 3, st));
    RC(eqd_launch_head_u_bwd(g, K, gpar[G_WK], gpar[G_WQ], S.qmean, S.qp, du_chunks > 1 ? W.dscores : W.du, ggrad[G_WK],
                             ggrad[G_WQ], W.dqm_part, st, W.head_part, defer, du_chunks));
    RC(eqd_launch_qmean_bwd(g, K, W.dqm_part, W.dhm, st));
    if (drop) {      // d(keep * s * LeakyReLU(z)): the dropout factor rides on the incoming gradient, the LeakyReLU
                     // derivative comes from the saved activation's sign as without dropout
        const size_t n = (size_t)N * 64;
        hipLaunchKernelGGL(k_mul_inplace, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st,
                           W.dhm, drop->head, n);
        RC(eqd_check_launch("k_mul_inplace"));
    }
    EqdLinJob j = lin_job(N, D.dh, dH_L, D.dh, slope, eps);
    lin_src(j, 0, W.dhm, 64, 64, gpar[G_WM], 1, D.dh, S.hm);
    j.nsrc = 1; j.R = W.dHk; j.ldr = 64; j.beta = 1.f;
    RC(eqd_linear(&j, 1, st));
    wjobs.push_back(atb_job(W.dhm, 64, 64, H, D.dh, D.dh, N, ggrad[G_WM], D.dh, ggrad[G_BM], slope, S.hm));
    return EQD_OK;
}

// Test / single-op entry point: the head's backward alone, from the state a forward left in `saved`.  d_h_L [N][d_hid],
// d_x_L [N][3] receive the gradient w.r.t. the last layer's node state; the head parameters' gradients are accumulated
// into grad_flat (the layer parameters' entries are not touched).
extern "C" int eqd_model_head_backward(const EqdModelDesc* m, const EqdGraph* g, const float* const* params,
                                       const EqdDropout* drop, const float* d_lig, const float* d_Ylig, const float* d_Yrec,
                                       const float* d_T, const float* d_b, float* grad_flat, const int64_t* grad_offsets,
                                       const void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                                       float* d_h_L, float* d_x_L, void* stream) {
    RC(eqd_model_check(m, g));
    if (!params || !grad_flat || !grad_offsets || !saved || !scratch || !d_h_L || !d_x_L) {
        eqd_set_error("eqd_model_head_backward: NULL argument");
        return EQD_ERR_NULL;
    }
    RC(drop_check(drop));
    hipStream_t st = (hipStream_t)stream;
    const Dims D = make_dims(m, g);
    g_bf16_mode = m->storage_bf16 ? 1 : 0;
    EqdArena As(const_cast<void*>(saved), saved_bytes);
    Saved S;
    carve_saved(D, g, As, S, m->storage_bf16 != 0, qkv_saved_bf16(D, m, g), edge_state_saved(m, g));
    EqdArena Aw(scratch, scratch_bytes);
    Scratch W;
    carve_scratch(D, m, g, Aw, W);
    if (!As.ok || !Aw.ok) {
        eqd_set_error("eqd_model_head_backward: workspace too small (saved %zu, scratch %zu needed)", As.off, Aw.off);
        return EQD_ERR_WORKSPACE;
    }
    const int nparams = EQD_PARAMS_PER_LAYER * D.L + EQD_GLOBAL_PARAMS;
    float* gptr[EQD_PARAMS_PER_LAYER * 64 + EQD_GLOBAL_PARAMS];
    for (int i = 0; i < nparams; ++i) gptr[i] = grad_flat + grad_offsets[i];
    const float* const* gpar = params + (size_t)EQD_PARAMS_PER_LAYER * D.L;
    float* const* ggrad = gptr + (size_t)EQD_PARAMS_PER_LAYER * D.L;
    EqdRedList* defer = new EqdRedList();
    defer->n = 0;
    struct DeferGuard {
        EqdRedList* p;
        ~DeferGuard() { delete p; }
    } defer_guard{defer};
    std::vector<EqdAtbJob> wjobs;
    float* dH = W.dH_all + (size_t)D.L * D.N * 80;
    RC(head_backward(m, g, D, S, W, gpar, ggrad, drop, d_lig, d_Ylig, d_Yrec, d_T, d_b, nullptr, nullptr, dH, W.dXa, st, defer,
                     wjobs));
    RC(eqd_atb(wjobs.data(), (int)wjobs.size(), W.atb_part, W.atb_bytes, st));
    RC(eqd_launch_reduce_segments(defer->seg, defer->n, st));
    HIPOK(hipMemcpyAsync(d_h_L, dH, (size_t)D.N * D.dh * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPOK(hipMemcpyAsync(d_x_L, W.dXa, (size_t)D.N * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return EQD_OK;
}

extern "C" int eqd_model_backward(const EqdModelDesc* m, const EqdGraph* g, const float* const* params,
                                  const EqdDropout* drop, const float* d_lig, const float* d_Ylig, const float* d_Yrec, const float* d_T,
                                  const float* d_b, const float* d_h_last, const float* d_x_last, float* grad_flat, const int64_t* grad_offsets, const void* saved,
                                  size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream, void* ctx) {
    RC(eqd_model_check(m, g));
    if (!params || !grad_flat || !grad_offsets || !saved || !scratch) {
        eqd_set_error("eqd_model_backward: NULL argument");
        return EQD_ERR_NULL;
    }
    hipStream_t st = (hipStream_t)stream;
    (void)ctx;
    RC(drop_check(drop));
    const Dims D = make_dims(m, g);
    g_bf16_mode = m->storage_bf16 ? 1 : 0;
    EqdArena As(const_cast<void*>(saved), saved_bytes);
    Saved S;
    carve_saved(D, g, As, S, m->storage_bf16 != 0, qkv_saved_bf16(D, m, g), edge_state_saved(m, g));
    EqdArena Aw(scratch, scratch_bytes);
    Scratch W;
    carve_scratch(D, m, g, Aw, W);
    if (!As.ok || !Aw.ok) {
        eqd_set_error("eqd_model_backward: workspace too small (saved %zu, scratch %zu needed)", As.off, Aw.off);
        return EQD_ERR_WORKSPACE;
    }
    const float slope = m->lrelu_slope, eps = m->ln_eps;
    const int N = D.N, B = D.B, K = D.K;
    const int nparams = EQD_PARAMS_PER_LAYER * D.L + EQD_GLOBAL_PARAMS;
    float* gptr[EQD_PARAMS_PER_LAYER * 64 + EQD_GLOBAL_PARAMS];
    for (int i = 0; i < nparams; ++i) gptr[i] = grad_flat + grad_offsets[i];
    const float* const* gpar = params + (size_t)EQD_PARAMS_PER_LAYER * D.L;
    float* const* ggrad = gptr + (size_t)EQD_PARAMS_PER_LAYER * D.L;

    // ---- head -------------------------------------------------------------------------------------------
    EqdRedList* defer = new EqdRedList();
    defer->n = 0;
    struct DeferGuard {
        EqdRedList* p;
        ~DeferGuard() { delete p; }
    } defer_guard{defer};
    const float* H = S.h[D.L];
    float* dXcur = W.dXa;   // grad wrt x[L]
    float* dXnext = W.dXb;
    const size_t NS = (size_t)N * 80, NP = (size_t)N * 64;
    auto dHof = [&](int i) -> float* { return W.dH_all + (size_t)i * NS; };   // grad wrt h[i]
    std::vector<EqdAtbJob> wjobs;   // weight-gradient GEMMs of the whole pass, launched together at the end
    wjobs.reserve((size_t)D.L * 10 + 1);
    RC(head_backward(m, g, D, S, W, gpar, ggrad, drop, d_lig, d_Ylig, d_Yrec, d_T, d_b, d_h_last, d_x_last, dHof(D.L), dXcur,
                     st, defer, wjobs));
    (void)H; (void)B; (void)K;
    // dh(l) = dz Wn1[:, :d] + dP W1a + dQ W1b + dq Wq + dk Wk + dv Wv + (1-s) dH   (dq, dk w.r.t. the pre-activations):
    // the gradient wrt h[l] once layer l's attention and edge backward are done
    auto dh_job = [&](int l) -> EqdLinJob {
        const float* const* p = params + (size_t)EQD_PARAMS_PER_LAYER * l;
        const int d = D.d_in(l);
        // dH(0) only feeds the embedding gradient: its node-feature columns (d_emb .. d0 - 1) are not computed
        EqdLinJob j = lin_job(N, l == 0 ? m->d_emb : d, dHof(l), d, slope, eps);
        int ns = 0;
        lin_src(j, ns++, W.dz_all + l * NS, d, d, p[P_WN1], 1, D.ldwn(l));
        lin_src(j, ns++, W.dP_all + l * NP, 64, 64, p[P_W1], 1, D.ldw1(l));
        lin_src(j, ns++, W.dQ_all + l * NP, 64, 64, p[P_W1] + d, 1, D.ldw1(l));
        if (m->cross_msgs) {
            const int da = D.d_att(l);
            lin_src(j, ns++, W.dq_all + l * NS, da, d, p[P_WQ], 1, d);      // already w.r.t. the pre-activations
            lin_src(j, ns++, W.dk_all + l * NS, da, d, p[P_WK], 1, d);
            lin_src(j, ns++, W.dv_all + l * NS, da, d, p[P_WV], 1, d);
        }
        j.nsrc = ns;
        if (d == D.dh) { j.R = dHof(l + 1); j.ldr = D.dh; j.beta = 1.f - m->skip_weight_h; }
        return j;
    };
    if (W.att_ds) RC(eqd_launch_seg_start(g, W.att_seg_start, st));      // (the dS hand-off form of the attention backward)
    // ---- layers, last to first ----------------------------------------------------------------------------
    for (int l = D.L - 1; l >= 0; --l) {
        const float* const* p = params + (size_t)EQD_PARAMS_PER_LAYER * l;
        float* const* gp = gptr + (size_t)EQD_PARAMS_PER_LAYER * l;
        const int d = D.d_in(l);
        const LayerSaved& Ls = S.lay[l];
        const bool skip = (d == D.dh);
        const float alpha = skip ? m->skip_weight_h : 1.f;
        const int ldn = D.ldwn(l);
        const int da = D.d_att(l);
        // (padding columns of d aggr_cross when da != d: the chain's job writes them as zeros, EqdLinJob.pad_to - they meet
        //  zeros in V, but must not be NaN bit patterns left in the scratch buffer)
        float *dz = W.dz_all + l * NS, *dq = W.dq_all + l * NS, *dk = W.dk_all + l * NS, *dv = W.dv_all + l * NS;
        float *dP = W.dP_all + l * NP, *dQ = W.dQ_all + l * NP;
        float* dHout = dHof(l + 1);
        // ONE row chain: [dh of the layer above ->] da1n = alpha dH Wn2 -> LeakyReLU/LayerNorm backward ->
        // d aggr_msg, d aggr_cross, d h0
        {
            EqdChainJob cj[8];
            int nj = 0;
            auto clear = [&](EqdChainJob& C) {
                memset(&C, 0, sizeof(C));
                for (int i = 0; i < EQD_MAX_SRC; ++i) C.src_local[i] = -1;
                C.out_local = -1;
            };
            const bool fused_dh = l < D.L - 1;
            if (l < D.L - 1) {
                EqdChainJob& C = cj[nj++];
                clear(C);
                C.lin = dh_job(l + 1);
                C.out_local = 2;
            }
            {
                EqdChainJob& C = cj[nj++];
                clear(C);
                C.lin = lin_job(N, d, nullptr, d, slope, eps);
                lin_src(C.lin, 0, dHout, D.dh, D.dh, p[P_WN2], 1, d);
                C.lin.nsrc = 1; C.lin.alpha = alpha;
                if (fused_dh) C.src_local[0] = 2;
                C.out_local = 0;
            }
            float* lnp = W.ln_part + (size_t)l * W.ln_part_stride;
            {
                EqdChainJob& C = cj[nj++];
                clear(C);
                C.type = 1;
                C.lin = lin_job(N, d, dz, d, slope, eps);
                lin_src(C.lin, 0, Ls.y_act, d, d, nullptr, 0, 0);
                C.lin.nsrc = 1; C.lin.ln_g = p[P_NLG];
                C.lin.mul = drop_node(D, drop, l); C.lin.ld_mul = d;
                C.src_local[0] = 0;
                C.out_local = 1;
                C.aux = lnp;
            }
            auto dx_job = [&](float* Y, int M, int ldy, int coff) -> EqdChainJob& {
                EqdChainJob& C = cj[nj++];
                clear(C);
                C.lin = lin_job(N, M, Y, ldy, slope, eps);
                lin_src(C.lin, 0, dz, d, d, p[P_WN1] + coff, 1, ldn);
                C.lin.nsrc = 1;
                C.src_local[0] = 1;
                return C;
            };
            dx_job(W.d_aggr_msg, 64, 64, d);
            if (m->cross_msgs) dx_job(W.d_aggr_cross, d, da, d + 64).lin.pad_to = da != d ? da : 0;
            // only the embedding columns of d h0 are ever read (k_embed_bwd: the trailing node features are inputs), so the
            // job computes m->d_emb of the D.d0 columns: a 64-wide job instead of a 69-wide one (the general body)
            EqdChainJob& C5 = dx_job(W.dh0acc, m->d_emb, D.d0, 2 * d + 64);
            if (l < D.L - 1) {      // the last layer (processed first) initialises the accumulator
                C5.lin.R = W.dh0acc; C5.lin.ldr = D.d0; C5.lin.beta = 1.f;
            }
            int nb = 0;      // partial rows of the LayerNorm-backward sums = workgroups of whichever kernel took the chain
            RC(eqd_launch_rowchain(cj, nj, N, st, &nb));
            if (defer->n + 2 <= 512) {
                defer->seg[defer->n++] = EqdRedSeg{lnp, nb, 256, d, gp[P_NLG], 0, 0, 0};
                defer->seg[defer->n++] = EqdRedSeg{lnp + 128, nb, 256, d, gp[P_NLB], 0, 0, 0};
            } else {
                EqdRedSeg segs[2] = {{lnp, nb, 256, d, gp[P_NLG], 0, 0, 0}, {lnp + 128, nb, 256, d, gp[P_NLB], 0, 0, 0}};
                RC(eqd_launch_reduce_segments(segs, 2, st));
            }
        }
        // The backward kernels of a layer each fill the chip (LDS-bound occupancy), so they run back to back
        // on ONE stream: side streams only added event latency here.  Order: edge backward, then the attention backward
        // with the edge backward's node gather riding in the same launch where that form exists (the two are independent;
        // eqd_launch_attention_bwd_gather issues them one after the other otherwise).
        {
            EqdEdgeParams ep = edge_params(D, m, l, p, drop);
            ep.xh_save = Ls.xh_e; ep.rstd_save = Ls.rstd_e; ep.zpos_save = Ls.zpos_e;      // saved by the forward, or NULL
            EqdEdgeGrads eg;
            memset(&eg, 0, sizeof(eg));
            eg.dW1 = gp[P_W1]; eg.ldw1 = D.ldw1(l); eg.dln_g = gp[P_LNG]; eg.dln_b = gp[P_LNB]; eg.dW2 = gp[P_W2];
            eg.db2 = gp[P_B2]; eg.dWc1 = gp[P_WC1]; eg.dbc1 = gp[P_BC1]; eg.dwc2 = gp[P_WC2]; eg.dbc2 = gp[P_BC2];
            EqdGatherCall gc;
            RC(eqd_edge_message_bwd_impl(g, &ep, Ls.P, Ls.Q, S.x[l], W.d_aggr_msg, dXcur, dP, dQ, dXnext, &eg, W.edge_ws,
                                         W.edge_ws_bytes, st, W.vecp_all + (size_t)l * W.vecp_stride, defer,
                                         m->cross_msgs ? &gc : nullptr));
            if (m->cross_msgs) {
                const bool qb = Ls.qa_b != nullptr;      // saved as bf16 (bf16 storage mode, dS hand-off form)
                RC(eqd_launch_attention_bwd_gather(g, da, qb ? (const float*)Ls.qa_b : Ls.qa, qb ? (const float*)Ls.ka_b : Ls.ka,
                                                   qb ? (const float*)Ls.va_b : Ls.va, Ls.aggr_cross, Ls.lse, W.d_aggr_cross, dq, dk,
                                                   dv, W.delta, m->lrelu_slope, m->storage_bf16 != 0, &gc, defer, st, W.att_ds,
                                                   W.att_seg_start, qb));
            }
        }
        {
            EqdAtbJob ajobs[16];
            const int na = node_atb_jobs(D, l, m, &S, dHout, dz, dP, dQ, dq, dk, dv, gp, ajobs);
            wjobs.insert(wjobs.end(), ajobs, ajobs + na);
        }
        float* t = dXcur; dXcur = dXnext; dXnext = t;
    }
    {
        // h[0] = h0 feeds layer 0 directly (dH of h[0]) as well as every layer's node_mlp (dh0acc).  DB5.5-sized batches: the
        // chain d h[0] -> embedding gradient rides in the weight-gradient launches it does not depend on
        EqdLinJob j = dh_job(0);
        if (eqd_atb_tail_wanted(&j, (int)wjobs.size())) {
            RC(eqd_atb_with_tail(wjobs.data(), (int)wjobs.size(), W.atb_part, W.atb_bytes, st, &j, g, W.dh0acc, dHof(0), D.d0,
                                 m->d_emb, ggrad[G_EMB], W.emb_part, defer));
        } else {
            RC(eqd_linear(&j, 1, st));
            RC(eqd_launch_embed_bwd(g, W.dh0acc, dHof(0), D.d0, m->d_emb, ggrad[G_EMB], W.emb_part, st, defer));
            RC(eqd_atb(wjobs.data(), (int)wjobs.size(), W.atb_part, W.atb_bytes, st));
        }
    }
    // deferred LayerNorm / coordinate-MLP vector reductions, edge weight-gradient partials and embedding tables
    RC(eqd_launch_reduce_segments(defer->seg, defer->n, st));
    return EQD_OK;
}

// ---------------------------------------------------------------------------------------------
// Node update of ONE IEGMN layer as an operator pair (rigid_docking_model.py:319-337):
//   a1n  = LayerNorm(mul * LeakyReLU([h | aggr_msg | aggr_cross | h0] Wn1^T + bn1))      (node_mlp.0 .. node_mlp.3)
//   u    = a1n Wn2^T + bn2                                                                (node_mlp.4)
//   h'   = s u + (1 - s) h   when d_in == d_out, else u                                    (:332-337)
// The same row-chain jobs eqd_model_forward / eqd_model_backward enqueue per layer (one launch forward; one chain + the
// weight-gradient GEMMs + one reduction backward), behind their own entry points.
// ---------------------------------------------------------------------------------------------
namespace {
// has_cross: the call carries an aggr_cross array (ld_cross is only read then; the workspace query has no arrays and
// does not depend on it)
int node_update_check(int rows, const EqdNodeUpdateParams* p, const char* who, bool has_cross) {
    if (!p || !p->Wn1 || !p->bn1 || !p->ln_g || !p->ln_b || !p->Wn2 || !p->bn2) {
        eqd_set_error("%s: NULL parameter pointer", who);
        return EQD_ERR_NULL;
    }
    if (rows < 0 || p->d_in < 4 || p->d_in > 80 || p->d0 < 4 || p->d0 > 80 || p->d_out != 64 ||
        (has_cross && p->ld_cross < p->d_in)) {
        eqd_set_error("%s: unsupported widths d_in=%d d0=%d d_out=%d ld_cross=%d (need 4..80, 4..80, 64, >= d_in)", who,
                      p->d_in, p->d0, p->d_out, p->ld_cross);
        return EQD_ERR_UNSUPPORTED;
    }
    return EQD_OK;
}
struct NodeUpdateWs {
    float *dz, *ln_part, *atb_part;
    size_t atb_bytes;
};
void node_update_atb_jobs(int rows, const EqdNodeUpdateParams* p, const float* h, const float* aggr_msg,
                          const float* aggr_cross, const float* h0, const float* a1n, const float* d_h_out, const float* dz,
                          const EqdNodeUpdateGrads* gr, EqdAtbJob* jobs, int* n_out) {
    const int d = p->d_in, ldn = p->d0 + 2 * d + 64;
    const float alpha = d == p->d_out ? p->skip_weight_h : 1.f;
    int n = 0;
    jobs[n++] = atb_job(d_h_out, p->d_out, p->d_out, a1n, d, d, rows, gr ? gr->dWn2 : nullptr, d, gr ? gr->dbn2 : nullptr,
                        p->slope, nullptr, alpha);
    jobs[n++] = atb_job(dz, d, d, aggr_msg, 64, 64, rows, gr ? gr->dWn1 + d : nullptr, ldn, nullptr, p->slope);
    if (aggr_cross || !gr)
        jobs[n++] = atb_job(dz, d, d, aggr_cross, p->ld_cross, d, rows, gr ? gr->dWn1 + d + 64 : nullptr, ldn, nullptr, p->slope);
    jobs[n++] = atb_job(dz, d, d, h0, p->d0, p->d0, rows, gr ? gr->dWn1 + 2 * d + 64 : nullptr, ldn, nullptr, p->slope);
    jobs[n++] = atb_job(dz, d, d, h, d, d, rows, gr ? gr->dWn1 : nullptr, ldn, gr ? gr->dbn1 : nullptr, p->slope);
    *n_out = n;
}
size_t node_update_carve(int rows, const EqdNodeUpdateParams* p, EqdArena& A, NodeUpdateWs& W) {
    W.dz = A.take<float>((size_t)rows * p->d_in);
    W.ln_part = A.take<float>(eqd_ln_act_bwd_partial_floats(rows, 80));
    EqdAtbJob jobs[8];
    int n = 0;
    // (sizing: the operand pointers are not read, the shapes are)
    static const float dummy = 0.f;
    node_update_atb_jobs(rows, p, &dummy, &dummy, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, jobs, &n);
    // (distinct, never dereferenced output addresses: jobs that share an output are split over launches)
    for (int i = 0; i < n; ++i) jobs[i].out = (float*)(uintptr_t)(4096 * (i + 1));
    W.atb_bytes = eqd_atb_partial_bytes(jobs, n);
    W.atb_part = (float*)A.take<char>(W.atb_bytes);
    return A.off + 256;
}
}  // namespace

extern "C" int eqd_node_update_fwd(int rows, const EqdNodeUpdateParams* p, const float* h, const float* aggr_msg,
                                   const float* aggr_cross, const float* h0, float* h_out, float* y_act, float* a1n,
                                   void* stream) {
    RC(node_update_check(rows, p, "eqd_node_update_fwd", aggr_cross != nullptr));
    if (!h || !aggr_msg || !h0 || !h_out || !y_act || !a1n) {
        eqd_set_error("eqd_node_update_fwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (rows == 0) return EQD_OK;
    g_bf16_mode = p->bf16 ? 1 : 0;
    const int d = p->d_in, ldn = p->d0 + 2 * d + 64;
    EqdLinJob j1 = lin_job(rows, d, a1n, d, p->slope, p->ln_eps);
    int ns = 0;
    lin_src(j1, ns++, h, d, d, p->Wn1, ldn, 1);
    lin_src(j1, ns++, aggr_msg, 64, 64, p->Wn1 + d, ldn, 1);
    if (aggr_cross) lin_src(j1, ns++, aggr_cross, p->ld_cross, d, p->Wn1 + d + 64, ldn, 1);      // NULL: cross_msgs off (zeros)
    lin_src(j1, ns++, h0, p->d0, p->d0, p->Wn1 + 2 * d + 64, ldn, 1);
    j1.nsrc = ns; j1.bias = p->bn1; j1.act = 1; j1.ln_g = p->ln_g; j1.ln_b = p->ln_b;
    j1.pre_ln = y_act; j1.ld_pre = d;
    j1.mul = p->drop_mul; j1.ld_mul = d;
    EqdLinJob j2 = lin_job(rows, p->d_out, h_out, p->d_out, p->slope, p->ln_eps);
    lin_src(j2, 0, a1n, d, d, p->Wn2, d, 1);
    j2.nsrc = 1; j2.bias = p->bn2;
    if (d == p->d_out) {
        j2.alpha = p->skip_weight_h; j2.beta = 1.f - p->skip_weight_h; j2.R = h; j2.ldr = d;
    }
    EqdChainJob cj[2];
    memset(cj, 0, sizeof(cj));
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < EQD_MAX_SRC; ++i) cj[k].src_local[i] = -1;
        cj[k].out_local = -1;
    }
    cj[0].lin = j1; cj[0].out_local = 0;
    cj[1].lin = j2; cj[1].src_local[0] = 0;
    return eqd_launch_rowchain(cj, 2, rows, (hipStream_t)stream);
}

extern "C" size_t eqd_node_update_bwd_workspace_bytes(int rows, const EqdNodeUpdateParams* p) {
    if (node_update_check(rows, p, "eqd_node_update_bwd_workspace_bytes", false)) return 0;
    EqdArena A(nullptr, 0);
    NodeUpdateWs W;
    return node_update_carve(rows, p, A, W);
}

extern "C" int eqd_node_update_bwd(int rows, const EqdNodeUpdateParams* p, const float* h, const float* aggr_msg,
                                   const float* aggr_cross, const float* h0, const float* y_act, const float* a1n,
                                   const float* d_h_out, float* d_h, float* d_aggr_msg, float* d_aggr_cross, float* d_h0,
                                   const EqdNodeUpdateGrads* grads, void* workspace, size_t ws_bytes, void* stream) {
    RC(node_update_check(rows, p, "eqd_node_update_bwd", aggr_cross != nullptr));
    if (!h || !aggr_msg || !h0 || !y_act || !a1n || !d_h_out || !d_h || !d_aggr_msg || !d_h0 || !grads ||
        (aggr_cross && !d_aggr_cross) || !grads->dWn1 || !grads->dbn1 || !grads->dln_g || !grads->dln_b || !grads->dWn2 ||
        !grads->dbn2) {
        eqd_set_error("eqd_node_update_bwd: NULL argument");
        return EQD_ERR_NULL;
    }
    if (rows == 0) return EQD_OK;
    g_bf16_mode = p->bf16 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    EqdArena A(workspace, ws_bytes);
    NodeUpdateWs W;
    node_update_carve(rows, p, A, W);
    if (!A.ok) {
        eqd_set_error("eqd_node_update_bwd: workspace too small (%zu needed)", A.off + 256);
        return EQD_ERR_WORKSPACE;
    }
    const int d = p->d_in, ldn = p->d0 + 2 * d + 64;
    const bool skip = d == p->d_out;
    const float alpha = skip ? p->skip_weight_h : 1.f;
    // ONE row chain: d a1n = alpha d_h_out Wn2 -> LeakyReLU / LayerNorm backward (dz) -> dz times the four column blocks
    // of Wn1 (the gradient w.r.t. h also takes the skip connection's (1 - s) d_h_out)
    EqdChainJob cj[EQD_CHAIN_MAXJOBS];
    int nj = 0;
    auto clear = [&](EqdChainJob& C) {
        memset(&C, 0, sizeof(C));
        for (int i = 0; i < EQD_MAX_SRC; ++i) C.src_local[i] = -1;
        C.out_local = -1;
    };
    {
        EqdChainJob& C = cj[nj++];
        clear(C);
        C.lin = lin_job(rows, d, nullptr, d, p->slope, p->ln_eps);
        lin_src(C.lin, 0, d_h_out, p->d_out, p->d_out, p->Wn2, 1, d);
        C.lin.nsrc = 1; C.lin.alpha = alpha;
        C.out_local = 0;
    }
    {
        EqdChainJob& C = cj[nj++];
        clear(C);
        C.type = 1;
        C.lin = lin_job(rows, d, W.dz, d, p->slope, p->ln_eps);
        lin_src(C.lin, 0, y_act, d, d, nullptr, 0, 0);
        C.lin.nsrc = 1; C.lin.ln_g = p->ln_g;
        C.lin.mul = p->drop_mul; C.lin.ld_mul = d;
        C.src_local[0] = 0;
        C.out_local = 1;
        C.aux = W.ln_part;
    }
    auto dx_job = [&](float* Y, int M, int ldy, int coff) -> EqdChainJob& {
        EqdChainJob& C = cj[nj++];
        clear(C);
        C.lin = lin_job(rows, M, Y, ldy, p->slope, p->ln_eps);
        lin_src(C.lin, 0, W.dz, d, d, p->Wn1 + coff, 1, ldn);
        C.lin.nsrc = 1;
        C.src_local[0] = 1;
        return C;
    };
    dx_job(d_aggr_msg, 64, 64, d);
    if (aggr_cross) dx_job(d_aggr_cross, d, p->ld_cross, d + 64).lin.pad_to = p->ld_cross != d ? p->ld_cross : 0;
    dx_job(d_h0, p->d0, p->d0, 2 * d + 64);
    {
        EqdChainJob& C = dx_job(d_h, d, d, 0);
        if (skip) { C.lin.R = d_h_out; C.lin.ldr = p->d_out; C.lin.beta = 1.f - p->skip_weight_h; }
    }
    int nb = 0;
    RC(eqd_launch_rowchain(cj, nj, rows, st, &nb));
    EqdRedSeg segs[2] = {{W.ln_part, nb, 256, d, grads->dln_g, 0, 0, 0}, {W.ln_part + 128, nb, 256, d, grads->dln_b, 0, 0, 0}};
    RC(eqd_launch_reduce_segments(segs, 2, st));
    EqdAtbJob jobs[8];
    int na = 0;
    node_update_atb_jobs(rows, p, h, aggr_msg, aggr_cross, h0, a1n, d_h_out, W.dz, grads, jobs, &na);
    return eqd_atb(jobs, na, W.atb_part, W.atb_bytes, st);
}
