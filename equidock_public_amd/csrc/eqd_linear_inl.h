// Row-local linear tile shared by k_linear (one job per workgroup) and k_rowchain (a sequence of jobs on
// the same 16 rows whose intermediate results stay in LDS).
#pragma once
#include "eqd_common.h"

#define LIN_KC 80   /* K chunk staged per step (every source of the IEGMN path has K <= 69) */
#define LIN_S 81    /* LDS row stride: odd -> the 16 rows of a fragment read hit 16 different banks */
#define LIN_LOCALS 4
// A workgroup owns 16 rows.  Per source chunk: ALL loads (X tile 16 x Kc, weight slice M x Kc) are
// issued together by the 256 threads as coalesced 64-byte row segments, parked in registers while the
// previous chunk is multiplied, then written to LDS; MFMA operands come from LDS.  Wave w owns output
// blocks mb = w and w + 4, so accumulators are complete (no cross-wave reduction); LayerNorm statistics
// are exchanged through LDS.  These GEMMs are tiny (3200 x 64..384 x 64 at config B): what matters is
// one memory round trip per source instead of one per 4 k-values.
struct LinRegs {
    float x[5], w[25];
};
struct LinSmem {
    float Xl[16 * LIN_S];
    float Wl[80 * LIN_S];
    float stat[EQD_WAVES][16];
};
__device__ __forceinline__ void lin_load(const EqdLinJob& J, const EqdLinSrc& S, bool local, int k0, int row0, int t,
                                         LinRegs& R) {
    const int Kc = (S.K - k0 < LIN_KC) ? S.K - k0 : LIN_KC;
    const int tr = t >> 4, tc = t & 15;
    const int row = row0 + tr;
    if (!local) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = tc + 16 * j;
            float v = 0.f;
            if (row < J.rows && k < Kc) {
                const size_t o = (size_t)row * S.ldx + k0 + k;
                v = S.X[o];
                if (S.mask) v *= lrelu_grad(S.mask[o], J.slope);
            }
            R.x[j] = v;
        }
    }
    const bool kfast = (S.w_cs == 1);
#pragma unroll
    for (int jm = 0; jm < 5; ++jm)
#pragma unroll
        for (int jk = 0; jk < 5; ++jk) {
            const int m = (kfast ? tr : tc) + 16 * jm;
            const int k = (kfast ? tc : tr) + 16 * jk;
            R.w[jm * 5 + jk] = (m < J.M && k < Kc) ? S.W[(size_t)m * S.w_rs + (size_t)(k0 + k) * S.w_cs] : 0.f;
        }
}
__device__ __forceinline__ void lin_store(const EqdLinSrc& S, bool local, int t, const LinRegs& R,
                                          float* __restrict__ Xl, float* __restrict__ Wl) {
    const int tr = t >> 4, tc = t & 15;
    if (!local) {
#pragma unroll
        for (int j = 0; j < 5; ++j) Xl[tr * LIN_S + tc + 16 * j] = R.x[j];
    }
    const bool kfast = (S.w_cs == 1);
#pragma unroll
    for (int jm = 0; jm < 5; ++jm)
#pragma unroll
        for (int jk = 0; jk < 5; ++jk) {
            const int m = (kfast ? tr : tc) + 16 * jm;
            const int k = (kfast ? tc : tr) + 16 * jk;
            Wl[m * LIN_S + k] = R.w[jm * 5 + jk];
        }
}

// One linear job on rows row0 .. row0+15.  src_local[i] >= 0: source i is the LDS tile Lb[src_local[i]]
// ([16][LIN_S], written by an earlier job of the chain; K <= 80); out_local >= 0: the result is also left
// in Lb[out_local].  Must be called by all 256 threads of the workgroup.
__device__ __forceinline__ void linear_tile(const EqdLinJob& J, const int* __restrict__ src_local, int out_local,
                                            LinSmem& sm, float (*Lb)[16 * LIN_S], int row0) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int M = J.M;
    const int mbn = (M + 15) >> 4;
    const int rowi = row0 + l15;
    const bool rv = rowi < J.rows;
    // this wave's output blocks and their epilogue operands (prefetched: latency hides under the GEMM)
    const int mbs[2] = {wave, wave + 4};
    const bool own[2] = {wave < mbn, wave + 4 < mbn};
    float bias[2][4], lg[2][4], lb[2][4], res[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * mbs[i] + 4 * g + r;
            const bool ok = own[i] && f < M;
            bias[i][r] = (ok && J.bias) ? J.bias[f] : 0.f;
            lg[i][r] = (ok && J.ln_g) ? J.ln_g[f] : 0.f;
            lb[i][r] = (ok && J.ln_g) ? J.ln_b[f] : 0.f;
            res[i][r] = (ok && J.R && rv) ? J.R[(size_t)rowi * J.ldr + f] : 0.f;
        }
    f32x4 acc[2] = {f4zero(), f4zero()}, acc2[2] = {f4zero(), f4zero()};

    // ---- pipelined (source, K chunk) steps ------------------------------------------------------------
    LinRegs R;
    int s = 0, k0 = 0;
    lin_load(J, J.s[0], src_local && src_local[0] >= 0, 0, row0, t, R);
    while (s < J.nsrc) {
        const EqdLinSrc& S = J.s[s];
        const bool local = src_local && src_local[s] >= 0;
        const int Kc = (S.K - k0 < LIN_KC) ? S.K - k0 : LIN_KC;
        __syncthreads();                  // previous chunk's fragment reads are done
        lin_store(S, local, t, R, sm.Xl, sm.Wl);
        __syncthreads();
        int ns = s, nk0 = k0 + LIN_KC;    // next step
        if (nk0 >= S.K) {
            ns = s + 1;
            nk0 = 0;
        }
        if (ns < J.nsrc) lin_load(J, J.s[ns], src_local && src_local[ns] >= 0, nk0, row0, t, R);
        const float* __restrict__ Xs = local ? &Lb[src_local[s]][k0] : sm.Xl;
        const int nks = (Kc + 3) >> 2;
        int ks = 0;
        // 4 k-steps per trip: the 12 LDS reads are issued together and the MFMAs alternate between two
        // accumulator sets (a single dependent chain would leave the matrix pipe idle 3 passes out of 4)
        for (; ks + 4 <= nks; ks += 4) {
            float b[4], a[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b[u] = Xs[l15 * LIN_S + 4 * (ks + u) + g];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i][u] = own[i] ? sm.Wl[(16 * mbs[i] + l15) * LIN_S + 4 * (ks + u) + g] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (own[i]) {
                    acc[i] = mfma4(a[i][0], b[0], acc[i]);
                    acc2[i] = mfma4(a[i][1], b[1], acc2[i]);
                    acc[i] = mfma4(a[i][2], b[2], acc[i]);
                    acc2[i] = mfma4(a[i][3], b[3], acc2[i]);
                }
        }
        for (; ks < nks; ++ks) {
            const float b = Xs[l15 * LIN_S + 4 * ks + g];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (own[i]) acc[i] = mfma4(sm.Wl[(16 * mbs[i] + l15) * LIN_S + 4 * ks + g], b, acc[i]);
        }
        s = ns;
        k0 = nk0;
    }

    // ---- epilogue in F-layout: feature f = 16 mb + 4 g + r, row = rowi ------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * mbs[i] + 4 * g + r;
            float v = (acc[i][r] + acc2[i][r]) + bias[i][r];
            if (J.act) v = lrelu(v, J.slope);
            acc[i][r] = (own[i] && f < M) ? v : 0.f;
        }
    if (J.ln_g) {
        const float invM = 1.f / (float)M;
        float sm_ = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm_ += acc[i][r];
        sm_ = group_sum(sm_);
        if (g == 0) sm.stat[wave][l15] = sm_;
        __syncthreads();
        const float mean = (sm.stat[0][l15] + sm.stat[1][l15] + sm.stat[2][l15] + sm.stat[3][l15]) * invM;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * mbs[i] + 4 * g + r;
                const float dlt = (own[i] && f < M) ? acc[i][r] - mean : 0.f;
                q += dlt * dlt;
            }
        q = group_sum(q);
        __syncthreads();
        if (g == 0) sm.stat[wave][l15] = q;
        __syncthreads();
        const float rstd =
            1.f / sqrtf((sm.stat[0][l15] + sm.stat[1][l15] + sm.stat[2][l15] + sm.stat[3][l15]) * invM + J.ln_eps);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * mbs[i] + 4 * g + r;
                if (own[i] && f < M) {
                    const float v = acc[i][r];
                    if (J.pre_ln && rv) J.pre_ln[(size_t)rowi * J.ld_pre + f] = v;
                    acc[i][r] = (v - mean) * rstd * lg[i][r] + lb[i][r];
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * mbs[i] + 4 * g + r;
            if (own[i] && f < M) {
                const float v = J.alpha * acc[i][r] + J.beta * res[i][r];
                if (J.Y && rv) J.Y[(size_t)rowi * J.ldy + f] = v;
                if (out_local >= 0) Lb[out_local][l15 * LIN_S + f] = v;
            }
        }
}
