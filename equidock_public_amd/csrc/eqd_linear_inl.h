// Row-local linear tile shared by k_linear (one job per workgroup) and k_rowchain (a sequence of jobs on
// the same 16 rows whose intermediate results stay in LDS).
#pragma once
#include "eqd_common.h"

#define LIN_KC 80   /* K chunk staged per step (every source of the IEGMN path has K <= 69) */
#define LIN_S 84    /* LDS row stride: 21 x 16 B (b128 stores), 20 l15 + g hits 64 different banks */
#define LIN_LOCALS 4
// A workgroup owns 16 rows.  Per source chunk: ALL loads (X tile 16 x Kc, weight slice M x Kc) are
// issued together by the 256 threads as 16-byte vectors along the contiguous axis (4-byte alignment is
// enough, see ld4u), parked in registers while the previous chunk is multiplied, then written to LDS; MFMA
// operands come from LDS.  Wave w owns output blocks mb = w and w + 4, so accumulators are complete (no
// cross-wave reduction); LayerNorm statistics are exchanged through LDS.  These GEMMs are tiny
// (3200 x 64..384 x 64 at config B): what matters is one memory round trip per source and few load
// instructions (a 64 x 64 weight slice is 4 vector loads per thread, not 16 scalar ones).
// Weights are addressed W[m * w_rs + k * w_cs] with w_cs == 1 (forward) or w_rs == 1 (transposed, backward).
#define LIN_TR(i) EQD_TR(i)
struct LinRegs {
    f32x4 x[2], xm[2], w[5][2];   // raw 16-byte loads; ld4u_fix is applied when they are written to LDS
};
struct alignas(16) LinSmem {
    float Xl[16 * LIN_S];
    float Wl[80 * LIN_S];
    float stat[EQD_WAVES][16];
};
// valid floats of the thread's X segment h / weight segment (j, h) of the step (source S, chunk k0)
__device__ __forceinline__ int lin_nx(const EqdLinJob& J, const EqdLinSrc& S, int k0, int row0, int t, int h) {
    const int Kc = (S.K - k0 < LIN_KC) ? S.K - k0 : LIN_KC;
    const int tr = t >> 4, tc = t & 15;
    return (row0 + tr < J.rows && (h == 0 || tc < 4)) ? Kc - (4 * tc + 64 * h) : 0;
}
__device__ __forceinline__ int lin_nw(const EqdLinJob& J, const EqdLinSrc& S, int k0, int t, int j, int h) {
    const int Kc = (S.K - k0 < LIN_KC) ? S.K - k0 : LIN_KC;
    const int tr = t >> 4, tc = t & 15;
    const bool kfast = (S.w_cs == 1);
    // a = index along the strided axis (tr + 16 j), c = first index of the 4-wide segment along the
    // contiguous axis (k when kfast, m otherwise)
    const int na = kfast ? J.M : Kc, nc = kfast ? Kc : J.M;
    return (tr + 16 * j < na && (h == 0 || tc < 4)) ? nc - (4 * tc + 64 * h) : 0;
}
// Interior steps (all 16 rows valid, a full 64-wide K chunk, 64 outputs - every step of layers >= 1) take a
// path without any per-lane predicate: 1 + 4 unconditional 16-byte loads per thread.  Conditional loads
// compile to exec-masked branches with a wait behind each, which serialises the batch (measured: 8600 shader
// clocks per step with predicated loads).
__device__ __forceinline__ bool lin_fast(const EqdLinJob& J, const EqdLinSrc& S, int k0, int row0) {
    return S.K - k0 == 64 && J.M == 64 && row0 + 16 <= J.rows;
}
__device__ __forceinline__ void lin_load(const EqdLinJob& J, const EqdLinSrc& S, bool local, int k0, int row0, int t,
                                         LinRegs& R) {
    const int tr = t >> 4, tc = t & 15;
    const int row = row0 + tr;
    const bool kfast = (S.w_cs == 1);
    const int astride = kfast ? S.w_rs : S.w_cs;
    const float* __restrict__ Wb = S.W + (size_t)k0 * S.w_cs;
    if (lin_fast(J, S, k0, row0)) {
        if (!local) {
            const size_t o = (size_t)row * S.ldx + k0 + 4 * tc;
            R.x[0] = *(const f4v*)(S.X + o);
            if (S.mask) R.xm[0] = *(const f4v*)(S.mask + o);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) R.w[j][0] = *(const f4v*)(Wb + (size_t)(tr + 16 * j) * astride + 4 * tc);
        return;
    }
    if (!local) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = lin_nx(J, S, k0, row0, t, h);
            const size_t o = (size_t)(row < J.rows ? row : 0) * S.ldx + k0 + 4 * tc + 64 * h;
            R.x[h] = ld4u_raw(S.X + o, n, S.X);
            if (S.mask) R.xm[h] = ld4u_raw(S.mask + o, n, S.mask);
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int a = tr + 16 * j;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = lin_nw(J, S, k0, t, j, h);
            R.w[j][h] = ld4u_raw(Wb + (size_t)(n > 0 ? a : 0) * astride + 4 * tc + 64 * h, n, S.W);
        }
    }
}
__device__ __forceinline__ void lin_store(const EqdLinJob& J, const EqdLinSrc& S, bool local, int k0, int row0, int t,
                                          const LinRegs& R, float* __restrict__ Xl, float* __restrict__ Wl) {
    const int tr = t >> 4, tc = t & 15;
    // weights always land as Wl[m][k]
    if (lin_fast(J, S, k0, row0)) {
        if (!local) {
            f32x4 v = R.x[0];
            if (S.mask) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(R.xm[0][i], J.slope);
            }
            *(f32x4*)&Xl[tr * LIN_S + 4 * tc] = v;
        }
        if (S.w_cs == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)&Wl[(tr + 16 * j) * LIN_S + 4 * tc] = R.w[j][0];
        } else {       // the vector runs along m: transpose on the way in (4-way bank conflict, 16 short stores)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) Wl[(4 * tc + i) * LIN_S + tr + 16 * j] = R.w[j][0][i];
        }
        return;
    }
    if (!local) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && tc >= 4) continue;
            const int n = lin_nx(J, S, k0, row0, t, h);
            float4 v = ld4u_fix(R.x[h], n);
            if (S.mask) {
                const float4 mk = ld4u_fix(R.xm[h], n);
                v.x *= lrelu_grad(mk.x, J.slope); v.y *= lrelu_grad(mk.y, J.slope);
                v.z *= lrelu_grad(mk.z, J.slope); v.w *= lrelu_grad(mk.w, J.slope);
            }
            *(float4*)&Xl[tr * LIN_S + 4 * tc + 64 * h] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int a = tr + 16 * j;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && tc >= 4) continue;
            const float4 v = ld4u_fix(R.w[j][h], lin_nw(J, S, k0, t, j, h));
            const int c = 4 * tc + 64 * h;
            if (S.w_cs == 1) {
                *(float4*)&Wl[a * LIN_S + c] = v;
            } else {       // a = k, c = m
                Wl[(c + 0) * LIN_S + a] = v.x;
                Wl[(c + 1) * LIN_S + a] = v.y;
                Wl[(c + 2) * LIN_S + a] = v.z;
                Wl[(c + 3) * LIN_S + a] = v.w;
            }
        }
    }
}

// acc (+)= W-fragment x X-fragment over the chunk for the wave's NOWN output blocks; no per-lane predicates.
// The 4 k-values of MFMA step j belong to lane groups g = 0..3 as k = 16 (j >> 2) + 4 g + (j & 3): any
// assignment is valid as long as both operands use it, and this one makes the four steps j = 4 q .. 4 q + 3 of
// a lane ONE 16-byte LDS read per operand.  All reads of the chunk are issued before the first MFMA; the MFMAs
// alternate between two accumulator sets (a single dependent chain leaves the matrix pipe idle).
template <int NOWN>
__device__ __forceinline__ void lin_mma(f32x4 (&acc)[2], f32x4 (&acc2)[2], const float* __restrict__ Xs,
                                        const float* __restrict__ Wl, const int (&mb)[2], int Kc, int l15, int g) {
    const int nq = (Kc + 15) >> 4;          // 16 k-values per q (zero padded in LDS up to LIN_KC = 80)
    if (nq == 4) {
        f32x4 b[4], a[NOWN][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b[q] = *(const f32x4*)&Xs[l15 * LIN_S + 16 * q + 4 * g];
#pragma unroll
            for (int i = 0; i < NOWN; ++i) a[i][q] = *(const f32x4*)&Wl[(16 * mb[i] + l15) * LIN_S + 16 * q + 4 * g];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NOWN; ++i) {
                acc[i] = mfma4(a[i][q][0], b[q][0], acc[i]);
                acc2[i] = mfma4(a[i][q][1], b[q][1], acc2[i]);
                acc[i] = mfma4(a[i][q][2], b[q][2], acc[i]);
                acc2[i] = mfma4(a[i][q][3], b[q][3], acc2[i]);
            }
        return;
    }
    for (int q = 0; q < nq; ++q) {
        const f32x4 b = *(const f32x4*)&Xs[l15 * LIN_S + 16 * q + 4 * g];
#pragma unroll
        for (int i = 0; i < NOWN; ++i) {
            const f32x4 a = *(const f32x4*)&Wl[(16 * mb[i] + l15) * LIN_S + 16 * q + 4 * g];
            acc[i] = mfma4(a[0], b[0], acc[i]);
            acc2[i] = mfma4(a[1], b[1], acc2[i]);
            acc[i] = mfma4(a[2], b[2], acc[i]);
            acc2[i] = mfma4(a[3], b[3], acc2[i]);
        }
    }
}

struct LinStep {
    int s, k0;
};
__device__ __forceinline__ LinStep lin_next(const EqdLinJob& J, LinStep c) {
    if (c.s >= J.nsrc) return c;
    c.k0 += LIN_KC;
    if (c.k0 >= J.s[c.s].K) {
        c.s += 1;
        c.k0 = 0;
    }
    return c;
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

    // ---- pipelined (source, K chunk) steps: the loads of step i + 2 are issued while step i is multiplied ----
    LinRegs RA, RB;
    int tr_i = 0;
    (void)tr_i;
    auto is_local = [&](int si) { return src_local && src_local[si] >= 0; };
    // one step: RX holds step `c`; after it has been written to LDS it is refilled with step `n2`
    auto step = [&](LinStep c, LinRegs& RX, LinStep n2) {
        const EqdLinSrc& S = J.s[c.s];
        const bool local = is_local(c.s);
        const int Kc = (S.K - c.k0 < LIN_KC) ? S.K - c.k0 : LIN_KC;
        __syncthreads();                  // previous chunk's fragment reads are done
        LIN_TR(tr_i++);
        lin_store(J, S, local, c.k0, row0, t, RX, sm.Xl, sm.Wl);
        LIN_TR(tr_i++);
        __syncthreads();
        LIN_TR(tr_i++);
        if (n2.s < J.nsrc) lin_load(J, J.s[n2.s], is_local(n2.s), n2.k0, row0, t, RX);
        LIN_TR(tr_i++);
        const float* __restrict__ Xs = local ? &Lb[src_local[c.s]][c.k0] : sm.Xl;
        if (own[1])
            lin_mma<2>(acc, acc2, Xs, sm.Wl, mbs, Kc, l15, g);
        else if (own[0])
            lin_mma<1>(acc, acc2, Xs, sm.Wl, mbs, Kc, l15, g);
        LIN_TR(tr_i++);
    };
    LinStep cur = {0, 0};
    LinStep nx = lin_next(J, cur);
    LIN_TR(tr_i++);
    lin_load(J, J.s[0], is_local(0), 0, row0, t, RA);
    if (nx.s < J.nsrc) lin_load(J, J.s[nx.s], is_local(nx.s), nx.k0, row0, t, RB);
    LIN_TR(tr_i++);
    while (cur.s < J.nsrc) {
        LinStep n2 = lin_next(J, nx);
        step(cur, RA, n2);
        cur = nx;
        nx = n2;
        if (cur.s >= J.nsrc) break;
        n2 = lin_next(J, nx);
        step(cur, RB, n2);
        cur = nx;
        nx = n2;
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
    LIN_TR(tr_i++);
}
