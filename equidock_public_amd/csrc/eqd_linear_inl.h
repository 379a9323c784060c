// Row-local linear tile shared by k_linear (one job per workgroup) and k_rowchain (a sequence of jobs on
// the same rows whose intermediate results stay in LDS).
//
// A workgroup (4 waves) owns RT tiles of 16 rows.  A job is a sum over sources of X_s W_s^T; every source is
// walked in K chunks of 64 ("full" steps) followed by chunks of <= 16 ("small" steps: the 5 columns left over
// from the 69-wide h0 / layer-0 features).  Per step the weights are staged ONCE for the RT row tiles:
//   * all global loads of a step are unconditional (rows and weight rows beyond the matrix are clamped to the
//     last valid one and zeroed when written to LDS): a predicated load compiles to an exec-masked branch with
//     a wait behind it, which turns a batch of loads into a chain of round trips (measured: 8 600 -> 3 600
//     shader clocks per step);
//   * full steps move 16-byte vectors along the contiguous axis (4-byte alignment is enough on gfx950);
//   * the loads of step i + 2 are in flight while step i is multiplied (two register sets);
//   * MFMA operands are read from LDS as b128: k-step j of a chunk belongs to lane group g as
//     k = 16 (j >> 2) + 4 g + (j & 3), valid because both operands use the same assignment;
//   * wave w owns output blocks mb = w and w + 4 (M <= 80), so accumulators are complete without a cross-wave
//     reduction; LayerNorm statistics go through LDS.
// Weights are addressed W[m * w_rs + k * w_cs] with w_cs == 1 (forward) or w_rs == 1 (transposed, backward).
#pragma once
#include "eqd_common.h"

#define LIN_S 84    /* LDS row stride: 21 x 16 B (aligned b128), 20 l15 + g hits 64 different banks */
#define LIN_LOCALS 3
#ifdef EQD_TRACE_FINE      /* per-step stamps perturb the pipeline; the chain experiment only stamps job boundaries */
#define LIN_TR(i) EQD_TR(i)
#else
#define LIN_TR(i) do { } while (0)
#endif

template <int RT>
struct LinRegs {
    f32x4 x[RT], xm[RT], w[5];
};
template <int RT>
struct alignas(16) LinSmem {
    float Xl[RT][16 * LIN_S];
    float Wl[80 * LIN_S];
    float stat[RT][EQD_WAVES][16];
};

struct LinStep {
    int s, k0, kc;   // source, first k of the chunk, chunk width (64 = full step, 1..16 = small step)
};
__device__ __forceinline__ int lin_chunk(int rem) { return rem >= 64 ? 64 : (rem < 16 ? rem : 16); }
__device__ __forceinline__ LinStep lin_first(const EqdLinJob& J) {
    LinStep c = {0, 0, lin_chunk(uni(J.s[0].K))};
    return c;
}
__device__ __forceinline__ LinStep lin_next(const EqdLinJob& J, LinStep c) {
    const int nsrc = uni(J.nsrc);
    if (c.s >= nsrc) return c;
    c.k0 += c.kc;
    if (c.k0 >= uni(J.s[c.s].K)) {
        c.s += 1;
        c.k0 = 0;
    }
    c.kc = c.s < nsrc ? lin_chunk(uni(J.s[c.s].K) - c.k0) : 0;
    return c;
}

// ---- loads of one step into registers (nothing is waited for here) -------------------------------------
template <int RT>
__device__ __forceinline__ void lin_load(const EqdLinJob& J, const EqdLinSrc& S_, bool local, LinStep c, int row0, int t,
                                         LinRegs<RT>& R) {
    const int tr = t >> 4, tc = t & 15;
    const EqdLinSrc S = uni(S_);
    const bool kfast = (S.w_cs == 1);
    const int M = uni(J.M), rows = uni(J.rows);
    if (c.kc == 64) {
        if (!local) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                int row = row0 + 16 * rt + tr;
                row = row < rows ? row : rows - 1;
                const size_t o = (size_t)row * S.ldx + c.k0 + 4 * tc;
                R.x[rt] = *(const EQD_GAS f4v*)(S.X + o);
                if (S.mask) R.xm[rt] = *(const EQD_GAS f4v*)(S.mask + o);
            }
        }
        if (kfast) {      // thread: weight rows m = tr + 16 j, columns k0 + 4 tc ..
            const float* __restrict__ Wb = S.W + c.k0 + 4 * tc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int m = tr + 16 * j;
                m = m < M ? m : M - 1;
                R.w[j] = *(const EQD_GAS f4v*)(Wb + (size_t)m * S.w_rs);
            }
            if (M > 64) {
                int m = 64 + tr;
                m = m < M ? m : M - 1;
                R.w[4] = *(const EQD_GAS f4v*)(Wb + (size_t)m * S.w_rs);
            }
        } else {          // transposed: thread: k = tr + 16 j, weight rows m = 4 tc .. (tail / beyond M: ld4u)
            const float* __restrict__ Wb = S.W + (size_t)c.k0 * S.w_cs;
            if (M >= 64) {                  // every 4-wide row segment below 64 is complete: plain vector loads
#pragma unroll
                for (int j = 0; j < 4; ++j) R.w[j] = *(const EQD_GAS f4v*)(Wb + (size_t)(tr + 16 * j) * S.w_cs + 4 * tc);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    R.w[j] = ld4u_raw(Wb + (size_t)(tr + 16 * j) * S.w_cs + 4 * tc, M - 4 * tc, S.W);
            }
            if (M > 64) {                   // rows 64 .. 79: k = t >> 2, m = 64 + 4 (t & 3)
                const int m = 64 + 4 * (t & 3);
                R.w[4] = ld4u_raw(Wb + (size_t)(t >> 2) * S.w_cs + m, M - m, S.W);
            }
        }
        return;
    }
    // small step: one element per (row, k) / (weight row, k); k clamped into the chunk
    const int kk = c.k0 + (tc < c.kc ? tc : c.kc - 1);
    if (!local) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int row = row0 + 16 * rt + tr;
            row = row < rows ? row : rows - 1;
            const size_t o = (size_t)row * S.ldx + kk;
            R.x[rt][0] = ((const EQD_GAS float*)S.X)[o];
            if (S.mask) R.xm[rt][0] = ((const EQD_GAS float*)S.mask)[o];
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int m = tr + 16 * j;
        m = m < M ? m : M - 1;
        R.w[j][0] = ((const EQD_GAS float*)S.W)[(size_t)m * S.w_rs + (size_t)kk * S.w_cs];
    }
}

// ---- registers -> LDS: Xl[rt][row][k]; Wl[m][k], except full steps of transposed sources: Wl[k][m]; zero padded ------------------------------
template <int RT>
__device__ __forceinline__ void lin_store(const EqdLinJob& J, const EqdLinSrc& S, bool local, LinStep c, int t,
                                          const LinRegs<RT>& R, LinSmem<RT>& sm) {
    const int tr = t >> 4, tc = t & 15;
    const bool kfast = (uni(S.w_cs) == 1);
    const bool masked = uni(S.mask) != nullptr;
    const int M = uni(J.M);
    const float slope = uni(J.slope);
    float* __restrict__ Wl = sm.Wl;
    if (c.kc == 64) {
        if (!local) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 v = R.x[rt];
                if (masked) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(R.xm[rt][i], slope);
                }
                *(f32x4*)&sm.Xl[rt][tr * LIN_S + 4 * tc] = v;
            }
        }
        const f32x4 z = f4zero();
        if (kfast) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)&Wl[(tr + 16 * j) * LIN_S + 4 * tc] = (tr + 16 * j < M) ? R.w[j] : z;
            if (M > 64) *(f32x4*)&Wl[(64 + tr) * LIN_S + 4 * tc] = (64 + tr < M) ? R.w[4] : z;
        } else {       // the vector runs along m: stored as Wl[k][m] (conflict-free b128; transposing here would be 16
                       // scalar stores with an 8-way bank conflict, +770 clocks per step); lin_mma reads it accordingly
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = R.w[j];
                if (M < 64) {
                    const float4 f = ld4u_fix(R.w[j], M - 4 * tc);
                    v = f32x4{f.x, f.y, f.z, f.w};
                }
                *(f32x4*)&Wl[(tr + 16 * j) * LIN_S + 4 * tc] = v;
            }
            if (M > 64) {
                const int m = 64 + 4 * (t & 3), k = t >> 2;
                const float4 f = ld4u_fix(R.w[4], M - m);
                *(f32x4*)&Wl[k * LIN_S + m] = f32x4{f.x, f.y, f.z, f.w};
            }
        }
        return;
    }
    const bool kv = tc < c.kc;
    if (!local) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float v = R.x[rt][0];
            if (masked) v *= lrelu_grad(R.xm[rt][0], slope);
            sm.Xl[rt][tr * LIN_S + tc] = kv ? v : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int m = tr + 16 * j;
        Wl[m * LIN_S + tc] = (kv && m < M) ? R.w[j][0] : 0.f;
    }
}

// acc[rt][i] (+)= W-fragment x X-fragment of the chunk for the wave's NOWN output blocks; no per-lane predicates.
// The weight fragments are read once for the RT row tiles.  MFMAs alternate between two accumulator sets (a single
// dependent chain leaves the matrix pipe idle).
template <int RT, int NOWN, int NQ, bool WT>
__device__ __forceinline__ void lin_mma(f32x4 (&acc)[RT][2], f32x4 (&acc2)[RT][2], const float* (&Xs)[RT],
                                        const float* __restrict__ Wl, const int (&mb)[2], int l15, int g) {
    // WT: the weights lie as Wl[k][m] (full steps of transposed sources): the 4 k-values of a lane are 4 scalar reads
    f32x4 a[NOWN][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < NOWN; ++i) {
            if (WT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[i][q][j] = Wl[(16 * q + 4 * g + j) * LIN_S + 16 * mb[i] + l15];
            } else {
                a[i][q] = *(const f32x4*)&Wl[(16 * mb[i] + l15) * LIN_S + 16 * q + 4 * g];
            }
        }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        f32x4 b[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[q] = *(const f32x4*)&Xs[rt][l15 * LIN_S + 16 * q + 4 * g];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < NOWN; ++i) {
                acc[rt][i] = mfma4(a[i][q][0], b[q][0], acc[rt][i]);
                acc2[rt][i] = mfma4(a[i][q][1], b[q][1], acc2[rt][i]);
                acc[rt][i] = mfma4(a[i][q][2], b[q][2], acc[rt][i]);
                acc2[rt][i] = mfma4(a[i][q][3], b[q][3], acc2[rt][i]);
            }
    }
}

// One linear job on rows row0 .. row0 + 16 RT - 1.  src_local[i] >= 0: source i is the LDS tile Lb[rt][src_local[i]]
// ([16][LIN_S], written by an earlier job of the chain; K <= 80); out_local >= 0: the result is also left in
// Lb[rt][out_local].  Must be called by all 256 threads of the workgroup.
// RA / have_first / has_next: a row chain hands the registers of this job's first step in already loaded (have_first)
// and names the next linear job (Jn, its LDS-source flags src_local_n), whose first step is fetched into RA before this
// job's epilogue - otherwise every job of a chain starts with a fully exposed memory round trip.
template <int RT>
__device__ __forceinline__ void linear_tile(const EqdLinJob& J, const int* __restrict__ src_local, int out_local,
                                            LinSmem<RT>& sm, float (*Lb)[LIN_LOCALS][16 * LIN_S], int row0,
                                            LinRegs<RT>& RA, bool have_first, bool has_next, const EqdLinJob& Jn,
                                            const int* __restrict__ src_local_n, int trace_slot = 255) {
    (void)trace_slot;
    EQD_TR(trace_slot);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // the job header, moved to SGPRs once (see uni())
    const int M = uni(J.M), rows = uni(J.rows), nsrc = uni(J.nsrc), act = uni(J.act);
    const float* const jbias = uni(J.bias);
    const float* const jlng = uni(J.ln_g);
    const float* const jlnb = uni(J.ln_b);
    const float* const jR = uni(J.R);
    float* const jY = uni(J.Y);
    float* const jpre = uni(J.pre_ln);
    const int ldr = uni(J.ldr), ldy = uni(J.ldy), ld_pre = uni(J.ld_pre);
    const float alpha = uni(J.alpha), beta = uni(J.beta), slope = uni(J.slope), ln_eps = uni(J.ln_eps);
    const int mbn = (M + 15) >> 4;
    // this wave's output blocks and their epilogue operands (fetched now: the latency hides under the GEMM)
    const int mbs[2] = {wave, wave + 4};
    const bool own[2] = {wave < mbn, wave + 4 < mbn};
    f32x4 bias[2], lg[2], lb[2], res[RT][2];
    int nf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f0 = 16 * mbs[i] + 4 * g;
        nf[i] = own[i] ? M - f0 : 0;            // valid features at f0 (<= 0: none)
        bias[i] = jbias ? ld4u_raw(jbias + f0, nf[i], jbias) : f4zero();
        lg[i] = jlng ? ld4u_raw(jlng + f0, nf[i], jlng) : f4zero();
        lb[i] = jlng ? ld4u_raw(jlnb + f0, nf[i], jlnb) : f4zero();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int row = row0 + 16 * rt + l15;
            row = row < rows ? row : rows - 1;
            res[rt][i] = jR ? ld4u_raw(jR + (size_t)row * ldr + f0, nf[i], jR) : f4zero();
        }
    }
    f32x4 acc[RT][2], acc2[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[rt][i] = acc2[rt][i] = f4zero();

    // ---- lean pipeline (M == 64, every source at least 64 wide - all jobs of the layers >= 1 except the 69-wide h0
    //      gradient): every load address is computed ONCE per job and kept in VGPRs, the source loop is unrolled, and a
    //      step is: barrier, 5 b128 LDS stores, barrier, 5 loads for the step after next, 8 LDS reads, 16 MFMAs.  The
    //      generic pipeline below spends ~3 500 clocks per step on descriptor handling and bookkeeping; the floor measured
    //      with static addressing is 1 350 (profiles/exp_step_floor.hip). ----------------------------------------------------
    bool lean = (RT == 1) && (M == 64);     // with two row tiles the extra address registers spill (256-VGPR budget)
#pragma unroll
    for (int si = 0; si < EQD_MAX_SRC; ++si)
        if (si < nsrc && uni(J.s[si].K) < 64) lean = false;
    LinRegs<RT> RB;
    int tr_i = 0;
    (void)tr_i;
    auto generic_pipeline = [&]() {
        // ---- pipelined (source, K chunk) steps: the loads of step i + 2 are issued while step i is multiplied ----
        auto is_local = [&](int si) { return src_local && uni(src_local[si]) >= 0; };
        // one step: RX holds step `c`; after it has been written to LDS it is refilled with step `n2`
        auto step = [&](LinStep c, LinRegs<RT>& RX, LinStep n2) {
            const EqdLinSrc& S = J.s[c.s];
            const bool local = is_local(c.s);
            __syncthreads();                  // previous chunk's fragment reads are done
            LIN_TR(tr_i++);
            lin_store<RT>(J, S, local, c, t, RX, sm);
            LIN_TR(tr_i++);
            __syncthreads();
            LIN_TR(tr_i++);
            if (n2.s < nsrc) lin_load<RT>(J, J.s[n2.s], is_local(n2.s), n2, row0, t, RX);
            LIN_TR(tr_i++);
            const float* Xs[RT];
            const int loc = local ? uni(src_local[c.s]) : 0;
    #pragma unroll
            for (int rt = 0; rt < RT; ++rt) Xs[rt] = local ? &Lb[rt][loc][c.k0] : sm.Xl[rt];
            if (c.kc == 64) {
                if (uni(S.w_cs) == 1) {
                    if (own[1])
                        lin_mma<RT, 2, 4, false>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    else if (own[0])
                        lin_mma<RT, 1, 4, false>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                } else {
                    if (own[1])
                        lin_mma<RT, 2, 4, true>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    else if (own[0])
                        lin_mma<RT, 1, 4, true>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                }
            } else {
                if (own[1])
                    lin_mma<RT, 2, 1, false>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                else if (own[0])
                    lin_mma<RT, 1, 1, false>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
            }
            LIN_TR(tr_i++);
        };
        LinStep cur = lin_first(J);
        LinStep nx = lin_next(J, cur);
        EQD_TR(trace_slot + 1);
        LIN_TR(tr_i++);
        if (!have_first) lin_load<RT>(J, J.s[0], is_local(0), cur, row0, t, RA);
        if (nx.s < nsrc) lin_load<RT>(J, J.s[nx.s], is_local(nx.s), nx, row0, t, RB);
        LIN_TR(tr_i++);
        while (cur.s < nsrc) {
            LinStep n2 = lin_next(J, nx);
            step(cur, RA, n2);
            cur = nx;
            nx = n2;
            if (cur.s >= nsrc) break;
            n2 = lin_next(J, nx);
            step(cur, RB, n2);
            cur = nx;
            nx = n2;
        }
    };
    if constexpr (RT == 1) {
        if (lean) {
            const int tr = t >> 4, tc = t & 15;
            // addresses of row tile 0; tile rt is drow[rt] rows further (rows beyond the matrix are clamped to its last row)
            unsigned long long xa[EQD_MAX_SRC], ma[EQD_MAX_SRC], wa[EQD_MAX_SRC];
            int wst[EQD_MAX_SRC], locs[EQD_MAX_SRC], ldxb[EQD_MAX_SRC];
            int rowc0 = row0 + tr, drow[RT];
            rowc0 = rowc0 < rows ? rowc0 : rows - 1;
    #pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                int r = row0 + 16 * rt + tr;
                r = r < rows ? r : rows - 1;
                drow[rt] = r - rowc0;
            }
            unsigned kfm = 0u, mkm = 0u, remm = 0u;
    #pragma unroll
            for (int si = 0; si < EQD_MAX_SRC; ++si) {
                wst[si] = 0;
                locs[si] = -1;
                ldxb[si] = 0;
                wa[si] = xa[si] = ma[si] = 0ull;
                if (si < nsrc) {
                    const EqdLinSrc S = uni(J.s[si]);
                    locs[si] = src_local ? uni(src_local[si]) : -1;
                    const bool kf = (S.w_cs == 1);
                    const int stride = kf ? S.w_rs : S.w_cs;
                    if (kf) kfm |= 1u << si;
                    if (S.mask) mkm |= 1u << si;
                    if (S.K > 64) remm |= 1u << si;
                    wa[si] = (unsigned long long)(S.W + (size_t)tr * stride + 4 * tc);
                    wst[si] = 64 * stride;      // bytes between weight rows tr + 16 j
                    if (locs[si] < 0) {
                        ldxb[si] = 4 * S.ldx;
                        xa[si] = (unsigned long long)(S.X + (size_t)rowc0 * S.ldx + 4 * tc);
                        ma[si] = S.mask ? (unsigned long long)(S.mask + (size_t)rowc0 * S.ldx + 4 * tc) : 0ull;
                    }
                }
            }
            auto load = [&](int si, LinRegs<RT>& R) {       // si is a compile-time constant at every call
                if (locs[si] < 0) {
    #pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const unsigned long long off = (unsigned long long)((long long)drow[rt] * ldxb[si]);
                        R.x[rt] = *(const EQD_GAS f4v*)(xa[si] + off);
                        if ((mkm >> si) & 1u) R.xm[rt] = *(const EQD_GAS f4v*)(ma[si] + off);
                    }
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) R.w[j] = *(const EQD_GAS f4v*)(wa[si] + (unsigned long long)(unsigned)(j * wst[si]));
            };
            EQD_TR(trace_slot + 1);
            LIN_TR(tr_i++);
            if (!have_first) load(0, RA);
            if (1 < nsrc) load(1, RB);
            LIN_TR(tr_i++);
    #pragma unroll
            for (int si = 0; si < EQD_MAX_SRC; ++si) {
                if (si < nsrc) {
                    LinRegs<RT>& RX = (si & 1) ? RB : RA;
                    __syncthreads();                  // previous step's fragment reads are done
                    LIN_TR(tr_i++);
                    if (locs[si] < 0) {
    #pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            f32x4 v = RX.x[rt];
                            if ((mkm >> si) & 1u) {
    #pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(RX.xm[rt][i], slope);
                            }
                            *(f32x4*)&sm.Xl[rt][tr * LIN_S + 4 * tc] = v;
                        }
                    }
    #pragma unroll
                    for (int j = 0; j < 4; ++j) *(f32x4*)&sm.Wl[(tr + 16 * j) * LIN_S + 4 * tc] = RX.w[j];   // Wl[m][k] or Wl[k][m]
                    LIN_TR(tr_i++);
                    __syncthreads();
                    LIN_TR(tr_i++);
                    if (si + 2 < EQD_MAX_SRC) {
                        if (si + 2 < nsrc) load((si + 2) % EQD_MAX_SRC, RX);
                    }
                    LIN_TR(tr_i++);
                    const float* Xs[RT];
    #pragma unroll
                    for (int rt = 0; rt < RT; ++rt) Xs[rt] = locs[si] >= 0 ? &Lb[rt][locs[si]][0] : sm.Xl[rt];
                    if ((kfm >> si) & 1u)
                        lin_mma<RT, 1, 4, false>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    else
                        lin_mma<RT, 1, 4, true>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    LIN_TR(tr_i++);
                }
            }
            // what is left of sources wider than 64 (the 5 columns beyond 64 of the 69-wide h0): 16-wide steps, not pipelined
            if (remm) {
    #pragma unroll
                for (int si = 0; si < EQD_MAX_SRC; ++si) {
                    if (si < nsrc && ((remm >> si) & 1u)) {
                        const EqdLinSrc S = uni(J.s[si]);
                        for (int k0 = 64; k0 < S.K;) {
                            const LinStep c = {si, k0, lin_chunk(S.K - k0)};
                            lin_load<RT>(J, J.s[si], locs[si] >= 0, c, row0, t, RB);
                            __syncthreads();
                            lin_store<RT>(J, J.s[si], locs[si] >= 0, c, t, RB, sm);
                            __syncthreads();
                            const float* Xs[RT];
    #pragma unroll
                            for (int rt = 0; rt < RT; ++rt) Xs[rt] = locs[si] >= 0 ? &Lb[rt][locs[si]][k0] : sm.Xl[rt];
                            lin_mma<RT, 1, 1, false>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                            k0 += c.kc;
                        }
                    }
                }
            }
        } else {
            generic_pipeline();
        }
    } else {
        generic_pipeline();
    }
    EQD_TR(trace_slot + 2);
    if (has_next) lin_load<RT>(Jn, Jn.s[0], src_local_n && uni(src_local_n[0]) >= 0, lin_first(Jn), row0, t, RA);
    EQD_TR(trace_slot + 3);
    // ---- epilogue in F-layout: feature f = 16 mb + 4 g + r, row = row0 + 16 rt + l15 -----------------------------
    // `plain`: M is a multiple of 4, so an owned 4-feature group is always complete - no tail handling, 16-byte stores
    const bool plain = (M & 3) == 0;
    float4 bs[2], lgv[2], lbv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (plain) {
            bs[i] = make_float4(bias[i][0], bias[i][1], bias[i][2], bias[i][3]);
            lgv[i] = make_float4(lg[i][0], lg[i][1], lg[i][2], lg[i][3]);
            lbv[i] = make_float4(lb[i][0], lb[i][1], lb[i][2], lb[i][3]);
        } else {
            bs[i] = ld4u_fix(bias[i], nf[i]);
            lgv[i] = ld4u_fix(lg[i], nf[i]);
            lbv[i] = ld4u_fix(lb[i], nf[i]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int rowi = row0 + 16 * rt + l15;
        const bool rv = rowi < rows;
        float v[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float bb[4] = {bs[i].x, bs[i].y, bs[i].z, bs[i].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = (acc[rt][i][r] + acc2[rt][i][r]) + bb[r];
                if (act) y = lrelu(y, slope);
                v[i][r] = r < nf[i] ? y : 0.f;
            }
        }
        if (jlng) {
            const float invM = 1.f / (float)M;
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) s1 += v[i][r];
            s1 = group_sum(s1);
            if (g == 0) sm.stat[rt][wave][l15] = s1;
            __syncthreads();
            const float mean =
                (sm.stat[rt][0][l15] + sm.stat[rt][1][l15] + sm.stat[rt][2][l15] + sm.stat[rt][3][l15]) * invM;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dlt = r < nf[i] ? v[i][r] - mean : 0.f;
                    q += dlt * dlt;
                }
            q = group_sum(q);
            __syncthreads();
            if (g == 0) sm.stat[rt][wave][l15] = q;
            __syncthreads();
            const float rstd = 1.f / sqrtf((sm.stat[rt][0][l15] + sm.stat[rt][1][l15] + sm.stat[rt][2][l15] +
                                            sm.stat[rt][3][l15]) * invM + ln_eps);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float gg[4] = {lgv[i].x, lgv[i].y, lgv[i].z, lgv[i].w};
                const float be[4] = {lbv[i].x, lbv[i].y, lbv[i].z, lbv[i].w};
                if (plain && nf[i] > 0 && jpre && rv)
                    *(EQD_GAS f4v*)&jpre[(size_t)rowi * ld_pre + 16 * mbs[i] + 4 * g] = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nf[i]) {
                        const int f = 16 * mbs[i] + 4 * g + r;
                        if (!plain && jpre && rv) ((EQD_GAS float*)jpre)[(size_t)rowi * ld_pre + f] = v[i][r];
                        v[i][r] = (v[i][r] - mean) * rstd * gg[r] + be[r];
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 rr;
            if (plain)
                rr = make_float4(res[rt][i][0], res[rt][i][1], res[rt][i][2], res[rt][i][3]);
            else
                rr = ld4u_fix(res[rt][i], nf[i]);
            const float rs[4] = {rr.x, rr.y, rr.z, rr.w};
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = alpha * v[i][r] + beta * rs[r];
            const int f0 = 16 * mbs[i] + 4 * g;
            if (plain) {
                if (nf[i] > 0) {
                    const f32x4 yv = {y[0], y[1], y[2], y[3]};
                    if (jY && rv) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = yv;
                    if (out_local >= 0) *(f32x4*)&Lb[rt][out_local][l15 * LIN_S + f0] = yv;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nf[i]) {
                        if (jY && rv) ((EQD_GAS float*)jY)[(size_t)rowi * ldy + f0 + r] = y[r];
                        if (out_local >= 0) Lb[rt][out_local][l15 * LIN_S + f0 + r] = y[r];
                    }
            }
        }
    }
    LIN_TR(tr_i++);
}
