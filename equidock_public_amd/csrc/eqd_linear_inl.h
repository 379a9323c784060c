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

// ---- job descriptor access ------------------------------------------------------------------------------
// The descriptors live in LDS.  Reading a field as an LDS load + v_readfirstlane costs an LDS round trip (>= 64 clocks,
// in practice serialised by the wait in front of every readfirstlane): the ~40 fields of a one-source job were 2 200
// clocks of prologue in front of a single pipeline step of 1 800 (profiles/exp_trace_chain.py).  Instead lane i fetches
// dword i and dword 64 + i of the descriptor ONCE - two LDS loads in flight together - and every field is a v_readlane
// from those two registers: no memory access, and the next job's words are fetched while the current job runs.
struct JobW {
    int w0, w1;
};
__device__ __forceinline__ JobW jobw_load(const void* desc, int ndw, int lane) {
    const int* p = (const int*)desc;
    JobW W;
    const int i1 = 64 + lane;
    W.w0 = p[lane < ndw ? lane : 0];
    W.w1 = p[i1 < ndw ? i1 : 0];
    return W;
}
__device__ __forceinline__ int jw_i(const JobW& W, int dw) {
    return __builtin_amdgcn_readlane(dw < 64 ? W.w0 : W.w1, dw & 63);
}
__device__ __forceinline__ float jw_f(const JobW& W, int dw) {
    const int v = jw_i(W, dw);
    float f;
    __builtin_memcpy(&f, &v, 4);
    return f;
}
template <class T>
__device__ __forceinline__ T* jw_p(const JobW& W, int dw) {
    const unsigned lo = (unsigned)jw_i(W, dw), hi = (unsigned)jw_i(W, dw + 1);
    return (T*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
#define JW_OFF(type, field) ((int)(offsetof(type, field) / 4))
#define JW_SRC_DW ((int)(sizeof(EqdLinSrc) / 4))
static_assert(sizeof(EqdChainJob) <= 128 * 4, "a job descriptor must fit two dwords per lane");
static_assert(offsetof(EqdChainJob, lin) == 0, "EqdChainJob starts with its EqdLinJob");
__device__ __forceinline__ EqdLinSrc jw_src(const JobW& W, int si) {
    const int b = JW_OFF(EqdLinJob, s) + si * JW_SRC_DW;
    EqdLinSrc S;
    S.X = jw_p<const float>(W, b + JW_OFF(EqdLinSrc, X));
    S.mask = jw_p<const float>(W, b + JW_OFF(EqdLinSrc, mask));
    S.W = jw_p<const float>(W, b + JW_OFF(EqdLinSrc, W));
    S.ldx = jw_i(W, b + JW_OFF(EqdLinSrc, ldx));
    S.K = jw_i(W, b + JW_OFF(EqdLinSrc, K));
    S.w_rs = jw_i(W, b + JW_OFF(EqdLinSrc, w_rs));
    S.w_cs = jw_i(W, b + JW_OFF(EqdLinSrc, w_cs));
    return S;
}

// ---- loads of one step into registers (nothing is waited for here) -------------------------------------
template <int RT>
__device__ __forceinline__ void lin_load_s(const EqdLinSrc S, const int M, const int rows, bool local, LinStep c, int row0,
                                           int t, LinRegs<RT>& R) {      // S, M, rows: already wave-uniform (SGPRs)
    const int tr = t >> 4, tc = t & 15;
    const bool kfast = (S.w_cs == 1);
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

template <int RT>
__device__ __forceinline__ void lin_load(const EqdLinJob& J, const EqdLinSrc& S_, bool local, LinStep c, int row0, int t,
                                         LinRegs<RT>& R) {
    lin_load_s<RT>(uni(S_), uni(J.M), uni(J.rows), local, c, row0, t, R);
}

// ---- registers -> LDS: Xl[rt][row][k]; Wl[m][k], except full steps of transposed sources: Wl[k][m]; zero padded ------------------------------
template <int RT>
__device__ __forceinline__ void lin_store_s(const EqdLinSrc S, const int M, const float slope, bool local, LinStep c, int t,
                                            const LinRegs<RT>& R, LinSmem<RT>& sm) {     // S, M, slope: wave-uniform
    const int tr = t >> 4, tc = t & 15;
    const bool kfast = (S.w_cs == 1);
    const bool masked = S.mask != nullptr;
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

template <int RT>
__device__ __forceinline__ void lin_store(const EqdLinJob& J, const EqdLinSrc& S, bool local, LinStep c, int t,
                                          const LinRegs<RT>& R, LinSmem<RT>& sm) {
    lin_store_s<RT>(uni(S), uni(J.M), uni(J.slope), local, c, t, R, sm);
}

// acc[rt][i] (+)= W-fragment x X-fragment of the chunk for the wave's NOWN output blocks; no per-lane predicates.
// The weight fragments are read once for the RT row tiles.  MFMAs alternate between two accumulator sets (a single
// dependent chain leaves the matrix pipe idle).
template <int RT, int NOWN, int NQ, bool WT, bool BF = false, bool LB = true>
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
#if !defined(EQD_HOSTSIM) && !defined(EQD_NO_LDS_BATCH) && !defined(EQD_NO_LDS_BATCH_ROWEDGE)
        // every fragment read of the step is issued before its first MFMA (left alone the scheduler sinks each read next to
        // its use: read - s_waitcnt lgkmcnt(0) - 4 MFMA, the LDS latency exposed sixteen times per step).  LB: off in the
        // instances that are capped at 256 registers (two workgroups per SIMD pair), where holding every fragment spills.
        if constexpr (LB) __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (BF) {
            // bf16 mode: the 4 k-values a lane holds for chunk q (k = 16 q + 4 g + j) are exactly one operand of
            // v_mfma_f32_16x16x16_bf16: four fp32 instructions become one (inputs rounded to bf16, fp32 accumulate)
            // ... and two consecutive chunks one v_mfma_f32_16x16x32_bf16 (gfx950: 2 x K at the same issue cost)
#pragma unroll
            for (int qp = 0; qp < NQ / 2; ++qp) {
                const s16x8 bp = cat_bf(pack_bf4(b[2 * qp][0], b[2 * qp][1], b[2 * qp][2], b[2 * qp][3]),
                                        pack_bf4(b[2 * qp + 1][0], b[2 * qp + 1][1], b[2 * qp + 1][2], b[2 * qp + 1][3]));
#pragma unroll
                for (int i = 0; i < NOWN; ++i) {
                    const s16x8 ap = cat_bf(pack_bf4(a[i][2 * qp][0], a[i][2 * qp][1], a[i][2 * qp][2], a[i][2 * qp][3]),
                                            pack_bf4(a[i][2 * qp + 1][0], a[i][2 * qp + 1][1], a[i][2 * qp + 1][2], a[i][2 * qp + 1][3]));
                    if (qp & 1) acc2[rt][i] = mfma_bf32(ap, bp, acc2[rt][i]);
                    else acc[rt][i] = mfma_bf32(ap, bp, acc[rt][i]);
                }
            }
            if constexpr (NQ % 2 == 1) {
                constexpr int q = NQ - 1;
                const s16x4 bp = pack_bf4(b[q][0], b[q][1], b[q][2], b[q][3]);
#pragma unroll
                for (int i = 0; i < NOWN; ++i) {
                    const s16x4 ap = pack_bf4(a[i][q][0], a[i][q][1], a[i][q][2], a[i][q][3]);
                    if ((NQ / 2) & 1) acc2[rt][i] = mfma_bf(ap, bp, acc2[rt][i]);
                    else acc[rt][i] = mfma_bf(ap, bp, acc[rt][i]);
                }
            }
        } else {
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
}

// ---- lean job (one row tile, M == 64, every source at least 64 wide: all jobs of the layers >= 1 except the 69-wide h0
//      gradient).  Everything that the general linear_tile below decides per element is a compile-time fact here (one
//      complete output block per wave, 16-byte accesses everywhere), every load address is computed ONCE per job and kept
//      in VGPRs, and the source loop is unrolled; a step is: barrier, 5 b128 LDS stores, barrier, 5 loads for the step
//      after next, 8 LDS reads, 16 MFMAs.  A workgroup at config B has nothing to overlap a job's bookkeeping with
//      (one wave per SIMD), so its instruction count is latency: the general path spends ~3 500 clocks per step and
//      ~2 000 per job on descriptor handling; the floor with static addressing is 1 350 per step
//      (profiles/exp_step_floor.hip).  Arithmetic (order of every sum) is the same as in linear_tile. -----------------
template <bool BF, bool LB = true>
__device__ __forceinline__ void linear_tile_lean(const JobW& W, bool chain, int out_local, LinSmem<1>& sm,
                                                 float (*Lb)[LIN_LOCALS][16 * LIN_S], int row0, LinRegs<1>& RA,
                                                 bool have_first, bool has_next, const JobW& Wn, int trace_slot) {
    (void)trace_slot;
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tr = t >> 4, tc = t & 15;
    const int rows = jw_i(W, LJ(rows)), nsrc = jw_i(W, LJ(nsrc));
    const float slope = jw_f(W, LJ(slope));
    // epilogue operands of this wave's output block (features f0 .. f0 + 3 of row row0 + l15), requested now
    const int f0 = 16 * wave + 4 * g;
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const int rowe = rv ? rowi : rows - 1;
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    const float* const jlng = jw_p<const float>(W, LJ(ln_g));
    const float* const jR = jw_p<const float>(W, LJ(R));
    f32x4 bias = f4zero(), lg = f4zero(), lb = f4zero(), res = f4zero();
    if (jbias) bias = *(const EQD_GAS f4v*)(jbias + f0);
    if (jlng) {
        lg = *(const EQD_GAS f4v*)(jlng + f0);
        lb = *(const EQD_GAS f4v*)(jw_p<const float>(W, LJ(ln_b)) + f0);
    }
    if (jR) res = *(const EQD_GAS f4v*)(jR + (size_t)rowe * jw_i(W, LJ(ldr)) + f0);
    f32x4 acc[1][2], acc2[1][2];
    acc[0][0] = acc[0][1] = acc2[0][0] = acc2[0][1] = f4zero();
    const int mbs[2] = {wave, wave + 4};

    // per-source addresses of this thread: X row tr (clamped), columns 4 tc ..; W rows tr + 16 j (or k rows when transposed)
    unsigned long long xa[EQD_MAX_SRC], ma[EQD_MAX_SRC], wa[EQD_MAX_SRC];
    int wst[EQD_MAX_SRC], locs[EQD_MAX_SRC];
    int rowc0 = row0 + tr;
    rowc0 = rowc0 < rows ? rowc0 : rows - 1;
    unsigned kfm = 0u, mkm = 0u, remm = 0u;
#pragma unroll
    for (int si = 0; si < EQD_MAX_SRC; ++si) {
        if (si < nsrc) {
            const EqdLinSrc S = jw_src(W, si);
            locs[si] = chain ? jw_i(W, JW_OFF(EqdChainJob, src_local) + si) : -1;
            const bool kf = (S.w_cs == 1);
            const int stride = kf ? S.w_rs : S.w_cs;
            if (kf) kfm |= 1u << si;
            if (S.mask) mkm |= 1u << si;
            if (S.K > 64) remm |= 1u << si;
            wa[si] = (unsigned long long)(S.W + (size_t)tr * stride + 4 * tc);
            wst[si] = 64 * stride;      // bytes between weight rows tr + 16 j
            if (locs[si] < 0) {
                xa[si] = (unsigned long long)(S.X + (size_t)rowc0 * S.ldx + 4 * tc);
                ma[si] = S.mask ? (unsigned long long)(S.mask + (size_t)rowc0 * S.ldx + 4 * tc) : 0ull;
            }
        }
    }
    auto load = [&](int si, LinRegs<1>& R) {       // si is a compile-time constant at every call
        if (locs[si] < 0) {
            R.x[0] = *(const EQD_GAS f4v*)(xa[si]);
            if ((mkm >> si) & 1u) R.xm[0] = *(const EQD_GAS f4v*)(ma[si]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) R.w[j] = *(const EQD_GAS f4v*)(wa[si] + (unsigned long long)(unsigned)(j * wst[si]));
    };
    LinRegs<1> RB;
    int tr_i = 0;
    (void)tr_i;
    EQD_TR(trace_slot + 1);
    LIN_TR(tr_i++);
    if (!have_first) load(0, RA);
    if (1 < nsrc) load(1, RB);
    LIN_TR(tr_i++);
#pragma unroll
    for (int si = 0; si < EQD_MAX_SRC; ++si) {
        if (si < nsrc) {
            LinRegs<1>& RX = (si & 1) ? RB : RA;
            __syncthreads();                  // previous step's fragment reads are done
            LIN_TR(tr_i++);
            if (locs[si] < 0) {
                f32x4 v = RX.x[0];
                if ((mkm >> si) & 1u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(RX.xm[0][i], slope);
                }
                *(f32x4*)&sm.Xl[0][tr * LIN_S + 4 * tc] = v;
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
            const float* Xs[1] = {locs[si] >= 0 ? &Lb[0][locs[si]][0] : sm.Xl[0]};
            if ((kfm >> si) & 1u)
                lin_mma<1, 1, 4, false, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
            else
                lin_mma<1, 1, 4, true, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
            LIN_TR(tr_i++);
        }
    }
    // what is left of sources wider than 64 (the 5 columns beyond 64 of the 69-wide h0): 16-wide steps, not pipelined
    if (remm) {
        const int M = 64;
#pragma unroll
        for (int si = 0; si < EQD_MAX_SRC; ++si) {
            if (si < nsrc && ((remm >> si) & 1u)) {
                const EqdLinSrc S = jw_src(W, si);
                for (int k0 = 64; k0 < S.K;) {
                    const LinStep c = {si, k0, lin_chunk(S.K - k0)};
                    lin_load_s<1>(S, M, rows, locs[si] >= 0, c, row0, t, RB);
                    __syncthreads();
                    lin_store_s<1>(S, M, slope, locs[si] >= 0, c, t, RB, sm);
                    __syncthreads();
                    const float* Xs[1] = {locs[si] >= 0 ? &Lb[0][locs[si]][k0] : sm.Xl[0]};
                    lin_mma<1, 1, 1, false, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    k0 += c.kc;
                }
            }
        }
    }
    EQD_TR(trace_slot + 2);
    if (has_next) {      // the next job's first step (same rows) -> RA
        const EqdLinSrc Sn = jw_src(Wn, 0);
        const bool local_n = chain && jw_i(Wn, JW_OFF(EqdChainJob, src_local)) >= 0;
        const LinStep c0 = {0, 0, lin_chunk(Sn.K)};
        lin_load_s<1>(Sn, jw_i(Wn, LJ(M)), rows, local_n, c0, row0, t, RA);
    }
    EQD_TR(trace_slot + 3);
    // ---- epilogue: feature f0 + r of row rowi ------------------------------------------------------------------------
    const int act = jw_i(W, LJ(act));
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = (acc[0][0][r] + acc2[0][0][r]) + bias[r];
        if (act) y = lrelu(y, slope);
        v[r] = y;
    }
    if (const float* const jmul = jw_p<const float>(W, LJ(mul))) {      // dropout factors (training mode only)
        const f32x4 mm = *(const EQD_GAS f4v*)(jmul + (size_t)rowe * jw_i(W, LJ(ld_mul)) + f0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= mm[r];
    }
    if (jlng) {
        const float invM = 1.f / 64.f;
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s1 += v[r];
        s1 = group_sum(s1);
        if (g == 0) sm.stat[0][wave][l15] = s1;
        __syncthreads();
        const float mean = (sm.stat[0][0][l15] + sm.stat[0][1][l15] + sm.stat[0][2][l15] + sm.stat[0][3][l15]) * invM;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dlt = v[r] - mean;
            q += dlt * dlt;
        }
        q = group_sum(q);
        __syncthreads();
        if (g == 0) sm.stat[0][wave][l15] = q;
        __syncthreads();
        const float rstd = 1.f / sqrtf((sm.stat[0][0][l15] + sm.stat[0][1][l15] + sm.stat[0][2][l15] + sm.stat[0][3][l15]) * invM +
                                       jw_f(W, LJ(ln_eps)));
        float* const jpre = jw_p<float>(W, LJ(pre_ln));
        const int ld_pre = jw_i(W, LJ(ld_pre));
        if (jpre && rv) *(EQD_GAS f4v*)&jpre[(size_t)rowi * ld_pre + f0] = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] - mean) * rstd * lg[r] + lb[r];
    }
    const float alpha = jw_f(W, LJ(alpha)), beta = jw_f(W, LJ(beta));
    f32x4 yv;
#pragma unroll
    for (int r = 0; r < 4; ++r) yv[r] = alpha * v[r] + beta * res[r];
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));
    if (jY && rv) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = yv;
    unsigned short* const jYb = jw_p<unsigned short>(W, LJ(Yb));      // bf16 copy (EqdLinJob.Yb), or NULL
    const int ldyb = jw_i(W, LJ(ldyb));
    if (jYb && rv) *(EQD_GAS s16x4*)&jYb[(size_t)rowi * ldyb + f0] = pack_bf4(yv[0], yv[1], yv[2], yv[3]);
    if (out_local >= 0) *(f32x4*)&Lb[0][out_local][l15 * LIN_S + f0] = yv;
    LIN_TR(tr_i++);
#undef LJ
}

// One linear job on rows row0 .. row0 + 16 RT - 1.  src_local[i] >= 0: source i is the LDS tile Lb[rt][src_local[i]]
// ([16][LIN_S], written by an earlier job of the chain; K <= 80); out_local >= 0: the result is also left in
// Lb[rt][out_local].  Must be called by all 256 threads of the workgroup.
// RA / have_first / has_next: a row chain hands the registers of this job's first step in already loaded (have_first)
// and names the next linear job (Jn, its LDS-source flags src_local_n), whose first step is fetched into RA before this
// job's epilogue - otherwise every job of a chain starts with a fully exposed memory round trip.
// W: the descriptor words of J (jobw_load); chain: J is the `lin` of an EqdChainJob, whose src_local / out_local words are
// in W too; Wn: the words of the next job Jn (only read when has_next).
template <int RT, bool BF = false, bool LB = true>
__device__ __forceinline__ void linear_tile(const EqdLinJob& J, const JobW& W, bool chain,
                                            const int* __restrict__ src_local, int out_local,
                                            LinSmem<RT>& sm, float (*Lb)[LIN_LOCALS][16 * LIN_S], int row0,
                                            LinRegs<RT>& RA, bool have_first, bool has_next, const JobW& Wn,
                                            int trace_slot = 255) {
    (void)trace_slot;
    EQD_TR(trace_slot);
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int M = jw_i(W, LJ(M)), nsrc = jw_i(W, LJ(nsrc));
    if constexpr (RT == 1) {     // with two row tiles the lean job's address registers spill (256-VGPR budget)
        bool lean = (M == 64);
#pragma unroll
        for (int si = 0; si < EQD_MAX_SRC; ++si)
            if (si < nsrc && jw_i(W, LJ(s) + si * JW_SRC_DW + JW_OFF(EqdLinSrc, K)) < 64) lean = false;
        if (lean) {
            linear_tile_lean<BF, LB>(W, chain, out_local, sm, Lb, row0, RA, have_first, has_next, Wn, trace_slot);
            return;
        }
    }
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // the job header: SGPRs, read out of the descriptor words
    const int rows = jw_i(W, LJ(rows)), act = jw_i(W, LJ(act));
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    const float* const jlng = jw_p<const float>(W, LJ(ln_g));
    const float* const jlnb = jw_p<const float>(W, LJ(ln_b));
    const float* const jR = jw_p<const float>(W, LJ(R));
    float* const jY = jw_p<float>(W, LJ(Y));
    float* const jpre = jw_p<float>(W, LJ(pre_ln));
    const int ldr = jw_i(W, LJ(ldr)), ldy = jw_i(W, LJ(ldy)), ld_pre = jw_i(W, LJ(ld_pre)), pad_to = jw_i(W, LJ(pad_to));
    const float alpha = jw_f(W, LJ(alpha)), beta = jw_f(W, LJ(beta)), slope = jw_f(W, LJ(slope)), ln_eps = jw_f(W, LJ(ln_eps));
    unsigned short* const jYb = jw_p<unsigned short>(W, LJ(Yb));      // bf16 copy of the output (EqdLinJob.Yb), or NULL
    const int ldyb = jw_i(W, LJ(ldyb));
    const int mbn = (M + 15) >> 4;
    // this wave's output blocks and their epilogue operands (fetched now: the latency hides under the GEMM)
    const int mbs[2] = {wave, wave + 4};
    const bool own[2] = {wave < mbn, wave + 4 < mbn};
    f32x4 bias[2], lg[2], lb[2], res[RT][2];
    int nf[2];
    const float* const jsafe = jw_p<const float>(W, LJ(s) + JW_OFF(EqdLinSrc, W));      // >= 16 readable bytes (source 0's weights)
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
            // (without a residual the load goes to a valid dummy address and the epilogue selects 0: a `jR ? load : 0` here
            //  kept a hoisted zero vector live across the whole pipeline - in scratch memory in the 256-register instances)
            res[rt][i] = ld4u_raw((jR ? jR : jsafe) + (jR ? (size_t)row * ldr + f0 : (size_t)0), jR ? nf[i] : 0, jR ? jR : jsafe);
        }
    }
    f32x4 acc[RT][2], acc2[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[rt][i] = acc2[rt][i] = f4zero();

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
                        lin_mma<RT, 2, 4, false, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    else if (own[0])
                        lin_mma<RT, 1, 4, false, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                } else {
                    if (own[1])
                        lin_mma<RT, 2, 4, true, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                    else if (own[0])
                        lin_mma<RT, 1, 4, true, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                }
            } else {
                if (own[1])
                    lin_mma<RT, 2, 1, false, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
                else if (own[0])
                    lin_mma<RT, 1, 1, false, BF, LB>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
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
    generic_pipeline();
    EQD_TR(trace_slot + 2);
    if (has_next) {      // the next job's first step (same rows) -> RA
        const EqdLinSrc Sn = jw_src(Wn, 0);
        const bool local_n = chain && jw_i(Wn, JW_OFF(EqdChainJob, src_local)) >= 0;
        const LinStep c0 = {0, 0, lin_chunk(Sn.K)};
        lin_load_s<RT>(Sn, jw_i(Wn, LJ(M)), rows, local_n, c0, row0, t, RA);
    }
    EQD_TR(trace_slot + 3);
    // ---- epilogue in F-layout: feature f = 16 mb + 4 g + r, row = row0 + 16 rt + l15 -----------------------------
    // `plain`: M is a multiple of 4, so an owned 4-feature group is always complete - no tail handling, 16-byte stores
    const bool plain = (M & 3) == 0;
    // The register-capped instances (!LB) derive the epilogue's lane constants (offsets, feature counts) AGAIN from an
    // opaque copy of the thread index: computed before the pipeline they stay live across it - hoisted out of a chain's job
    // loop, even - and were what these instances spilled to scratch memory.  A dozen integer instructions per job instead.
    int te = t;
#ifndef EQD_HOSTSIM
    if constexpr (!LB) asm volatile("" : "+v"(te));
#endif
    const int l15e = te & 15, ge = (te >> 4) & 3, wavee = LB ? wave : (te >> 6);
    const int mbe[2] = {wavee, wavee + 4};
    const bool owne[2] = {wavee < mbn, wavee + 4 < mbn};
    int nfe[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) nfe[i] = owne[i] ? M - (16 * mbe[i] + 4 * ge) : 0;
    float4 bs[2], lgv[2], lbv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (plain) {
            bs[i] = make_float4(bias[i][0], bias[i][1], bias[i][2], bias[i][3]);
            lgv[i] = make_float4(lg[i][0], lg[i][1], lg[i][2], lg[i][3]);
            lbv[i] = make_float4(lb[i][0], lb[i][1], lb[i][2], lb[i][3]);
        } else {
            bs[i] = ld4u_fix(bias[i], nfe[i]);
            lgv[i] = ld4u_fix(lg[i], nfe[i]);
            lbv[i] = ld4u_fix(lb[i], nfe[i]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int rowi = row0 + 16 * rt + l15e;
        const bool rv = rowi < rows;
        float v[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float bb[4] = {bs[i].x, bs[i].y, bs[i].z, bs[i].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = (acc[rt][i][r] + acc2[rt][i][r]) + bb[r];
                if (act) y = lrelu(y, slope);
                v[i][r] = r < nfe[i] ? y : 0.f;
            }
        }
        if (const float* const jmul = jw_p<const float>(W, LJ(mul))) {      // dropout factors (training mode only)
            const int ld_mul = jw_i(W, LJ(ld_mul));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 mm = ld4u_fix(ld4u_raw(jmul + (size_t)(rv ? rowi : rows - 1) * ld_mul + 16 * mbe[i] + 4 * ge, nfe[i], jmul),
                                           nfe[i]);
                v[i][0] *= mm.x; v[i][1] *= mm.y; v[i][2] *= mm.z; v[i][3] *= mm.w;
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
            if (ge == 0) sm.stat[rt][wavee][l15e] = s1;
            __syncthreads();
            const float mean =
                (sm.stat[rt][0][l15e] + sm.stat[rt][1][l15e] + sm.stat[rt][2][l15e] + sm.stat[rt][3][l15e]) * invM;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dlt = r < nfe[i] ? v[i][r] - mean : 0.f;
                    q += dlt * dlt;
                }
            q = group_sum(q);
            __syncthreads();
            if (ge == 0) sm.stat[rt][wavee][l15e] = q;
            __syncthreads();
            const float rstd = 1.f / sqrtf((sm.stat[rt][0][l15e] + sm.stat[rt][1][l15e] + sm.stat[rt][2][l15e] +
                                            sm.stat[rt][3][l15e]) * invM + ln_eps);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float gg[4] = {lgv[i].x, lgv[i].y, lgv[i].z, lgv[i].w};
                const float be[4] = {lbv[i].x, lbv[i].y, lbv[i].z, lbv[i].w};
                if (plain && nfe[i] > 0 && jpre && rv)
                    *(EQD_GAS f4v*)&jpre[(size_t)rowi * ld_pre + 16 * mbe[i] + 4 * ge] = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nfe[i]) {
                        const int f = 16 * mbe[i] + 4 * ge + r;
                        if (!plain && jpre && rv) ((EQD_GAS float*)jpre)[(size_t)rowi * ld_pre + f] = v[i][r];
                        v[i][r] = (v[i][r] - mean) * rstd * gg[r] + be[r];
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 rr;
            if (!jR)
                rr = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (plain)
                rr = make_float4(res[rt][i][0], res[rt][i][1], res[rt][i][2], res[rt][i][3]);
            else
                rr = ld4u_fix(res[rt][i], nfe[i]);
            const float rs[4] = {rr.x, rr.y, rr.z, rr.w};
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = alpha * v[i][r] + beta * rs[r];
            const int f0 = 16 * mbe[i] + 4 * ge;
            if (plain) {
                if (nfe[i] > 0) {
                    const f32x4 yv = {y[0], y[1], y[2], y[3]};
                    if (jY && rv) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = yv;
                    if (jYb && rv) *(EQD_GAS s16x4*)&jYb[(size_t)rowi * ldyb + f0] = pack_bf4(y[0], y[1], y[2], y[3]);
                    if (out_local >= 0) *(f32x4*)&Lb[rt][out_local][l15e * LIN_S + f0] = yv;
                } else if (owne[i] && f0 < pad_to) {      // zero padding of the row (EqdLinJob.pad_to; pad_to is a multiple of 16)
                    if (jY && rv) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = f4zero();
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nfe[i]) {
                        if (jY && rv) ((EQD_GAS float*)jY)[(size_t)rowi * ldy + f0 + r] = y[r];
                        if (jYb && rv) ((EQD_GAS unsigned short*)jYb)[(size_t)rowi * ldyb + f0 + r] = f2bf(y[r]);
                        if (out_local >= 0) Lb[rt][out_local][l15e * LIN_S + f0 + r] = y[r];
                    } else {
                        if (f0 + r < pad_to) {      // zero padding of the row (EqdLinJob.pad_to)
                            if (jY && rv) ((EQD_GAS float*)jY)[(size_t)rowi * ldy + f0 + r] = 0.f;
                        }
                        // (the bf16 copy: zeros up to the end of the last started 4-column group, what a vector load of it covers)
                        if (jYb && rv && owne[i] && nfe[i] > 0 && f0 + r < ldyb)
                            ((EQD_GAS unsigned short*)jYb)[(size_t)rowi * ldyb + f0 + r] = 0;
                    }
            }
        }
    }
    LIN_TR(tr_i++);
#undef LJ
}
