// k_rowres: the row-local job chains with the weights RESIDENT in LDS - the form for many tiles per CU.
//
// k_rowwave (one wave per 16-row tile, weights loaded by every wave for itself) showed where the node-level chains are
// bound once the barriers are gone: a CU pulls ~10-12 B/clock through its vector-memory path whatever the pattern
// (profiles/r02_exp_trace_rowwave_*.txt: ~2 000-5 000 clocks to issue the 12 KB of a 32-column sub-step), and a chain
// needs 86-168 KB of weights PER TILE that way.  Here a persistent 8-wave workgroup per CU stages each source's weights
// (<= 21 KB) into LDS ONCE for all the tiles it owns - double buffered, the next source's weights are in flight while
// the current one is multiplied - and the waves read their MFMA A operands from LDS (128 B/clock per CU).  Global
// traffic per CU drops to the rows themselves (~1.5 KB per row and chain) + one copy of the weights.
//
// Work split: workgroup b owns `tps` consecutive tiles, wave w of it the tiles w, w + 8 (<= RR_TMAX per wave: two
// accumulator sets).  Loop order: job -> source -> the wave's tiles, so the job's accumulators live in registers across
// its sources; the chain's intermediate tile (EqdChainJob.out_local) lives in one LDS tile per (wave, tile slot) - the
// chains this kernel takes (rr_eligible, host) only ever read the tile written last, so every local id names that tile.
// One workgroup barrier per source (the weight hand-over), none inside.  Arithmetic, layouts (S / P), epilogue and
// LayerNorm backward are k_rowwave's (eqd_rowwave_inl.h).
#pragma once
#include "eqd_rowwave_inl.h"

#define RR_WAVES 8
#define RR_TMAX 2
#define RR_WFLOATS (64 * 84)      /* one source: [64][KP] (k-contiguous weights, KP = 68 or 84) or [80][64] (m-contiguous) */

struct RrSmem {
    float Wl[2][RR_WFLOATS];
    float tile[RR_WAVES][RR_TMAX][16 * RW_S];
    float red[RR_WAVES][256];
};

// global -> registers of one source's weights (all 512 threads; 16-byte vectors, coalesced along the contiguous axis)
struct RrStage {
    f32x4 v[3];
};
__device__ __forceinline__ int rr_kp(int K) { return K > 64 ? 84 : 68; }      // 4 x odd: conflict-free b128 fragment reads

__device__ __forceinline__ void rr_stage_load(const EqdLinSrc S, int t, RrStage& R) {
    const bool tp = S.w_cs != 1;
    if (!tp) {          // W[m][k]: idx -> (m, 4-column group c4); tails of a 69-wide row through the unaligned-tail load
        const int nc4 = S.K > 64 ? 20 : 16;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 64 * RR_WAVES * j;
            const int m = idx / nc4, c4 = idx - m * nc4;
            const int mm = m < 64 ? m : 63;
            R.v[j] = ld4u_raw(S.W + (size_t)mm * S.w_rs + 4 * c4, S.K - 4 * c4, S.W);
        }
    } else {            // W[k][m] (m contiguous): idx -> (k, 4-column group of m); rows k >= K become zeros in LDS
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 64 * RR_WAVES * j;
            const int k = idx >> 4, c4 = idx & 15;
            const int kk = k < S.K ? k : S.K - 1;
            R.v[j] = *(const EQD_GAS f4v*)(S.W + (size_t)kk * S.w_cs + 4 * c4);
        }
    }
}
__device__ __forceinline__ void rr_stage_store(const EqdLinSrc S, int t, const RrStage& R, float* Wl) {
    const bool tp = S.w_cs != 1;
    if (!tp) {
        const int nc4 = S.K > 64 ? 20 : 16, KP = rr_kp(S.K);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 64 * RR_WAVES * j;
            const int m = idx / nc4, c4 = idx - m * nc4;
            if (m < 64) {
                const float4 f = ld4u_fix(R.v[j], S.K - 4 * c4);
                *(f32x4*)&Wl[m * KP + 4 * c4] = f32x4{f.x, f.y, f.z, f.w};
            }
        }
    } else {
        const int nk = S.K > 64 ? 80 : 64;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 64 * RR_WAVES * j;
            const int k = idx >> 4, c4 = idx & 15;
            if (k < nk) *(f32x4*)&Wl[k * 64 + 4 * c4] = k < S.K ? R.v[j] : f4zero();
        }
    }
}

// bf16 mode: the staged weights are rounded ONCE, when they are written to LDS, as bf16 [m][k] (k contiguous, row stride
// 72 / 88 bf16) whatever the orientation in memory - an MFMA A operand is then one ds_read_b64 and the inner loop has no
// weight conversions (round 2 staged fp32 and converted per use: 4 ds_read_b128 + 8 v_cvt_pk_bf16_f32 per 4 MFMAs and
// tile, the LDS pipe busy ~4x longer than the MFMA pipes).  m-contiguous weights (the backward's dX = dY W) are transposed
// by the staging stores (12 two-byte stores per thread and source, once per workgroup), so that every job runs the
// k-contiguous form and leaves its result in the S layout.
__device__ __forceinline__ int rr_kp16(int K) { return K > 64 ? 88 : 72; }
__device__ __forceinline__ void rr_stage_store_bf(const EqdLinSrc S, int t, const RrStage& R, unsigned short* Wl) {
    const bool tp = S.w_cs != 1;
    const int KP = rr_kp16(S.K);
    if (!tp) {
        const int nc4 = S.K > 64 ? 20 : 16;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 64 * RR_WAVES * j;
            const int m = idx / nc4, c4 = idx - m * nc4;
            if (m < 64) {
                const float4 f = ld4u_fix(R.v[j], S.K - 4 * c4);
                *(s16x4*)&Wl[m * KP + 4 * c4] = pack_bf4(f.x, f.y, f.z, f.w);
            }
        }
    } else {
        const int nk = S.K > 64 ? 80 : 64;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 64 * RR_WAVES * j;
            const int k = idx >> 4, c4 = idx & 15;
            if (k < nk) {
                const f32x4 v = k < S.K ? R.v[j] : f4zero();
#pragma unroll
                for (int u = 0; u < 4; ++u) Wl[(4 * c4 + u) * KP + k] = f2bf(v[u]);
            }
        }
    }
}
// rows of one (source, tile) item as loaded: 64 columns + the 4 columns 64 + 4 g .. of a wide source
struct RrRows {
    f32x4 x[4], xr;
};
__device__ __forceinline__ void rr_rows_load(const EqdLinSrc S, int rowc, int g, RrRows& R) {
    const char* const xb = (const char*)S.X;
    const unsigned xl = 4u * (unsigned)(rowc * S.ldx + 4 * g);
#pragma unroll
    for (int a = 0; a < 4; ++a) R.x[a] = rw_ld(xb + 64 * a, xl);
    if (S.K > 64) {
        const int n = S.K - 64 - 4 * g;
        const int sh = (n > 0 && n < 4) ? 4 - n : 0;
        R.xr = rw_ld(xb, n > 0 ? 4u * (unsigned)(rowc * S.ldx + 64 + 4 * g - sh) : xl);
    }
}
// the copy the MFMAs read (they never read a register a load in flight targets: eqd_rowwave_inl.h), with the LeakyReLU
// mask applied - mask rows are fetched here, not ahead (two of the ten sources of the backward chain carry one)
__device__ __forceinline__ void rr_rows_take(const EqdLinSrc S, float slope, int rowc, int g, const RrRows& R, RrRows& C) {
#pragma unroll
    for (int a = 0; a < 4; ++a) C.x[a] = R.x[a];
    C.xr = R.xr;
    if (S.mask) {
        const char* const mb = (const char*)S.mask;
        const unsigned xl = 4u * (unsigned)(rowc * S.ldx + 4 * g);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 m = rw_ld(mb + 64 * a, xl);
#pragma unroll
            for (int b = 0; b < 4; ++b) C.x[a][b] *= lrelu_grad(m[b], slope);
        }
    }
}

// one (source, tile) item: B operands from the item's rows (global source) or the LDS tile, A operands from the staged
// weights; 64 (fp32) / 16 (bf16) MFMAs per 64 columns
template <bool BF>
__device__ __forceinline__ void rr_item(const float* __restrict__ Wl, int K, bool tp, bool local, const float* T,
                                        const RrRows& R, int l15, int g, f32x4 (&acc)[4]) {
    const int KP = rr_kp(K);
    const int na = K > 64 ? 5 : 4;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        if (a < na) {
            f32x4 bv;
            if (a == 4) {
                const float4 xf = ld4u_fix(R.xr, K - 64 - 4 * g);
                bv = f32x4{xf.x, xf.y, xf.z, xf.w};
            } else if (local) {
                bv = *(const f32x4*)(T + l15 * RW_S + 16 * a + 4 * g);
            } else {
                bv = R.x[a < 4 ? a : 0];
            }
            f32x4 w[4];
            if (!tp) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) w[mb] = *(const f32x4*)&Wl[(16 * mb + l15) * KP + 16 * a + 4 * g];
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) w[b] = *(const f32x4*)&Wl[(16 * a + 4 * g + b) * 64 + 4 * l15];
            }
            if constexpr (BF) {
                const s16x4 bp = pack_bf4(bv[0], bv[1], bv[2], bv[3]);
                if (!tp) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma_bf(pack_bf4(w[mb][0], w[mb][1], w[mb][2], w[mb][3]), bp, acc[mb]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = mfma_bf(pack_bf4(w[0][j], w[1][j], w[2][j], w[3][j]), bp, acc[j]);
                }
            } else {
                if (!tp) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma4(w[mb][b], bv[b], acc[mb]);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = mfma4(w[b][j], bv[b], acc[j]);
                }
            }
        }
    }
}

// one (source, tile) item with bf16 weights in LDS: 4 MFMAs per 16 columns, A operands straight from LDS
__device__ __forceinline__ void rr_item_bf(const unsigned short* __restrict__ Wl, int K, bool local, const float* T,
                                           const RrRows& R, int l15, int g, f32x4 (&acc)[4]) {
    const int KP = rr_kp16(K);
    // columns 0..63: two 32-deep chunks of v_mfma_f32_16x16x32_bf16 (two of the lane's 16-column blocks each)
#pragma unroll
    for (int ap = 0; ap < 2; ++ap) {
        f32x4 b0, b1;
        if (local) {
            b0 = *(const f32x4*)(T + l15 * RW_S + 32 * ap + 4 * g);
            b1 = *(const f32x4*)(T + l15 * RW_S + 32 * ap + 16 + 4 * g);
        } else {
            b0 = R.x[2 * ap];
            b1 = R.x[2 * ap + 1];
        }
        const s16x8 bp = cat_bf(pack_bf4(b0[0], b0[1], b0[2], b0[3]), pack_bf4(b1[0], b1[1], b1[2], b1[3]));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            acc[mb] = mfma_bf32(cat_bf(*(const s16x4*)&Wl[(16 * mb + l15) * KP + 32 * ap + 4 * g],
                                       *(const s16x4*)&Wl[(16 * mb + l15) * KP + 32 * ap + 16 + 4 * g]), bp, acc[mb]);
    }
    if (K > 64) {      // the 69-wide sources' remainder columns: one 16-deep chunk
        const float4 xf = ld4u_fix(R.xr, K - 64 - 4 * g);
        const s16x4 bp = pack_bf4(xf.x, xf.y, xf.z, xf.w);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            acc[mb] = mfma_bf(*(const s16x4*)&Wl[(16 * mb + l15) * KP + 64 + 4 * g], bp, acc[mb]);
    }
}

template <bool BF>
__global__ __launch_bounds__(64 * RR_WAVES, 1) void k_rowres(EqdChainArg A_, int tps) {
    __shared__ __attribute__((aligned(16))) EqdChainArg A;
    __shared__ __attribute__((aligned(16))) RrSmem sm;
    kernarg_to_lds(A, EQD_KERNARG_PTR(A_), 0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    float* const red = sm.red[wave];
    for (int i = lane; i < 256; i += 64) red[i] = 0.f;
    __syncthreads();
    constexpr int CJ_DW = (int)(sizeof(EqdChainJob) / 4);
    const int njobs = uni(A.njobs);
    JobW Wc = jobw_load(&A.j[0], CJ_DW, lane);
    const int rows = jw_i(Wc, JW_OFF(EqdLinJob, rows));
    const int ntiles = (rows + 15) >> 4;
    const int tile0 = (int)blockIdx.x * tps;
    int nt_wg = ntiles - tile0;
    nt_wg = nt_wg < tps ? nt_wg : tps;
    // this wave's tiles: slot s -> tile tile0 + wave + RR_WAVES s
    int nslots = 0;
    int row0s[RR_TMAX], rowcs[RR_TMAX];
#pragma unroll
    for (int s = 0; s < RR_TMAX; ++s) {
        const bool has = wave + RR_WAVES * s < nt_wg;
        nslots += has ? 1 : 0;
        row0s[s] = (tile0 + wave + RR_WAVES * s) * 16;
        int rc = row0s[s] + l15;
        rowcs[s] = has ? (rc < rows ? rc : rows - 1) : 0;
    }
    nslots = uni(nslots);
    float* aux = nullptr;
    int jj = 0;
    // LayerNorm-backward jobs in front of the first linear job
    while (jj < njobs && jw_i(Wc, JW_OFF(EqdChainJob, type)) != 0) {
#pragma unroll
        for (int s = 0; s < RR_TMAX; ++s)
            if (s < nslots) rw_lnbwd(Wc, sm.tile[wave][s], red, row0s[s], l15, g, 0);
        aux = jw_p<float>(Wc, JW_OFF(EqdChainJob, aux));
        ++jj;
        if (jj < njobs) Wc = jobw_load(&A.j[jj], CJ_DW, lane);
    }
    int buf = 0;
    int trc = 0;      // source counter of the phase-trace experiments (profiles/exp_trace_rowwave.py)
    (void)trc;
    RrStage WS;
    RrRows XR[RR_TMAX];
    const float* held_x = nullptr;      // the global source whose rows XR holds (wave-uniform)
    int held_ld = 0, held_k = 0;
    if (jj < njobs) {      // the first source's weights
        const EqdLinSrc S0 = jw_src(Wc, 0);
        rr_stage_load(S0, t, WS);
        if constexpr (BF) rr_stage_store_bf(S0, t, WS, (unsigned short*)sm.Wl[0]);
        else rr_stage_store(S0, t, WS, sm.Wl[0]);
    }
    __syncthreads();
    while (jj < njobs) {
        const int nsrc = jw_i(Wc, JW_OFF(EqdLinJob, nsrc));
        const bool tp = jw_i(Wc, JW_OFF(EqdLinJob, s) + JW_OFF(EqdLinSrc, w_cs)) != 1;
        const float slope = jw_f(Wc, JW_OFF(EqdLinJob, slope));
        const int next_lin = jw_i(Wc, JW_OFF(EqdChainJob, next_lin));
        JobW Wnl = Wc;
        if (next_lin >= 0) Wnl = jobw_load(&A.j[next_lin], CJ_DW, lane);
        f32x4 acc[RR_TMAX][4];
#pragma unroll
        for (int s = 0; s < RR_TMAX; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[s][q] = f4zero();
        // rows of the job's first source for every tile slot (global sources only); from then on the rows of source
        // si + 1 are fetched while source si is multiplied
        // (XR lives across jobs: `held` names the rows it holds, so the five projection jobs of a layer - same h rows -
        // fetch them once, and a job whose first source is global finds its rows already requested by the job before)
        {
            const EqdLinSrc S0 = jw_src(Wc, 0);
            if (jw_i(Wc, JW_OFF(EqdChainJob, src_local)) < 0 && !(held_x == S0.X && held_ld == S0.ldx && held_k == S0.K)) {
#pragma unroll
                for (int s = 0; s < RR_TMAX; ++s)
                    if (s < nslots) rr_rows_load(S0, rowcs[s], g, XR[s]);
                held_x = S0.X; held_ld = S0.ldx; held_k = S0.K;
            }
        }
        for (int si = 0; si < nsrc; ++si) {
            EQD_TR(100 + 5 * trc);
            const EqdLinSrc S = jw_src(Wc, si);
            const int loc = jw_i(Wc, JW_OFF(EqdChainJob, src_local) + si);
            const bool last_src = si + 1 >= nsrc;
            const bool have_next = !last_src || next_lin >= 0;
            const EqdLinSrc Sn = last_src ? jw_src(Wnl, 0) : jw_src(Wc, si + 1);
            const int nloc = last_src ? 0 : jw_i(Wc, JW_OFF(EqdChainJob, src_local) + si + 1);
            RrRows XC[RR_TMAX];
            if (loc < 0) {
#pragma unroll
                for (int s = 0; s < RR_TMAX; ++s)
                    if (s < nslots) rr_rows_take(S, slope, rowcs[s], g, XR[s], XC[s]);      // waits for the rows
            }
            EQD_TR(101 + 5 * trc);
            // behind the copies (in front of them the copies would wait for these loads too): the next source's weights
            // and, within the job, its rows
            if (have_next) rr_stage_load(Sn, t, WS);
            const bool next_rows = last_src ? (next_lin >= 0 && jw_i(Wnl, JW_OFF(EqdChainJob, src_local)) < 0) : nloc < 0;
            if (next_rows && !(held_x == Sn.X && held_ld == Sn.ldx && held_k == Sn.K)) {
#pragma unroll
                for (int s = 0; s < RR_TMAX; ++s)
                    if (s < nslots) rr_rows_load(Sn, rowcs[s], g, XR[s]);
                held_x = Sn.X; held_ld = Sn.ldx; held_k = Sn.K;
            }
            EQD_TR(102 + 5 * trc);
#pragma unroll
            for (int s = 0; s < RR_TMAX; ++s)
                if (s < nslots) {
                    if constexpr (BF)
                        rr_item_bf((const unsigned short*)sm.Wl[buf], S.K, loc >= 0, sm.tile[wave][s], XC[s], l15, g, acc[s]);
                    else
                        rr_item<BF>(sm.Wl[buf], S.K, tp, loc >= 0, sm.tile[wave][s], XC[s], l15, g, acc[s]);
                }
            EQD_TR(103 + 5 * trc);
            if (have_next) {
                if constexpr (BF) rr_stage_store_bf(Sn, t, WS, (unsigned short*)sm.Wl[buf ^ 1]);
                else rr_stage_store(Sn, t, WS, sm.Wl[buf ^ 1]);
            }
            EQD_TR(104 + 5 * trc);
            __syncthreads();      // every wave is done with Wl[buf]; Wl[buf ^ 1] is complete
            buf ^= 1;
            ++trc;
        }
        // (fetching both tiles' residual rows and the parameter vectors ahead of the arithmetic - the epilogues and job
        // transitions are 24-31 % of a launch, profiles/r02_exp_trace_rowres_C*.txt - was measured SLOWER: the kernel is at
        // its register limit and the extra live values spill, k_rowres 975 -> 1 213 us per step at C)
#pragma unroll
        for (int s = 0; s < RR_TMAX; ++s)
            if (s < nslots) rw_epilogue(Wc, BF ? false : tp, acc[s], sm.tile[wave][s], row0s[s], l15, g, 0);      // (bf16: S layout always)
        // the LayerNorm-backward jobs behind it, then the next linear job
        ++jj;
        while (jj < njobs) {
            Wc = jobw_load(&A.j[jj], CJ_DW, lane);
            if (jw_i(Wc, JW_OFF(EqdChainJob, type)) == 0) break;
#pragma unroll
            for (int s = 0; s < RR_TMAX; ++s)
                if (s < nslots) rw_lnbwd(Wc, sm.tile[wave][s], red, row0s[s], l15, g, 0);
            aux = jw_p<float>(Wc, JW_OFF(EqdChainJob, aux));
            ++jj;
        }
    }
    __syncthreads();
    if (aux) {      // (every wave walks the whole job list, so every wave knows aux)
        float* ap = aux + (size_t)blockIdx.x * 256;
        if (t < 256) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < RR_WAVES; ++w) sacc += sm.red[w][t];
            ap[t] = sacc;
        }
    }
}
