// k_rowwave: the row-local job chains (EqdChainJob, eqd_common.h) with ONE WAVE per 16-row tile and the whole chain in
// registers - the throughput form of k_rowchain, which spreads a tile over four waves and pays two workgroup barriers
// and an LDS staging of the weights per 64-wide K step (1 350 clocks of step overhead around 512 clocks of fp32 MFMA
// work, 32 in bf16 mode: MfmaUtil 4 % at config B, 15 % at config C).
//
// Transposed formulation, as everywhere in this library: features on the MFMA M axis, the tile's 16 rows on the N axis.
// A wave owns all 64 output features of its rows, so
//   * nothing is shared between waves: no barrier, no staging - the A operand (weights) is loaded from global memory
//     (L2 resident: <= 170 KB per chain) straight into the lanes that feed the MFMA.  A source is walked in sub-steps of
//     32 columns: 12 x 16-byte loads (8 weight vectors, 2 row vectors, 2 mask vectors - always 12, see rw_load) issued
//     one sub-step ahead of the 32 (fp32) / 8 (bf16) MFMAs that consume them, through a copy into MFMA-only registers
//     (rw_take) so that the matrix instructions never wait for loads in flight;
//   * the D layout of one job is the B layout of the next: lane (l15, g) holds, of row l15, 16 features v[a][b],
//       S layout: feature 16 a + 4 g + b      (weights W[m][k], k contiguous:  acc[mb][r] = v[mb][r])
//       P layout: feature 16 g + 4 a + b      (weights W[k][m], m contiguous:  acc[j][r]  = v[r][j])
//     and MFMA (a, b) contracts k = feature(a) + b, whichever layout the producer left.  With k-contiguous weights the
//     lane loads W[16 mb + l15][feature(a) ..+3] (4 MFMAs b = 0..3 per load and output block mb); with m-contiguous
//     weights (the backward's dX = dY W) it loads W[feature(a) + b][4 l15 ..+3] and MFMA j of the four uses component j:
//     output block j then holds features 4 i + j on its M index i, i.e. the P layout.  No transposes anywhere;
//   * tiles that later jobs of the chain read (EqdChainJob.out_local) go to a wave-private LDS tile in [row][feature]
//     order; LayerNorm statistics are 16 in-lane values + two cross-lane steps.
// Arithmetic: the same products as k_rowchain, summed in another order (fp32: within the north-star tolerance, and
// checked against the same oracle; bf16: inputs rounded at the same points).
//
// Eligible chains (rw_eligible, host): every job 64 outputs wide; sources 64 wide, or 65..80 wide when read from global
// memory without mask (the 69-wide h0 of node_mlp.0); one weight orientation per job; no global source that an earlier
// job of the chain writes.  Everything else - layer 0 with its 69-wide features - stays on k_rowchain / k_linear.
// This kernel is the middle form: measured faster than the four-wave kernels only from a few tiles per CU, and slower
// than k_rowres (weights resident in LDS, eqd_rowres_inl.h) there, because every wave streams the chain's weights for
// itself and a CU accepts only ~10-12 B/clock of loads.  It stays selectable (EQD_ROWWAVE=1), it takes the chains
// k_rowres cannot (more than one live intermediate tile), and its helpers (layouts, epilogue, LayerNorm backward) are
// k_rowres's.
#pragma once
#include "eqd_linear_inl.h"

#define RW_S 68          /* LDS row stride of a wave-private tile: 4 l15 + c covers the 64 banks once per 16 lanes */
#define RW_WAVES 4

struct RwBuf {           // operands of one SUB-step as loaded (nothing is waited for at load time).  A source is walked in
                         // sub-steps h = 0, 1: features a = 2 h, 2 h + 1 (32 columns each); h = 2: columns 64 .. K - 1
    f32x4 w[8];          // k-contiguous weights: w[4 i + mb] for a = 2 h + i; m-contiguous: w[4 i + b]; h = 2: w[mb] raw
    f32x4 x[2], xm[2];   // global source row, S layout (features 16 a + 4 g ..), and its LeakyReLU mask row; h = 2: x[0] raw
};

struct RwSmem {
    float tile[RW_WAVES][LIN_LOCALS][16 * RW_S];
    float red[RW_WAVES][256];      // LayerNorm-backward partial sums of the wave's rows: [d gamma 0..127 | d beta 128..255]
};

// Loads of one sub-step.  ALWAYS the same 12 instructions in the same register order - 8 weight vectors, 2 source-row
// vectors, 2 mask-row vectors - whatever the source looks like: what a sub-step does not need (rows of an LDS-resident
// source, an absent mask, the second half of the columns 64 .. K - 1, everything when the chain has no further sub-step)
// is fetched from a harmless address (the start of the weight matrix: a cache hit).  With a load count that depends on
// the descriptor the compiler cannot tell how many loads are in flight when the PREVIOUS buffer is consumed and waits
// for all of them (s_waitcnt vmcnt(0)): the prefetch would be serialised behind the MFMAs it is meant to overlap
// (measured: 2 000-3 500 clocks of issue stall per sub-step, profiles/r02_exp_trace_rowwave_*.txt).  Every address is a
// wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset, so a load costs one scalar add and no 64-bit VALU work.
__device__ __forceinline__ f32x4 rw_ld(const char* base, unsigned off) { return *(const EQD_GAS f4v*)(base + off); }

__device__ __forceinline__ void rw_load(const JobW& W, int si, int h, bool local, bool valid, int rowc, int l15, int g,
                                        RwBuf& R) {
    const EqdLinSrc S = jw_src(W, valid ? si : 0);
    const bool tp = S.w_cs != 1;
    const char* const wb = (const char*)S.W;
    const unsigned lane16 = 16u * (unsigned)(l15 + 16 * g);
    const bool xreal = valid && !local;
    const char* const xb = xreal ? (const char*)S.X : wb;
    const char* const mb_ = (xreal && S.mask) ? (const char*)S.mask : xb;
    if (valid && h == 2) {        // columns 64 .. K - 1: k = 64 + 4 g + b of lane group g, zeros beyond K (rw_mma_rest)
        const int n = S.K - 64 - 4 * g;         // valid columns of this lane's 4
        if (tp) {                 // W[k][4 l15 ..]: whole vectors; rows beyond K are read at row K - 1 and dropped later
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int k = 64 + 4 * g + b;
                k = k < S.K ? k : S.K - 1;
                R.w[b] = rw_ld(wb, 4u * (unsigned)(k * S.w_cs + 4 * l15));
            }
        } else {                  // W[16 mb + l15][64 + 4 g ..]: unaligned-tail loads (ld4u_raw: vector ENDING at the last column)
            const int sh = (n > 0 && n < 4) ? 4 - n : 0;
            const unsigned wl = n > 0 ? 4u * (unsigned)(l15 * S.w_rs + 64 + 4 * g - sh) : 4u * (unsigned)(l15 * S.w_rs);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) R.w[mb] = rw_ld(wb + (size_t)(16 * mb) * S.w_rs * 4, wl);
        }
#pragma unroll
        for (int q = 4; q < 8; ++q) R.w[q] = rw_ld(wb, lane16);
        {
            const int sh = (n > 0 && n < 4) ? 4 - n : 0;
            const unsigned xl = n > 0 ? 4u * (unsigned)(rowc * S.ldx + 64 + 4 * g - sh) : 4u * (unsigned)(rowc * S.ldx);
            R.x[0] = rw_ld(xb, xreal ? xl : lane16);
        }
        R.x[1] = rw_ld(wb, lane16);
        R.xm[0] = rw_ld(wb, lane16);
        R.xm[1] = rw_ld(wb, lane16);
        return;
    }
    const int hh = valid ? h : 0;
    if (tp) {
        const unsigned wl = 4u * (unsigned)(4 * g * S.w_cs + 4 * l15);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) R.w[4 * i + b] = rw_ld(wb + (size_t)(32 * hh + 16 * i + b) * S.w_cs * 4, wl);
    } else {
        const unsigned wl = 4u * (unsigned)(l15 * S.w_rs + 4 * g);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                R.w[4 * i + mb] = rw_ld(wb + ((size_t)(16 * mb) * S.w_rs + 32 * hh + 16 * i) * 4, wl);
    }
    const unsigned xl = xreal ? 4u * (unsigned)(rowc * S.ldx + 4 * g) : lane16;
    const size_t xu = xreal ? (size_t)(32 * hh) * 4 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) R.x[i] = rw_ld(xb + xu + (xreal ? 64 * i : 0), xl);
#pragma unroll
    for (int i = 0; i < 2; ++i) R.xm[i] = rw_ld(mb_ + xu + (xreal ? 64 * i : 0), xl);
}

// MFMA operands of one sub-step, COPIED out of the load buffer (RwBuf) once its loads have landed: the copy is what
// decouples the matrix instructions from the loads in flight - they read registers no load ever targets, so the compiler
// has nothing to wait for in front of them, whatever the control flow around the prefetch looks like (with the MFMAs
// reading the load buffers directly it waited for the NEXT sub-step's loads as well: 4 800 clocks per sub-step instead of
// ~1 400, profiles/r02_exp_trace_rowwave_*.txt).  The mask product and, in bf16 mode, the rounding happen in the copy.
template <bool BF>
struct RwOps;
template <>
struct RwOps<false> {
    f32x4 w[8], x[2];
};
template <>
struct RwOps<true> {
    s16x4 w[8], x[2];      // non-transposed: w[4 i + mb] = the lane's 4 k-values; transposed: w[4 i + j] gathered over b
};

template <bool BF>
__device__ __forceinline__ void rw_take(const RwBuf& R, bool tp, bool masked, float slope, bool rest, int n_rest,
                                        RwOps<BF>& O) {
    f32x4 x[2] = {R.x[0], R.x[1]};
    f32x4 w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = R.w[q];
    if (rest) {      // columns 64 .. K - 1: zeros beyond K; non-transposed weights arrive as unaligned-tail vectors
        const float4 xf = ld4u_fix(R.x[0], n_rest);
        x[0] = f32x4{xf.x, xf.y, xf.z, xf.w};
        x[1] = f4zero();
        if (!tp) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 wf = ld4u_fix(R.w[mb], n_rest);
                w[mb] = f32x4{wf.x, wf.y, wf.z, wf.w};
            }
        }
    } else if (masked) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) x[i][b] *= lrelu_grad(R.xm[i][b], slope);
    }
    if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < 2; ++i) O.x[i] = pack_bf4(x[i][0], x[i][1], x[i][2], x[i][3]);
        if (!tp) {
#pragma unroll
            for (int q = 0; q < 8; ++q) O.w[q] = pack_bf4(w[q][0], w[q][1], w[q][2], w[q][3]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) O.w[4 * i + j] = pack_bf4(w[4 * i][j], w[4 * i + 1][j], w[4 * i + 2][j], w[4 * i + 3][j]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) O.x[i] = x[i];
#pragma unroll
        for (int q = 0; q < 8; ++q) O.w[q] = w[q];
    }
}

// the 32 (fp32) / 8 (bf16) MFMAs of a sub-step; Bl: B operand of an LDS-resident source (fp32), else O.x.  ni: 2, or 1
// for the columns 64 .. K - 1 (only a = 4 exists)
template <bool BF>
__device__ __forceinline__ void rw_mma(const RwOps<BF>& O, bool local, const f32x4 (&Bl)[2], bool tp, int ni, f32x4 (&acc)[4]) {
    if constexpr (BF) {
        if (ni == 2) {      // both 16-deep halves of the sub-step at once: v_mfma_f32_16x16x32_bf16
            const s16x4 b0 = local ? pack_bf4(Bl[0][0], Bl[0][1], Bl[0][2], Bl[0][3]) : O.x[0];
            const s16x4 b1 = local ? pack_bf4(Bl[1][0], Bl[1][1], Bl[1][2], Bl[1][3]) : O.x[1];
            const s16x8 bp = cat_bf(b0, b1);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma_bf32(cat_bf(O.w[q], O.w[4 + q]), bp, acc[q]);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < ni) {
            if constexpr (BF) {
                const s16x4 bp = local ? pack_bf4(Bl[i][0], Bl[i][1], Bl[i][2], Bl[i][3]) : O.x[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma_bf(O.w[4 * i + q], bp, acc[q]);
            } else {
                const f32x4 bv = local ? Bl[i] : O.x[i];
                if (!tp) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma4(O.w[4 * i + mb][b], bv[b], acc[mb]);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = mfma4(O.w[4 * i + b][j], bv[b], acc[j]);
                }
            }
        }
    }
}

// Y = alpha * f(acc + bias) + beta * R of one job for the wave's rows (EqdLinJob semantics, see equidock_hip.h).
// In two steps so that a caller with several tiles (k_rowres) can put ALL the epilogue's loads in flight first: the
// job's parameter vectors once (RwEpiParams), each tile's residual rows (rw_epi_rows), then the arithmetic per tile -
// one load latency per job instead of two or three per tile (they were 24-31 % of a k_rowres launch,
// profiles/r02_exp_trace_rowres_C*.txt).
struct RwEpiParams {
    f32x4 bias[4], lg[4], lb[4];
};
__device__ __forceinline__ void rw_epi_feat(bool tp, int g, int (&ft)[4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) ft[a] = tp ? 16 * g + 4 * a : 16 * a + 4 * g;
}
__device__ __forceinline__ void rw_epi_params(const JobW& W, bool tp, int g, RwEpiParams& P) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    int ft[4];
    rw_epi_feat(tp, g, ft);
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    const float* const jlng = jw_p<const float>(W, LJ(ln_g));
    const float* const jlnb = jw_p<const float>(W, LJ(ln_b));
    if (jbias) {
#pragma unroll
        for (int a = 0; a < 4; ++a) P.bias[a] = *(const EQD_GAS f4v*)(jbias + ft[a]);
    }
    if (jlng) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            P.lg[a] = *(const EQD_GAS f4v*)(jlng + ft[a]);
            P.lb[a] = *(const EQD_GAS f4v*)(jlnb + ft[a]);
        }
    }
#undef LJ
}
__device__ __forceinline__ void rw_epi_rows(const JobW& W, bool tp, int row0, int l15, int g, f32x4 (&res)[4]) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int rows = jw_i(W, LJ(rows));
    const int rowi = row0 + l15;
    const int rowe = rowi < rows ? rowi : rows - 1;
    int ft[4];
    rw_epi_feat(tp, g, ft);
    const float* const jR = jw_p<const float>(W, LJ(R));
    const int ldr = jw_i(W, LJ(ldr));
    if (jR) {
        const float* rp = jR + (size_t)rowe * ldr;
#pragma unroll
        for (int a = 0; a < 4; ++a) res[a] = *(const EQD_GAS f4v*)(rp + ft[a]);
    }
#undef LJ
}
// tstride: floats between the LDS tiles of consecutive local ids (0: every id names the same tile, k_rowres)
// epl (k_rowres): the job's parameter vectors in LDS ([bias | ln_g | ln_b] x 64 floats) instead of P
__device__ __forceinline__ void rw_epi_finish(const JobW& W, bool tp, const f32x4 (&acc)[4], const RwEpiParams& P,
                                              const f32x4 (&res)[4], float* tiles, int row0, int l15, int g, int tstride,
                                              const float* epl = nullptr) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int rows = jw_i(W, LJ(rows));
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    int ft[4];
    rw_epi_feat(tp, g, ft);
    const bool has_bias = jw_p<const float>(W, LJ(bias)) != nullptr;
    const bool has_ln = jw_p<const float>(W, LJ(ln_g)) != nullptr;
    const bool has_res = jw_p<const float>(W, LJ(R)) != nullptr;
    const float slope = jw_f(W, LJ(slope));
    f32x4 v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) v[a][b] = tp ? acc[b][a] : acc[a][b];
    if (has_bias) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 bv = epl ? *(const f32x4*)(epl + ft[a]) : P.bias[a];
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] += bv[b];
        }
    }
    if (jw_i(W, LJ(act))) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] = lrelu(v[a][b], slope);
    }
    if (const float* const jmul = jw_p<const float>(W, LJ(mul))) {      // dropout factors (training mode only)
        const float* mp = jmul + (size_t)(rv ? rowi : rows - 1) * jw_i(W, LJ(ld_mul));
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 mm = *(const EQD_GAS f4v*)(mp + ft[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] *= mm[b];
        }
    }
    if (has_ln) {
        float* const jpre = jw_p<float>(W, LJ(pre_ln));
        const int ld_pre = jw_i(W, LJ(ld_pre));      // (descriptor reads are wave operations: never under a lane predicate)
        if (jpre && rv) {
            float* pp = jpre + (size_t)rowi * ld_pre;
#pragma unroll
            for (int a = 0; a < 4; ++a) *(EQD_GAS f4v*)(pp + ft[a]) = v[a];
        }
        float s1 = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) s1 += (v[a][0] + v[a][1]) + (v[a][2] + v[a][3]);
        const float mean = group_sum(s1) * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float dlt = v[a][b] - mean;
                q += dlt * dlt;
            }
        const float rstd = 1.f / sqrtf(group_sum(q) * (1.f / 64.f) + jw_f(W, LJ(ln_eps)));
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 lg = epl ? *(const f32x4*)(epl + 64 + ft[a]) : P.lg[a];
            const f32x4 lb = epl ? *(const f32x4*)(epl + 128 + ft[a]) : P.lb[a];
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] = (v[a][b] - mean) * rstd * lg[b] + lb[b];
        }
    }
    const float alpha = jw_f(W, LJ(alpha)), beta = jw_f(W, LJ(beta));
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) v[a][b] = alpha * v[a][b] + (has_res ? beta * res[a][b] : 0.f);
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));
    if (jY && rv) {
        float* yp = jY + (size_t)rowi * ldy;
#pragma unroll
        for (int a = 0; a < 4; ++a) *(EQD_GAS f4v*)(yp + ft[a]) = v[a];
    }
    unsigned short* const jYb = jw_p<unsigned short>(W, LJ(Yb));      // bf16 copy of the output (EqdLinJob.Yb), or NULL
    const int ldyb = jw_i(W, LJ(ldyb));
    if (jYb && rv) {
        unsigned short* yp = jYb + (size_t)rowi * ldyb;
#pragma unroll
        for (int a = 0; a < 4; ++a) *(EQD_GAS s16x4*)(yp + ft[a]) = pack_bf4(v[a][0], v[a][1], v[a][2], v[a][3]);
    }
    const int out_l = jw_i(W, JW_OFF(EqdChainJob, out_local));
    if (out_l >= 0) {
        float* T = tiles + out_l * tstride + l15 * RW_S;
#pragma unroll
        for (int a = 0; a < 4; ++a) *(f32x4*)(T + ft[a]) = v[a];
        wave_lds_fence();
    }
#undef LJ
}
__device__ __forceinline__ void rw_epilogue(const JobW& W, bool tp, const f32x4 (&acc)[4], float* tiles, int row0, int l15,
                                            int g, int tstride = 16 * RW_S) {
    RwEpiParams P;
    f32x4 res[4];
    rw_epi_rows(W, tp, row0, l15, g, res);
    rw_epi_params(W, tp, g, P);
    rw_epi_finish(W, tp, acc, P, res, tiles, row0, l15, g, tstride);
}

// LeakyReLU -> LayerNorm backward of the wave's rows (EqdChainJob.type 1; same mathematics as chain_lnbwd64); the loads
// of a tile's saved activations (rw_lnbwd_rows) apart from the arithmetic, for the same reason as in the epilogue
__device__ __forceinline__ void rw_lnbwd_rows(const JobW& W, int row0, int l15, int g, f32x4 (&y)[4]) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int rows = jw_i(W, LJ(rows));
    const int rowi = row0 + l15;
    const float* yp = jw_p<const float>(W, LJ(s) + JW_OFF(EqdLinSrc, X)) +
                      (size_t)(rowi < rows ? rowi : rows - 1) * jw_i(W, LJ(s) + JW_OFF(EqdLinSrc, ldx)) + 4 * g;
#pragma unroll
    for (int a = 0; a < 4; ++a) y[a] = *(const EQD_GAS f4v*)(yp + 16 * a);
#undef LJ
}
__device__ __forceinline__ void rw_lnbwd_finish(const JobW& W, const f32x4 (&yin)[4], const f32x4 (&gam)[4], float* tiles,
                                                float* red, int row0, int l15, int g, int tstride) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int rows = jw_i(W, LJ(rows));
    const int src_l = jw_i(W, JW_OFF(EqdChainJob, src_local)), out_l = jw_i(W, JW_OFF(EqdChainJob, out_local));
    const float slope = jw_f(W, LJ(slope)), ln_eps = jw_f(W, LJ(ln_eps));
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    f32x4 y[4], o[4];
    const float* T = tiles + src_l * tstride + l15 * RW_S + 4 * g;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        o[a] = *(const f32x4*)(T + 16 * a);
        y[a] = yin[a];
    }
    if (!rv) {
#pragma unroll
        for (int a = 0; a < 4; ++a) o[a] = y[a] = f4zero();
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) s += (y[a][0] + y[a][1]) + (y[a][2] + y[a][3]);
    const float mean = group_sum(s) * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float c = y[a][b] - mean;
            q += c * c;
        }
    const float rstd = 1.f / sqrtf(group_sum(q) * (1.f / 64.f) + ln_eps);
    float p1 = 0.f, p2 = 0.f;
    f32x4 xh[4], dx[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            xh[a][b] = (y[a][b] - mean) * rstd;
            dx[a][b] = o[a][b] * gam[a][b];
            p1 += dx[a][b];
            p2 += dx[a][b] * xh[a][b];
        }
    const float s1 = group_sum(p1) * (1.f / 64.f), s2 = group_sum(p2) * (1.f / 64.f);
    f32x4 z[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            z[a][b] = rv ? rstd * (dx[a][b] - s1 - xh[a][b] * s2) : 0.f;
    if (const float* const jmul = jw_p<const float>(W, LJ(mul))) {      // dropout factors of the forward (training mode):
        const float* mp = jmul + (size_t)(rv ? rowi : rows - 1) * jw_i(W, LJ(ld_mul)) + 4 * g;      // d LeakyReLU * keep * s
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 mm = *(const EQD_GAS f4v*)(mp + 16 * a);
#pragma unroll
            for (int b = 0; b < 4; ++b) z[a][b] *= lrelu_grad(y[a][b], slope) * mm[b];
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) z[a][b] *= lrelu_grad(y[a][b], slope);
    }
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));
    if (jY && rv) {
        float* zp = jY + (size_t)rowi * ldy + 4 * g;
#pragma unroll
        for (int a = 0; a < 4; ++a) *(EQD_GAS f4v*)(zp + 16 * a) = z[a];
    }
    if (out_l >= 0) {
        float* To = tiles + out_l * tstride + l15 * RW_S + 4 * g;
#pragma unroll
        for (int a = 0; a < 4; ++a) *(f32x4*)(To + 16 * a) = z[a];
        wave_lds_fence();
    }
    // d gamma / d beta over the wave's 16 rows: lane l15 ends with the sum of entry l15 = 4 a + b (feature 16 a + 4 g + b)
    float vg[16], vb[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            vg[4 * a + b] = o[a][b] * xh[a][b];      // rows beyond the matrix carry o = 0
            vb[4 * a + b] = o[a][b];
        }
    const float dg = reduce16x16(vg, l15), db = reduce16x16(vb, l15);
    const int f = 16 * (l15 >> 2) + 4 * g + (l15 & 3);
    red[f] += dg;
    red[128 + f] += db;
#undef LJ
}
__device__ __forceinline__ void rw_lnbwd_gamma(const JobW& W, int g, f32x4 (&gam)[4]) {
    const float* const jg = jw_p<const float>(W, JW_OFF(EqdLinJob, ln_g));
#pragma unroll
    for (int a = 0; a < 4; ++a) gam[a] = *(const EQD_GAS f4v*)(jg + 16 * a + 4 * g);
}
__device__ __forceinline__ void rw_lnbwd(const JobW& W, float* tiles, float* red, int row0, int l15, int g,
                                         int tstride = 16 * RW_S) {
    f32x4 y[4], gam[4];
    rw_lnbwd_rows(W, row0, l15, g, y);
    rw_lnbwd_gamma(W, g, gam);
    rw_lnbwd_finish(W, y, gam, tiles, red, row0, l15, g, tstride);
}

// One wave per 16-row tile; workgroup = RW_WAVES independent waves (they only meet to add their LayerNorm-backward
// partial sums: one row of `aux` per workgroup, like k_rowchain).
template <bool BF>
__global__ __launch_bounds__(64 * RW_WAVES, 2) void k_rowwave(EqdChainArg A_) {
    __shared__ __attribute__((aligned(16))) EqdChainArg A;
    __shared__ __attribute__((aligned(16))) RwSmem sm;
    kernarg_to_lds(A, EQD_KERNARG_PTR(A_), 0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    float* const red = sm.red[wave];
    for (int i = lane; i < 256; i += 64) red[i] = 0.f;
    __syncthreads();
    constexpr int CJ_DW = (int)(sizeof(EqdChainJob) / 4);
    const int njobs = uni(A.njobs);
    const int row0 = ((int)blockIdx.x * RW_WAVES + wave) * 16;
    float* const tiles = &sm.tile[wave][0][0];
    float* aux = nullptr;
    JobW Wc = jobw_load(&A.j[0], CJ_DW, lane);
    const int rows = jw_i(Wc, JW_OFF(EqdLinJob, rows));
    if (row0 < rows) {      // wave-uniform
        int rowc = row0 + l15;
        rowc = rowc < rows ? rowc : rows - 1;
        int jj = 0;
        // LayerNorm-backward jobs in front of the first linear job
        while (jj < njobs && jw_i(Wc, JW_OFF(EqdChainJob, type)) != 0) {
            rw_lnbwd(Wc, tiles, red, row0, l15, g);
            aux = jw_p<float>(Wc, JW_OFF(EqdChainJob, aux));
            ++jj;
            if (jj < njobs) Wc = jobw_load(&A.j[jj], CJ_DW, lane);
        }
        int trc = 0;      // sub-step counter of the phase-trace experiments (profiles/exp_trace_rowwave.py)
        (void)trc;
        // linear jobs, one after the other; per job a two-stage pipeline over its sub-steps (32 source columns each):
        // the loads of sub-step s + 1 are in flight while sub-step s is multiplied
        while (jj < njobs) {
            const int nsrc = jw_i(Wc, JW_OFF(EqdLinJob, nsrc));
            const bool tp = jw_i(Wc, JW_OFF(EqdLinJob, s) + JW_OFF(EqdLinSrc, w_cs)) != 1;
            const float slope = jw_f(Wc, JW_OFF(EqdLinJob, slope));
            f32x4 acc[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
            RwBuf R;
            RwOps<BF> O;
            int si = 0, h = 0;
            rw_load(Wc, 0, 0, jw_i(Wc, JW_OFF(EqdChainJob, src_local)) >= 0, true, rowc, l15, g, R);
            for (;;) {
                EQD_TR(100 + 3 * trc);
                const int sb = JW_OFF(EqdLinJob, s) + si * JW_SRC_DW;
                const int Ks = jw_i(Wc, sb + JW_OFF(EqdLinSrc, K));
                const int loc = jw_i(Wc, JW_OFF(EqdChainJob, src_local) + si);
                const bool masked = jw_p<const float>(Wc, sb + JW_OFF(EqdLinSrc, mask)) != nullptr;
                rw_take<BF>(R, tp, masked && loc < 0, slope, h == 2, Ks - 64 - 4 * g, O);      // waits for the loads
                const bool src_end = h == 2 || (h == 1 && Ks <= 64);
                const bool more = !(src_end && si + 1 >= nsrc);
                const int nsi = src_end ? si + 1 : si, nh = src_end ? 0 : h + 1;
                if (more) rw_load(Wc, nsi, nh, jw_i(Wc, JW_OFF(EqdChainJob, src_local) + nsi) >= 0, true, rowc, l15, g, R);
                EQD_TR(101 + 3 * trc);
                f32x4 Bl[2] = {f4zero(), f4zero()};
                if (loc >= 0) {
                    const float* T = tiles + loc * (16 * RW_S) + l15 * RW_S + 4 * g + 32 * h;
#pragma unroll
                    for (int i = 0; i < 2; ++i) Bl[i] = *(const f32x4*)(T + 16 * i);
                }
                rw_mma<BF>(O, loc >= 0, Bl, tp, h == 2 ? 1 : 2, acc);
                EQD_TR(102 + 3 * trc);
                ++trc;
                if (!more) break;
                si = nsi;
                h = nh;
            }
            rw_epilogue(Wc, tp, acc, tiles, row0, l15, g);
            // the LayerNorm-backward jobs behind it, then the next linear job
            ++jj;
            while (jj < njobs) {
                Wc = jobw_load(&A.j[jj], CJ_DW, lane);
                if (jw_i(Wc, JW_OFF(EqdChainJob, type)) == 0) break;
                rw_lnbwd(Wc, tiles, red, row0, l15, g);
                aux = jw_p<float>(Wc, JW_OFF(EqdChainJob, aux));
                ++jj;
            }
        }
    }
    __syncthreads();
    // (wave 0 always has rows: its tile index is 4 blockIdx.x < number of tiles)
    if (wave == 0 && aux) {
        float* ap = aux + (size_t)blockIdx.x * 256;
        for (int i = lane; i < 256; i += 64) ap[i] = (sm.red[0][i] + sm.red[1][i]) + (sm.red[2][i] + sm.red[3][i]);
    }
}
