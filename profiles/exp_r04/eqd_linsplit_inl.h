// Split job: the form of a lean linear job (one 16-row tile, M == 64, every source 64..80 wide, one weight orientation)
// for launches with at most one tile per CU (DB5.5-sized batches: 200 tiles on 256 CUs, one wave per SIMD).
//
// There a job's time is the length of its dependent chain, not a throughput: the lean body walks a job source by source
// and every step is  global load -> LDS store -> barrier -> LDS read -> 16 MFMAs on one accumulator pair -> barrier
// (1 750 clocks, profiles/exp_step_floor.hip), because the four waves share a staged copy of the step's weights and each
// owns 16 of the 64 outputs.  Here the four waves split the CONTRACTION instead: wave w takes columns 16 w .. 16 w + 15
// of every 64-wide source and computes all 64 outputs of them.
//   * Its operands come from global memory straight into the lanes that feed the MFMA - 16-byte loads along the
//     contiguous axis in both weight orientations (below) - so there is no staging, no barrier inside the contraction,
//     and every load of the job (all sources) is in flight before the first MFMA; the weights of the next job of a
//     chain are requested at the same time (LinSplitCarry).
//   * 16 MFMAs per source on FOUR independent accumulators.
//   * One exchange per job: every wave leaves its four partial blocks in LDS (conflict-free b128), a barrier, and wave w
//     sums the four partials of output block w in a fixed order - after that the registers hold exactly what the lean
//     body's accumulators hold (lane = row l15, features 16 w + 4 g .. + 3) and the epilogue is the same code.
// k-step j of a wave's slice belongs to lane group g as k = 16 w + 4 g + j in both operands, so
//   w_cs == 1 (forward, W[m][k]):     lane (l15, g) loads W[16 i + l15][16 w + 4 g ..] for the four blocks i; component j
//                                     is the A operand of k-step j for block i;
//   w_rs == 1 (backward, W[k][m]):    lane (l15, g) loads W[16 w + 4 g + j][4 l15 ..] for the four k-steps j; component i
//                                     is the A operand of an MFMA whose output row l15 is feature 4 l15 + i, so
//                                     accumulator i holds, in lane (l15, g) component r, feature 16 g + 4 r + i: block g,
//                                     and the exchange stores it as such.
// The sums run in a different order than in the lean body (four partial sums over k instead of one): results differ at
// fp32 rounding, each launch is deterministic.
#pragma once

#define LSP_RS 20      /* exchange row stride in floats: 20 l15 + 4 g (mod 64 banks) is conflict-free for 16 lanes x 16 B */
struct alignas(16) LinSplitSmem {
    float xr[4][EQD_WAVES][16 * LSP_RS];      // [output block][source wave][row][16 features]
};
template <bool ON>
struct LinSplitSmemOpt {      // kernels without the split body carry no exchange buffer
    LinSplitSmem s;
};
template <>
struct LinSplitSmemOpt<false> {
    float s[4];
};
struct LinSplitCarry {      // the first two sources' weights of a chain's next linear job, requested a job ahead
    f32x4 w[2][4];
};
template <int AHEAD>
struct LspFrag {
    f32x4 w[AHEAD][4], x[AHEAD], xm[AHEAD];
};

#define LJ(f) JW_OFF(EqdLinJob, f)
#define LSRC(si, f) (LJ(s) + (si) * JW_SRC_DW + JW_OFF(EqdLinSrc, f))
// which jobs the split body takes (all of a descriptor's facts are wave-uniform)
__device__ __forceinline__ bool lsp_eligible(const JobW& W) {
    const int nsrc = jw_i(W, LJ(nsrc));
    bool ok = jw_i(W, LJ(M)) == 64 && nsrc >= 1 && nsrc <= EQD_MAX_SRC && jw_i(W, LJ(pad_to)) <= 64;
    const bool kf = jw_i(W, LSRC(0, w_cs)) == 1;
#pragma unroll
    for (int si = 0; si < EQD_MAX_SRC; ++si)
        if (si < nsrc) {
            const int K = jw_i(W, LSRC(si, K));
            if (K < 64 || K > 80) ok = false;
            if ((jw_i(W, LSRC(si, w_cs)) == 1) != kf) ok = false;
            if (!kf && jw_i(W, LSRC(si, w_rs)) != 1) ok = false;
        }
    return ok;
}
__device__ __forceinline__ void lsp_load_w(const EqdLinSrc& S, int wave, int l15, int g, f32x4 (&w)[4]) {
    if (S.w_cs == 1) {
        const float* p = S.W + (size_t)l15 * S.w_rs + 16 * wave + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = *(const EQD_GAS f4v*)(p + (size_t)(16 * i) * S.w_rs);
    } else {
        const float* p = S.W + (size_t)(16 * wave + 4 * g) * S.w_cs + 4 * l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *(const EQD_GAS f4v*)(p + (size_t)j * S.w_cs);
    }
}

// CHAIN: the job is an EqdChainJob (LDS tiles as sources / result; Lb is not touched otherwise).
// AHEAD: sources whose loads are in flight together (a register budget: 24 per source).  have_carry / cin: the weights of
// sources 0 and 1 arrived with the previous job; want_next / Wn / cout: request those of the next linear job Wn (the
// caller checked lsp_eligible(Wn)).  Must be called by all 256 threads of the workgroup.
template <int AHEAD, bool CHAIN>
__device__ __forceinline__ void linear_tile_split(const JobW& W, int out_local, LinSplitSmem& xs,
                                                  float (*stat)[16], float (*Lb)[LIN_LOCALS][16 * LIN_S], int row0,
                                                  bool have_carry, const LinSplitCarry& cin, bool want_next, const JobW& Wn,
                                                  LinSplitCarry& cout, int trace_slot = 255) {
    (void)trace_slot;
    EQD_TR(trace_slot);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int rows = jw_i(W, LJ(rows)), nsrc = jw_i(W, LJ(nsrc));
    const float slope = jw_f(W, LJ(slope));
    const bool kf = jw_i(W, LSRC(0, w_cs)) == 1;
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const int rowe = rv ? rowi : rows - 1;
    const int kc = 16 * wave + 4 * g;          // this lane's four columns of a 64-wide source
    LspFrag<AHEAD> F;
    int locs[EQD_MAX_SRC];
    unsigned mkm = 0u, remm = 0u;
#pragma unroll
    for (int si = 0; si < EQD_MAX_SRC; ++si) {
        locs[si] = -1;
        if (si < nsrc) {
            if constexpr (CHAIN) locs[si] = jw_i(W, JW_OFF(EqdChainJob, src_local) + si);
            if (jw_p<const float>(W, LSRC(si, mask))) mkm |= 1u << si;
            if (jw_i(W, LSRC(si, K)) > 64) remm |= 1u << si;
        }
    }
    auto load = [&](int si, int slot) __attribute__((always_inline)) {       // si, slot: compile-time constants at every call
        const EqdLinSrc S = jw_src(W, si);
        if (si < 2 && have_carry) {
#pragma unroll
            for (int i = 0; i < 4; ++i) F.w[slot][i] = cin.w[si][i];
        } else {
            lsp_load_w(S, wave, l15, g, F.w[slot]);
        }
        if (locs[si] < 0) {
            const size_t o = (size_t)rowe * S.ldx + kc;
            F.x[slot] = *(const EQD_GAS f4v*)(S.X + o);
            if ((mkm >> si) & 1u) F.xm[slot] = *(const EQD_GAS f4v*)(S.mask + o);
        }
    };
#pragma unroll
    for (int si = 0; si < AHEAD; ++si)
        if (si < nsrc) load(si, si);
    if (want_next) {
        const int nn = jw_i(Wn, LJ(nsrc));
#pragma unroll
        for (int si = 0; si < 2; ++si)
            if (si < nn) {
                const EqdLinSrc Sn = jw_src(Wn, si);
                lsp_load_w(Sn, wave, l15, g, cout.w[si]);
            }
    }
    // epilogue operands of this wave's output block (features f0 .. f0 + 3 of row rowi): behind the job's own loads
    const int f0 = 16 * wave + 4 * g;
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    const float* const jlng = jw_p<const float>(W, LJ(ln_g));
    const float* const jR = jw_p<const float>(W, LJ(R));
    const float* const jmul = jw_p<const float>(W, LJ(mul));
    f32x4 bias = f4zero(), lg = f4zero(), lb = f4zero(), res = f4zero(), mm = {1.f, 1.f, 1.f, 1.f};
    if (jbias) bias = *(const EQD_GAS f4v*)(jbias + f0);
    if (jlng) {
        lg = *(const EQD_GAS f4v*)(jlng + f0);
        lb = *(const EQD_GAS f4v*)(jw_p<const float>(W, LJ(ln_b)) + f0);
    }
    if (jR) res = *(const EQD_GAS f4v*)(jR + (size_t)rowe * jw_i(W, LJ(ldr)) + f0);
    if (jmul) mm = *(const EQD_GAS f4v*)(jmul + (size_t)rowe * jw_i(W, LJ(ld_mul)) + f0);

    EQD_TR(trace_slot + 1);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4zero();
#pragma unroll
    for (int si = 0; si < EQD_MAX_SRC; ++si) {
        if (si < nsrc) {
            const int slot = si % AHEAD;
            f32x4 b = F.x[slot];
            if constexpr (CHAIN) {
                if (locs[si] >= 0) b = *(const f32x4*)&Lb[0][locs[si]][l15 * LIN_S + kc];
            }
            if ((mkm >> si) & 1u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] *= lrelu_grad(F.xm[slot][i], slope);
            }
            if (kf) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = mfma4(F.w[slot][i][j], b[j], acc[i]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = mfma4(F.w[slot][j][i], b[j], acc[i]);
            }
            if (si + AHEAD < EQD_MAX_SRC) {
                if (si + AHEAD < nsrc) load(si + AHEAD < EQD_MAX_SRC ? si + AHEAD : 0, slot);
            }
        }
    }
    // what is left of sources wider than 64 (the 5 columns beyond 64 of the 69-wide h0): wave w takes k = 64 + 4 w + g
    if (remm) {
#pragma unroll
        for (int si = 0; si < EQD_MAX_SRC; ++si) {
            if (si < nsrc && ((remm >> si) & 1u)) {
                const EqdLinSrc S = jw_src(W, si);
                if (64 + 4 * wave < S.K) {
                    const int k = 64 + 4 * wave + g;
                    const bool kv = k < S.K;
                    const int ke = kv ? k : S.K - 1;
                    float a[4], b;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = kf ? 16 * i + l15 : 4 * l15 + i;
                        a[i] = ((const EQD_GAS float*)S.W)[(size_t)m * S.w_rs + (size_t)ke * S.w_cs];
                    }
                    if (locs[si] < 0) {
                        b = ((const EQD_GAS float*)S.X)[(size_t)rowe * S.ldx + ke];
                        if ((mkm >> si) & 1u) b *= lrelu_grad(((const EQD_GAS float*)S.mask)[(size_t)rowe * S.ldx + ke], slope);
                    } else {
                        b = 0.f;
                        if constexpr (CHAIN) b = Lb[0][locs[si]][l15 * LIN_S + ke];
                    }
                    if (!kv) b = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = mfma4(kv ? a[i] : 0.f, b, acc[i]);
                }
            }
        }
    }
    EQD_TR(trace_slot + 2);
    // ---- exchange: the partial sums of output block i, from wave `wave`, row l15 -------------------------------------
    if (kf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)&xs.xr[i][wave][l15 * LSP_RS + 4 * g] = acc[i];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(f32x4*)&xs.xr[g][wave][l15 * LSP_RS + 4 * r] = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    }
    __syncthreads();
    f32x4 sum;
    {
        const f32x4 p0 = *(const f32x4*)&xs.xr[wave][0][l15 * LSP_RS + 4 * g];
        const f32x4 p1 = *(const f32x4*)&xs.xr[wave][1][l15 * LSP_RS + 4 * g];
        const f32x4 p2 = *(const f32x4*)&xs.xr[wave][2][l15 * LSP_RS + 4 * g];
        const f32x4 p3 = *(const f32x4*)&xs.xr[wave][3][l15 * LSP_RS + 4 * g];
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = (p0[r] + p1[r]) + (p2[r] + p3[r]);
    }
    EQD_TR(trace_slot + 3);
    // ---- epilogue: feature f0 + r of row rowi (as in linear_tile_lean) ------------------------------------------------
    const int act = jw_i(W, LJ(act));
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = sum[r] + bias[r];
        if (act) y = lrelu(y, slope);
        v[r] = y;
    }
    if (jmul) {      // dropout factors (training mode only)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= mm[r];
    }
    if (jlng) {
        const float invM = 1.f / 64.f;
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s1 += v[r];
        s1 = group_sum(s1);
        if (g == 0) stat[wave][l15] = s1;
        __syncthreads();
        const float mean = (stat[0][l15] + stat[1][l15] + stat[2][l15] + stat[3][l15]) * invM;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dlt = v[r] - mean;
            q += dlt * dlt;
        }
        q = group_sum(q);
        __syncthreads();
        if (g == 0) stat[wave][l15] = q;
        __syncthreads();
        const float rstd = 1.f / sqrtf((stat[0][l15] + stat[1][l15] + stat[2][l15] + stat[3][l15]) * invM + jw_f(W, LJ(ln_eps)));
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
    if constexpr (CHAIN) {
        if (out_local >= 0) *(f32x4*)&Lb[0][out_local][l15 * LIN_S + f0] = yv;
    }
}
#undef LSRC
#undef LJ
