// k_rowres80: k_rowres (eqd_rowres_inl.h: persistent 8-wave workgroups, a source's weights resident in LDS) for the
// 69-wide FIRST layer's forward jobs in bf16 mode - outputs up to 80 wide (five 16-feature blocks per wave), LDS-resident
// intermediate tiles up to 80 wide.  Round 5: at 64 x (300, 300) the first layer's projection group and
// node-update chain ran on the four-wave kernels (k_linear<2> 77 us, k_rowchain 135 us per launch against 36 us for a 64-wide
// layer's chain on k_rowres) because k_rowres takes 64-wide outputs only.
//
// Scope (rw80_eligible, host): bf16 mode; outputs M <= 80; global sources 64..80 wide without mask, LDS-resident sources (the
// tile written last) up to 80 wide; weights of either orientation (k-contiguous: the forward Linears; m-contiguous: the
// backward's dX = dY W); LayerNorm-backward jobs up to 80 features - i.e. the first layer's projection group, node-update
// chain and backward chain.  Differences to k_rowres:
//   * weights staged as bf16 [80][88] (rows >= M and columns >= K zero) whatever their orientation in memory, four 16-byte
//     loads per thread and source;
//   * five accumulator blocks per tile slot; a job with M <= 64 uses four (wave-uniform);
//   * tiles of 84 floats per row (80 + 4: 16-byte aligned rows, 21 l15 + g covers the banks), columns >= M written as zeros so
//     that a consumer's remainder chunk (columns 64..79) needs no mask;
//   * the epilogue handles rows of any stride (69-float rows of y_act: scalar stores) and LayerNorm over M features.
// Same products as the four-wave kernels, summed in another order (bf16 mode: inputs rounded at the same points).
#pragma once
#include "eqd_rowres_inl.h"

#define R8_KP 88         /* bf16 row stride of the staged weights: 44 dwords = 12 mod 32 -> b64 fragment reads at the LDS rate */
#define R8_TS 84         /* float row stride of a wave-private tile */
#define R8_WROWS 80

struct Rr80Smem {
    unsigned short Wl[2][R8_WROWS * R8_KP];
    float tile[RR_WAVES][RR_TMAX][16 * R8_TS];
    float red[RR_WAVES][256];      // LayerNorm-backward partial sums of a wave's rows: [d gamma 0..127 | d beta 128..255]
};

struct Rr80Stage {
    f32x4 v[4];
};
// k-contiguous W[m][k] (w_cs == 1; rows may be unaligned: 69-float rows): idx -> (m, 4-column group c4 of 20 along k);
// m-contiguous W[k][m] (w_rs == 1): idx -> (k, 4-column group c4 of 20 along m), transposed by the staging stores
__device__ __forceinline__ void rr80_stage_load(const EqdLinSrc S, int M, int t, Rr80Stage& R) {
    const bool tp = S.w_cs != 1;
    const int nrow = tp ? S.K : M, ncol = tp ? M : S.K;      // rows / columns of the matrix as it lies in memory
    const int stride = tp ? S.w_cs : S.w_rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = t + 64 * RR_WAVES * j;
        const int r = idx / 20, c4 = idx - r * 20;
        const int rr = r < nrow ? r : nrow - 1;
        R.v[j] = ld4u_raw(S.W + (size_t)rr * stride + 4 * c4, idx < R8_WROWS * 20 ? ncol - 4 * c4 : 0, S.W);
    }
}
__device__ __forceinline__ void rr80_stage_store(const EqdLinSrc S, int M, int t, const Rr80Stage& R, unsigned short* Wl) {
    const bool tp = S.w_cs != 1;
    const int nrow = tp ? S.K : M, ncol = tp ? M : S.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = t + 64 * RR_WAVES * j;
        const int r = idx / 20, c4 = idx - r * 20;
        if (idx < R8_WROWS * 20) {
            const float4 f = ld4u_fix(R.v[j], r < nrow ? ncol - 4 * c4 : 0);      // zeros beyond the matrix
            if (!tp) {
                *(s16x4*)&Wl[r * R8_KP + 4 * c4] = pack_bf4(f.x, f.y, f.z, f.w);
            } else {      // (r = k, columns 4 c4 .. = m): element (m, k) -> Wl[m][k]
                Wl[(4 * c4 + 0) * R8_KP + r] = f2bf(f.x);
                Wl[(4 * c4 + 1) * R8_KP + r] = f2bf(f.y);
                Wl[(4 * c4 + 2) * R8_KP + r] = f2bf(f.z);
                Wl[(4 * c4 + 3) * R8_KP + r] = f2bf(f.w);
            }
        }
    }
}

// one (source, tile) item: MB output blocks (4 or 5, wave-uniform), K <= 80
__device__ __forceinline__ void rr80_item(const unsigned short* __restrict__ Wl, int K, int nmb, bool local, const float* T,
                                          const RrRows& R, int l15, int g, f32x4 (&acc)[5]) {
#pragma unroll
    for (int ap = 0; ap < 2; ++ap) {
        f32x4 b0, b1;
        if (local) {
            b0 = *(const f32x4*)(T + l15 * R8_TS + 32 * ap + 4 * g);
            b1 = *(const f32x4*)(T + l15 * R8_TS + 32 * ap + 16 + 4 * g);
        } else {
            b0 = R.x[2 * ap];
            b1 = R.x[2 * ap + 1];
        }
        const s16x8 bp = cat_bf(pack_bf4(b0[0], b0[1], b0[2], b0[3]), pack_bf4(b1[0], b1[1], b1[2], b1[3]));
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
            if (mb < nmb)
                acc[mb] = mfma_bf32(cat_bf(*(const s16x4*)&Wl[(16 * mb + l15) * R8_KP + 32 * ap + 4 * g],
                                           *(const s16x4*)&Wl[(16 * mb + l15) * R8_KP + 32 * ap + 16 + 4 * g]), bp, acc[mb]);
    }
    if (K > 64) {      // remainder columns 64 .. K - 1: one 16-deep chunk (zeros beyond K on both sides)
        s16x4 bp;
        if (local) {
            const f32x4 b = *(const f32x4*)(T + l15 * R8_TS + 64 + 4 * g);
            bp = pack_bf4(b[0], b[1], b[2], b[3]);
        } else {
            const float4 xf = ld4u_fix(R.xr, K - 64 - 4 * g);
            bp = pack_bf4(xf.x, xf.y, xf.z, xf.w);
        }
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
            if (mb < nmb) acc[mb] = mfma_bf(*(const s16x4*)&Wl[(16 * mb + l15) * R8_KP + 64 + 4 * g], bp, acc[mb]);
    }
}

// Y = alpha * f(acc + bias) + beta * R for the wave's 16 rows, M <= 80 outputs: lane (l15, g) holds features 16 a + 4 g + b of
// row row0 + l15 (the S layout).  EqdLinJob semantics (include/equidock_hip.h); every row pointer may have any stride.
__device__ __forceinline__ void rr80_epilogue(const JobW& W, int nmb, const f32x4 (&acc)[5], float* tile, int row0, int l15,
                                              int g) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int rows = jw_i(W, LJ(rows)), M = jw_i(W, LJ(M));
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const size_t rowe = (size_t)(rv ? rowi : rows - 1);
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    const float* const jlng = jw_p<const float>(W, LJ(ln_g));
    const float* const jlnb = jw_p<const float>(W, LJ(ln_b));
    const float* const jR = jw_p<const float>(W, LJ(R));
    const float* const jmul = jw_p<const float>(W, LJ(mul));
    const int ldr = jw_i(W, LJ(ldr)), ld_mul = jw_i(W, LJ(ld_mul));
    const float slope = jw_f(W, LJ(slope));
    const int act = jw_i(W, LJ(act));
    // a 4-feature group f0 .. f0 + 3 of a parameter vector / a row, zeros beyond M (vector ENDING at the last valid element)
    auto ld4 = [&](const float* p, int f0) -> f32x4 {
        const int n = M - f0;
        const float4 f = ld4u_fix(ld4u_raw(p + f0, n, p), n);
        return f32x4{f.x, f.y, f.z, f.w};
    };
    f32x4 v[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int f0 = 16 * a + 4 * g;
        v[a] = f4zero();
        if (a < nmb) {
            v[a] = acc[a];
            if (jbias) v[a] += ld4(jbias, f0);
            if (act) {
#pragma unroll
                for (int b = 0; b < 4; ++b) v[a][b] = lrelu(v[a][b], slope);
            }
            if (jmul) {      // dropout factors (training mode)
                const f32x4 mm = ld4(jmul + rowe * ld_mul, f0);
#pragma unroll
                for (int b = 0; b < 4; ++b) v[a][b] *= mm[b];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] = f0 + b < M ? v[a][b] : 0.f;
        }
    }
    if (jlng) {
        float* const jpre = jw_p<float>(W, LJ(pre_ln));
        const int ld_pre = jw_i(W, LJ(ld_pre));
        if (jpre && rv) {
            float* pp = jpre + (size_t)rowi * ld_pre;
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (a < nmb && 16 * a + 4 * g + b < M) ((EQD_GAS float*)pp)[16 * a + 4 * g + b] = v[a][b];
        }
        const float invM = 1.f / (float)M;
        float s1 = 0.f;
#pragma unroll
        for (int a = 0; a < 5; ++a) s1 += (v[a][0] + v[a][1]) + (v[a][2] + v[a][3]);      // (zeros beyond M)
        const float mean = group_sum(s1) * invM;
        float q = 0.f;
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float dlt = 16 * a + 4 * g + b < M ? v[a][b] - mean : 0.f;
                q += dlt * dlt;
            }
        const float rstd = 1.f / sqrtf(group_sum(q) * invM + jw_f(W, LJ(ln_eps)));
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            if (a < nmb) {
                const int f0 = 16 * a + 4 * g;
                const f32x4 lg = ld4(jlng, f0), lb = ld4(jlnb, f0);
#pragma unroll
                for (int b = 0; b < 4; ++b) v[a][b] = f0 + b < M ? (v[a][b] - mean) * rstd * lg[b] + lb[b] : 0.f;
            }
        }
    }
    const float alpha = jw_f(W, LJ(alpha)), beta = jw_f(W, LJ(beta));
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        if (a < nmb) {
            const int f0 = 16 * a + 4 * g;
            f32x4 res = f4zero();
            if (jR) res = ld4(jR + rowe * ldr, f0);
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] = f0 + b < M ? alpha * v[a][b] + (jR ? beta * res[b] : 0.f) : 0.f;
        }
    }
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy)), pad_to = jw_i(W, LJ(pad_to));
    if (jY && rv) {
        float* yp = jY + (size_t)rowi * ldy;
        const bool vec = (ldy & 3) == 0 && ((((uintptr_t)jY) & 15) == 0);      // 16-byte aligned rows: whole groups as vectors
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const int f0 = 16 * a + 4 * g;
            if (a < nmb) {
                if (vec && (f0 + 4 <= M || f0 + 4 <= pad_to)) {      // (columns M .. pad_to - 1 are zeros already)
                    *(EQD_GAS f4v*)(yp + f0) = v[a];
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (f0 + b < M || f0 + b < pad_to) ((EQD_GAS float*)yp)[f0 + b] = v[a][b];
                }
            }
        }
    }
    unsigned short* const jYb = jw_p<unsigned short>(W, LJ(Yb));
    const int ldyb = jw_i(W, LJ(ldyb));
    if (jYb && rv) {      // bf16 copy: whole 4-column groups that start below M (zeros beyond M), rows 8-byte aligned (ldyb % 4 == 0)
        unsigned short* yp = jYb + (size_t)rowi * ldyb;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const int f0 = 16 * a + 4 * g;
            if (a < nmb && f0 < M && f0 + 4 <= ldyb) *(EQD_GAS s16x4*)(yp + f0) = pack_bf4(v[a][0], v[a][1], v[a][2], v[a][3]);
        }
    }
    const int out_l = jw_i(W, JW_OFF(EqdChainJob, out_local));
    if (out_l >= 0) {      // all 80 columns: zeros beyond M (a consumer's remainder chunk reads 64 .. 79)
        float* T = tile + l15 * R8_TS;
#pragma unroll
        for (int a = 0; a < 5; ++a) *(f32x4*)(T + 16 * a + 4 * g) = v[a];
        wave_lds_fence();
    }
#undef LJ
}

// LeakyReLU -> LayerNorm backward of the wave's rows over M <= 80 features (EqdChainJob.type 1; the mathematics of
// rw_lnbwd_finish / chain_lnbwd): incoming gradient = the wave's tile, saved activations y_act = lin.s[0].X (rows of any
// stride), dz -> the tile (zeros beyond M) and lin.Y, per-wave sums of d gamma / d beta -> red
__device__ __forceinline__ void rr80_lnbwd(const JobW& W, float* tile, float* red, int row0, int l15, int g) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int rows = jw_i(W, LJ(rows)), M = jw_i(W, LJ(M));
    const int nmb = (M + 15) >> 4;
    const float slope = jw_f(W, LJ(slope)), ln_eps = jw_f(W, LJ(ln_eps));
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const size_t rowe = (size_t)(rv ? rowi : rows - 1);
    const float* const yrow = jw_p<const float>(W, LJ(s) + JW_OFF(EqdLinSrc, X)) + rowe * jw_i(W, LJ(s) + JW_OFF(EqdLinSrc, ldx));
    const float* const jg = jw_p<const float>(W, LJ(ln_g));
    const float* const jmul = jw_p<const float>(W, LJ(mul));
    const int ld_mul = jw_i(W, LJ(ld_mul));
    auto ld4 = [&](const float* p, int f0) -> f32x4 {
        const int n = M - f0;
        const float4 f = ld4u_fix(ld4u_raw(p + f0, n, p), n);
        return f32x4{f.x, f.y, f.z, f.w};
    };
    // (register diet: only y and o stay live across the passes - gamma is folded into o's copy dx, the normalised
    //  activations are recomputed where they are used, the dropout factors are read per block in the last pass)
    f32x4 y[5], o[5], dx[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int f0 = 16 * a + 4 * g;
        y[a] = o[a] = dx[a] = f4zero();
        if (a < nmb) {
            y[a] = ld4(yrow, f0);
            o[a] = *(const f32x4*)(tile + l15 * R8_TS + f0);      // (zeros beyond M: the producer's epilogue)
            const f32x4 gam = ld4(jg, f0);
#pragma unroll
            for (int b = 0; b < 4; ++b) dx[a][b] = o[a][b] * gam[b];
        }
        if (!rv) o[a] = y[a] = dx[a] = f4zero();
    }
    const float invM = 1.f / (float)M;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a) s += (y[a][0] + y[a][1]) + (y[a][2] + y[a][3]);
    const float mean = group_sum(s) * invM;
    float q = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float c = 16 * a + 4 * g + b < M ? y[a][b] - mean : 0.f;
            q += c * c;
        }
    const float rstd = 1.f / sqrtf(group_sum(q) * invM + ln_eps);
    auto xh_of = [&](int a, int b) { return 16 * a + 4 * g + b < M ? (y[a][b] - mean) * rstd : 0.f; };
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            p1 += dx[a][b];
            p2 += dx[a][b] * xh_of(a, b);
        }
    const float s1 = group_sum(p1) * invM, s2 = group_sum(p2) * invM;
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));
    float* const zp = jY ? jY + (size_t)(rv ? rowi : 0) * ldy : nullptr;
    float* const T = tile + l15 * R8_TS;
#pragma unroll
    for (int a = 0; a < 5; ++a) {      // dz -> lin.Y and the tile (zeros beyond M; out_local names the wave's one tile)
        const int f0 = 16 * a + 4 * g;
        f32x4 mm = {1.f, 1.f, 1.f, 1.f};
        if (jmul && a < nmb) mm = ld4(jmul + rowe * ld_mul, f0);
        f32x4 z;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool ok = rv && f0 + b < M;
            z[b] = ok ? rstd * (dx[a][b] - s1 - xh_of(a, b) * s2) * (lrelu_grad(y[a][b], slope) * mm[b]) : 0.f;
            if (zp && rv && a < nmb && f0 + b < M) ((EQD_GAS float*)zp)[f0 + b] = z[b];
        }
        *(f32x4*)(T + f0) = z;
    }
    wave_lds_fence();
    // d gamma / d beta over the wave's 16 rows: blocks 0..3 through the 16 x 16 butterfly (lane l15 ends with entry l15 = 4 a +
    // b, feature 16 a + 4 g + b), block 4 (features 64 + 4 g + b) with four plain 16-lane sums
    float vg[16], vb[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            vg[4 * a + b] = o[a][b] * xh_of(a, b);      // rows beyond the matrix carry o = 0
            vb[4 * a + b] = o[a][b];
        }
    const float dg = reduce16x16(vg, l15), db = reduce16x16(vb, l15);
    const int f = 16 * (l15 >> 2) + 4 * g + (l15 & 3);
    red[f] += dg;
    red[128 + f] += db;
    if (nmb > 4) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float sg = l16_sum(o[4][b] * xh_of(4, b)), sb = l16_sum(o[4][b]);
            if (l15 == b) {
                red[64 + 4 * g + b] += sg;
                red[128 + 64 + 4 * g + b] += sb;
            }
        }
    }
#undef LJ
}

__global__ __launch_bounds__(64 * RR_WAVES, 1) void k_rowres80(EqdChainArg A_, int tps) {
    __shared__ __attribute__((aligned(16))) EqdChainArg A;
    __shared__ __attribute__((aligned(16))) Rr80Smem sm;
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
    float* aux = nullptr;
    int nt_wg = ntiles - tile0;
    nt_wg = nt_wg < tps ? nt_wg : tps;
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
    int buf = 0;
    Rr80Stage WS;
    RrRows XR[RR_TMAX];
    const float* held_x = nullptr;      // the global source whose rows XR holds (wave-uniform)
    int held_ld = 0, held_k = 0;
    // (the host only sends chains whose FIRST job is a linear one: rw80_eligible)
    {      // the first source's weights
        const EqdLinSrc S0 = jw_src(Wc, 0);
        const int M0 = jw_i(Wc, JW_OFF(EqdLinJob, M));
        rr80_stage_load(S0, M0, t, WS);
        rr80_stage_store(S0, M0, t, WS, sm.Wl[0]);
    }
    __syncthreads();
    int jj = 0;
    while (jj < njobs) {
        // the next LINEAR job behind this one (LayerNorm-backward jobs in between run in this job's tail), or -1
        int nl = jj + 1;
        while (nl < njobs && uni(A.j[nl].type) != 0) ++nl;
        const bool have_nj = nl < njobs;
        const int nsrc = jw_i(Wc, JW_OFF(EqdLinJob, nsrc));
        const int M = jw_i(Wc, JW_OFF(EqdLinJob, M));
        const int nmb = (M + 15) >> 4;
        JobW Wnl = Wc;
        if (have_nj) Wnl = jobw_load(&A.j[nl], CJ_DW, lane);
        const int Mn = jw_i(Wnl, JW_OFF(EqdLinJob, M));
        f32x4 acc[RR_TMAX][5];
#pragma unroll
        for (int s = 0; s < RR_TMAX; ++s)
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[s][q] = f4zero();
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
            const EqdLinSrc S = jw_src(Wc, si);
            const int loc = jw_i(Wc, JW_OFF(EqdChainJob, src_local) + si);
            const bool last_src = si + 1 >= nsrc;
            const bool have_next = !last_src || have_nj;
            const EqdLinSrc Sn = last_src ? jw_src(Wnl, 0) : jw_src(Wc, si + 1);
            const int nloc = last_src ? 0 : jw_i(Wc, JW_OFF(EqdChainJob, src_local) + si + 1);
            RrRows XC[RR_TMAX];
            if (loc < 0) {
#pragma unroll
                for (int s = 0; s < RR_TMAX; ++s)
                    if (s < nslots) rr_rows_take(S, 0.f, rowcs[s], g, XR[s], XC[s]);      // waits for the rows (no masks here)
            }
            if (have_next) rr80_stage_load(Sn, last_src ? Mn : M, t, WS);
            const bool next_rows = last_src ? (have_nj && jw_i(Wnl, JW_OFF(EqdChainJob, src_local)) < 0) : nloc < 0;
            if (next_rows && !(held_x == Sn.X && held_ld == Sn.ldx && held_k == Sn.K)) {
#pragma unroll
                for (int s = 0; s < RR_TMAX; ++s)
                    if (s < nslots) rr_rows_load(Sn, rowcs[s], g, XR[s]);
                held_x = Sn.X; held_ld = Sn.ldx; held_k = Sn.K;
            }
#pragma unroll
            for (int s = 0; s < RR_TMAX; ++s)
                if (s < nslots) rr80_item(sm.Wl[buf], S.K, nmb, loc >= 0, sm.tile[wave][s], XC[s], l15, g, acc[s]);
            if (have_next) rr80_stage_store(Sn, last_src ? Mn : M, t, WS, sm.Wl[buf ^ 1]);
            __syncthreads();      // every wave is done with Wl[buf]; Wl[buf ^ 1] is complete
            buf ^= 1;
        }
#pragma unroll
        for (int s = 0; s < RR_TMAX; ++s)
            if (s < nslots) rr80_epilogue(Wc, nmb, acc[s], sm.tile[wave][s], row0s[s], l15, g);
        // the LayerNorm-backward jobs behind it, then the next linear job
        if (nl > jj + 1) {
            for (int q = jj + 1; q < nl; ++q) {
                const JobW Wl_ = jobw_load(&A.j[q], CJ_DW, lane);
#pragma unroll
                for (int s = 0; s < RR_TMAX; ++s)
                    if (s < nslots) rr80_lnbwd(Wl_, sm.tile[wave][s], red, row0s[s], l15, g);
                aux = jw_p<float>(Wl_, JW_OFF(EqdChainJob, aux));
            }
            // rows fetched ahead for the next job are dropped (it re-requests them): the 40 registers of the row buffers are
            // then dead across the LayerNorm backward instead of being parked in scratch memory around it
            held_x = nullptr;
#pragma unroll
            for (int s = 0; s < RR_TMAX; ++s) {
#pragma unroll
                for (int a = 0; a < 4; ++a) XR[s].x[a] = f4zero();
                XR[s].xr = f4zero();
            }
        }
        Wc = Wnl;
        jj = nl;
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
