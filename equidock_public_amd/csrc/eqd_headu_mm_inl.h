// The per-head key / query maps of the keypoint head (rigid_docking_model.py:524-541) as matrix products.
//
//   forward   qp[s][k] = W_Q^(k) qmean[partner(s)],   u[s][k] = W_K^(k)T qp[s][k] / 8
//   backward  dqp = W_K^(k) du / 8,  dqm_part[s][k] = W_Q^(k)T dqp,  dW_K^(k) += sum_s qp (x) du / 8,  dW_Q^(k) += sum_s dqp (x) qmean
// Both are the same pair of products - Y1 = A1 X, Y2 = A2^T Y1 with 64 x 64 matrices per head and one column per segment -
// and the backward adds two outer-product sums over the segments.  The first kernels computed them with scalar FMAs: one
// 64-thread workgroup per (segment, head) walking weight rows (forward: 6 400 workgroups re-reading a head's 32 KB 128 times at
// 64 pairs), operands through LDS four bytes at a time (backward).  Here a workgroup is (head, block of up to 32 segments),
// items = segments on the MFMA N axis:
//   Y1 (wave w = rows 16 w .. 16 w + 15 of A1): both operands are 16-byte global loads (k = 16 q + 4 g + j on either side);
//   Y2: wave w contracts ITS 16 rows of Y1 - the accumulator of the first product IS the B operand of the second - against
//       A2 rows loaded 16 bytes along the output axis (component cb of the load feeds the MFMA whose output row l15 is column
//       4 l15 + cb), and the four waves' partial Y2 meet in LDS once per 16 segments;
//   outer products: k = segments; the first factor of dW_Q is dqp TRANSPOSED, which is the same first product with its operands
//       swapped (16 more MFMAs on registers that are already there) instead of a trip through LDS.
// fp32; sums over the contraction index / the segments run in MFMA order (the first kernels: sequential).
#pragma once
#include "eqd_common.h"

#define HUM_XS 68      /* exchange row stride (floats) */
struct alignas(16) HuMmSmem {
    float xs[EQD_WAVES][16 * HUM_XS];      // [wave][segment][64 outputs]
};

// A1 fragment of wave `wave`: rows 16 wave + l15, columns 16 q + 4 g ..
__device__ __forceinline__ void hum_load_a1(const float* __restrict__ A1, int wave, int l15, int g, f32x4 (&a)[4]) {
    const float* p = A1 + (size_t)(16 * wave + l15) * 64 + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *(const EQD_GAS f4v*)(p + 16 * q);
}
// A2 fragment: rows 16 wave + 4 g + r (r = 0..3), columns 4 l15 ..
__device__ __forceinline__ void hum_load_a2(const float* __restrict__ A2, int wave, int l15, int g, f32x4 (&a)[4]) {
    const float* p = A2 + (size_t)(16 * wave + 4 * g) * 64 + 4 * l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = *(const EQD_GAS f4v*)(p + 64 * r);
}
// y1[r] = Y1[segment l15][16 wave + 4 g + r] = sum_c A1[16 wave + 4 g + r][c] X[l15][c]
__device__ __forceinline__ f32x4 hum_y1(const f32x4 (&a1)[4], const f32x4 (&x)[4]) {
    f32x4 acc = f4zero();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma4(a1[q][j], x[q][j], acc);
    return acc;
}
// the same product transposed: y1t[r] = Y1[segment 4 g + r][16 wave + l15]
__device__ __forceinline__ f32x4 hum_y1t(const f32x4 (&a1)[4], const f32x4 (&x)[4]) {
    f32x4 acc = f4zero();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma4(x[q][j], a1[q][j], acc);
    return acc;
}
// Y2 = A2^T Y1 for 16 segments: every wave adds its 16 rows of Y1, the partial sums meet in LDS; thread t then holds
// Y2[segment t >> 4][4 (t & 15) .. + 3].  Two barriers; all 256 threads.
__device__ __forceinline__ f32x4 hum_y2(const f32x4 (&a2)[4], const f32x4 y1, HuMmSmem& sm, int wave, int l15, int g, int t) {
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        acc[cb] = f4zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[cb] = mfma4(a2[r][cb], y1[r], acc[cb]);
    }
    // acc[cb][r']: output 4 (4 g + r') + cb of segment l15
    __syncthreads();      // the previous block's sums have been read
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
        *(f32x4*)&sm.xs[wave][l15 * HUM_XS + 16 * g + 4 * rr] = f32x4{acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]};
    __syncthreads();
    const int o = (t >> 4) * HUM_XS + 4 * (t & 15);
    const f32x4 p0 = *(const f32x4*)&sm.xs[0][o], p1 = *(const f32x4*)&sm.xs[1][o], p2 = *(const f32x4*)&sm.xs[2][o],
                p3 = *(const f32x4*)&sm.xs[3][o];
    f32x4 s;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = (p0[r] + p1[r]) + (p2[r] + p3[r]);
    return s;
}

// forward: grid (K, ceil(2 B / segs_per_group))
__global__ __launch_bounds__(EQD_BLOCK) void k_head_u_mm(int B, int K, const float* __restrict__ Wk, const float* __restrict__ Wq,
                                                         const float* __restrict__ qmean, float* __restrict__ qp,
                                                         float* __restrict__ u, int segs_per_group) {
    __shared__ HuMmSmem sm;
    const int k = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int S = 2 * B, Sbeg = (int)blockIdx.y * segs_per_group;
    const int Send = S < Sbeg + segs_per_group ? S : Sbeg + segs_per_group;
    f32x4 a1[4], a2[4];
    hum_load_a1(Wq + (size_t)k * 4096, wave, l15, g, a1);
    hum_load_a2(Wk + (size_t)k * 4096, wave, l15, g, a2);
    for (int s0 = Sbeg; s0 < Send; s0 += 16) {
        const int s = s0 + l15;
        const bool sv = s < Send;
        const int sc = sv ? s : Send - 1;
        const int partner = sc < B ? sc + B : sc - B;
        f32x4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q] = *(const EQD_GAS f4v*)(qmean + (size_t)partner * 64 + 16 * q + 4 * g);
            if (!sv) x[q] = f4zero();
        }
        const f32x4 y1 = hum_y1(a1, x);
        if (sv) *(EQD_GAS f4v*)(qp + ((size_t)s * K + k) * 64 + 16 * wave + 4 * g) = y1;
        f32x4 y2 = hum_y2(a2, y1, sm, wave, l15, g, t);
        const int so = s0 + (t >> 4);
        if (so < Send) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y2[r] *= 0.125f;
            *(EQD_GAS f4v*)(u + ((size_t)so * K + k) * 64 + 4 * (t & 15)) = y2;
        }
    }
}

// backward: grid (K, groups); part == NULL: the block adds its sums to dWk / dWq itself, else it writes the partial
// [group][Wk | Wq][head][64 x 64] (same layout as the first kernel).  du_chunks > 1: du = the keypoint backward's partial blocks.
__global__ __launch_bounds__(EQD_BLOCK) void k_head_u_bwd_mm(int B, int K, const float* __restrict__ Wk,
                                                             const float* __restrict__ Wq, const float* __restrict__ qmean,
                                                             const float* __restrict__ qp, const float* __restrict__ du,
                                                             int du_chunks, float* __restrict__ dWk, float* __restrict__ dWq,
                                                             float* __restrict__ dqm_part, float* __restrict__ part,
                                                             int segs_per_group) {
    __shared__ HuMmSmem sm;
    const int k = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int S = 2 * B, Sbeg = (int)blockIdx.y * segs_per_group;
    const int Send = S < Sbeg + segs_per_group ? S : Sbeg + segs_per_group;
    f32x4 a1[4], a2[4];
    hum_load_a1(Wk + (size_t)k * 4096, wave, l15, g, a1);
    hum_load_a2(Wq + (size_t)k * 4096, wave, l15, g, a2);
    // the gradient rows this lane finishes: j = 16 wave + 4 g + r, columns 4 l15 .. (requested now, added at the end)
    f32x4 oldK[4], oldQ[4];
    if (!part) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t o = ((size_t)k * 64 + 16 * wave + 4 * g + r) * 64 + 4 * l15;
            oldK[r] = *(const EQD_GAS f4v*)(dWk + o);
            oldQ[r] = *(const EQD_GAS f4v*)(dWq + o);
        }
    }
    auto du_at = [&](int s, int col) {      // du[s][k][col .. col + 3] / 8, chunks summed in order
        f32x4 d = f4zero();
        if (du_chunks <= 1) {
            d = *(const EQD_GAS f4v*)(du + ((size_t)s * K + k) * 64 + col);
        } else {
            for (int ch = 0; ch < du_chunks; ++ch) {
                const f32x4 v = *(const EQD_GAS f4v*)(du + (((size_t)s * du_chunks + ch) * K + k) * 64 + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] += v[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] *= 0.125f;
        return d;
    };
    f32x4 accK[4], accQ[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) accK[cb] = accQ[cb] = f4zero();
    for (int s0 = Sbeg; s0 < Send; s0 += 16) {
        // operands by segment l15 (the two products) ...
        const int s = s0 + l15;
        const bool sv = s < Send;
        const int sc = sv ? s : Send - 1;
        f32x4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q] = du_at(sc, 16 * q + 4 * g);
            if (!sv) x[q] = f4zero();
        }
        // ... and by segment 4 g + r (the outer products: k-step r of the lane group)
        float qpa[4];
        f32x4 dub[4], qmb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s2 = s0 + 4 * g + r;
            const bool v2 = s2 < Send;
            const int c2 = v2 ? s2 : Send - 1;
            const int partner = c2 < B ? c2 + B : c2 - B;
            qpa[r] = v2 ? qp[((size_t)c2 * K + k) * 64 + 16 * wave + l15] : 0.f;
            dub[r] = du_at(c2, 4 * l15);
            qmb[r] = *(const EQD_GAS f4v*)(qmean + (size_t)partner * 64 + 4 * l15);
            if (!v2) {
                dub[r] = f4zero();
                qmb[r] = f4zero();
            }
        }
        const f32x4 y1 = hum_y1(a1, x);        // dqp[segment l15][16 wave + 4 g + r]
        const f32x4 y1t = hum_y1t(a1, x);      // dqp[segment 4 g + r][16 wave + l15]
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                accK[cb] = mfma4(qpa[r], dub[r][cb], accK[cb]);
                accQ[cb] = mfma4(y1t[r], qmb[r][cb], accQ[cb]);
            }
        const f32x4 y2 = hum_y2(a2, y1, sm, wave, l15, g, t);
        const int so = s0 + (t >> 4);
        if (so < Send) *(EQD_GAS f4v*)(dqm_part + ((size_t)so * K + k) * 64 + 4 * (t & 15)) = y2;
    }
    // accK[cb][r]: row 16 wave + 4 g + r, column 4 l15 + cb
    float* pk = part ? part + ((size_t)blockIdx.y * 2 * K + k) * 4096 : dWk + (size_t)k * 4096;
    float* pq = part ? pk + (size_t)K * 4096 : dWq + (size_t)k * 4096;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t o = (size_t)(16 * wave + 4 * g + r) * 64 + 4 * l15;
        f32x4 vk = {accK[0][r], accK[1][r], accK[2][r], accK[3][r]}, vq = {accQ[0][r], accQ[1][r], accQ[2][r], accQ[3][r]};
        if (!part) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vk[i] += oldK[r][i];
                vq[i] += oldQ[r][i];
            }
        }
        *(EQD_GAS f4v*)(pk + o) = vk;
        *(EQD_GAS f4v*)(pq + o) = vq;
    }
}
