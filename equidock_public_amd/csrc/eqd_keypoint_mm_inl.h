// Keypoint pooling (rigid_docking_model.py:521-567) as matrix products.
//
// scores[i][k] = h_i . u[s][k]  is a [nodes of the segment] x 64 x [K heads] product, and its backward is two more:
// du[k][:] = sum_i dscores[i][k] h_i (contraction over the nodes) and dH[i][:] = sum_k dscores[i][k] u[k][:] (contraction
// over the heads).  The first kernels of this stage ran one workgroup per (segment, head): every head re-read the
// segment's h rows (50 x), a thread walked a whole 256-byte row, scores went out as 4-byte stores K floats apart - 0.03 of
// the HBM roofline and nothing on the matrix cores (profiles/r04_y9_bench_*.log).  Here:
//   forward   one workgroup per (segment, block of 16 heads); a wave takes every fourth 16-row tile: 16 MFMAs
//             (items on N, heads on M, u in registers for the whole launch), scores leave as 16-byte stores, the softmax
//             statistics and Y = att^T Z are lane-local sums reduced once per workgroup (reduce16x16 + one LDS exchange).
//   backward  one workgroup per segment, wave w = heads 16 w .. 16 w + 15 over ALL rows: the softmax backward in the
//             layout the forward MFMA left (lane = row, 4 heads per lane group) - its row sums need no exchange at all,
//             the per-head dot over the rows is a 16-lane sum -, dH's partial product straight from those registers,
//             du's after a wave-private 16 x 16 LDS transpose; the four waves' dH / dZ partials of a tile meet in LDS
//             (double buffered: one barrier per tile).  No dscores array in HBM.
// fp32 throughout, as the first kernels (the head is fp32 in every mode); sums run in a different order.
#pragma once
#include "eqd_common.h"

#define KPM_TB 4      /* forward: row tiles of a wave whose loads are in flight together */

// NW: waves per workgroup - 4, or 16 for batches of few long segments (4 x 2 000 residues: 32 workgroups of 125 tiles)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_keypoint_mm(const int32_t* __restrict__ seg_off, int K,
                                                           const float* __restrict__ u, const float* __restrict__ H,
                                                           const float* __restrict__ Z, float* __restrict__ Y,
                                                           float* __restrict__ Yl_out, float* __restrict__ Yr_out, int B,
                                                           float* __restrict__ scores, float* __restrict__ lse,
                                                           float* __restrict__ Yc) {
    // Yc (optional): the keypoints RELATIVE to the segment's first node, Yc = Y - Z[n0], for the backward.  The sums run over
    // z - Z[n0] in any case: real structures lie 100 - 300 A from the origin of their PDB frame, and the softmax backward
    // needs z - Y to better than fp32(Y) can carry at that magnitude (k_keypoint_bwd_mm).
    __shared__ float sred[NW][16][4];
    __shared__ float smx[16];
    const int s = blockIdx.x, hb = blockIdx.y, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = seg_off[s], n1 = seg_off[s + 1];
    const int nt = (n1 - n0 + 15) >> 4;
    // A operand: head 16 hb + l15 (clamped; the surplus heads of the last block are never stored), k = 16 q + 4 g + j
    int ha = 16 * hb + l15;
    ha = ha < K ? ha : K - 1;
    f32x4 au[4];
    {
        const float* up = u + ((size_t)s * K + ha) * 64 + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) au[q] = *(const EQD_GAS f4v*)(up + 16 * q);
    }
    const int h0 = 16 * hb + 4 * g;      // this lane's result heads h0 + r
    const int nh = K - h0;               // r < nh are real
    float mx[4] = {EQD_NEG_BIG, EQD_NEG_BIG, EQD_NEG_BIG, EQD_NEG_BIG};
    for (int t0 = wave; t0 < nt; t0 += NW * KPM_TB) {
        f32x4 b[KPM_TB][4];
#pragma unroll
        for (int i = 0; i < KPM_TB; ++i) {
            int row = n0 + 16 * (t0 + NW * i) + l15;
            row = row < n1 ? row : n1 - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) b[i][q] = *(const EQD_GAS f4v*)(H + (size_t)row * 64 + 16 * q + 4 * g);
        }
#pragma unroll
        for (int i = 0; i < KPM_TB; ++i) {
            const int tile = t0 + NW * i;
            if (tile < nt) {
                f32x4 acc = f4zero();
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = mfma4(au[q][j], b[i][q][j], acc);
                const int row = n0 + 16 * tile + l15;
                if (row < n1) {
                    float* sp = scores + (size_t)row * K + h0;
                    if (nh >= 4) {
                        *(EQD_GAS f4v*)sp = acc;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r < nh) ((EQD_GAS float*)sp)[r] = acc[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = mx[r];
        v = fmaxf(v, lane_xor<1>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<8>(v));
        mx[r] = v;
    }
    if (l15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sred[wave][4 * g + r][0] = mx[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        float m = sred[0][4 * g + r][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, sred[w][4 * g + r][0]);
        mx[r] = m;
    }
    if (wave == 0 && l15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) smx[4 * g + r] = mx[r];
    }
    __syncthreads();
    // second pass over the wave's own tiles (it reads back what its lanes stored): v[4 r + 0] = sum of exp, v[4 r + 1 + c] = Y
    const float c0 = ((const EQD_GAS float*)Z)[(size_t)n0 * 3], c1 = ((const EQD_GAS float*)Z)[(size_t)n0 * 3 + 1],
                c2 = ((const EQD_GAS float*)Z)[(size_t)n0 * 3 + 2];
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
#pragma unroll 2
    for (int tile = wave; tile < nt; tile += NW) {
        const int row = n0 + 16 * tile + l15;
        const bool rv = row < n1;
        const int rowc = rv ? row : n1 - 1;
        const float* sp = scores + (size_t)rowc * K + h0;
        f32x4 sc = f4zero();
        if (nh >= 4) {
            sc = *(const EQD_GAS f4v*)sp;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nh) sc[r] = ((const EQD_GAS float*)sp)[r];
        }
        const float z0 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3] - c0, z1 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3 + 1] - c1,
                    z2 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3 + 2] - c2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = (rv && r < nh) ? expf(sc[r] - mx[r]) : 0.f;
            v[4 * r] += p;
            v[4 * r + 1] += p * z0;
            v[4 * r + 2] += p * z1;
            v[4 * r + 3] += p * z2;
        }
    }
    const float tot = reduce16x16(v, l15);      // lane l15: value (l15 & 3) of head 4 g + (l15 >> 2)
    sred[wave][4 * g + (l15 >> 2)][l15 & 3] = tot;
    __syncthreads();
    if (t < 16 && 16 * hb + t < K) {
        const int k = 16 * hb + t;
        float q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w += 4) a += (sred[w][t][c] + sred[w + 1][t][c]) + (sred[w + 2][t][c] + sred[w + 3][t][c]);
            q[c] = a;
        }
        const float se = q[0];
        const float inv = se > 0.f ? 1.f / se : 0.f;
        const float yc0 = q[1] * inv, yc1 = q[2] * inv, yc2 = q[3] * inv;
        const float y0 = se > 0.f ? c0 + yc0 : 0.f, y1 = se > 0.f ? c1 + yc1 : 0.f, y2 = se > 0.f ? c2 + yc2 : 0.f;
        float* y = Y + ((size_t)s * K + k) * 3;
        y[0] = y0; y[1] = y1; y[2] = y2;
        if (Yc) {
            float* yc = Yc + ((size_t)s * K + k) * 3;
            yc[0] = yc0; yc[1] = yc1; yc[2] = yc2;
        }
        float* yo = s < B ? (Yl_out ? Yl_out + ((size_t)s * K + k) * 3 : nullptr)
                          : (Yr_out ? Yr_out + ((size_t)(s - B) * K + k) * 3 : nullptr);
        if (yo) {
            yo[0] = y0; yo[1] = y1; yo[2] = y2;
        }
        lse[(size_t)s * K + k] = se > 0.f ? smx[t] + logf(se) : 0.f;
    }
}

// ---- backward: one workgroup per segment, wave w = heads 16 w .. 16 w + 15 (K <= 64) -----------------------------------
#define KPB_RS 20      /* exchange row stride (floats): 20 l15 + 4 g is conflict-free for 16 lanes x 16 B */
__global__ __launch_bounds__(EQD_BLOCK) void k_keypoint_bwd_mm(const int32_t* __restrict__ seg_off, int K,
                                                               const float* __restrict__ H, const float* __restrict__ Z,
                                                               const float* __restrict__ scores,
                                                               const float* __restrict__ lse, const float* __restrict__ u,
                                                               const float* __restrict__ dY, const float* __restrict__ Yk,
                                                               float* __restrict__ du, float* __restrict__ dH,
                                                               float* __restrict__ dZ) {
    // gridDim.y = NC row chunks per segment: chunk c takes the segment's tiles [c tpc, (c + 1) tpc) and writes its du to
    // du[(s NC + c) K + k][:] (NC == 1: the result itself, else partial sums that k_head_u_bwd adds up in chunk order)
    __shared__ __attribute__((aligned(16))) float xh[2][EQD_WAVES][4][16 * KPB_RS];      // dH partials [buffer][wave][c block][row][16 c]
    __shared__ float xz[2][EQD_WAVES][16][4];
    __shared__ float tr[EQD_WAVES][16][17];                                                // wave-private [head][row]
    const int s = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = seg_off[s], n1 = seg_off[s + 1];
    const int nt = (n1 - n0 + 15) >> 4;
    // lane = row l15 of a tile, heads hB + j
    const int hB = 16 * wave + 4 * g;
    const int nh = K - hB;      // j < nh are real
    float Lj[4], dyj[4][3];
    float uaj[4][4];            // uaj[cb][j] = u[head hB + j][16 cb + l15]: A operand of dH's product
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int hc = hB + j;
        hc = hc < K ? hc : K - 1;
        Lj[j] = lse[(size_t)s * K + hc];
#pragma unroll
        for (int c = 0; c < 3; ++c) dyj[j][c] = dY[((size_t)s * K + hc) * 3 + c];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) uaj[cb][j] = u[((size_t)s * K + hc) * 64 + 16 * cb + l15];
    }
    auto load_sc = [&](int rowc) {
        const float* sp = scores + (size_t)rowc * K + hB;
        f32x4 sc = f4zero();
        if (nh >= 4) {
            sc = *(const EQD_GAS f4v*)sp;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nh) sc[j] = ((const EQD_GAS float*)sp)[j];
        }
        return sc;
    };
    // The softmax backward is dscore = att (dY[k] . z - sum_rows att (dY[k] . z)), and the sum is dY[k] . Y[k] (Y = att^T Z, the
    // forward's result).  It is evaluated as att * (dY[k] . (z - Y[k])) - the coordinates are differenced FIRST: real structures
    // lie 100 - 300 A from the origin of their PDB frame, both dot products are then ~ |dY| x 250 and their difference - what
    // matters - ~ |dY| x 20, a decimal digit lost before du sums 1 270 of them to nearly zero (round 6: the head parameters'
    // gradients of a 1 270 + 40-residue complex were 7e-4 from a float64 evaluation where torch's fp32 is 5e-5; now 4e-5).
    // Both z and Y are taken RELATIVE to the segment's first node c = Z[n0] (fp32(Y) at |Y| = 250 is only good to 1.5e-5 A -
    // with a sharp softmax z - Y of the dominant row is 1e-3 A): yk[j] = Yc[hB + j] = Y - c, written by k_keypoint_mm; callers
    // without it (the operator-level entry point) get it from a pass over the segment's rows.
    const float c0 = ((const EQD_GAS float*)Z)[(size_t)n0 * 3], c1 = ((const EQD_GAS float*)Z)[(size_t)n0 * 3 + 1],
                c2 = ((const EQD_GAS float*)Z)[(size_t)n0 * 3 + 2];
    float yk[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) yk[j][0] = yk[j][1] = yk[j][2] = 0.f;
    if (Yk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int hc = hB + j;
            hc = hc < K ? hc : K - 1;
            const float* y = Yk + ((size_t)s * K + hc) * 3;
            yk[j][0] = y[0]; yk[j][1] = y[1]; yk[j][2] = y[2];
        }
    }
#pragma unroll 2
    for (int tile = 0; tile < (Yk ? 0 : nt); ++tile) {
        const int row = n0 + 16 * tile + l15;
        const bool rv = row < n1;
        const int rowc = rv ? row : n1 - 1;
        const f32x4 sc = load_sc(rowc);
        const float z0 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3] - c0, z1 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3 + 1] - c1,
                    z2 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3 + 2] - c2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = (rv && j < nh) ? expf(sc[j] - Lj[j]) : 0.f;
            yk[j][0] += a * z0;
            yk[j][1] += a * z1;
            yk[j][2] += a * z2;
        }
    }
    if (!Yk) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) yk[j][c] = l16_sum(yk[j][c]);
    }
    const int NC = (int)gridDim.y, chunk = (int)blockIdx.y;
    const int tpc = (nt + NC - 1) / NC;
    const int tb = chunk * tpc, te = tb + tpc < nt ? tb + tpc : nt;
    // pass 2
    f32x4 accU[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) accU[cb] = f4zero();
    for (int tile = tb; tile < te; ++tile) {
        const int p = tile & 1;
        const int row = n0 + 16 * tile + l15;
        const bool rv = row < n1;
        const int rowc = rv ? row : n1 - 1;
        const f32x4 sc = load_sc(rowc);
        const float z0 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3] - c0, z1 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3 + 1] - c1,
                    z2 = ((const EQD_GAS float*)Z)[(size_t)rowc * 3 + 2] - c2;
        f32x4 hv[4];      // rows 4 ks + g of the tile, columns 4 l15 ..: B operand of du's product
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            int r2 = n0 + 16 * tile + 4 * ks + g;
            r2 = r2 < n1 ? r2 : n1 - 1;
            hv[ks] = *(const EQD_GAS f4v*)(H + (size_t)r2 * 64 + 4 * l15);
        }
        float ds[4], dz0 = 0.f, dz1 = 0.f, dz2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = rv && j < nh;
            const float a = ok ? expf(sc[j] - Lj[j]) : 0.f;
            ds[j] = a * (dyj[j][0] * (z0 - yk[j][0]) + dyj[j][1] * (z1 - yk[j][1]) + dyj[j][2] * (z2 - yk[j][2]));
            dz0 += a * dyj[j][0];
            dz1 += a * dyj[j][1];
            dz2 += a * dyj[j][2];
        }
        // dH partial of this wave's heads: D[c][row]
        f32x4 accH[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            accH[cb] = f4zero();
#pragma unroll
            for (int j = 0; j < 4; ++j) accH[cb] = mfma4(uaj[cb][j], ds[j], accH[cb]);
        }
        dz0 = group_sum(dz0);
        dz1 = group_sum(dz1);
        dz2 = group_sum(dz2);
        // dscores of the tile, transposed inside the wave: A operand of du's product (head l15, rows 4 ks + g)
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 4; ++j) tr[wave][4 * g + j][l15] = ds[j];
        wave_lds_fence();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float a = tr[wave][l15][4 * ks + g];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) accU[cb] = mfma4(a, hv[ks][cb], accU[cb]);
        }
        // the four waves' partials of the tile meet in LDS
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *(f32x4*)&xh[p][wave][cb][l15 * KPB_RS + 4 * g] = accH[cb];
        if (g == 0) {
            xz[p][wave][l15][0] = dz0;
            xz[p][wave][l15][1] = dz1;
            xz[p][wave][l15][2] = dz2;
        }
        __syncthreads();
        {
            const int rr = t >> 4, c4 = (t & 15) * 4, cb = c4 >> 4, off = c4 & 15;
            const int orow = n0 + 16 * tile + rr;
            if (orow < n1) {
                f32x4 o;
                const f32x4 p0 = *(const f32x4*)&xh[p][0][cb][rr * KPB_RS + off], p1 = *(const f32x4*)&xh[p][1][cb][rr * KPB_RS + off],
                            p2 = *(const f32x4*)&xh[p][2][cb][rr * KPB_RS + off], p3 = *(const f32x4*)&xh[p][3][cb][rr * KPB_RS + off];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (p0[r] + p1[r]) + (p2[r] + p3[r]);
                *(EQD_GAS f4v*)(dH + (size_t)orow * 64 + c4) = o;
            }
            if (t < 48) {
                const int zr = t / 3, zc = t - 3 * zr;
                const int zrow = n0 + 16 * tile + zr;
                if (zrow < n1) dZ[(size_t)zrow * 3 + zc] = (xz[p][0][zr][zc] + xz[p][1][zr][zc]) + (xz[p][2][zr][zc] + xz[p][3][zr][zc]);
            }
        }
    }
    // du[head 16 w + 4 g + r][4 l15 + cb]
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r < nh) {
            const f32x4 o = {accU[0][r], accU[1][r], accU[2][r], accU[3][r]};
            *(EQD_GAS f4v*)(du + (((size_t)s * NC + chunk) * K + hB + r) * 64 + 4 * l15) = o;
        }
}
