// Device bodies of the by-source / by-destination gather (k_node_gather) and of the fixed-order partial reduction
// (reduce_block), shared by eqd_node_kernels.hip (their own launches) and eqd_attn_kernels.hip (the launch in which the
// gather of a layer rides beside the layer's attention backward, k_attn_bwd_gather).
#pragma once
#include "eqd_common.h"

template <int PL>
__device__ __forceinline__ void reduce_block(const EqdRedArg& A, int blk, float (*red)[68], float (*red2)[64]) {
    int ch = 0;
    while (ch + 1 < A.nchains && blk >= A.chain_blk0[ch + 1]) ++ch;
    ch = uni(ch);
    const int first = A.chain_first[ch], len = A.chain_len[ch];
    const int t = threadIdx.x, cg = t & 15, pl = t >> 4;
    const int c0 = (blk - A.chain_blk0[ch]) * 64;
    const int n = A.s[first].n;
    const int col = c0 + 4 * cg;            // this thread's columns col .. col + 3
    const int nv = n - col;                 // how many of them exist (<= 0: none)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n >= 4) {
        for (int j = 0; j < len; ++j) {
            const EqdRedSeg& S = A.s[first + j];
            const float* __restrict__ base = S.partial + (nv > 0 ? col : 0);
            const int np = S.nparts;
            const size_t ps = (size_t)S.pstride;
            for (int p0 = pl; p0 < np; p0 += 8 * PL) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + PL * u;
                    v[u] = ld4u_raw(base + (size_t)(p < np ? p : 0) * ps, nv, S.partial);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 f = ld4u_fix(v[u], p0 + PL * u < np ? nv : 0);
                    acc.x += f.x;
                    acc.y += f.y;
                    acc.z += f.z;
                    acc.w += f.w;
                }
            }
        }
    } else if (cg == 0) {                    // 1 .. 3 columns in all (a scalar bias): scalar loads
        float a3[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < len; ++j) {
            const EqdRedSeg& S = A.s[first + j];
            for (int p = pl; p < S.nparts; p += PL)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = S.partial[(size_t)p * S.pstride + (c < n ? c : 0)];
                    a3[c] += c < n ? v : 0.f;
                }
        }
        acc = make_float4(a3[0], a3[1], a3[2], 0.f);
    }
    *(float4*)&red[pl][4 * cg] = acc;
    __syncthreads();
    if (t < 256) {
        const int c = t & 63, qd = t >> 6;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < PL / 4; ++j) s += red[(PL / 4) * qd + j][c];
        red2[qd][c] = s;
    }
    __syncthreads();
    const int i = c0 + t;
    if (t < 64 && i < n) {
        const float s = (red2[0][t] + red2[1][t]) + (red2[2][t] + red2[3][t]);
        const EqdRedSeg& S0 = A.s[first];
        if (S0.cols > 0) {
            const int row = i / S0.cols, cc = i - row * S0.cols;
            if (cc < S0.cols_valid) S0.out[(size_t)row * S0.ld_out + cc] += s;
        } else {
            S0.out[i] += s;
        }
    }
}

template <bool BF>
__device__ __forceinline__ f32x4 gather_dz4(const float* __restrict__ dz, size_t e, int c4) {
    if constexpr (BF) {
        typedef unsigned gather_u32x2 __attribute__((ext_vector_type(2)));
        const gather_u32x2 h = *(const gather_u32x2*)((const unsigned short*)dz + e * 64 + 4 * c4);      // 4 bf16
        return f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                     __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
    } else {
        return *(const f32x4*)(dz + e * 64 + 4 * c4);
    }
}
#define GATHER_NODES 16      /* nodes per workgroup: 4 waves x 4 */
// what one gather launch needs (eqd_launch_node_gather; filled by the edge backward for whoever launches it)
struct EqdGatherArgs {
    const int32_t *csc_ptr, *csc_eid, *rowptr;
    int n;
    const float *dz, *dxrel, *d_xnew;
    float a;
    float *dP, *dQ, *dx;
    int ngather;       // workgroups of the gather itself; the launch's workgroups beyond them run reduce_block
    int bf16;          // dz holds bf16 rows
};
template <bool BF>
__device__ __forceinline__ void node_gather_body(const EqdGatherArgs& GA, int blk) {
    const int32_t* __restrict__ csc_ptr = GA.csc_ptr;
    const int32_t* __restrict__ csc_eid = GA.csc_eid;
    const int32_t* __restrict__ rowptr = GA.rowptr;
    const int n = GA.n;
    const float* __restrict__ dz = GA.dz;
    const float* __restrict__ dxrel = GA.dxrel;
    const float* __restrict__ d_xnew = GA.d_xnew;
    const float a = GA.a;
    float* __restrict__ dP = GA.dP;
    float* __restrict__ dQ = GA.dQ;
    float* __restrict__ dx = GA.dx;
    const int jr = blk * GATHER_NODES + (threadIdx.x >> 4);
    const int c4 = threadIdx.x & 15;
    const bool live = jr < n;
    const int j = live ? jr : n - 1;      // lanes beyond the last node repeat it (unconditional loads) and store nothing
    const int s0 = csc_ptr[j], s1 = csc_ptr[j + 1];
    const int d0 = rowptr[j], d1 = rowptr[j + 1];
    const int nout = s1 - s0, nin = d1 - d0;
    f32x4 sp = f4zero(), sq = f4zero();
    float sx = 0.f;                       // component c4 & 3 of the dx sum (lanes c4 < 3 store it)
    // First 16 edges of both directions in TWO dependent round trips: the 16 by-source edge ids (CSC) and the 16
    // by-destination rows (CSR: consecutive edge ids, no index needed) are issued together, then the 16 by-source rows.
    // Every load is unconditional on a clamped index and masked afterwards (a predicated load is an exec-masked branch
    // with its own wait).  Sums are taken in edge order.
    constexpr int GB = 16;
    if (nout > 0 || nin > 0) {       // (a node without any edge reads nothing: dz may be empty)
        const int so = nout > 0 ? s0 : 0, no1 = nout > 0 ? nout - 1 : 0;
        const int di = nin > 0 ? d0 : (nout > 0 ? csc_eid[s0] : 0), ni1 = nin > 0 ? nin - 1 : 0;
        int e[GB];
        f32x4 vq[GB], vp[GB];
        float wq[GB], wp[GB];
#pragma unroll
        for (int i = 0; i < GB; ++i) e[i] = nout > 0 ? csc_eid[so + (i < no1 ? i : no1)] : di;
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            const size_t ee = (size_t)(di + (i < ni1 ? i : ni1));
            vq[i] = gather_dz4<BF>(dz, ee, c4);
            wq[i] = dxrel[ee * 4 + (c4 & 3)];
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            vp[i] = gather_dz4<BF>(dz, (size_t)e[i], c4);
            wp[i] = dxrel[(size_t)e[i] * 4 + (c4 & 3)];
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            if (i < nout) {
                sp += vp[i];
                sx += wp[i];
            }
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            if (i < nin) {
                sq += vq[i];
                sx -= wq[i];
            }
        }
    }
    // degrees beyond 16 (out-degree is unbounded; in-degree <= 32): the remaining edges, 8 at a time
    for (int q0 = s0 + GB; q0 < s1; q0 += 8) {
        int e[8];
        f32x4 v[8];
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = csc_eid[q0 + i < s1 ? q0 + i : s1 - 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = gather_dz4<BF>(dz, (size_t)e[i], c4);
            w[i] = dxrel[(size_t)e[i] * 4 + (c4 & 3)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (q0 + i < s1) {
                sp += v[i];
                sx += w[i];
            }
        }
    }
    for (int e0 = d0 + GB; e0 < d1; e0 += 8) {
        f32x4 v[8];
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t ee = (size_t)(e0 + i < d1 ? e0 + i : d1 - 1);
            v[i] = gather_dz4<BF>(dz, ee, c4);
            w[i] = dxrel[ee * 4 + (c4 & 3)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (e0 + i < d1) {
                sq += v[i];
                sx -= w[i];
            }
        }
    }
    if (live) {
        *(f32x4*)(dP + (size_t)j * 64 + 4 * c4) = sp;
        *(f32x4*)(dQ + (size_t)j * 64 + 4 * c4) = sq;
        if (c4 < 3) dx[(size_t)j * 3 + c4] = a * d_xnew[(size_t)j * 3 + c4] + sx;
    }
}
