// Cross-attention tile helpers and the forward body (included by eqd_attn_kernels.hip and, for the launch that runs the
// forward attention of small batches beside the edge-message forward, by eqd_edge_kernels.hip).
#pragma once
#include "eqd_common.h"

// The backward passes recompute p = exp(S - lse) (arguments <= 0 up to rounding) with the hardware exponential
// (v_exp_f32 on x log2 e: ~4e-6 relative error at |x| ~ 80, far inside the gradient tolerance) - 32 of them per
// 32 x 32 score tile otherwise cost as many VALU cycles as a third of the tile's MFMAs.  The forward keeps expf's bits (exp_nooverflow).
#ifndef EQD_NATIVE_EXP
#define EQD_NATIVE_EXP(x) __expf(x)
#endif
__device__ __forceinline__ float bwd_exp(float x) { return EQD_NATIVE_EXP(x); }

template <int DB>
struct AttnCfg {
    enum { DS = 16 * DB + 4, KS = 4 * DB, NL = 8 * DB, TILE = 32 * (16 * DB + 4), RED = DB * 2 * 4 * 64 };
};

// FAST: d == 16 DB exactly (64, or the 69-wide first layer zero-padded to 80 by the caller), 16-byte aligned rows
template <int DB, bool FAST>
struct TileRegs {
    float4 q[FAST ? 8 : 1];            // FAST: row (lane >> 4) + 4 j, columns 4 (lane & 15) .. (the first 64 columns)
    float4 qx[(FAST && DB > 4) ? 2 : 1];   // FAST, d == 80: row (lane + 64 j) >> 2, columns 64 + 4 ((lane + 64 j) & 3) ..
    f32x4 qv[FAST ? 1 : 2 * DB];       // any d: elements 4 (lane + 64 j) .. + 3 of the contiguous [rows][d] slab (raw)
    int nvalid;
};

// rows r0 .. r0+31 of M (row stride d), clipped at r1 -> registers (zeros beyond)
template <int DB, bool FAST>
__device__ __forceinline__ void tile_load(TileRegs<DB, FAST>& R, const float* __restrict__ M, int d, int r0, int r1,
                                          int lane) {
    int nrows = r1 - r0;
    nrows = nrows < 0 ? 0 : (nrows > 32 ? 32 : nrows);
    const int nvalid = nrows * d;
    const float* __restrict__ base = M + (size_t)r0 * d;
    R.nvalid = nvalid;
    if (FAST) {
        // unpredicated: buffer loads whose descriptor ends behind the range's last row - the hardware returns zeros for the
        // lanes beyond it (a select between the loaded value and 0 comes back from hipcc as an exec-masked load behind a
        // branch with the destination zeroed in front: 8 extra instructions per load); a tile entirely beyond the range
        // is not fetched at all (wave-uniform)
        if (nrows > 0) {
#if defined(EQD_HOSTSIM) || defined(EQD_NO_BUFFER_LOADS) || defined(EQD_NO_BUFFER_LOADS_TILE)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i4 = lane + 64 * j;
                const int row = i4 >> 4;
                const float4 v = ((const float4*)base)[(row < nrows ? row : nrows - 1) * (4 * DB) + (i4 & 15)];
                R.q[j] = row < nrows ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if constexpr (DB > 4) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = lane + 64 * j;
                    const int row = i >> 2;
                    const float4 v = ((const float4*)base)[(row < nrows ? row : nrows - 1) * (4 * DB) + 16 + (i & 3)];
                    R.qx[j] = row < nrows ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#else
            // (one descriptor per tensor and range end - loop-invariant, built once: r1 is the same for the whole workgroup but
            //  arrives in a vector register, hence the readfirstlane; the tile's first row goes into the lanes' offsets.
            //  Byte offsets are 32-bit: tensors of up to 2^32 / (4 d) rows; FAST: d == 16 DB)
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)M, 0, __builtin_amdgcn_readfirstlane(r1) * (64 * DB), 0x00020000);
            const int vo = r0 * (64 * DB);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i4 = lane + 64 * j;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16 * ((i4 >> 4) * (4 * DB) + (i4 & 15)), 0, 0);
                R.q[j] = __builtin_bit_cast(float4, v);
            }
            if constexpr (DB > 4) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = lane + 64 * j;
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16 * ((i >> 2) * (4 * DB) + 16 + (i & 3)), 0, 0);
                    R.qx[j] = __builtin_bit_cast(float4, v);
                }
            }
#endif
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) R.q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DB > 4) R.qx[0] = R.qx[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        // the 32 rows are one contiguous slab of 32 d floats: 16-byte vectors (4-byte alignment is enough), all in
        // flight at once; tail / beyond-the-range handling in ld4u_raw / ld4u_fix
        if (nvalid > 0) {
#pragma unroll
            for (int j = 0; j < 2 * DB; ++j) {
                const int e = 4 * (lane + 64 * j);
                R.qv[j] = ld4u_raw(base + (e < nvalid ? e : 0), nvalid - e, M);
            }
        }
    }
}
template <int DB, bool FAST>
__device__ __forceinline__ void tile_store(const TileRegs<DB, FAST>& R, float* __restrict__ L, int d, int lane) {
    constexpr int DS = AttnCfg<DB>::DS;
    if (FAST) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i4 = lane + 64 * j;
            *(float4*)&L[(i4 >> 4) * DS + 4 * (i4 & 15)] = R.q[j];
        }
        if constexpr (DB > 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = lane + 64 * j;
                *(float4*)&L[(i >> 2) * DS + 64 + 4 * (i & 3)] = R.qx[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2 * DB; ++j) {
            const int e = 4 * (lane + 64 * j);
            const float4 f = R.nvalid > 0 ? ld4u_fix(R.qv[j], R.nvalid - e) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float vv[4] = {f.x, f.y, f.z, f.w};
            int row = e / d, col = e - row * d;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (col >= d) {
                    col -= d;
                    ++row;
                }
                if (row < 32) L[row * DS + col] = vv[u];
                ++col;
            }
        }
    }
}
// cooperative (256 threads) [32][d] tile -> LDS, used once per workgroup for the block's own rows
__device__ __forceinline__ void block_tile_stage(const float* __restrict__ M, int d, int DS, int r0, int r1,
                                                 float* __restrict__ L, int t) {
    int nrows = r1 - r0;
    nrows = nrows > 32 ? 32 : nrows;
    const int nvalid = nrows * d;       // >= d >= 4: a block has at least one row
    const float* __restrict__ base = M + (size_t)r0 * d;
    f32x4 v[3];                         // 3 x 256 vectors >= 32 x 80 floats
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = 4 * (t + 256 * j);
        v[j] = ld4u_raw(base + (e < nvalid ? e : 0), nvalid - e, M);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = 4 * (t + 256 * j);
        const float4 f = ld4u_fix(v[j], nvalid - e);
        const float vv[4] = {f.x, f.y, f.z, f.w};
        int row = e / d, col = e - row * d;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (col >= d) {
                col -= d;
                ++row;
            }
            if (row < 32) L[row * DS + col] = vv[u];
            ++col;
        }
    }
}
// d == 64, 16-byte aligned rows: the [32][64] tile is 512 float4, two per thread, unpredicated (rows beyond the block are
// fetched from its last row and written as zeros).  Columns 64.. of the LDS tile are never read when d == 64.
template <int DB>
__device__ __forceinline__ void block_tile_stage_fast(const float* __restrict__ M, int DS, int r0, int r1,
                                                      float* __restrict__ L, int t) {
    int nrows = r1 - r0;
    nrows = nrows > 32 ? 32 : nrows;
    const float4* __restrict__ base = (const float4*)(M + (size_t)r0 * (16 * DB));
    float4 v[2], vx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i4 = t + 256 * j, row = i4 >> 4;
        v[j] = base[(row < nrows ? row : nrows - 1) * (4 * DB) + (i4 & 15)];
    }
    if constexpr (DB > 4) {      // columns 64 .. 79: 32 rows x 4 vectors, threads 0 .. 127
        const int row = (t & 127) >> 2;
        vx = base[(row < nrows ? row : nrows - 1) * (4 * DB) + 16 + (t & 3)];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i4 = t + 256 * j, row = i4 >> 4;
        *(float4*)&L[row * DS + 4 * (i4 & 15)] = row < nrows ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (DB > 4) {
        const int row = (t & 127) >> 2;
        if (t < 128) *(float4*)&L[row * DS + 64 + 4 * (t & 3)] = row < nrows ? vx : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void zero_fill(float* __restrict__ L, int n, int t, int nthreads) {
    for (int i = t; i < n; i += nthreads) L[i] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// NB = 2: one workgroup per work item (a block of up to 32 rows).  NB = 1: TWO workgroups per item, 16 rows each - for
// small batches, whose 32-row blocks do not give every CU a workgroup (DB5.5: 112 items, 256 CUs): twice the workgroups
// with half the matrix work and the same loads each; the arithmetic of a row is unchanged (its keys are split over the
// waves and merged in the same order).  (The backward uses half blocks for another reason: k_attn_bwd.)
// ---- MFMA steps shared by the forward and backward bodies, fp32 (BF = false) or bf16 inputs (BF = true) ----------------
// A bf16 MFMA (v_mfma_f32_16x16x16_bf16, fp32 accumulate) contracts 16 k per instruction, 4 per lane; which 4 a lane
// supplies does not matter as long as A and B agree, so four consecutive fp32 steps (lane group g supplying k = 4 ks + g,
// ks = 4 c .. 4 c + 3) become ONE bf16 instruction on the same operand loads, packed (v_cvt_pk_bf16_f32).
template <bool BF, int KS>
struct KFrag {                       // a lane's B-operand values for KS k-steps: floats, or bf16 packed four steps at a time
    float v[BF ? 1 : KS];
    s16x4 p[BF ? KS / 4 : 1];
};
// fragment of row `row` of an LDS tile (row stride DS): element ks = T[row * DS + 4 ks + g]
template <bool BF, int KS>
__device__ __forceinline__ void kfrag_load(KFrag<BF, KS>& F, const float* __restrict__ T, int row, int DS, int g) {
    if constexpr (BF) {
#pragma unroll
        for (int c = 0; c < KS / 4; ++c)
            F.p[c] = pack_bf4(T[row * DS + 16 * c + g], T[row * DS + 16 * c + 4 + g], T[row * DS + 16 * c + 8 + g],
                              T[row * DS + 16 * c + 12 + g]);
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) F.v[ks] = T[row * DS + 4 * ks + g];
    }
}
// acc[nb] += sum_ks A[arow * DS + 4 ks + g] * F[nb][ks]   (A: a streamed LDS tile, one row per lane)
template <bool BF, int KS, int NB>
__device__ __forceinline__ void mma_k(f32x4 (&acc)[NB], const float* __restrict__ A, int arow, int DS, int g,
                                      const KFrag<BF, KS> (&F)[NB]) {
    if constexpr (BF) {
        auto chunk = [&](int c) {
            return pack_bf4(A[arow * DS + 16 * c + g], A[arow * DS + 16 * c + 4 + g], A[arow * DS + 16 * c + 8 + g],
                            A[arow * DS + 16 * c + 12 + g]);
        };
#pragma unroll
        for (int cp = 0; cp < KS / 8; ++cp) {      // pairs of 16-deep chunks: v_mfma_f32_16x16x32_bf16
            const s16x8 a = cat_bf(chunk(2 * cp), chunk(2 * cp + 1));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf32(a, cat_bf(F[nb].p[2 * cp], F[nb].p[2 * cp + 1]), acc[nb]);
        }
        if constexpr ((KS / 4) % 2 == 1) {          // the 80-wide first layer: five chunks
            const s16x4 a = chunk(KS / 4 - 1);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf(a, F[nb].p[KS / 4 - 1], acc[nb]);
        }
    } else {
        // (all A fragments first: with the read next to its use hipcc emitted  ds_read2_b32 - s_waitcnt lgkmcnt(0) - 4 MFMA
        //  sixteen times per score block, the full LDS latency in front of every fourth MFMA)
        float a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = A[arow * DS + 4 * ks + g];
#if !defined(EQD_HOSTSIM) && !defined(EQD_NO_LDS_BATCH)
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks every read back next to its MFMAs)
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma4(a[ks], F[nb].v[ks], acc[nb]);
        }
    }
}
// acc[db][nb] += sum_r A[(row0 + r) * DS + 16 db + l15] * B[nb][r]   (B: an F-layout tile, e.g. softmax weights)
// both 16-row halves of a 32-row tile at once (bf16: ONE 32-deep chunk of v_mfma_f32_16x16x32_bf16):
// acc[db][nb] += sum over h, r of A[(16 h + 4 g + r) * DS + 16 db + l15] * B[h][nb][r]
template <bool BF, int DB, int NB>
__device__ __forceinline__ void mma_r2(f32x4 (&acc)[DB][NB], const float* __restrict__ A, int g, int DS, int l15,
                                       const f32x4 (&B)[2][NB]);
template <bool BF, int DB, int NB>
__device__ __forceinline__ void mma_r(f32x4 (&acc)[DB][NB], const float* __restrict__ A, int row0, int DS, int l15,
                                      const f32x4 (&B)[NB]) {
    if constexpr (BF) {
        s16x4 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = pack_bf4(B[nb][0], B[nb][1], B[nb][2], B[nb][3]);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const s16x4 a = pack_bf4(A[row0 * DS + 16 * db + l15], A[(row0 + 1) * DS + 16 * db + l15],
                                     A[(row0 + 2) * DS + 16 * db + l15], A[(row0 + 3) * DS + 16 * db + l15]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[db][nb] = mfma_bf(a, b[nb], acc[db][nb]);
        }
    } else {
        float a[4][DB];      // (all A fragments first, see mma_k)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int db = 0; db < DB; ++db) a[r][db] = A[(row0 + r) * DS + 16 * db + l15];
#if !defined(EQD_HOSTSIM) && !defined(EQD_NO_LDS_BATCH)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[db][nb] = mfma4(a[r][db], B[nb][r], acc[db][nb]);
            }
    }
}
template <bool BF, int DB, int NB>
__device__ __forceinline__ void mma_r2(f32x4 (&acc)[DB][NB], const float* __restrict__ A, int g, int DS, int l15,
                                       const f32x4 (&B)[2][NB]) {
    if constexpr (BF) {
        s16x8 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            b[nb] = cat_bf(pack_bf4(B[0][nb][0], B[0][nb][1], B[0][nb][2], B[0][nb][3]),
                           pack_bf4(B[1][nb][0], B[1][nb][1], B[1][nb][2], B[1][nb][3]));
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const int r0 = 4 * g, r1 = 16 + 4 * g;
            const s16x8 a = cat_bf(pack_bf4(A[r0 * DS + 16 * db + l15], A[(r0 + 1) * DS + 16 * db + l15],
                                            A[(r0 + 2) * DS + 16 * db + l15], A[(r0 + 3) * DS + 16 * db + l15]),
                                   pack_bf4(A[r1 * DS + 16 * db + l15], A[(r1 + 1) * DS + 16 * db + l15],
                                            A[(r1 + 2) * DS + 16 * db + l15], A[(r1 + 3) * DS + 16 * db + l15]));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[db][nb] = mfma_bf32(a, b[nb], acc[db][nb]);
        }
    } else {
        mma_r<false, DB, NB>(acc, A, 4 * g, DS, l15, B[0]);
        mma_r<false, DB, NB>(acc, A, 16 + 4 * g, DS, l15, B[1]);
    }
}

// LDS of one 4-wave attention-forward group.  The merge buffer aliases the waves' own K tiles (C::RED <= C::TILE; written
// only after the wave's last tile): 79 KB instead of 111 KB, i.e. two groups per CU, so that one wave's softmax overlaps
// another's MFMAs.
template <int DB>
struct alignas(16) AttnFwdSmem {
    typedef AttnCfg<DB> C;
    float Qt[C::TILE];
    float Kt[EQD_WAVES][C::TILE];
    float Vt[EQD_WAVES][C::TILE];
    float sm_m[EQD_WAVES][32], sm_l[EQD_WAVES][32];
    static_assert(C::RED <= C::TILE, "merge buffer must fit a K tile");
};

// Half-block launches (two workgroups per item): workgroup b runs on XCD b % 8 and the work list is laid out so that item
// i belongs to XCD class i % 8 (graph.py: _xcd_interleave; n_att_items is a multiple of 8).  Both halves of an item must
// therefore sit at workgroup ids with the item's residue: b = 16 (i / 8) + 8 half + i % 8.
__device__ __forceinline__ int att_half_item(int b) { return ((b >> 4) << 3) + (b & 7); }
__device__ __forceinline__ int att_half_of(int b) { return (b >> 3) & 1; }

// The forward of one work item (NB = 2) or of one 16-row half of it (NB = 1, half = 0 | 1) by a group of 4 waves;
// t = the thread's index within the group (0..255).  `group_barriers`: the group shares its workgroup with other groups
// that execute the same number of __syncthreads() (k_edge_attn_fwd: the two halves of one item) - an empty half then
// still has to pass them.
template <int DB, bool FAST, int NB, bool BF = false>
__device__ __forceinline__ void attn_fwd_body(AttnFwdSmem<DB>& sm, const EqdGraph& G, int item, int half, int t, int d,
                                              const float* __restrict__ q, const float* __restrict__ k,
                                              const float* __restrict__ v, float* __restrict__ out,
                                              float* __restrict__ lse, bool group_barriers = false) {
    typedef AttnCfg<DB> C;
    constexpr int DS = C::DS, KS = C::KS;
    float* Qt = sm.Qt;
    float (*Kt)[C::TILE] = sm.Kt;
    float (*Vt)[C::TILE] = sm.Vt;
    float (*sm_m)[32] = sm.sm_m;
    float (*sm_l)[32] = sm.sm_l;
    float (*red)[C::TILE] = Kt;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) {   // at most 16 rows in the item's block, or a padding item of the XCD-interleaved list (uniform)
        if (group_barriers) {
            static_assert(!(NB == 1) || FAST, "half blocks exist on the float4 path only");
            __syncthreads();
            __syncthreads();
        }
        return;
    }
    int rowq[NB];
    bool qv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowq[nb] = b0 + 16 * nb + l15;
        qv[nb] = rowq[nb] < b1;
    }

    EQD_TR_WG();
    EQD_TR(0);
    TileRegs<DB, FAST> rk, rv;
    int kt = o0 + 32 * wave;
    tile_load<DB, FAST>(rk, k, d, kt, o1, lane);
    tile_load<DB, FAST>(rv, v, d, kt, o1, lane);
    if (FAST) {      // every element that is read later is written by the tile stores: no zero fill needed
        EQD_TR(1);
        block_tile_stage_fast<DB>(q, DS, b0, b1, Qt, t);
    } else {
        zero_fill(Qt, C::TILE, t, EQD_BLOCK);
        zero_fill(Kt[wave], C::TILE, lane, 64);
        zero_fill(Vt[wave], C::TILE, lane, 64);
        __syncthreads();
        EQD_TR(1);
        block_tile_stage(q, d, DS, b0, b1, Qt, t);
    }
    __syncthreads();
    EQD_TR(2);
    KFrag<BF, KS> qf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) kfrag_load<BF, KS>(qf[nb], Qt, 16 * nb + l15, DS, g);

    f32x4 O[DB][NB];
    float mrun[NB], lrun[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int db = 0; db < DB; ++db) O[db][nb] = f4zero();
        mrun[nb] = EQD_NEG_BIG;
        lrun[nb] = 0.f;
    }
    const float* __restrict__ Kw = Kt[wave];
    const float* __restrict__ Vw = Vt[wave];
    EQD_TR(3);
    for (; kt < o1; kt += 32 * EQD_WAVES) {
        wave_lds_fence();
        tile_store<DB, FAST>(rk, Kt[wave], d, lane);
        tile_store<DB, FAST>(rv, Vt[wave], d, lane);
        wave_lds_fence();
        EQD_TR(4);
        tile_load<DB, FAST>(rk, k, d, kt + 32 * EQD_WAVES, o1, lane);   // prefetch the wave's next tile
        tile_load<DB, FAST>(rv, v, d, kt + 32 * EQD_WAVES, o1, lane);
        EQD_TR(5);
        f32x4 S[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) S[mb][nb] = f4zero();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) mma_k<BF, KS, NB>(S[mb], Kw, 16 * mb + l15, DS, g, qf);
        EQD_TR(6);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float mx = EQD_NEG_BIG;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    // bf16 mode: logits in log2 units (the running maximum is then kept as an INTEGER, see below)
                    const float sv = BF ? S[mb][nb][r] * 1.44269504088896341f : S[mb][nb][r];
                    const float s = key < o1 ? sv : EQD_NEG_BIG;
                    S[mb][nb][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = group_max(mx);
            // bf16 mode: the un-normalised weights p = 2^(s - M) are rounded to bf16 for the P V product; with M an
            // integer every tile / wave / merge step rescales by an exact power of two, so WHICH bf16 value a weight
            // rounds to does not depend on the tile schedule (and the oracle can state the rounding point:
            // 2^(s log2 e - ceil(max)) rounded to bf16)
            const float mnew = fmaxf(mrun[nb], BF ? ceilf(mx) : mx);
            const float alpha = BF ? exp2_flush(mrun[nb] - mnew) : exp_nooverflow(mrun[nb] - mnew);
            float ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float p = key < o1 ? (BF ? exp2_flush(S[mb][nb][r] - mnew) : exp_nooverflow(S[mb][nb][r] - mnew)) : 0.f;
                    S[mb][nb][r] = p;
                    ps += p;
                }
            lrun[nb] = lrun[nb] * alpha + group_sum(ps);
            mrun[nb] = mnew;
#pragma unroll
            for (int db = 0; db < DB; ++db) O[db][nb] *= alpha;
        }
        EQD_TR(7);
        mma_r2<BF, DB, NB>(O, Vw, g, DS, l15, S);
        EQD_TR(8);
    }
    EQD_TR(9);
    // ---- merge the 4 waves' partial softmax states -----------------------------------------------
    wave_lds_fence();       // this wave's reads of its K tile are done before it is overwritten
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][((db * 2 + nb) * 4 + r) * 64 + lane] = O[db][nb][r];
    if (g == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            sm_m[wave][16 * nb + l15] = mrun[nb];
            sm_l[wave][16 * nb + l15] = lrun[nb];
        }
    }
    __syncthreads();
    float sc[NB][EQD_WAVES], inv[NB], mtot[NB], ltot[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float mm = EQD_NEG_BIG;
#pragma unroll
        for (int w = 0; w < EQD_WAVES; ++w) mm = fmaxf(mm, sm_m[w][16 * nb + l15]);
        float ll = 0.f;
#pragma unroll
        for (int w = 0; w < EQD_WAVES; ++w) {
            sc[nb][w] = BF ? exp2_flush(sm_m[w][16 * nb + l15] - mm) : exp_nooverflow(sm_m[w][16 * nb + l15] - mm);
            ll += sm_l[w][16 * nb + l15] * sc[nb][w];
        }
        mtot[nb] = mm;
        ltot[nb] = ll;
        inv[nb] = ll > 0.f ? 1.f / ll : 0.f;
    }
    for (int db = wave; db < DB; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = 0.f;
#pragma unroll
                for (int w = 0; w < EQD_WAVES; ++w) o += red[w][((db * 2 + nb) * 4 + r) * 64 + lane] * sc[nb][w];
                const int f = 16 * db + 4 * g + r;
                if (f < d) out[(size_t)rowq[nb] * d + f] = o * inv[nb];
            }
        }
    if (wave == 0 && g == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            if (qv[nb])
                lse[rowq[nb]] = ltot[nb] > 0.f ? (BF ? mtot[nb] * 0.693147180559945309f : mtot[nb]) + logf(ltot[nb]) : 0.f;
    }
    EQD_TR(10);
    EQD_TR_WG_END();
}

template <int DB, bool FAST, int NB, bool BF = false>
__global__ __launch_bounds__(EQD_BLOCK) void k_attn_fwd(EqdGraph G, int d, const float* __restrict__ q,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        float* __restrict__ out, float* __restrict__ lse) {
    __shared__ AttnFwdSmem<DB> sm;
    const int item = NB == 1 ? att_half_item((int)blockIdx.x) : (int)blockIdx.x;
    attn_fwd_body<DB, FAST, NB, BF>(sm, G, item, att_half_of((int)blockIdx.x), (int)threadIdx.x, d, q, k, v, out, lse);
}
