// bf16 mode of the cross-attention kernels with the streamed tiles held in LDS AS bf16 ("lb": LDS bf16), d = 64.
//
// The first bf16 version (eqd_attn_fwd_inl.h / eqd_attn_kernels.hip with BF = true) kept the fp32 tiles in LDS and
// rounded to bf16 when an MFMA operand was formed: every v_mfma_f32_16x16x16_bf16 then cost four ds_read_b32 and two
// v_cvt_pk_bf16_f32 for its A operand, and the counters said what that does - MfmaUtil 6 % at 64 x (300, 300), the LDS pipe
// (128 B / clock per CU, shared by the CU's 8 waves) busy ~6x longer than the MFMA pipes.  Here a streamed tile is rounded
// ONCE, when the wave parks it in LDS, in the layout(s) its contractions read:
//   row-major  [32 rows][64]  (row stride LB_RS): A operand of a contraction over FEATURES (S = K Q^T, dP = V dO^T):
//                              lane (l15, g) reads features 16 c + 4 g .. + 3 of row l15 as ONE ds_read_b64;
//   transposed [64][32 rows]  (row stride LB_TS): A operand of a contraction over the tile's ROWS (O += V^T P, dQ += K^T dS,
//                              dV += dO^T P, dK += Q^T dS): rows 4 g .. 4 g + 3 of feature l15, ONE ds_read_b64.
// The global loads give a lane four CONSECUTIVE rows x eight columns (instead of eight rows x four columns), so that both
// layouts are written with vector stores (8 x b64 row-major, 8 x b64 transposed).  Same rounding points as before (each
// element of K / V / Q / dO rounded to nearest-even bf16, softmax weights rounded relative to an integer maximum, fp32
// accumulation); the lanes' k-slot assignment inside an MFMA differs, i.e. fp32 summation order only.
// Half the LDS bytes per tile: 46 KB (forward) / 58-75 KB (backward) per workgroup.
#pragma once
#include "eqd_attn_fwd_inl.h"

#define LB_RS 72   /* bf16 row stride of a row-major tile: 36 dwords = 4 mod 32 (as the staged bf16 weights, WSB) */
#define LB_TS 36   /* bf16 row stride of a transposed tile: 18 dwords - the 16 rows of a b64 fragment read hit 16 distinct
                      bank pairs (18 i mod 32 runs over all even banks), 72 B rows keep the b64 accesses 8-byte aligned */
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct LbRegs {
    float4 lo[4], hi[4];      // rows 4 a + r (a = lane >> 3); lo: columns 4 cg .. + 3, hi: columns 32 + 4 cg .. + 3 (cg = lane & 7)
    int nrows;                // rows of the tile inside the range: rows beyond hold a copy of the last valid row (RAW loads)
};
// (The lane's two 16-byte pieces of a row are 128 B apart, not adjacent: one load instruction then covers 128 contiguous
// bytes per row for the 8 lanes of a row group, and in the transposed tile the lanes of a half wave write features
// 4 cg + j, whose rows lie 8 banks apart (72 cg dwords): the minimal two lanes per bank pair.)

// rows r0 .. r0 + 31 of M ([.][64] fp32, 16-byte aligned rows), clipped at r1 -> registers, RAW: rows beyond the range are
// fetched from its last row and zeroed when the tile is stored (a select next to the load turns it into an exec-masked
// branch with a wait behind it - and, here, parked half-loaded vectors in scratch memory: the kernel ran 1.4x slower)
// On the GPU the loads are BUFFER loads whose descriptor ends behind the range's last row: the hardware returns zeros for
// the rows beyond it (LB_ZERO_FILLED), no address clamp per row and no zeroing selects when the tile is stored (those were
// 64 v_cndmask per tile); the descriptor is wave-uniform (r1, the end of the work item's range, is the same for the whole
// workgroup but arrives in a vector register, hence the readfirstlane).  Byte offsets are 32-bit signed: tensors of up to 2^23 rows (the launchers refuse more than EQD_ATT_MAX_ROWS, eqd_attn_kernels.hip).
// The host simulator keeps the clamped plain loads.
#if defined(EQD_HOSTSIM) || defined(EQD_NO_BUFFER_LOADS)
#define LB_ZERO_FILLED 0
#else
#define LB_ZERO_FILLED 1
#endif
__device__ __forceinline__ void lb_load(LbRegs& R, const float* __restrict__ M, int r0, int r1, int lane) {
    int nrows = r1 - r0;
    nrows = nrows < 0 ? 0 : (nrows > 32 ? 32 : nrows);
    R.nrows = nrows;
    const int a = lane >> 3, cg = lane & 7;
#if LB_ZERO_FILLED && !defined(EQD_NO_BUFFER_LOADS_LB)
    // (one descriptor per tensor and range end - loop-invariant, built once; the tile's first row goes into the lanes' offsets)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)M, 0, __builtin_amdgcn_readfirstlane(r1) * 256, 0x00020000);
    const int vo = (r0 + 4 * a) * 256 + cg * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 l = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 256 * r, 0, 0));
        const f32x4 h = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 256 * r + 128, 0, 0));
        R.lo[r] = make_float4(l[0], l[1], l[2], l[3]);
        R.hi[r] = make_float4(h[0], h[1], h[2], h[3]);
    }
#else
    if (nrows > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * a + r;
            const float4* __restrict__ p = (const float4*)(M + (size_t)(r0 + (row < nrows ? row : nrows - 1)) * 64 + 4 * cg);
            R.lo[r] = p[0];
            R.hi[r] = p[8];
            if (LB_ZERO_FILLED && row >= nrows) R.lo[r] = R.hi[r] = make_float4(0.f, 0.f, 0.f, 0.f);      // (bisecting builds only)
        }
    } else if (LB_ZERO_FILLED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) R.lo[r] = R.hi[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#endif
}
// The same tile from a SAVED bf16 tensor ([.][64] bf16 rows, 128 B: the bf16 storage mode keeps q / k / v of the 64-wide layers
// that way for the backward): 8-byte loads, exact conversion - rounding the values again when the tile is parked in LDS
// gives back the same bits, so the kernels' results do not depend on which form they were handed.
__device__ __forceinline__ float4 lb_cvt4(s16x4 h) {
    return make_float4(bf2f((unsigned short)h[0]), bf2f((unsigned short)h[1]), bf2f((unsigned short)h[2]), bf2f((unsigned short)h[3]));
}
// RAW like lb_load: the 8 bytes of a piece are parked in .x / .y of its float4 as they come; lb_fix_bf converts them in
// place when the tile is consumed (a conversion next to the load would make the loads of the NEXT tile - issued a whole
// iteration ahead - wait at once)
__device__ __forceinline__ float4 lb_raw8(const unsigned short* __restrict__ p) {
    const unsigned long long h = *(const unsigned long long*)p;      // (one 8-byte load)
    return make_float4(__builtin_bit_cast(float, (unsigned)h), __builtin_bit_cast(float, (unsigned)(h >> 32)), 0.f, 0.f);
}
__device__ __forceinline__ float4 lb_fix8(float4 raw) {
    const unsigned lo = __builtin_bit_cast(unsigned, raw.x), hi = __builtin_bit_cast(unsigned, raw.y);
    return make_float4(__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                       __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u));
}
__device__ __forceinline__ void lb_load_bf(LbRegs& R, const unsigned short* __restrict__ M, int r0, int r1, int lane) {
    int nrows = r1 - r0;
    nrows = nrows < 0 ? 0 : (nrows > 32 ? 32 : nrows);
    R.nrows = nrows;
    const int a = lane >> 3, cg = lane & 7;
#if LB_ZERO_FILLED && !defined(EQD_NO_BUFFER_LOADS_LBBF)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)M, 0, __builtin_amdgcn_readfirstlane(r1) * 128, 0x00020000);
    const int vo = (r0 + 4 * a) * 128 + cg * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const eqd_u32x2 l = __builtin_amdgcn_raw_buffer_load_b64(rs, vo + 128 * r, 0, 0);
        const eqd_u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs, vo + 128 * r + 64, 0, 0);
        // (scalar temporaries: __builtin_bit_cast applied to an element of an ext-vector returns element 0 with this hipcc)
        const unsigned l0 = l.x, l1 = l.y, h0 = h.x, h1 = h.y;
        R.lo[r] = make_float4(__builtin_bit_cast(float, l0), __builtin_bit_cast(float, l1), 0.f, 0.f);
        R.hi[r] = make_float4(__builtin_bit_cast(float, h0), __builtin_bit_cast(float, h1), 0.f, 0.f);
    }
#else
    if (nrows > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * a + r;
            const unsigned short* __restrict__ p = M + (size_t)(r0 + (row < nrows ? row : nrows - 1)) * 64 + 4 * cg;
            R.lo[r] = lb_raw8(p);
            R.hi[r] = lb_raw8(p + 32);
            if (LB_ZERO_FILLED && row >= nrows) R.lo[r] = R.hi[r] = make_float4(0.f, 0.f, 0.f, 0.f);      // (bisecting builds only)
        }
    } else if (LB_ZERO_FILLED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) R.lo[r] = R.hi[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#endif
}
template <bool QB>
__device__ __forceinline__ void lb_load_any(LbRegs& R, const float* __restrict__ M, int r0, int r1, int lane) {
    if constexpr (QB) lb_load_bf(R, (const unsigned short*)M, r0, r1, lane);
    else lb_load(R, M, r0, r1, lane);
}
template <bool QB>
__device__ __forceinline__ void lb_fix_any(LbRegs& R) {      // before the first use of a tile loaded by lb_load_any<true>
    if constexpr (QB) {
        if (LB_ZERO_FILLED || R.nrows > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                R.lo[r] = lb_fix8(R.lo[r]);
                R.hi[r] = lb_fix8(R.hi[r]);
            }
        }
    }
}
// [32][64] block tile of a saved bf16 tensor -> the fp32 LDS tile block_tile_stage_fast<4> makes (rows beyond the block: zeros)
__device__ __forceinline__ void block_tile_stage_bf(const unsigned short* __restrict__ M, int DS, int r0, int r1,
                                                    float* __restrict__ L, int t) {
    int nrows = r1 - r0;
    nrows = nrows > 32 ? 32 : nrows;
    const int row = t >> 3, c8 = t & 7;
    typedef short s16x8v __attribute__((ext_vector_type(8)));
    const s16x8v h = *(const s16x8v*)(M + (size_t)(r0 + (row < nrows ? row : nrows - 1)) * 64 + 8 * c8);
    const bool ok = row < nrows;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (ok) {
        lo = make_float4(bf2f((unsigned short)h[0]), bf2f((unsigned short)h[1]), bf2f((unsigned short)h[2]), bf2f((unsigned short)h[3]));
        hi = make_float4(bf2f((unsigned short)h[4]), bf2f((unsigned short)h[5]), bf2f((unsigned short)h[6]), bf2f((unsigned short)h[7]));
    }
    *(float4*)&L[row * DS + 8 * c8] = lo;
    *(float4*)&L[row * DS + 8 * c8 + 4] = hi;
}
__device__ __forceinline__ s16x4 lb_pack(float a, float b, float c, float d, bool ok) {
    return pack_bf4(ok ? a : 0.f, ok ? b : 0.f, ok ? c : 0.f, ok ? d : 0.f);
}
__device__ __forceinline__ void lb_store_rm(const LbRegs& R, unsigned short* __restrict__ rm, int lane) {
    const int a = lane >> 3, cg = lane & 7;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = LB_ZERO_FILLED || 4 * a + r < R.nrows;
        *(s16x4*)&rm[(4 * a + r) * LB_RS + 4 * cg] = lb_pack(R.lo[r].x, R.lo[r].y, R.lo[r].z, R.lo[r].w, ok);
        *(s16x4*)&rm[(4 * a + r) * LB_RS + 32 + 4 * cg] = lb_pack(R.hi[r].x, R.hi[r].y, R.hi[r].z, R.hi[r].w, ok);
    }
}
__device__ __forceinline__ s16x4 lb_pack4(float a, float b, float c, float d, int row0, int nrows) {
#if LB_ZERO_FILLED
    // (the empty asm keeps the four values scalar: left alone, hipcc vectorizes this 4 x 4 transposition through a stack
    //  object - 96 .. 144 B of scratch plus 5 KB of promoted LDS per workgroup, and the forward ran 12 % slower)
    asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return pack_bf4(a, b, c, d);
#endif
    return pack_bf4(row0 < nrows ? a : 0.f, row0 + 1 < nrows ? b : 0.f, row0 + 2 < nrows ? c : 0.f, row0 + 3 < nrows ? d : 0.f);
}
__device__ __forceinline__ void lb_store_tr(const LbRegs& R, unsigned short* __restrict__ tr, int lane) {
    const int a = lane >> 3, cg = lane & 7, n = R.nrows;
    unsigned short* const t0 = tr + (4 * cg) * LB_TS + 4 * a;
    unsigned short* const t1 = tr + (32 + 4 * cg) * LB_TS + 4 * a;
    *(s16x4*)&t0[0 * LB_TS] = lb_pack4(R.lo[0].x, R.lo[1].x, R.lo[2].x, R.lo[3].x, 4 * a, n);
    *(s16x4*)&t0[1 * LB_TS] = lb_pack4(R.lo[0].y, R.lo[1].y, R.lo[2].y, R.lo[3].y, 4 * a, n);
    *(s16x4*)&t0[2 * LB_TS] = lb_pack4(R.lo[0].z, R.lo[1].z, R.lo[2].z, R.lo[3].z, 4 * a, n);
    *(s16x4*)&t0[3 * LB_TS] = lb_pack4(R.lo[0].w, R.lo[1].w, R.lo[2].w, R.lo[3].w, 4 * a, n);
    *(s16x4*)&t1[0 * LB_TS] = lb_pack4(R.hi[0].x, R.hi[1].x, R.hi[2].x, R.hi[3].x, 4 * a, n);
    *(s16x4*)&t1[1 * LB_TS] = lb_pack4(R.hi[0].y, R.hi[1].y, R.hi[2].y, R.hi[3].y, 4 * a, n);
    *(s16x4*)&t1[2 * LB_TS] = lb_pack4(R.hi[0].z, R.hi[1].z, R.hi[2].z, R.hi[3].z, 4 * a, n);
    *(s16x4*)&t1[3 * LB_TS] = lb_pack4(R.hi[0].w, R.hi[1].w, R.hi[2].w, R.hi[3].w, 4 * a, n);
}
// sum over the lane's 8 columns of row r of the element-wise product of two tiles held in the same layout (0 beyond the range)
__device__ __forceinline__ float lb_rowdot(const LbRegs& A, const LbRegs& B, int r, int lane) {
    const float s = (A.lo[r].x * B.lo[r].x + A.lo[r].y * B.lo[r].y + A.lo[r].z * B.lo[r].z + A.lo[r].w * B.lo[r].w) +
                    (A.hi[r].x * B.hi[r].x + A.hi[r].y * B.hi[r].y + A.hi[r].z * B.hi[r].z + A.hi[r].w * B.hi[r].w);
    return (LB_ZERO_FILLED || 4 * (lane >> 3) + r < A.nrows) ? s : 0.f;
}
// B-operand fragments of row `row` of an fp32 block tile (row stride DS): chunk c = features 16 c + 4 g .. + 3
__device__ __forceinline__ void lb_frag(s16x4 (&F)[4], const float* __restrict__ T, int row, int DS, int g) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 v = *(const f32x4*)&T[row * DS + 16 * c + 4 * g];
        F[c] = pack_bf4(v[0], v[1], v[2], v[3]);
    }
}
// acc[nb] += sum_f A[arow][f] * F[nb][f]   (A: a row-major bf16 tile)
template <int NB>
__device__ __forceinline__ void lb_mma_k(f32x4 (&acc)[NB], const unsigned short* __restrict__ rm, int arow, int g,
                                         const s16x4 (&F)[NB][4]) {
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {      // 64 features = two 32-deep chunks of v_mfma_f32_16x16x32_bf16
        const s16x8 a = cat_bf(*(const s16x4*)&rm[arow * LB_RS + 32 * cp + 4 * g], *(const s16x4*)&rm[arow * LB_RS + 32 * cp + 16 + 4 * g]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf32(a, cat_bf(F[nb][2 * cp], F[nb][2 * cp + 1]), acc[nb]);
    }
}
// acc[db][nb] += sum_r A[row0 + r][16 db + l15] * B[nb][r]   (A: the transposed bf16 copy of the tile; row0 = 16 mb + 4 g)
// (the tile's 32 rows are ONE 32-deep chunk of v_mfma_f32_16x16x32_bf16: B[h] = the F-layout block of rows 16 h + 4 g .. + 3)
template <int NB>
__device__ __forceinline__ void lb_mma_r(f32x4 (&acc)[4][NB], const unsigned short* __restrict__ tr, int g, int l15,
                                         const f32x4 (&B)[2][NB]) {
    s16x8 b[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        b[nb] = cat_bf(pack_bf4(B[0][nb][0], B[0][nb][1], B[0][nb][2], B[0][nb][3]),
                       pack_bf4(B[1][nb][0], B[1][nb][1], B[1][nb][2], B[1][nb][3]));
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const s16x8 a = cat_bf(*(const s16x4*)&tr[(16 * db + l15) * LB_TS + 4 * g], *(const s16x4*)&tr[(16 * db + l15) * LB_TS + 16 + 4 * g]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[db][nb] = mfma_bf32(a, b[nb], acc[db][nb]);
    }
}

// the same contraction with the A operand read out of the ROW-MAJOR copy of the tile by the transposing LDS read
// (lds_tr16, eqd_common.h): lane (l15, g) points at features 16 db + 4 (l15 & 3) .. + 3 of row 4 g + (l15 >> 2) and receives
// rows 4 g .. 4 g + 3 of feature 16 db + l15 - what lb_mma_r reads from a transposed copy that then need not be written
// (8 packs + 8 ds_write_b64 per tile and operand).  Same operand values, same instruction.
template <int NB>
__device__ __forceinline__ void lb_mma_rt(f32x4 (&acc)[4][NB], const unsigned short* __restrict__ rm, int g, int l15,
                                          const f32x4 (&B)[2][NB]) {
    s16x8 b[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        b[nb] = cat_bf(pack_bf4(B[0][nb][0], B[0][nb][1], B[0][nb][2], B[0][nb][3]),
                       pack_bf4(B[1][nb][0], B[1][nb][1], B[1][nb][2], B[1][nb][3]));
    const unsigned short* const p0 = rm + (4 * g + (l15 >> 2)) * LB_RS + 4 * (l15 & 3);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const s16x8 a = cat_bf(lds_tr16(p0 + 16 * db), lds_tr16(p0 + 16 * LB_RS + 16 * db));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[db][nb] = mfma_bf32(a, b[nb], acc[db][nb]);
    }
}

// ---------------------------------------------------------------------------------------------
// forward (same schedule as attn_fwd_body: the 4 waves split the partner's 32-row tiles, online softmax, LDS merge)
// ---------------------------------------------------------------------------------------------
struct alignas(16) LbFwdWave {
    unsigned short krm[32 * LB_RS];     // K tile, row-major
    unsigned short vtr[64 * LB_TS];     // V tile, transposed
};
struct alignas(16) AttnFwdSmemLb {
    float Qt[AttnCfg<4>::TILE];
    LbFwdWave w[EQD_WAVES];
    float sm_m[EQD_WAVES][32], sm_l[EQD_WAVES][32];
    static_assert(sizeof(LbFwdWave) >= AttnCfg<4>::RED * sizeof(float), "merge buffer must fit a wave's tiles");
};

// QB: q, k, v point at bf16 rows (the saved form of the bf16 storage mode: the projections then write no fp32 q / k / v at all)
template <int NB, bool QB = false>
__device__ __forceinline__ void attn_fwd_body_lb(AttnFwdSmemLb& sm, const EqdGraph& G, int item, int half, int t,
                                                 const float* __restrict__ q, const float* __restrict__ k,
                                                 const float* __restrict__ v, float* __restrict__ out,
                                                 float* __restrict__ lse) {
    typedef AttnCfg<4> C;
    constexpr int DS = C::DS, d = 64;
    const int lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;
    int rowq[NB];
    bool qv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowq[nb] = b0 + 16 * nb + l15;
        qv[nb] = rowq[nb] < b1;
    }
    LbRegs rk, rv;
    int kt = o0 + 32 * wave;
    lb_load_any<QB>(rk, k, kt, o1, lane);
    lb_load_any<QB>(rv, v, kt, o1, lane);
    if constexpr (QB) block_tile_stage_bf((const unsigned short*)q, DS, b0, b1, sm.Qt, t);
    else block_tile_stage_fast<4>(q, DS, b0, b1, sm.Qt, t);
    __syncthreads();
    s16x4 qf[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) lb_frag(qf[nb], sm.Qt, 16 * nb + l15, DS, g);
    f32x4 O[4][NB];
    float mrun[NB], lrun[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int db = 0; db < 4; ++db) O[db][nb] = f4zero();
        mrun[nb] = EQD_NEG_BIG;
        lrun[nb] = 0.f;
    }
    unsigned short* const Kw = sm.w[wave].krm;
    unsigned short* const Vw = sm.w[wave].vtr;
    for (; kt < o1; kt += 32 * EQD_WAVES) {
        wave_lds_fence();
        lb_fix_any<QB>(rk);
        lb_fix_any<QB>(rv);
        lb_store_rm(rk, Kw, lane);
        lb_store_tr(rv, Vw, lane);
        wave_lds_fence();
        lb_load_any<QB>(rk, k, kt + 32 * EQD_WAVES, o1, lane);      // prefetch the wave's next tile
        lb_load_any<QB>(rv, v, kt + 32 * EQD_WAVES, o1, lane);
        f32x4 S[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) S[mb][nb] = f4zero();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) lb_mma_k<NB>(S[mb], Kw, 16 * mb + l15, g, qf);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float mx = EQD_NEG_BIG;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float s = key < o1 ? S[mb][nb][r] * 1.44269504088896341f : EQD_NEG_BIG;      // log2 units
                    S[mb][nb][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = group_max(mx);
            const float mnew = fmaxf(mrun[nb], ceilf(mx));       // integer running maximum (see attn_fwd_body)
            const float alpha = exp2_flush(mrun[nb] - mnew);
            float ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float p = key < o1 ? exp2_flush(S[mb][nb][r] - mnew) : 0.f;
                    S[mb][nb][r] = p;
                    ps += p;
                }
            lrun[nb] = lrun[nb] * alpha + group_sum(ps);
            mrun[nb] = mnew;
#pragma unroll
            for (int db = 0; db < 4; ++db) O[db][nb] *= alpha;
        }
        lb_mma_r<NB>(O, Vw, g, l15, S);
    }
    // ---- merge the 4 waves' partial softmax states (as attn_fwd_body) ------------------------------------------------
    wave_lds_fence();
    float* const redw = (float*)&sm.w[wave];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) redw[((db * 2 + nb) * 4 + r) * 64 + lane] = O[db][nb][r];
    if (g == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            sm.sm_m[wave][16 * nb + l15] = mrun[nb];
            sm.sm_l[wave][16 * nb + l15] = lrun[nb];
        }
    }
    __syncthreads();
    float sc[NB][EQD_WAVES], inv[NB], mtot[NB], ltot[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float mm = EQD_NEG_BIG;
#pragma unroll
        for (int w = 0; w < EQD_WAVES; ++w) mm = fmaxf(mm, sm.sm_m[w][16 * nb + l15]);
        float ll = 0.f;
#pragma unroll
        for (int w = 0; w < EQD_WAVES; ++w) {
            sc[nb][w] = exp2_flush(sm.sm_m[w][16 * nb + l15] - mm);
            ll += sm.sm_l[w][16 * nb + l15] * sc[nb][w];
        }
        mtot[nb] = mm;
        ltot[nb] = ll;
        inv[nb] = ll > 0.f ? 1.f / ll : 0.f;
    }
    for (int db = wave; db < 4; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = 0.f;
#pragma unroll
                for (int w = 0; w < EQD_WAVES; ++w) o += ((const float*)&sm.w[w])[((db * 2 + nb) * 4 + r) * 64 + lane] * sc[nb][w];
                out[(size_t)rowq[nb] * d + 16 * db + 4 * g + r] = o * inv[nb];
            }
        }
    if (wave == 0 && g == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            if (qv[nb]) lse[rowq[nb]] = ltot[nb] > 0.f ? mtot[nb] * 0.693147180559945309f + logf(ltot[nb]) : 0.f;
    }
}

// (launch bound 3: the kernel then fits 153 - 164 registers without a spill - it took 184 - 196 unasked - and three workgroups
//  share a CU's 160 KB of LDS instead of two: round 6, profiles/r06_y_occupancy_ab.txt)
template <int NB, bool QB = false>
__global__ __launch_bounds__(EQD_BLOCK, 3) void k_attn_fwd_lb(EqdGraph G, const float* __restrict__ q,
                                                           const float* __restrict__ k, const float* __restrict__ v,
                                                           float* __restrict__ out, float* __restrict__ lse) {
    __shared__ AttnFwdSmemLb sm;
    const int item = NB == 1 ? att_half_item((int)blockIdx.x) : (int)blockIdx.x;
    attn_fwd_body_lb<NB, QB>(sm, G, item, att_half_of((int)blockIdx.x), (int)threadIdx.x, q, k, v, out, lse);
}

// ---------------------------------------------------------------------------------------------
// backward: workgroups [0, n) of the launch run the dq pass, [n, 2 n) the dk / dv pass (as k_attn_bwd)
// ---------------------------------------------------------------------------------------------
struct alignas(16) LbBwdWave {
    unsigned short a_rm[32 * LB_RS];    // dq pass: K row-major   | kv pass: Q row-major
    unsigned short b_rm[32 * LB_RS];    //          V row-major   |          dO row-major
    // (until round 3 also transposed copies of K | Q and dO for the contractions over the tile's rows: they are read out of
    //  the row-major copies by the transposing LDS read now, lb_mma_rt)
};
struct alignas(16) AttnBwdSmemLb {
    LbBwdWave w[EQD_WAVES];
    float dls[EQD_WAVES][32];
    // the block's two fp32 tiles (dead once their fragments sit in registers) live in the streamed tiles of waves 2 and 3,
    // each wave's part of the merge buffer in its own tiles (written after its last tile)
    __device__ __forceinline__ float* blk(int i) { return (float*)&w[2 + i]; }
    __device__ __forceinline__ float* red(int wv) { return (float*)&w[wv]; }
};
static_assert(sizeof(LbBwdWave) >= AttnCfg<4>::TILE * sizeof(float), "a block tile must fit a wave's streamed tiles");
static_assert(sizeof(LbBwdWave) >= AttnCfg<4>::RED * sizeof(float), "the merge buffer must fit a wave's streamed tiles");

template <int NB>
__device__ __forceinline__ void attn_bwd_q_body_lb(AttnBwdSmemLb& sm, const EqdGraph& G, int item,
                                                   const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, const float* __restrict__ out,
                                                   const float* __restrict__ lse, const float* __restrict__ d_out,
                                                   float* __restrict__ dq, float* __restrict__ delta, int half,
                                                   float qk_slope) {
    typedef AttnCfg<4> C;
    constexpr int DS = C::DS, d = 64;
    float* Qt = sm.blk(0);
    float* Gt = sm.blk(1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;
    int rowq[NB];
    bool qv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowq[nb] = b0 + 16 * nb + l15;
        qv[nb] = rowq[nb] < b1;
    }
    LbRegs rk, rv;
    int kt = o0 + 32 * wave;
    lb_load(rk, k, kt, o1, lane);
    lb_load(rv, v, kt, o1, lane);
    float dl[NB], lq[NB];
    {
        float4 a[NB][4], b[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const size_t ro = (size_t)(qv[nb] ? rowq[nb] : b1 - 1) * 64;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                a[nb][qq] = *(const float4*)&d_out[ro + 16 * qq + 4 * g];
                b[nb][qq] = *(const float4*)&out[ro + 16 * qq + 4 * g];
            }
            lq[nb] = lse[qv[nb] ? rowq[nb] : b1 - 1];
        }
        block_tile_stage_fast<4>(q, DS, b0, b1, Qt, t);
        block_tile_stage_fast<4>(d_out, DS, b0, b1, Gt, t);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float s = 0.f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                s += a[nb][qq].x * b[nb][qq].x + a[nb][qq].y * b[nb][qq].y + a[nb][qq].z * b[nb][qq].z +
                     a[nb][qq].w * b[nb][qq].w;
            dl[nb] = qv[nb] ? s : 0.f;
            lq[nb] = qv[nb] ? lq[nb] : 0.f;
        }
    }
    __syncthreads();
    s16x4 qf[NB][4], dof[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        lb_frag(qf[nb], Qt, 16 * nb + l15, DS, g);
        lb_frag(dof[nb], Gt, 16 * nb + l15, DS, g);
        dl[nb] = group_sum(dl[nb]);
        if (wave == 0 && g == 0 && qv[nb]) delta[rowq[nb]] = dl[nb];
    }
    __syncthreads();      // every wave has its fragments: the block tiles (waves 2 and 3's tiles) may be overwritten
    f32x4 dQ[4][NB];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dQ[db][nb] = f4zero();
    LbBwdWave& W = sm.w[wave];
    for (; kt < o1; kt += 32 * EQD_WAVES) {
        wave_lds_fence();
        lb_store_rm(rk, W.a_rm, lane);
        lb_store_rm(rv, W.b_rm, lane);
        wave_lds_fence();
        lb_load(rk, k, kt + 32 * EQD_WAVES, o1, lane);
        lb_load(rv, v, kt + 32 * EQD_WAVES, o1, lane);
        f32x4 S[2][NB], dP[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) S[mb][nb] = dP[mb][nb] = f4zero();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            lb_mma_k<NB>(S[mb], W.a_rm, 16 * mb + l15, g, qf);
            lb_mma_k<NB>(dP[mb], W.b_rm, 16 * mb + l15, g, dof);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * mb + 4 * g + r;
                    const float p = key < o1 ? bwd_exp(S[mb][nb][r] - lq[nb]) : 0.f;
                    S[mb][nb][r] = p * (dP[mb][nb][r] - dl[nb]);
                }
        lb_mma_rt<NB>(dQ, W.a_rm, g, l15, S);
    }
    wave_lds_fence();
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm.red(wave)[((db * 2 + nb) * 4 + r) * 64 + lane] = dQ[db][nb][r];
    __syncthreads();
    for (int db = wave; db < 4; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                const size_t at = (size_t)rowq[nb] * d + 16 * db + 4 * g + r;      // x LeakyReLU'(q): see attn_bwd_q_body
                dq[at] = (sm.red(0)[o] + sm.red(1)[o] + sm.red(2)[o] + sm.red(3)[o]) * lrelu_grad(q[at], qk_slope);
            }
        }
}

// WDS: the dS hand-off (eqd_attn_kernels.hip: attn_bwd_kv_body) - the pass also writes its dS tiles (fp32) for the dq pass
// QB: q, k, v point at saved bf16 tensors (bf16 storage mode); DSB: the dS workspace holds bf16 (the dq pass rounds dS to
// bf16 when it forms its MFMA operand: the same bits, half the round trip)
template <int NB, bool WDS = false, bool QB = false, bool DSB = false>
__device__ __forceinline__ void attn_bwd_kv_body_lb(AttnBwdSmemLb& sm, const EqdGraph& G, int item,
                                                    const float* __restrict__ q, const float* __restrict__ k,
                                                    const float* __restrict__ v, const float* __restrict__ out,
                                                    const float* __restrict__ lse, const float* __restrict__ d_out,
                                                    float* __restrict__ dk, float* __restrict__ dv, int half,
                                                    float qk_slope, float* __restrict__ ds = nullptr, int ds_stride = 0,
                                                    const int32_t* __restrict__ seg_start = nullptr) {
    typedef AttnCfg<4> C;
    constexpr int DS = C::DS, d = 64;
    float* Kb = sm.blk(0);
    float* Vb = sm.blk(1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;
    int rowk[NB];
    bool kvd[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowk[nb] = b0 + 16 * nb + l15;
        kvd[nb] = rowk[nb] < b1;
    }
    int kcol[NB];      // WDS: this lane's key columns inside a dS row
    if constexpr (WDS) {
        const int y0 = seg_start[b0];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) kcol[nb] = rowk[nb] - y0;
    }
    LbRegs rq, rg, ro;
    int qt = o0 + 32 * wave;
    lb_load_any<QB>(rq, q, qt, o1, lane);
    lb_load(rg, d_out, qt, o1, lane);
    lb_load(ro, out, qt, o1, lane);
    float lr[2][4];      // lse of the tile's query rows 16 mb + 4 g + r
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = qt + 16 * mb + 4 * g + r;
            const int qc = o1 > o0 ? (qr < o1 ? qr : o1 - 1) : 0;
            const float lv = lse[qc];
            lr[mb][r] = qr < o1 ? lv : 0.f;
        }
    if constexpr (QB) {
        block_tile_stage_bf((const unsigned short*)k, DS, b0, b1, Kb, t);
        block_tile_stage_bf((const unsigned short*)v, DS, b0, b1, Vb, t);
    } else {
        block_tile_stage_fast<4>(k, DS, b0, b1, Kb, t);
        block_tile_stage_fast<4>(v, DS, b0, b1, Vb, t);
    }
    __syncthreads();
    s16x4 kf[NB][4], vf[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        lb_frag(kf[nb], Kb, 16 * nb + l15, DS, g);
        lb_frag(vf[nb], Vb, 16 * nb + l15, DS, g);
    }
    __syncthreads();
    f32x4 dK[4][NB], dV[4][NB];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dK[db][nb] = dV[db][nb] = f4zero();
    LbBwdWave& W = sm.w[wave];
    const int a8 = lane >> 3;
    for (; qt < o1; qt += 32 * EQD_WAVES) {
        wave_lds_fence();
        lb_fix_any<QB>(rq);
        lb_store_rm(rq, W.a_rm, lane);
        lb_store_rm(rg, W.b_rm, lane);
        // delta = rowsum(dO * O) of the streamed query rows (fp32, from the rows as loaded): the lane's 8 columns of its
        // rows 4 a + r, summed over the 8 lanes that share a row group
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p = lb_rowdot(rg, ro, r, lane);
            p += lane_xor<1>(p);
            p += lane_xor<2>(p);
            p += lane_xor<4>(p);
            if ((lane & 7) == 0) sm.dls[wave][4 * a8 + r] = p;
        }
        float lc[2][4], dc[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) lc[mb][r] = lr[mb][r];
        wave_lds_fence();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) dc[mb][r] = sm.dls[wave][16 * mb + 4 * g + r];
        const int qn = qt + 32 * EQD_WAVES;
        lb_load_any<QB>(rq, q, qn, o1, lane);
        lb_load(rg, d_out, qn, o1, lane);
        lb_load(ro, out, qn, o1, lane);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qn + 16 * mb + 4 * g + r;
                const int qc = o1 > o0 ? (qr < o1 ? qr : o1 - 1) : 0;
                const float lv = lse[qc];
                lr[mb][r] = qr < o1 ? lv : 0.f;
            }
        f32x4 S[2][NB], dP[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) S[mb][nb] = dP[mb][nb] = f4zero();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            lb_mma_k<NB>(S[mb], W.a_rm, 16 * mb + l15, g, kf);
            lb_mma_k<NB>(dP[mb], W.b_rm, 16 * mb + l15, g, vf);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = qt + 16 * mb + 4 * g + r < o1;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float p = ok ? bwd_exp(S[mb][nb][r] - lc[mb][r]) : 0.f;
                    S[mb][nb][r] = p;
                    dP[mb][nb][r] = p * (dP[mb][nb][r] - dc[mb][r]);
                }
            }
        if constexpr (WDS) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qr = qt + 16 * mb + 4 * g + r;
                    if (qr < o1) {
                        if constexpr (DSB) {
                            unsigned short* __restrict__ row = (unsigned short*)ds + (size_t)qr * ds_stride;
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                if (kvd[nb]) row[kcol[nb]] = f2bf(dP[mb][nb][r]);
                        } else {
                            float* __restrict__ row = ds + (size_t)qr * ds_stride;
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                if (kvd[nb]) row[kcol[nb]] = dP[mb][nb][r];
                        }
                    }
                }
        }
        lb_mma_rt<NB>(dV, W.b_rm, g, l15, S);
        lb_mma_rt<NB>(dK, W.a_rm, g, l15, dP);
    }
    wave_lds_fence();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {     // dK then dV through the same LDS buffer
        if (pass) __syncthreads();
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sm.red(wave)[((db * 2 + nb) * 4 + r) * 64 + lane] = pass ? dV[db][nb][r] : dK[db][nb][r];
        __syncthreads();
        float* __restrict__ dst = pass ? dv : dk;
        for (int db = wave; db < 4; db += EQD_WAVES)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (!kvd[nb]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                    const size_t at = (size_t)rowk[nb] * d + 16 * db + 4 * g + r;
                    const float s = sm.red(0)[o] + sm.red(1)[o] + sm.red(2)[o] + sm.red(3)[o];
                    float kval;      // (only its sign is used: LeakyReLU'(k))
                    if constexpr (QB) kval = bf2f(((const unsigned short*)k)[at]);
                    else kval = k[at];
                    dst[at] = pass ? s : s * lrelu_grad(kval, qk_slope);
                }
            }
    }
}

// dq pass of the dS hand-off in bf16 mode: dq[query] = sum over the partner's keys of dS[query][key] K[key] with K tiles
// parked transposed in LDS as bf16 and dS rounded to bf16 when the MFMA operand is formed (what the recompute form does
// with the dS it computes itself); structure of attn_bwd_qds_body (eqd_attn_kernels.hip)
struct alignas(16) AttnQdsSmemLb {
    struct alignas(16) Wave {
        unsigned short a_tr[64 * LB_TS];
        float pad[(AttnCfg<4>::RED * 4 > 64 * LB_TS * 2 ? (AttnCfg<4>::RED * 4 - 64 * LB_TS * 2) / 4 : 0)];      // the merge buffer must fit
    } w[EQD_WAVES];
    __device__ __forceinline__ float* red(int wv) { return (float*)&w[wv]; }
};
template <int NB, bool QB = false, bool DSB = false>
__device__ __forceinline__ void attn_bwd_qds_body_lb(AttnQdsSmemLb& sm, const EqdGraph& G, int item, const float* __restrict__ q,
                                                     const float* __restrict__ k, const float* __restrict__ ds, int ds_stride,
                                                     float* __restrict__ dq, int half, float qk_slope) {
    constexpr int d = 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b0 = G.att_items[item * 4 + 0], b1 = G.att_items[item * 4 + 1];
    const int o0 = G.att_items[item * 4 + 2], o1 = G.att_items[item * 4 + 3];
    if (NB == 1) {
        b0 += 16 * half;
        b1 = b1 < b0 + 16 ? b1 : b0 + 16;
    }
    if (b0 >= b1) return;
    int rowq[NB];
    bool qv[NB];
    const float* __restrict__ dsr[NB];      // (DSB: in units of bf16 elements, see ds_load)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rowq[nb] = b0 + 16 * nb + l15;
        qv[nb] = rowq[nb] < b1;
        const size_t ro = (size_t)(qv[nb] ? rowq[nb] : b1 - 1) * ds_stride;      // + (key - o0)
        dsr[nb] = DSB ? (const float*)((const unsigned short*)ds + ro) : ds + ro;
    }
    LbRegs rk;
    int kt = o0 + 32 * wave;
    lb_load_any<QB>(rk, k, kt, o1, lane);
    f32x4 sv[2][NB];
    auto ds_load = [&](int kt_) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = kt_ < o1 ? kt_ - o0 + 16 * mb + 4 * g : 0;
                if constexpr (DSB) {
                    const float4 f = lb_raw8((const unsigned short*)dsr[nb] + col);
                    sv[mb][nb] = f32x4{f.x, f.y, f.z, f.w};
                } else {
                    sv[mb][nb] = *(const f32x4*)(dsr[nb] + col);
                }
            }
    };
    ds_load(kt);
    f32x4 dQ[4][NB];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dQ[db][nb] = f4zero();
    unsigned short* __restrict__ Kt = sm.w[wave].a_tr;
    for (; kt < o1; kt += 32 * EQD_WAVES) {
        wave_lds_fence();
        lb_fix_any<QB>(rk);
        lb_store_tr(rk, Kt, lane);
        wave_lds_fence();
        f32x4 S[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 cur = sv[mb][nb];
                if constexpr (DSB) {      // (the raw 8 bytes of four bf16, converted now)
                    const float4 f = lb_fix8(make_float4(cur[0], cur[1], 0.f, 0.f));
                    cur = f32x4{f.x, f.y, f.z, f.w};
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) S[mb][nb][r] = kt + 16 * mb + 4 * g + r < o1 ? cur[r] : 0.f;
            }
        lb_load_any<QB>(rk, k, kt + 32 * EQD_WAVES, o1, lane);
        ds_load(kt + 32 * EQD_WAVES);
        lb_mma_r<NB>(dQ, Kt, g, l15, S);
    }
    wave_lds_fence();
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm.red(wave)[((db * 2 + nb) * 4 + r) * 64 + lane] = dQ[db][nb][r];
    __syncthreads();
    for (int db = wave; db < 4; db += EQD_WAVES)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!qv[nb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((db * 2 + nb) * 4 + r) * 64 + lane;
                const int f = 16 * db + 4 * g + r;
                float qval;      // (only its sign is used: LeakyReLU'(q))
                if constexpr (QB) qval = bf2f(((const unsigned short*)q)[(size_t)rowq[nb] * d + f]);
                else qval = q[(size_t)rowq[nb] * d + f];
                dq[(size_t)rowq[nb] * d + f] = (sm.red(0)[o] + sm.red(1)[o] + sm.red(2)[o] + sm.red(3)[o]) *
                                               lrelu_grad(qval, qk_slope);
            }
        }
}

template <int NB>
__global__ __launch_bounds__(EQD_BLOCK, NB == 1 ? 2 : 1) void k_attn_bwd_lb(EqdGraph G, const float* __restrict__ q,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        const float* __restrict__ out, const float* __restrict__ lse,
                                                        const float* __restrict__ d_out, float* __restrict__ dq,
                                                        float* __restrict__ dk, float* __restrict__ dv,
                                                        float* __restrict__ delta, float qk_slope) {
    __shared__ AttnBwdSmemLb sm;
    const int per = NB == 1 ? 2 * G.n_att_items : G.n_att_items;      // workgroups per pass
    const bool kv = (int)blockIdx.x >= per;
    const int idx = kv ? (int)blockIdx.x - per : (int)blockIdx.x;
    const int item = NB == 1 ? att_half_item(idx) : idx, half = NB == 1 ? att_half_of(idx) : 0;
    if (!kv)
        attn_bwd_q_body_lb<NB>(sm, G, item, q, k, v, out, lse, d_out, dq, delta, half, qk_slope);
    else
        attn_bwd_kv_body_lb<NB>(sm, G, item, q, k, v, out, lse, d_out, dk, dv, half, qk_slope);
}
