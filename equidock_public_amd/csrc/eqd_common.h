// Shared device helpers + host-side launcher declarations for libequidock_hip.so (gfx950).
//
// MFMA convention used by every matrix kernel in this library ("transposed formulation"):
// a wave owns a tile of ITEMS (edges or nodes) that live on the N axis of
// v_mfma_f32_16x16x4_f32 and FEATURES that live on the M axis, i.e. it computes
//      D^T[feature][item] = W[feature][k] * X^T[k][item].
// With the gfx950 register layout (A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D[row=4*(lane>>4)+reg][col=lane&15]) lane (l15, g) then holds, for item 16*nb+l15, the
// features f = 16*mb + 4*g + r in acc[mb][nb][r] ("F-layout").  Two consequences:
//   * per-item reductions over features (LayerNorm, dot products, softmax over keys) are
//     16 in-lane adds + 2 cross-lane steps (xor 16, 32);
//   * an F-layout tile IS the B operand of the next GEMM if that GEMM walks its K dimension
//     in the order k(s=4*mbi+r, g) = 16*mbi + 4*g + r -- so MLP chains (edge_mlp, coors_mlp,
//     attention P.V, all backward data GEMMs) never leave registers; only the weight (A)
//     operand is read, as one float4 per (mb_out, mb_in).
// f32-input MFMA is bit-for-bit an fmaf chain at the fp32 vector peak rate (157 TF), so the
// 1e-4 fp32 parity bar is met with fp32 summation-order differences only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/equidock_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define EQD_BLOCK 256
#define EQD_WAVES 4
#define EQD_NEG_BIG (-1.0e30f)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 16-byte global load that only needs 4-byte alignment (weight rows with odd leading dimensions, node rows
// of width 69): gfx950 runs global memory in unaligned-access mode, the compiler emits one
// global_load_dwordx4.  `n` = number of valid floats at p: >= 4 full vector; 1..3 tail, fetched as the 4 floats
// that END at the last valid one (the row must hold >= 4 floats before p + n) and shifted; <= 0 zeros (the
// load then goes to `safe`, any address with 16 readable bytes).  Branch-free on purpose: a divergent tail
// path would put a wait behind every load instead of letting a whole batch fly together.
// Phase timestamps for the latency experiments (profiles/exp_trace_*.py build a separate library with
// -DEQD_TRACE; the product library never defines it): lane 0 of wave 0 of workgroup 0 stores clock64().
#ifdef EQD_TRACE
extern __device__ long long eqd_trace_buf[1024];
#define EQD_TR(slot)                                                                          \
    do {                                                                                      \
        const int trs_ = (slot);                                                              \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && trs_ < 256) {           \
            eqd_trace_buf[2 * trs_] = clock64();                                              \
            eqd_trace_buf[2 * trs_ + 1] = wall_clock64();                                     \
        }                                                                                     \
    } while (0)
#define EQD_TR_WG()                                                                           \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 256)                          \
            eqd_trace_buf[512 + 2 * blockIdx.x] = wall_clock64();                             \
    } while (0)
#define EQD_TR_WG_END()                                                                       \
    do {                                                                                      \
        __syncthreads();                                                                      \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 256)                          \
            eqd_trace_buf[512 + 2 * blockIdx.x + 1] = wall_clock64();                         \
    } while (0)
#else
#define EQD_TR(slot) do { } while (0)
#define EQD_TR_WG() do { } while (0)
#define EQD_TR_WG_END() do { } while (0)
#endif

// Pointers that were read back from memory (descriptors copied to LDS) are "generic" to the compiler, and generic
// accesses compile to flat_load / flat_store, which also go through the LDS aperture check and both wait counters.
// Every such pointer in this library addresses HBM: EQD_GAS marks the access as global.
#ifndef EQD_GAS
#define EQD_GAS __attribute__((address_space(1)))
#endif

// Split in two so that a batch of loads can be issued back to back: ld4u_raw is ONE unconditional load
// instruction; ld4u_fix (shift + zero fill, selects) runs later, when the data is consumed.
typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 ld4u_raw(const float* __restrict__ p, int n, const float* __restrict__ safe) {
    const int sh = (n > 0 && n < 4) ? 4 - n : 0;
    return *(const EQD_GAS f4v*)(n > 0 ? p - sh : safe);
}
__device__ __forceinline__ float4 ld4u_fix(f32x4 v, int n) {
    const int sh = (n > 0 && n < 4) ? 4 - n : 0;
    float4 r;
    r.x = n > 0 ? (sh == 0 ? v[0] : sh == 1 ? v[1] : sh == 2 ? v[2] : v[3]) : 0.f;
    r.y = n > 1 ? (sh == 0 ? v[1] : sh == 1 ? v[2] : v[3]) : 0.f;
    r.z = n > 2 ? (sh == 0 ? v[2] : v[3]) : 0.f;
    r.w = n > 3 ? v[3] : 0.f;
    return r;
}
// Kernels that take a large by-value descriptor (row chains: 3.3 KB of job descriptions) copy it from the kernarg
// segment into LDS once, with vector loads: reading the fields on demand through scalar loads costs a scalar-cache /
// L2 round trip (500+ clocks) at every job boundary and pipeline step of a workgroup that has nothing else to do.
#ifndef EQD_KERNARG_PTR
#define EQD_KERNARG_PTR(first_param) ((const void*)__builtin_amdgcn_kernarg_segment_ptr())
#endif
// Values read back from that LDS copy are wave-uniform but live in VGPRs, which would turn every `if (J.act)` into an
// exec-masked region; uni() moves them to SGPRs (v_readfirstlane) so that control flow on them is scalar again.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <class T>
__device__ __forceinline__ T* uni(T* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ EqdLinSrc uni(const EqdLinSrc& S) {
    EqdLinSrc U;
    U.X = uni(S.X); U.mask = uni(S.mask); U.W = uni(S.W);
    U.ldx = uni(S.ldx); U.K = uni(S.K); U.w_rs = uni(S.w_rs); U.w_cs = uni(S.w_cs);
    return U;
}
template <class T>
__device__ __forceinline__ void kernarg_to_lds(T& dst, const void* kernarg, int byte_offset) {
    const int* __restrict__ src = (const int*)((const char*)kernarg + byte_offset);
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 4); i += blockDim.x) ((int*)&dst)[i] = src[i];
}

// ---- bf16 MFMA path (edge-message kernels, storage_bf16 mode) ---------------------------------------------------
// v_mfma_f32_16x16x16_bf16: lane (i = lane & 15, g = lane >> 4) supplies A[i][4g..4g+3] and B[4g..4g+3][i] as 4 bf16
// and receives D[4g + r][i] - the same D layout as the fp32 instruction, so with features on the M axis an output
// tile (4 fp32 per 16-feature block and lane) packed to bf16 IS the B operand of the next GEMM's k-chunk.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short f2bf(float f) {      // round to nearest even (finite inputs)
#ifdef EQD_HOSTSIM
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
#else
    return __builtin_bit_cast(unsigned short, (__bf16)f);      // v_cvt_pk_bf16_f32
#endif
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
#ifdef EQD_HOSTSIM
__device__ __forceinline__ s16x4 pack_bf4(float a, float b, float c, float d) {
    s16x4 r;
    r[0] = (short)f2bf(a); r[1] = (short)f2bf(b); r[2] = (short)f2bf(c); r[3] = (short)f2bf(d);
    return r;
}
#else
// gfx950: v_cvt_pk_bf16_f32 rounds two floats to nearest-even bf16 in ONE instruction (the shift-and-add form above is
// 4 VALU operations per value - more than the bf16 MFMA they feed); same results for finite inputs
typedef float eqd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 eqd_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned eqd_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 pack_bf4(float a, float b, float c, float d) {
    const eqd_f32x2 lo = {a, b}, hi = {c, d};
    const eqd_bf16x2 l = __builtin_convertvector(lo, eqd_bf16x2), h = __builtin_convertvector(hi, eqd_bf16x2);
    const eqd_u32x2 r = {__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, h)};
    return __builtin_bit_cast(s16x4, r);
}
#endif
__device__ __forceinline__ f32x4 mfma_bf(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16 - the gfx950 shape (2 x K per instruction at the issue cost of the 16x16x16 one): a lane supplies
// 8 bf16 of its A row and 8 of its B column, D as above.  WHICH eight k a lane supplies is free as long as A and B agree, so
// two consecutive k-chunks of the 16x16x16 formulation - lane group g holding k = 16 c + 4 g .. + 3 of chunk c - are ONE
// instruction on the concatenated operands: slots 0..3 = chunk c, slots 4..7 = chunk c + 1, for A and B alike.  Same
// products, fp32 accumulate (the hardware's summation order inside an instruction differs: rounding level).
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x8 cat_bf(s16x4 lo, s16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ f32x4 mfma_bf32(s16x8 a, s16x8 b, f32x4 c) {
#if defined(EQD_HOSTSIM) || defined(EQD_MFMA_K16)
    const s16x4 a0 = __builtin_shufflevector(a, a, 0, 1, 2, 3), a1 = __builtin_shufflevector(a, a, 4, 5, 6, 7);
    const s16x4 b0 = __builtin_shufflevector(b, b, 0, 1, 2, 3), b1 = __builtin_shufflevector(b, b, 4, 5, 6, 7);
    return mfma_bf(a1, b1, mfma_bf(a0, b0, c));
#else
    typedef __bf16 eqd_bf16x8 __attribute__((ext_vector_type(8)));
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(eqd_bf16x8, a), __builtin_bit_cast(eqd_bf16x8, b), c, 0, 0, 0);
    // The A operand stays live PAST the instruction, so that hipcc (ROCm 7.2) cannot allocate vdst onto it - which it does
    // whenever A is dead at the MFMA.  Why: builds of this library that carried such instructions were run-to-run
    // NONDETERMINISTIC in k_edge_bwd<bf16, dropout> twice (round 4: "29 % wrong, run-to-run different"; round 6: weight
    // gradients 1e-3 relative apart at 8 x (200, 200), found by comparing two runs bit for bit, profiles/r06_zy_det*.txt), builds
    // without them never.  The overlap looks necessary, not sufficient: micro-tests compute vdst == srcA correctly, alone and
    // under load (profiles/exp_r06_mfma_*.hip; round 4's micro-test, which reported it wrong, has a register copy directly in
    // front of its inline-asm MFMA), and variants of the failing build that kept the overlap but changed the schedule were
    // deterministic (HISTORY.md, round 6) - the mechanism is open.  Round 4 wrote the guard as an empty asm reading `a` behind
    // the builtin; nothing ordered it after the MFMA, the scheduler moved it in front, and 19 kernels carried 1 - 8 such
    // instructions again.  Now the asm also reads one register of the RESULT: it cannot move in front of the instruction that
    // defines d, a is live there, so a and d cannot share registers (one register, not all four: a "+v"(d) pulls accumulators
    // out of the AGPRs, + 655 VALU instructions in k_atb against + 226).  tests/test_abi_and_graph.py scans the shipped code
    // object for the overlap; tests/test_gpu_parity.py repeats seeded steps and compares bits.
    asm volatile("" ::"v"(a), "v"(d[0]));
    return d;
#endif
}

// ds_read_b64_tr_b16 (gfx950): a transposing LDS read.  Every lane supplies the address of 4 consecutive bf16 (8-byte
// aligned); inside each group of 16 lanes, lane i receives element (i & 3) of the reads of lanes (i >> 2), 4 + (i >> 2),
// 8 + (i >> 2), 12 + (i >> 2) - MEASURED on MI355X (profiles/exp_r04/trb16.hip, profiles/r04_p_trb16.txt: three address
// patterns, every lane and element).  With lanes 4 j .. 4 j + 3 pointing at 16 consecutive features of row j of a row-major
// tile, lane i therefore gets rows 0 .. 3 of feature column i: an MFMA operand of a contraction over the tile's ROWS comes
// straight out of the row-major copy, and the second, transposed copy of the tile need not exist.
__device__ __forceinline__ s16x4 lds_tr16(const unsigned short* p) {
#ifdef EQD_HOSTSIM
    const unsigned long long a = (unsigned long long)p;
    const int lane = (int)(threadIdx.x & 63), grp = lane & ~15, i = lane & 15;
    s16x4 r;
    for (int j = 0; j < 4; ++j) {
        const int src = grp + 4 * j + (i >> 2);
        const unsigned lo = (unsigned)__shfl((int)(unsigned)a, src), hi = (unsigned)__shfl((int)(unsigned)(a >> 32), src);
        const unsigned short* q = (const unsigned short*)(((unsigned long long)hi << 32) | lo);
        r[j] = (short)q[i & 3];
    }
    return r;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
#endif
}

__device__ __forceinline__ f32x4 f4zero() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}
__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float lrelu_grad(float y, float s) { return y > 0.f ? 1.f : s; }

// ---- lane exchanges without the LDS pipe (round 4) -------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32: an address computation, a trip through the LDS pipe (~50+ clocks) and an
// s_waitcnt per exchange - 93 of them per 16-edge tile of k_edge_bwd.  Inside a row of 16 lanes the same exchange is a DPP
// move (VALU, no wait): lane ^ 1 / ^ 2 = quad_perm, lane ^ 8 = row_ror:8, lane ^ 4 = row_half_mirror followed by a quad
// reverse; across rows v_permlane16_swap / v_permlane32_swap (gfx950) return (even rows | odd rows) resp. (lower | upper
// half) duplicated, so that v[l] + v[l ^ 16] is the sum of the two results.  All checked on MI355X against __shfl_xor
// (profiles/exp_r04/dpp.hip, profiles/r04_t_dpp.txt).  Pure data movement, and the sums are commutative: same results.
template <int M>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "inside a row of 16 lanes");
#if defined(EQD_HOSTSIM) || defined(EQD_NO_DPP)
    return __shfl_xor(v, M);
#else
    const int i = __builtin_bit_cast(int, v);
    int r;
    // (bound_ctrl = true with full row / bank masks: every lane has a valid source in these permutations, and the compiler
    //  may then treat the "old" operand as undefined - with bound_ctrl = false it zeroed the destination before each move,
    //  103 v_mov per 16-edge tile of k_edge_bwd<bf16> - and fold the move into the consuming add as v_add_f32_dpp)
#ifdef EQD_DPP_ZEROED_OLD
    constexpr bool BC = false;
#else
    constexpr bool BC = true;
#endif
    if constexpr (M == 1) r = __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, BC);            // quad_perm [1,0,3,2]
    else if constexpr (M == 2) r = __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, BC);       // quad_perm [2,3,0,1]
    else if constexpr (M == 8) r = __builtin_amdgcn_update_dpp(0, i, 0x128, 0xf, 0xf, BC);      // row_ror:8
    else r = __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, i, 0x141, 0xf, 0xf, BC), 0x1B, 0xf, 0xf,
                                         BC);                                                    // half mirror, quad reverse
    return __builtin_bit_cast(float, r);
#endif
}
// v[l] (+ | max) v[l ^ M], M = 16 | 32, through v_permlane{16,32}_swap: with both operands holding v, the instruction leaves
// (even rows | odd rows) resp. (lower | upper half) of v duplicated in the two registers.
// The instruction is written as inline assembly: through __builtin_amdgcn_permlane{16,32}_swap hipcc (ROCm 7.2) used the FIRST
// result for both elements of the returned pair in these kernels (v + v instead of v_even + v_odd in the code object; 15 of
// 16 operator tests failed on the GPU, profiles/r04_v_bisect.txt), with or without an opaque copy of the second operand.
template <int M>
__device__ __forceinline__ float lane_swap_sum(float v, int want_max) {
#if defined(EQD_HOSTSIM) || defined(EQD_NO_PLSWAP)
    const float o = __shfl_xor(v, M);
    return want_max ? fmaxf(v, o) : v + o;
#else
    // (inline assembly: the instruction rewrites BOTH registers; s_nop 1 = the wait states hipcc itself puts between a VALU
    //  write of an operand and the swap.  The other direction needs none: for the builtin hipcc (ROCm 7.2, -O3, gfx950) emits
    //  `s_nop 1; v_permlane16_swap_b32 v1, v2; v_add_f32 v1, ...` - the VALU read of the swapped register follows the swap
    //  directly, LLVM's gfx950 hazard recognizer has no wait state after it (checked in the round-5 ADVICE pass by compiling
    //  the builtin form and reading the assembly); tests/parity_common.py: check_lane_exchanges compares group_sum / group_max
    //  with __shfl_xor sums on the GPU)
    float x = v, y = v;
    if constexpr (M == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return want_max ? fmaxf(x, y) : x + y;
#endif
}
// exp(x) for arguments that cannot overflow (softmax: x = score - running maximum <= 0; RBF features: x = -d^2 / sigma).
// hipcc expands expf to  p = x log2e;  e = fma(x, log2e, -p) + x log2e_lo;  n = rint(p);  ldexp(exp2((p - n) + e), n)  and then
// two compare / select pairs for x > 88.7 (-> inf) and x < -103.28 (-> 0): 13 VALU instructions per value, 18 values per
// 32-key step of the attention forward.  Without the guards (one clamp instead, which keeps n inside the integers and the
// error term e small for the -1e30 sentinels of masked keys) the sequence returns the same bits for every x in
// [-103.28, 88.7] and 0 below -103.98; in between (the last half binade above the underflow) it returns the smallest
// denormal, 2^-149, where expf returns 0 - next to the row maximum's exp(0) = 1 that changes no sum.
// tests/parity_common.py: check_lane_exchanges compares it with expf on the GPU, denormal results included.
__device__ __forceinline__ float exp_nooverflow(float x) {
#if defined(EQD_HOSTSIM) || defined(EQD_LIBM_EXP)
    return expf(x);
#else
    const float c = __builtin_bit_cast(float, 0x3fb8aa3bu), cl = __builtin_bit_cast(float, 0x32a5705fu);
    x = x < -1000.f ? -1000.f : x;      // (a compare + select, not fmaxf: fmaxf(NaN, -1000) is -1000, and a NaN score or
                                        //  distance would vanish from the softmax / the RBF features without a trace)
    const float p = x * c;
    float e = __builtin_fmaf(x, c, -p);
    e = __builtin_fmaf(x, cl, e);
    const float n = __builtin_rintf(p);
    const float f = (p - n) + e;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
#endif
}
// 2^x on v_exp_f32 alone.  exp2f adds a compare, two selects, an add and an ldexp per value to return denormal results
// for x < -126; the instruction itself returns 0 there.  bf16 mode's softmax: p < 2^-126 next to the row maximum's p = 1
// changes neither the row sum nor the bf16-rounded probabilities.
__device__ __forceinline__ float exp2_flush(float x) {
#if defined(EQD_HOSTSIM) || defined(EQD_LIBM_EXP)
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
// sum / max over the 4 lane groups (same l15): after this every lane of the column has the total
__device__ __forceinline__ float group_sum(float v) {
#if defined(EQD_HOSTSIM) || defined(EQD_NO_PLSWAP)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
#else
    const float s = lane_swap_sum<16>(v, 0);
    return lane_swap_sum<32>(s, 0);
#endif
}
__device__ __forceinline__ float group_max(float v) {
#if defined(EQD_HOSTSIM) || defined(EQD_NO_PLSWAP)
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
#else
    const float s = lane_swap_sum<16>(v, 1);
    return lane_swap_sum<32>(s, 1);
#endif
}
// sum over the 16 lanes of a lane group (same g)
__device__ __forceinline__ float l16_sum(float v) {
    v += lane_xor<1>(v);
    v += lane_xor<2>(v);
    v += lane_xor<4>(v);
    v += lane_xor<8>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(l16_sum(v)); }
// Sixteen sums over the 16 lanes of a lane group in 15 exchanges (instead of 16 x 4): every lane brings v[0..15],
// lane l15 returns sum over the group's lanes of v[l15] (butterfly that halves the number of live values per step).
__device__ __forceinline__ float reduce16x16(const float (&v)[16], int l15) {
    const bool b3 = (l15 & 8) != 0, b2 = (l15 & 4) != 0, b1 = (l15 & 2) != 0, b0 = (l15 & 1) != 0;
    float w8[8], w4[4], w2[2];
#if defined(EQD_HOSTSIM) || defined(EQD_NO_DPP) || defined(EQD_NO_DPP_ASM)
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = (b3 ? v[i + 8] : v[i]) + lane_xor<8>(b3 ? v[i] : v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = (b2 ? w8[i + 4] : w8[i]) + lane_xor<4>(b2 ? w8[i] : w8[i + 4]);
#else
    // The first two steps as DPP adds with BANK masks (a bank = 4 consecutive lanes of the row): the lanes with bit 3 clear
    // (banks 0, 1) want v[i] + v[i] of lane ^ 8, the others (banks 2, 3) v[i + 8] + v[i + 8] of lane ^ 8 - two masked
    // v_add_f32_dpp per output instead of two selects, a DPP move and an add (the compiler cannot form them: a masked DPP
    // operand folds into an add only when the untouched lanes' result is the add's other operand).  Same two addends per
    // output as the plain form: same bits.  Lane ^ 4 = row_half_mirror, then the quads reversed (lane_xor<4>), bit 2
    // = banks 1, 3.  s_nop 1: the two wait states between a VALU write and a DPP read of the same register (the inputs
    // may have just been written; the hazard recognizer does not look inside an asm block).  Inside the second block every
    // DPP read is at least 7 instructions behind the move that wrote its register.  (The other DPP hazard - a VALU write
    // of EXEC less than 5 wait states earlier - cannot occur: on gfx9 hipcc changes EXEC with scalar instructions only,
    // there is no v_cmpx in any code object of the library.)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(w8[0]), "=&v"(w8[1]), "=&v"(w8[2]), "=&v"(w8[3]), "=&v"(w8[4]), "=&v"(w8[5]), "=&v"(w8[6]), "=&v"(w8[7])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    {
        float t[8];
        asm("s_nop 1\n\t"
            "v_mov_b32_dpp %4, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %5, %13 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %6, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %7, %15 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %8, %16 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %9, %17 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %10, %18 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %11, %19 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %12 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %8, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %5, %13 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %9, %17 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %6, %14 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %10, %18 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %7, %15 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %11, %19 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xa"
            : "=&v"(w4[0]), "=&v"(w4[1]), "=&v"(w4[2]), "=&v"(w4[3]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]),
              "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
            : "v"(w8[0]), "v"(w8[1]), "v"(w8[2]), "v"(w8[3]), "v"(w8[4]), "v"(w8[5]), "v"(w8[6]), "v"(w8[7]));
    }
    (void)b3;
    (void)b2;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = (b1 ? w4[i + 2] : w4[i]) + lane_xor<2>(b1 ? w4[i] : w4[i + 2]);
    return (b0 ? w2[1] : w2[0]) + lane_xor<1>(b0 ? w2[0] : w2[1]);
}

// Ordering point for wave-private LDS traffic (one wave writes, other lanes of the SAME wave
// read).  The hardware executes a wave's LDS instructions in order; this only stops the
// compiler from reordering them.  (The host simulator turns it into a wave rendezvous.)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void eqd_set_error(const char* fmt, ...);
int eqd_check_launch(const char* what);
const char* eqd_tunable(const char* name);   // EQD_* experiment switch, snapshotted once per process (eqd_tunables_reload)
int eqd_num_cus();   // compute units of the current HIP device (256 on MI355X; queried once per device)

static inline size_t eqd_align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// bump allocator over a caller-provided workspace
struct EqdArena {
    char* base;
    size_t size, off;
    bool ok;
    EqdArena(void* p, size_t n) : base((char*)p), size(n), off(0), ok(true) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = eqd_align_up(count * sizeof(T));
        if (base == nullptr || off + bytes > size) {
            ok = false;
            off += bytes;
            return nullptr;
        }
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
};

// deterministic column reduction of per-block partials: out[i] += sum_p partial[p * pstride + i]
#define EQD_RED_MAXSEG 64
struct EqdRedSeg {
    const float* partial;
    int nparts, pstride, n;
    float* out;
    int cols, cols_valid, ld_out;   // cols > 0: element i = (row i / cols, col i % cols) -> out[row * ld_out + col],
                                    // columns >= cols_valid are skipped; cols == 0: out[i]
};
// segments that accumulate into the SAME output (shared layers) form a chain handled by one
// workgroup column, so that no two workgroups ever update the same address
struct EqdRedArg {
    EqdRedSeg s[EQD_RED_MAXSEG];
    int chain_first[EQD_RED_MAXSEG], chain_len[EQD_RED_MAXSEG];
    int chain_blk0[EQD_RED_MAXSEG];   // first workgroup of the chain (one workgroup per 64 output columns)
    int nchains;
};
int eqd_launch_reduce_segments(const EqdRedSeg* segs, int nseg, hipStream_t st);
// list of pending reductions, flushed in a few launches at the end of the backward pass
struct EqdRedList {
    EqdRedSeg seg[512];
    int n;
};

// row-local job chains (k_rowchain)
#define EQD_CHAIN_MAXJOBS 8
struct EqdChainJob {
    EqdLinJob lin;
    int src_local[EQD_MAX_SRC];   // >= 0: source i is the LDS tile left by an earlier job, else global lin.s[i].X
    int out_local;                // >= 0: keep the result in this LDS tile (0..3)
    int type;                     // 0 linear; 1 LeakyReLU->LayerNorm backward (see chain_lnbwd)
    float* aux;                   // type 1: per-workgroup partial sums [blocks][256]
    int prefetch_next;            // filled by eqd_launch_rowchain: linear job whose first step is fetched early, or -1
    int next_lin;                 // filled by eqd_launch_rowchain: the next linear job of the chain, or -1 (k_rowwave)
};
struct EqdChainArg {
    EqdChainJob j[EQD_CHAIN_MAXJOBS];
    int njobs;
};
// partial_rows (optional): rows of LayerNorm-backward partial sums the launch wrote (= its workgroups)
int eqd_launch_rowchain(const EqdChainJob* jobs, int njobs, int rows, hipStream_t st, int* partial_rows = nullptr);
size_t eqd_atb_batch_partial_bytes(int rows);
// the end of the backward riding in the weight-gradient launches (eqd_node_kernels.hip)
int eqd_atb_tail_wanted(const EqdLinJob* dh0_job, int n_atb_jobs);
int eqd_atb_with_tail(const EqdAtbJob* jobs, int njobs, void* partial, size_t partial_bytes, hipStream_t st,
                      const EqdLinJob* dh0_job, const EqdGraph* g, const float* dh0acc, const float* dh0b, int ld, int d_emb,
                      float* demb, float* emb_partial, EqdRedList* defer);
int eqd_rows_resident(int rows);
int eqd_rowres80_on();                // 1 unless EQD_ROWRES80=0 (the first layer's bf16 forward chains on k_rowres80)      // 1: row chains of this size run on k_rowres (eqd_node_kernels.hip)
int eqd_row_tiles(int rows);          // 16-row tiles per workgroup of the row kernels
int eqd_rowchain_blocks(int rows);    // = workgroups of a row-chain launch = LayerNorm-backward partial rows

// internal launchers (defined across the .hip files)
int eqd_launch_vec_reduce(const float* partial, int nparts, int pstride, int n, float* out, hipStream_t st);
int eqd_launch_embed_fwd(const EqdGraph* g, const float* emb, int d_emb, int use_mu, float* h0, int ld, hipStream_t st);
struct EqdRedList;
int eqd_launch_embed_bwd(const EqdGraph* g, const float* dh0, const float* dh0b, int ld, int d_emb, float* demb,
                         float* partial, hipStream_t st, EqdRedList* defer = nullptr);
size_t eqd_embed_bwd_partial_floats(const EqdGraph* g, int d_emb);
// one pending gather (the edge backward's per-edge outputs -> per-node sums), for whoever launches it
struct EqdGatherCall {
    const float *dz, *dxrel, *d_xnew;
    float a;
    float *dP, *dQ, *dx;
    int dz_bf16;
};
struct EqdGatherArgs;      // eqd_gather_inl.h
int eqd_gather_plan(const EqdGraph* g, const EqdGatherCall* c, EqdRedList* pending, EqdGatherArgs* GA, EqdRedArg* RA, int* nblk);
int eqd_gather_rest(EqdRedList* pending, hipStream_t st);
int eqd_attention_bwd_gather_fused(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                   const float* d_out, bool bf16);
// dS hand-off form of the attention backward (eqd_attn_kernels.hip): workspace layout and when it is taken
int eqd_attention_ds_stride(const EqdGraph* g);
size_t eqd_attention_ds_bytes(const EqdGraph* g);
int eqd_attention_ds_wanted(const EqdGraph* g, int d, bool bf16);
int eqd_launch_seg_start(const EqdGraph* g, int32_t* seg_start, hipStream_t st);
int eqd_attention_fwd_bf16_impl(const EqdGraph* g, int d, const float* q, const float* k, const float* v, float* out,
                                float* lse, hipStream_t stream, bool qkv_bf16);
int eqd_launch_attention_bwd_gather(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                    const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                                    float qk_slope, bool bf16, const EqdGatherCall* gc, EqdRedList* pending, hipStream_t st,
                                    float* ds = nullptr, const int32_t* seg_start = nullptr, bool qkv_bf16 = false);
int eqd_launch_node_gather(const EqdGraph* g, const float* dz, const float* dxrel, const float* d_xnew, float a,
                           float* dP, float* dQ, float* dx, hipStream_t st, EqdRedList* pending = nullptr,
                           bool dz_bf16 = false);
int eqd_launch_ln_act_bwd(const float* y_act, const float* d_out, const float* gamma, int rows, int d, int ld,
                          float slope, float eps, float* dz, float* dgamma, float* dbeta, float* partial,
                          hipStream_t st, EqdRedList* defer = nullptr);
int eqd_edge_message_bwd_impl(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q, const float* x,
                              const float* d_aggr_msg, const float* d_xnew, float* dP, float* dQ, float* dx,
                              const EqdEdgeGrads* grads, void* workspace, size_t ws_bytes, hipStream_t st,
                              float* part_override, EqdRedList* defer, EqdGatherCall* hold_gather = nullptr);
size_t eqd_edge_bwd_vecp_floats(const EqdGraph* g);
int eqd_edge_attn_fwd(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q, const float* x,
                      float* aggr_msg, float* x_new, int d_att, const float* q, const float* k, const float* v,
                      float* att_out, float* lse, hipStream_t st);
int eqd_keypoint_pool_fwd_impl(const EqdGraph* g, int n_heads, const float* Wk, const float* Wq, const float* qmean,
                               const float* H, const float* Z, float* Y, float* Y_lig_out, float* Y_rec_out,
                               float* scores, float* lse, float* qp, float* u, hipStream_t st, float* Yc = nullptr);
int eqd_kabsch_fwd_impl(int n_pairs, int n_heads, const float* Y, const float* svd_draws, int svd_seed, float* T,
                        float* T2, float* b, float* A_out, int32_t* status, hipStream_t st, const EqdGraph* g = nullptr,
                        float* lig_out = nullptr, double* usv = nullptr);
int eqd_kabsch_bwd_impl(int n_pairs, int n_heads, const float* Y, const float* A, const float* T, const float* dT,
                        const float* db, const float* dYl_ext, const float* dYr_ext, float* dY, hipStream_t st,
                        const EqdGraph* g = nullptr, const float* d_lig = nullptr, const double* usv = nullptr);
int eqd_rigid_apply_bwd_impl(const EqdGraph* g, const float* d_lig, const float* dT_ext, const float* db_ext, float* dT,
                             float* db, hipStream_t st);
size_t eqd_ln_act_bwd_partial_floats(int rows, int d);
int eqd_launch_fill(float* p, float v, size_t n, hipStream_t st);
int eqd_launch_axpy(float* y, const float* x, float a, size_t n, hipStream_t st);
// eqd_cross_attention_bwd[_bf16] with dq, dk multiplied by LeakyReLU'(q), LeakyReLU'(k) (eqd_attn_kernels.hip)
int eqd_launch_attention_bwd_act(const EqdGraph* g, int d, const float* q, const float* k, const float* v, const float* out,
                                 const float* lse, const float* d_out, float* dq, float* dk, float* dv, float* delta,
                                 float qk_slope, bool bf16, hipStream_t st);
int eqd_launch_seg_mean(const EqdGraph* g, const float* hm, float* qmean, hipStream_t st);
size_t eqd_head_u_bwd_partial_floats(int n_pairs, int K);      // 0 when the batch is not split over segment groups
// part / defer: partial buffer of that size and the pass's pending-reduction list (both or neither)
int eqd_launch_head_u_bwd(const EqdGraph* g, int K, const float* Wk, const float* Wq, const float* qmean,
                          const float* qp, const float* du, float* dWk, float* dWq, float* dqm_part, hipStream_t st,
                          float* part, EqdRedList* defer, int du_chunks = 1);      // du_chunks > 1: du = partial blocks (k_keypoint_bwd_mm)
int eqd_launch_qmean_bwd(const EqdGraph* g, int K, const float* dqm_part, float* dhm, hipStream_t st);
int eqd_launch_keypoint_bwd(const EqdGraph* g, int K, const float* H, const float* Z, const float* scores,
                            const float* lse, const float* u, const float* dY, float* dscores, float* du,
                            float* dH, float* dZ, hipStream_t st, const float* Y = nullptr,      // Y: the forward's keypoints, or NULL
                            int* du_chunks = nullptr);
