// Node-level building blocks: multi-source linear (fwd and dX), A^T B weight gradients with a
// deterministic two-stage reduction, LayerNorm+LeakyReLU backward, embedding fwd/bwd, CSC gather.
//
// Reference arithmetic replaced: every nn.Linear / nn.LayerNorm / nn.Embedding on the hot path
// (src/model/rigid_docking_model.py:119-159, 382, 427-438) and their autograd backward.
#include "eqd_common.h"
#include "eqd_linear_inl.h"
#include "eqd_rowwave_inl.h"
#include "eqd_rowres_inl.h"
#include "eqd_rowres80_inl.h"
#include "eqd_gather_inl.h"

#include <mutex>
#include <vector>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// ------------------------------------------------------------------------------------------
// error plumbing (host)
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void eqd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// ------------------------------------------------------------------------------------------
// experiment / test switches (EQD_* environment variables): read ONCE per process and kept, so that a process that
// changes its environment between a forward and its backward - or between a graph capture and an eager fallback - cannot
// silently switch kernel forms in the middle of a step.  eqd_tunables_reload() (tests, A/B measurements) forgets the
// snapshot; the next launch re-reads the environment.
// ------------------------------------------------------------------------------------------
namespace {
struct EqdTunable {
    char name[32];
    char value[32];
    bool set;
};
EqdTunable g_tun[32];
int g_tun_n = 0;
std::mutex g_tun_mu;
}  // namespace
const char* eqd_tunable(const char* name) {
    // the value is copied out while the lock is held: the returned pointer is the calling thread's own buffer (a ring of
    // eight, so that a caller may hold a few lookups at once), never a slot a concurrent eqd_tunables_reload() rewrites
    static thread_local char ring[8][sizeof(((EqdTunable*)0)->value)];
    static thread_local int ring_at = 0;
    std::lock_guard<std::mutex> lk(g_tun_mu);
    const EqdTunable* hit = nullptr;
    for (int i = 0; i < g_tun_n && !hit; ++i)
        if (strcmp(g_tun[i].name, name) == 0) hit = &g_tun[i];
    if (!hit) {
        const char* v = getenv(name);
        // (table full, a name or a value that does not fit a slot: uncached, never truncated, never wrong)
        if (g_tun_n >= 32 || strlen(name) >= sizeof(g_tun[0].name) || (v && strlen(v) >= sizeof(g_tun[0].value))) return v;
        EqdTunable& t = g_tun[g_tun_n++];
        snprintf(t.name, sizeof(t.name), "%s", name);
        t.set = v != nullptr;
        snprintf(t.value, sizeof(t.value), "%s", v ? v : "");
        hit = &t;
    }
    if (!hit->set) return nullptr;
    char* out = ring[ring_at];
    ring_at = (ring_at + 1) & 7;
    memcpy(out, hit->value, sizeof(hit->value));
    return out;
}
extern "C" void eqd_tunables_reload(void) {
    std::lock_guard<std::mutex> lk(g_tun_mu);
    g_tun_n = 0;
}

// ------------------------------------------------------------------------------------------
// per-launch timing (eqd_profile_*, include/equidock_hip.h): between begin and end every launch of this library on the
// profiled stream is followed by an event record; the time between consecutive events is that launch's duration.
// ------------------------------------------------------------------------------------------
namespace {
struct EqdProfiler {
    bool on = false;
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;
    std::vector<const char*> name;
    std::vector<float> us;
    int n = 0, cap = 0;
};
// process-wide (torch runs the backward on its autograd thread, not on the thread that called eqd_profile_begin);
// only launches on the profiled stream are recorded
EqdProfiler g_prof;
std::mutex g_prof_mu;
}  // namespace

extern "C" int eqd_profile_begin(void* stream, int max_launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EqdProfiler& P = g_prof;
    if (P.on || max_launches < 1) {
        eqd_set_error("eqd_profile_begin: already profiling, or max_launches < 1");
        return EQD_ERR_SHAPE;
    }
    while ((int)P.ev.size() < max_launches + 1) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            eqd_set_error("eqd_profile_begin: hipEventCreate failed");
            return EQD_ERR_LAUNCH;
        }
        P.ev.push_back(e);
    }
    P.name.assign(max_launches, nullptr);
    P.us.assign(max_launches, 0.f);
    P.st = (hipStream_t)stream;
    P.n = 0;
    P.cap = max_launches;
    if (hipEventRecord(P.ev[0], P.st) != hipSuccess) return EQD_ERR_LAUNCH;
    P.on = true;
    return EQD_OK;
}
extern "C" int eqd_profile_end(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EqdProfiler& P = g_prof;
    if (!P.on) return 0;
    P.on = false;
    if (P.n > 0 && hipEventSynchronize(P.ev[P.n]) != hipSuccess) return -1;
    for (int i = 0; i < P.n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, P.ev[i], P.ev[i + 1]) != hipSuccess) return -1;
        P.us[i] = ms * 1000.f;
    }
    return P.n;
}
extern "C" int eqd_profile_mark(const char* label) {
    // work of OTHER libraries on the stream since the last event (torch's loss kernels, fills) gets its own interval
    static std::vector<char*> interned;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EqdProfiler& P = g_prof;
    if (!P.on || P.n >= P.cap || !label) return EQD_OK;
    const char* keep = nullptr;
    for (char* s : interned)
        if (strcmp(s, label) == 0) keep = s;
    if (!keep) {
        interned.push_back(strdup(label));
        keep = interned.back();
    }
    P.name[P.n] = keep;
    (void)hipEventRecord(P.ev[P.n + 1], P.st);
    ++P.n;
    return EQD_OK;
}
extern "C" const char* eqd_profile_name(int i) { return (i >= 0 && i < g_prof.n) ? g_prof.name[i] : ""; }
extern "C" float eqd_profile_us(int i) { return (i >= 0 && i < g_prof.n) ? g_prof.us[i] : 0.f; }

int eqd_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        eqd_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return EQD_ERR_LAUNCH;
    }
    EqdProfiler& P = g_prof;
    if (P.on) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (P.on && P.n < P.cap) {
            P.name[P.n] = what;
            (void)hipEventRecord(P.ev[P.n + 1], P.st);
            ++P.n;
        }
    }
    return EQD_OK;
}
int eqd_num_cus() {
#ifdef EQD_NUM_CUS_FIXED
    return EQD_NUM_CUS_FIXED;      // host simulator
#else
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
#endif
}
extern "C" const char* eqd_last_error(void) { return g_err; }
extern "C" int eqd_abi_version(void) { return EQD_ABI_VERSION; }
extern "C" int eqd_tile_edges(void) { return EQD_TILE_EDGES; }

// ------------------------------------------------------------------------------------------
// Self test of the lane exchanges of eqd_common.h (DPP moves, v_permlane{16,32}_swap through inline assembly) against
// the plain __shfl_xor forms they replace (and of exp_nooverflow / exp2_flush against expf / exp2f), on lane-distinct values and at several points of one kernel (the hazards
// around the swap instruction depend on the neighbouring instructions).  mismatch[0] = number of (lane, check) pairs
// whose bits differ.  Test aid (tests/parity_common.py: check_lane_exchanges).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EQD_BLOCK) void k_selftest_lanes(const float* __restrict__ in, int* __restrict__ mismatch,
                                                              float* __restrict__ out) {
    const int t = threadIdx.x, lane = t & 63;
    int bad = 0;
    auto same = [&](float a, float b) { bad += __builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b); };
    auto ref_sum = [&](float v, int lo, int hi) {      // butterfly over masks lo .. hi (the order the helpers use)
        for (int m = lo; m <= hi; m <<= 1) v += __shfl_xor(v, m);
        return v;
    };
    float v = in[t];
    float acc = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        v = v * 1.25f + (float)rep;                        // a VALU write right in front of the exchanges
        const float gs = group_sum(v);
        same(gs, ref_sum(v, 16, 32));
        const float w = gs * 0.5f - v;                     // a VALU read right behind them
        same(group_max(w), fmaxf(fmaxf(w, __shfl_xor(w, 16)), __shfl_xor(fmaxf(w, __shfl_xor(w, 16)), 32)));
        same(l16_sum(w), ref_sum(w, 1, 8));
        same(wave_sum(v), ref_sum(ref_sum(v, 1, 8), 16, 32));
        same(lane_xor<1>(w), __shfl_xor(w, 1));
        same(lane_xor<2>(w), __shfl_xor(w, 2));
        same(lane_xor<4>(w), __shfl_xor(w, 4));
        same(lane_xor<8>(w), __shfl_xor(w, 8));
        float a16[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a16[i] = w * (float)(i + 1) + v;
        float r16 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float si = ref_sum(a16[i], 1, 8);
            r16 = (lane & 15) == i ? si : r16;
        }
        // (reduce16x16 sums in a different order than a plain butterfly: compared to rounding, not bit for bit)
        const float got = reduce16x16(a16, lane & 15);
        bad += !(fabsf(got - r16) <= 1e-5f * (fabsf(r16) + 1.f));
        acc += gs + w;
    }
    // exp_nooverflow against expf, bit for bit: ordinary softmax arguments, the denormal results around -87 .. -104, the
    // underflow threshold, far below it and the -1e30 sentinel (mismatch[1]); exp2_flush against exp2f where the result is
    // normal (mismatch[2]) and 0 at the sentinel (mismatch[3])
    int bad_e = 0, bad_e2 = 0, bad_s = 0;
    {
        const float u = fabsf(in[t]) + 1e-3f * (float)t;
        const float xs[8] = {-u, -7.f * u, -40.f * u - 1.f, -87.f - 0.07f * (float)t, -103.f - 0.01f * (float)t, -150.f * u - 104.f,
                             -1e30f, 0.25f * u};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = exp_nooverflow(xs[i]), b = expf(xs[i]);
            // (x in (-103.98, -103.28): 2^-149 where expf returns 0, see exp_nooverflow)
            const bool last_half_binade = xs[i] < -103.27f && xs[i] > -103.99f && b == 0.f && a <= 1.5e-45f;
            if (__builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b) && !last_half_binade) {
                ++bad_e;
                acc = xs[i];      // (out256 then shows an argument that differs)
            }
            const float x2 = fmaxf(xs[i], -125.f);
            bad_e2 += __builtin_bit_cast(unsigned, exp2_flush(x2)) != __builtin_bit_cast(unsigned, exp2f(x2));
        }
        bad_s += exp2_flush(-1e30f) != 0.f;
        bad_s += exp2_flush(in[t] - 1e30f) != 0.f;
        // a NaN argument stays a NaN (a clamp written as fmaxf would turn it into exp(-1000) = 0); the NaN is made from the
        // input so that the compiler cannot fold the test
        const float qnan = __builtin_bit_cast(float, 0x7fc00000u | (__builtin_bit_cast(unsigned, in[t]) & 0xffu));
        {
            const float en = exp_nooverflow(qnan), e2n = exp2_flush(qnan);
            bad_e += !(en != en);
            bad_s += !(e2n != e2n);
        }
    }
    if (bad_e) atomicAdd(mismatch + 1, bad_e);
    if (bad_e2) atomicAdd(mismatch + 2, bad_e2);
    if (bad_s) atomicAdd(mismatch + 3, bad_s);
    out[t] = acc;
    if (bad) atomicAdd(mismatch, bad);
}
extern "C" int eqd_selftest_lane_exchanges(const float* in256, int* mismatch, float* out256, void* stream) {
    if (!in256 || !mismatch || !out256) {
        eqd_set_error("eqd_selftest_lane_exchanges: NULL argument");
        return EQD_ERR_NULL;
    }
    hipLaunchKernelGGL(k_selftest_lanes, dim3(1), dim3(EQD_BLOCK), 0, (hipStream_t)stream, in256, mismatch, out256);
    return eqd_check_launch("k_selftest_lanes");
}

// ------------------------------------------------------------------------------------------
// k_linear: Y = alpha * f(sum_s (X_s * lrelu'(mask_s)) W_s^T + bias) + beta * R
// workgroup tile: 16 rows (items on the MFMA N axis) x up to 80 outputs (M axis, MB = 5 blocks)
// ------------------------------------------------------------------------------------------
#define LIN_MAXJOBS 8
struct LinJobsArg {
    EqdLinJob j[LIN_MAXJOBS];
};

template <int RT, bool BF = false>
__global__ __launch_bounds__(EQD_BLOCK, 2) void k_linear(LinJobsArg jobs) {
    __shared__ LinSmem<RT> sm;
    __shared__ __attribute__((aligned(16))) EqdLinJob Jl;      // this workgroup's job, copied out of the kernarg segment
    kernarg_to_lds(Jl, EQD_KERNARG_PTR(jobs), (int)(blockIdx.y * sizeof(EqdLinJob)));
    __syncthreads();
    const EqdLinJob& J = Jl;
    const int row0 = (int)blockIdx.x * 16 * RT;
    if (row0 >= J.rows) return;      // uniform for the whole workgroup
    EQD_TR_WG();
    LinRegs<RT> RA;
    const JobW W = jobw_load(&Jl, (int)(sizeof(EqdLinJob) / 4), threadIdx.x & 63);
    linear_tile<RT, BF, RT == 1>(J, W, false, nullptr, -1, sm, nullptr, row0, RA, false, false, W);
    EQD_TR_WG_END();
}
// k_linear_simple (round 6): the jobs that need none of linear_tile's generality - ONE 64-wide source with k-contiguous
// weights, 64 outputs, optional bias / LeakyReLU, no mask, LayerNorm, residual or dropout factor: the five node projections
// (P, Q, q, k, v) of every 64-wide layer, i.e. 8 of the 10 k_linear launches of a DB5.5-sized step.  k_linear<1> allocates
// 218 registers for its lean + general bodies (six sources' addresses, two register sets, LayerNorm state): two workgroups
// per CU, and the 1 000 workgroups of a projection launch (5 jobs x 200 row tiles) ran as TWO rounds of 512.  This body
// needs < 64: with a third of the LDS it runs five workgroups per CU and the launch is one round.  Same loads, the same
// lin_mma call and the same epilogue expressions as linear_tile_lean: same bits.
struct alignas(16) LinSimpleSmem {
    float Xl[16 * LIN_S];
    float Wl[64 * LIN_S];
};
template <bool BF>
__global__ __launch_bounds__(EQD_BLOCK, 4) void k_linear_simple(LinJobsArg jobs) {
    __shared__ LinSimpleSmem sm;
    __shared__ __attribute__((aligned(16))) EqdLinJob Jl;
    kernarg_to_lds(Jl, EQD_KERNARG_PTR(jobs), (int)(blockIdx.y * sizeof(EqdLinJob)));
    __syncthreads();
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4, tr = t >> 4, tc = t & 15;
    const JobW W = jobw_load(&Jl, (int)(sizeof(EqdLinJob) / 4), lane);
    const int rows = jw_i(W, LJ(rows));
    const int row0 = (int)blockIdx.x * 16;
    if (row0 >= rows) return;      // uniform for the whole workgroup
    const EqdLinSrc S = jw_src(W, 0);
    // this thread's part of the step: X row tr (clamped), columns 4 tc ..; weight rows tr + 16 j, columns 4 tc ..
    int rowc = row0 + tr;
    rowc = rowc < rows ? rowc : rows - 1;
    const f32x4 xv = *(const EQD_GAS f4v*)(S.X + (size_t)rowc * S.ldx + 4 * tc);
    f32x4 wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = *(const EQD_GAS f4v*)(S.W + (size_t)(tr + 16 * j) * S.w_rs + 4 * tc);
    // epilogue operands of this wave's output block (features f0 .. f0 + 3 of row row0 + l15)
    const int f0 = 16 * wave + 4 * g;
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    f32x4 bias = f4zero();
    if (jbias) bias = *(const EQD_GAS f4v*)(jbias + f0);
    *(f32x4*)&sm.Xl[tr * LIN_S + 4 * tc] = xv;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)&sm.Wl[(tr + 16 * j) * LIN_S + 4 * tc] = wv[j];
    __syncthreads();
    f32x4 acc[1][2], acc2[1][2];
    acc[0][0] = acc[0][1] = acc2[0][0] = acc2[0][1] = f4zero();
    const int mbs[2] = {wave, wave + 4};
    const float* Xs[1] = {sm.Xl};
    lin_mma<1, 1, 4, false, BF, true>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
    const int act = jw_i(W, LJ(act));
    const float slope = jw_f(W, LJ(slope)), alpha = jw_f(W, LJ(alpha)), beta = jw_f(W, LJ(beta));
    f32x4 yv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = (acc[0][0][r] + acc2[0][0][r]) + bias[r];
        if (act) y = lrelu(y, slope);
        yv[r] = alpha * y + beta * 0.f;      // (no residual: linear_tile_lean's expression with res = 0)
    }
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));      // (descriptor reads are lane exchanges: outside the predicated stores)
    if (jY && rv) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = yv;
    unsigned short* const jYb = jw_p<unsigned short>(W, LJ(Yb));      // bf16 copy (EqdLinJob.Yb), or NULL
    const int ldyb = jw_i(W, LJ(ldyb));
    if (jYb && rv) *(EQD_GAS s16x4*)&jYb[(size_t)rowi * ldyb + f0] = pack_bf4(yv[0], yv[1], yv[2], yv[3]);
#undef LJ
}
// may every job of the launch run on k_linear_simple?  (EQD_LINEAR_SIMPLE=0 keeps k_linear: tests, A/B runs)
static bool lin_simple_eligible(const EqdLinJob* jobs, int n) {
    const char* f = eqd_tunable("EQD_LINEAR_SIMPLE");
    if (f && f[0] == '0' && f[1] == 0) return false;
    for (int i = 0; i < n; ++i) {
        const EqdLinJob& J = jobs[i];
        if (J.nsrc != 1 || J.M != 64 || J.s[0].K != 64 || J.s[0].w_cs != 1 || J.s[0].mask || !J.s[0].W || !J.s[0].X || J.ln_g ||
            J.R || J.mul || J.pre_ln || J.pad_to != 0 || J.rows <= 0)
            return false;
    }
    return n > 0;
}

// k_linear_simple80 (round 6): the same idea for the FIRST layer's projection group - one source of 65 .. 80 columns (69:
// residue embedding + surface features), 64 .. 80 outputs, rows zero-padded to pad_to (the attention operands' 80-wide rows).
// On k_linear these jobs take the general body: two pipeline steps (64 columns, then the 5 left over) with two barriers each,
// element-wise stores for the 69-wide outputs, 218 registers - 16.6 us per launch at 8 x (200, 200) against 5 us for a 64-wide
// layer's group.  Here both steps' operands are requested up front, the small step's columns go to columns 64 .. 79 of the same
// LDS rows (the row stride has room for 80), ONE barrier, and lin_mma is called as the general body calls it (4 chunks, then 1):
// the same products in the same order, the same epilogue expressions - same bits.  Whole 4-column groups only (eligibility:
// the row is padded at least to the end of M's last group), so every store is 16 bytes.
struct alignas(16) LinSimple80Smem {
    float Xl[16 * LIN_S];
    float Wl[80 * LIN_S];
};
template <bool BF>
__global__ __launch_bounds__(EQD_BLOCK, 4) void k_linear_simple80(LinJobsArg jobs) {
    __shared__ LinSimple80Smem sm;
    __shared__ __attribute__((aligned(16))) EqdLinJob Jl;
    kernarg_to_lds(Jl, EQD_KERNARG_PTR(jobs), (int)(blockIdx.y * sizeof(EqdLinJob)));
    __syncthreads();
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4, tr = t >> 4, tc = t & 15;
    const JobW W = jobw_load(&Jl, (int)(sizeof(EqdLinJob) / 4), lane);
    const int rows = jw_i(W, LJ(rows)), M = jw_i(W, LJ(M));
    const int row0 = (int)blockIdx.x * 16;
    if (row0 >= rows) return;      // uniform for the whole workgroup
    const EqdLinSrc S = jw_src(W, 0);
    const int kc = S.K - 64;       // the small step's width, 1 .. 16
    int rowc = row0 + tr;
    rowc = rowc < rows ? rowc : rows - 1;
    // full step: X row tr, columns 4 tc ..; weight rows tr + 16 j (clamped into the matrix), columns 4 tc ..
    const float* __restrict__ xrow = S.X + (size_t)rowc * S.ldx;
    const f32x4 xv = *(const EQD_GAS f4v*)(xrow + 4 * tc);
    f32x4 wv[5];
    int mrow[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int m = tr + 16 * j;
        mrow[j] = m < M ? m : M - 1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = *(const EQD_GAS f4v*)(S.W + (size_t)mrow[j] * S.w_rs + 4 * tc);
    wv[4] = f4zero();
    if (M > 64) wv[4] = *(const EQD_GAS f4v*)(S.W + (size_t)mrow[4] * S.w_rs + 4 * tc);
    // small step: one element per (row, k) / (weight row, k), k clamped into the chunk
    const int kk = 64 + (tc < kc ? tc : kc - 1);
    const float xs = ((const EQD_GAS float*)xrow)[kk];
    float ws[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) ws[j] = ((const EQD_GAS float*)S.W)[(size_t)mrow[j] * S.w_rs + kk];
    // epilogue operands of this wave's output blocks (wave 0 also owns block 4 when M > 64)
    const int mbn = (M + 15) >> 4;
    const int mbs[2] = {wave, wave + 4};
    const bool own[2] = {wave < mbn, wave + 4 < mbn};
    const float* const jbias = jw_p<const float>(W, LJ(bias));
    f32x4 bias[2];
    int nf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f0 = 16 * mbs[i] + 4 * g;
        nf[i] = own[i] ? M - f0 : 0;            // valid features at f0 (<= 0: none)
        bias[i] = jbias ? ld4u_raw(jbias + f0, nf[i], jbias) : f4zero();
    }
    const f32x4 z = f4zero();
    const bool kv = tc < kc;
    *(f32x4*)&sm.Xl[tr * LIN_S + 4 * tc] = xv;
    sm.Xl[tr * LIN_S + 64 + tc] = kv ? xs : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)&sm.Wl[(tr + 16 * j) * LIN_S + 4 * tc] = (tr + 16 * j < M) ? wv[j] : z;
    if (M > 64) *(f32x4*)&sm.Wl[(64 + tr) * LIN_S + 4 * tc] = (64 + tr < M) ? wv[4] : z;
#pragma unroll
    for (int j = 0; j < 5; ++j) sm.Wl[(tr + 16 * j) * LIN_S + 64 + tc] = (kv && tr + 16 * j < M) ? ws[j] : 0.f;
    __syncthreads();
    f32x4 acc[1][2], acc2[1][2];
    acc[0][0] = acc[0][1] = acc2[0][0] = acc2[0][1] = f4zero();
    const float* Xs[1] = {sm.Xl};
    const float* Xs2[1] = {sm.Xl + 64};
    if (own[1]) {
        lin_mma<1, 2, 4, false, BF, true>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
        lin_mma<1, 2, 1, false, BF, true>(acc, acc2, Xs2, sm.Wl + 64, mbs, l15, g);
    } else if (own[0]) {
        lin_mma<1, 1, 4, false, BF, true>(acc, acc2, Xs, sm.Wl, mbs, l15, g);
        lin_mma<1, 1, 1, false, BF, true>(acc, acc2, Xs2, sm.Wl + 64, mbs, l15, g);
    }
    const int act = jw_i(W, LJ(act)), pad_to = jw_i(W, LJ(pad_to));
    const float slope = jw_f(W, LJ(slope)), alpha = jw_f(W, LJ(alpha)), beta = jw_f(W, LJ(beta));
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));      // (descriptor reads are lane exchanges: outside the predicated stores)
    unsigned short* const jYb = jw_p<unsigned short>(W, LJ(Yb));
    const int ldyb = jw_i(W, LJ(ldyb));
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const bool plain = (M & 3) == 0;
    const int lim = pad_to > M ? pad_to : M;      // columns written in every row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 bs = plain ? make_float4(bias[i][0], bias[i][1], bias[i][2], bias[i][3]) : ld4u_fix(bias[i], nf[i]);
        const float bb[4] = {bs.x, bs.y, bs.z, bs.w};
        f32x4 yv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y = (acc[0][i][r] + acc2[0][i][r]) + bb[r];
            if (act) y = lrelu(y, slope);
            const float v = r < nf[i] ? y : 0.f;
            yv[r] = alpha * v + beta * 0.f;      // (no residual: linear_tile's expression with res = 0)
        }
        const int f0 = 16 * mbs[i] + 4 * g;
        if (own[i] && f0 < lim && rv) {
            if (jY) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = yv;
            if (jYb && nf[i] > 0) *(EQD_GAS s16x4*)&jYb[(size_t)rowi * ldyb + f0] = pack_bf4(yv[0], yv[1], yv[2], yv[3]);
        }
    }
#undef LJ
}
// EQD_LINEAR_SIMPLE80: 0 = off, 1 = only where k_linear would run one row tile per workgroup, 2 = at every size
static int lin_simple80_mode() {
    const char* f = eqd_tunable("EQD_LINEAR_SIMPLE80");
    if (f && f[0] >= '0' && f[0] <= '2' && f[1] == 0) return f[0] - '0';
    return 2;
}
static bool lin_simple80_eligible(const EqdLinJob* jobs, int n) {
    for (int i = 0; i < n; ++i) {
        const EqdLinJob& J = jobs[i];
        if (J.nsrc != 1 || J.M < 64 || J.M > 80 || J.s[0].K <= 64 || J.s[0].K > 80 || J.s[0].w_cs != 1 || J.s[0].mask || !J.s[0].W ||
            !J.s[0].X || J.ln_g || J.R || J.mul || J.pre_ln || J.rows <= 0)
            return false;
        const int m4 = (J.M + 3) & ~3, m16 = (J.M + 15) & ~15;
        if (J.pad_to != 0 && ((J.pad_to & 3) || J.pad_to < m4 || J.pad_to > m16)) return false;
        if ((J.M & 3) && (J.pad_to == 0 || J.Yb)) return false;      // a started 4-column group must be written whole
        if (J.Y && ((J.ldy & 3) || ((uintptr_t)J.Y & 15))) return false;
        if (J.Yb && ((J.ldyb & 3) || ((uintptr_t)J.Yb & 7))) return false;
    }
    return n > 0;
}

// 16-row tiles per workgroup.  One tile everywhere: with the lean (precomputed-address) step pipeline, which only fits
// the register budget with one tile, 16-row workgroups at two per CU measured faster than 32-row workgroups at every
// size (config C: 6 040 vs 5 950 pairs/s).  EQD_ROW_TILES=2 selects the two-tile kernels (kept and tested: they stage a
// step's weights once per 32 rows, which pays off once weights stop fitting the L2, i.e. for wider models).
int eqd_row_tiles(int rows) {
    (void)rows;
    const char* f = eqd_tunable("EQD_ROW_TILES");
    if (f && (f[0] == '1' || f[0] == '2') && f[1] == 0) return f[0] - '0';
    return 1;
}
int eqd_rowchain_blocks(int rows) {
    const int per = 16 * eqd_row_tiles(rows);
    return (rows + per - 1) / per;
}
// k_linear (what is left on it at large sizes: layer 0's 69-wide projection group and dh job, on the general body):
// two tiles from 3 tiles per CU - the general body's ~3 500 clocks per step are then paid once per 32 rows (C bf16:
// 156 -> 129 us per step, E: 78 -> 67; k_rowchain does not gain and keeps one tile)
static int linear_row_tiles(int rows) {
    const char* f = eqd_tunable("EQD_ROW_TILES");
    if (f && (f[0] == '1' || f[0] == '2') && f[1] == 0) return f[0] - '0';
    return (rows + 15) / 16 >= 3 * eqd_num_cus() ? 2 : 1;
}

// ------------------------------------------------------------------------------------------
// k_rowchain: a sequence of row-local jobs on the same 16 rows in ONE launch; intermediate tiles stay
// in LDS.  Replaces  node_mlp.0 -> LayerNorm -> node_mlp.4 (+skip) -> next layer's P/Q/q/k/v
// (7 jobs, was 3 launches) in the forward, and  node_mlp.4^T -> LeakyReLU/LayerNorm backward ->
// d aggr_msg / d aggr_cross / d h0  (5 jobs, was 3 launches + a reduction) in the backward.
// ------------------------------------------------------------------------------------------
// d == 64, one row tile: the workgroup works in the layout of the linear jobs' epilogue (lane = row l15, features
// 16 wave + 4 g .. + 3): a row statistic is an in-lane sum of 4, two cross-lane steps and an exchange between the four
// waves through LDS - three rounds (mean; variance; the two projections) instead of the 16 six-step wave reductions per
// wave of the one-wave-per-row layout, which ran one after the other (9 800 of the chain's 47 600 clocks).  yp: the
// job's y_act values, fetched by the caller before the previous job (have_y) or here.
__device__ __forceinline__ void chain_lnbwd64(const JobW& W, float (*Lb)[LIN_LOCALS][16 * LIN_S], float (*stat)[EQD_WAVES][16],
                                              int row0, bool have_y, f32x4 yp) {
#define LJ(f) JW_OFF(EqdLinJob, f)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int rows = jw_i(W, LJ(rows));
    const int src_l = jw_i(W, JW_OFF(EqdChainJob, src_local)), out_l = jw_i(W, JW_OFF(EqdChainJob, out_local));
    const float slope = jw_f(W, LJ(slope)), ln_eps = jw_f(W, LJ(ln_eps));
    const int f0 = 16 * wave + 4 * g;
    const int rowi = row0 + l15;
    const bool rv = rowi < rows;
    const f32x4 gam = *(const EQD_GAS f4v*)(jw_p<const float>(W, LJ(ln_g)) + f0);
    if (!have_y) {
        const int ldx = jw_i(W, LJ(s) + JW_OFF(EqdLinSrc, ldx));
        yp = *(const EQD_GAS f4v*)(jw_p<const float>(W, LJ(s) + JW_OFF(EqdLinSrc, X)) + (size_t)(rv ? rowi : rows - 1) * ldx + f0);
    }
    float* const jY = jw_p<float>(W, LJ(Y));
    const int ldy = jw_i(W, LJ(ldy));
    float* const jaux = jw_p<float>(W, JW_OFF(EqdChainJob, aux));
    f32x4 mulv = {1.f, 1.f, 1.f, 1.f};       // dropout factors of the forward (training mode): d LeakyReLU * keep * s
    if (const float* const jmul = jw_p<const float>(W, LJ(mul)))
        mulv = *(const EQD_GAS f4v*)(jmul + (size_t)(rv ? rowi : rows - 1) * jw_i(W, LJ(ld_mul)) + f0);
#undef LJ
    f32x4 o = *(const f32x4*)&Lb[0][src_l][l15 * LIN_S + f0];
    f32x4 y = yp;
    if (!rv) {
        o = f4zero();
        y = f4zero();
    }
    const float invd = 1.f / 64.f;
    auto row_total = [&](float v, int slot) {      // sum over the row's 64 features; all 256 threads call it
        v = group_sum(v);
        if (g == 0) stat[slot][wave][l15] = v;
        __syncthreads();
        return (stat[slot][0][l15] + stat[slot][1][l15]) + (stat[slot][2][l15] + stat[slot][3][l15]);
    };
    const float mean = row_total((y[0] + y[1]) + (y[2] + y[3]), 0) * invd;
    f32x4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = y[r] - mean;
    const float rstd = 1.f / sqrtf(row_total((c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]), 1) * invd + ln_eps);
    f32x4 xh, dx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xh[r] = c[r] * rstd;
        dx[r] = o[r] * gam[r];
    }
    // the two projections share one exchange (slots 2 and 3)
    float p1 = group_sum((dx[0] + dx[1]) + (dx[2] + dx[3]));
    float p2 = group_sum((dx[0] * xh[0] + dx[1] * xh[1]) + (dx[2] * xh[2] + dx[3] * xh[3]));
    if (g == 0) {
        stat[2][wave][l15] = p1;
        stat[3][wave][l15] = p2;
    }
    __syncthreads();
    const float s1 = ((stat[2][0][l15] + stat[2][1][l15]) + (stat[2][2][l15] + stat[2][3][l15])) * invd;
    const float s2 = ((stat[3][0][l15] + stat[3][1][l15]) + (stat[3][2][l15] + stat[3][3][l15])) * invd;
    f32x4 z;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = rv ? rstd * (dx[r] - s1 - xh[r] * s2) * (lrelu_grad(y[r], slope) * mulv[r]) : 0.f;
    *(f32x4*)&Lb[0][out_l][l15 * LIN_S + f0] = z;
    if (rv && jY) *(EQD_GAS f4v*)&jY[(size_t)rowi * ldy + f0] = z;
    // d gamma / d beta of the workgroup's 16 rows: 8 values per lane, summed over the 16 lanes of the group by a halving
    // butterfly (8 exchanges); lane l15 ends with value (l15 >> 1): 0..3 = d gamma of feature f0 + r, 4..7 = d beta
    float v8[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v8[r] = o[r] * xh[r];        // rows beyond the matrix carry o = 0
        v8[4 + r] = o[r];
    }
    const bool b3 = (l15 & 8) != 0, b2 = (l15 & 4) != 0, b1 = (l15 & 2) != 0;
    float w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = (b3 ? v8[i + 4] : v8[i]) + lane_xor<8>(b3 ? v8[i] : v8[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = (b2 ? w4[i + 2] : w4[i]) + lane_xor<4>(b2 ? w4[i] : w4[i + 2]);
    float w1 = (b1 ? w2[1] : w2[0]) + lane_xor<2>(b1 ? w2[0] : w2[1]);
    w1 += lane_xor<1>(w1);
    if ((l15 & 1) == 0) {
        const int idx = l15 >> 1;          // b3 b2 b1
        jaux[(size_t)blockIdx.x * 256 + (idx < 4 ? 0 : 128) + f0 + (idx & 3)] = w1;
    }
    // (the workgroup's partial row is [d gamma 0..127 | d beta 128..255]; columns of features >= 64 are never read)
}

template <int RT>
__device__ __forceinline__ void chain_lnbwd(const EqdChainJob& C, float (*Lb)[LIN_LOCALS][16 * LIN_S], float (*red)[256],
                                            int row0) {
    // y_act = LeakyReLU(z) (saved by the forward) is lin.s[0].X; the incoming gradient is the LDS tile
    // src_local[0]; dz goes to LDS tile out_local and to lin.Y; per-workgroup (d gamma | d beta) to aux.
    const EqdLinJob& J = C.lin;
    // descriptor fields come from the LDS copy: scalarise them, address HBM as global (see uni / EQD_GAS)
    const EQD_GAS float* const jlng = (const EQD_GAS float*)uni(J.ln_g);
    const EQD_GAS float* const jX = (const EQD_GAS float*)uni(J.s[0].X);
    EQD_GAS float* const jY = (EQD_GAS float*)uni(J.Y);
    EQD_GAS float* const jaux = (EQD_GAS float*)uni(C.aux);
    const EQD_GAS float* const jmul = (const EQD_GAS float*)uni(J.mul);      // dropout factors of the forward, or NULL
    const int ld_mul = uni(J.ld_mul);
    const int rows = uni(J.rows), ldx = uni(J.s[0].ldx), ldy = uni(J.ldy), src_l = uni(C.src_local[0]), out_l = uni(C.out_local);
    const float slope = uni(J.slope), ln_eps = uni(J.ln_eps);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int d = uni(J.M);
    const int f0 = lane, f1 = lane + 64;
    const bool v0 = f0 < d, v1 = f1 < d;
    const float g0 = v0 ? jlng[f0] : 0.f, g1 = v1 ? jlng[f1] : 0.f;
    const float invd = 1.f / (float)d;
    float dg0 = 0.f, dg1 = 0.f, db0 = 0.f, db1 = 0.f;
    // all y_act rows of the wave first (unpredicated, clamped), then the arithmetic
    float y0s[RT][4], y1s[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            int row = row0 + 16 * rt + 4 * wave + rr;
            row = row < rows ? row : rows - 1;
            const size_t o = (size_t)row * ldx;
            y0s[rt][rr] = jX[o + (v0 ? f0 : 0)];
            y1s[rt][rr] = jX[o + (v1 ? f1 : 0)];
        }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float* __restrict__ din = Lb[rt][src_l];
        float* __restrict__ dout = Lb[rt][out_l];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = 4 * wave + rr;
            const int row = row0 + 16 * rt + lr;
            const bool rv = row < rows;
            const float y0 = (rv && v0) ? y0s[rt][rr] : 0.f, y1 = (rv && v1) ? y1s[rt][rr] : 0.f;
            const float o0 = (rv && v0) ? din[lr * LIN_S + f0] : 0.f, o1 = (rv && v1) ? din[lr * LIN_S + f1] : 0.f;
            const float mean = wave_sum(y0 + y1) * invd;
            const float c0 = v0 ? y0 - mean : 0.f, c1 = v1 ? y1 - mean : 0.f;
            const float rstd = 1.f / sqrtf(wave_sum(c0 * c0 + c1 * c1) * invd + ln_eps);
            const float xh0 = c0 * rstd, xh1 = c1 * rstd;
            const float dx0 = o0 * g0, dx1 = o1 * g1;
            const float s1 = wave_sum(dx0 + dx1) * invd;
            const float s2 = wave_sum(dx0 * xh0 + dx1 * xh1) * invd;
            float z0 = rstd * (dx0 - s1 - xh0 * s2) * lrelu_grad(y0, slope);
            float z1 = rstd * (dx1 - s1 - xh1 * s2) * lrelu_grad(y1, slope);
            if (jmul) {
                const size_t mo = (size_t)(rv ? row : rows - 1) * ld_mul;
                z0 *= jmul[mo + (v0 ? f0 : 0)];
                z1 *= jmul[mo + (v1 ? f1 : 0)];
            }
            if (v0) dout[lr * LIN_S + f0] = rv ? z0 : 0.f;
            if (v1) dout[lr * LIN_S + f1] = rv ? z1 : 0.f;
            if (rv && jY) {
                if (v0) jY[(size_t)row * ldy + f0] = z0;
                if (v1) jY[(size_t)row * ldy + f1] = z1;
            }
            if (rv) {
                dg0 += o0 * xh0;
                dg1 += o1 * xh1;
                db0 += o0;
                db1 += o1;
            }
        }
    }
    red[wave][lane] = dg0;
    red[wave][64 + lane] = dg1;
    red[wave][128 + lane] = db0;
    red[wave][192 + lane] = db1;
    __syncthreads();
    jaux[(size_t)blockIdx.x * 256 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// OCC: workgroups per CU the register budget is sized for.  2 (256 registers) wherever a launch has more 16-row tiles than
// the chip has CUs; 1 (up to 512) for DB5.5-sized batches - 200 tiles on 256 CUs run one workgroup per CU whatever the
// budget, so the cap would only cost spills (RT = 2: 46 - 59 spilled registers at 256) and a shallower load pipeline.
// Same instructions in the same order per output element: bit-identical results.
template <int RT, bool BF = false, int OCC = 2>
__global__ __launch_bounds__(EQD_BLOCK, OCC) void k_rowchain(EqdChainArg A_) {
    __shared__ __attribute__((aligned(16))) EqdChainArg A;      // job descriptions: kernarg segment -> LDS, once
    kernarg_to_lds(A, EQD_KERNARG_PTR(A_), 0);
    __shared__ LinSmem<RT> sm;
    __shared__ __attribute__((aligned(16))) float Lb[RT][LIN_LOCALS][16 * LIN_S];
    __shared__ float red[EQD_WAVES][256];
    float (*sm4)[EQD_WAVES][16] = (float (*)[EQD_WAVES][16])&red[0][0];     // chain_lnbwd64's exchange slots (4 x 4 x 16)
    const int row0 = (int)blockIdx.x * 16 * RT;
    EQD_TR_WG();
    EQD_TR(200);
    for (int i = threadIdx.x; i < RT * LIN_LOCALS * 16 * LIN_S; i += EQD_BLOCK) (&Lb[0][0][0])[i] = 0.f;
    __syncthreads();
    LinRegs<RT> RA;
    bool have = false;
    const int njobs = uni(A.njobs);
    constexpr int CJ_DW = (int)(sizeof(EqdChainJob) / 4);
    const int lane = threadIdx.x & 63;
    JobW Wc = jobw_load(&A.j[0], CJ_DW, lane);      // descriptor words of the current job (see JobW)
    // The LayerNorm-backward job's saved activations (one 16-byte vector per thread) are requested BEFORE the linear job in
    // front of it runs (DB5.5-sized batches, the 512-register instances): the job then starts on data that has landed
    // instead of with an exposed round trip to memory - one per backward chain.  Same loads, same bits.
    f32x4 yp_pre = f4zero();
    bool have_yp = false;
    for (int jj = 0; jj < njobs; ++jj) {
        const EqdChainJob& C = A.j[jj];
        const JobW Wn = jobw_load(&A.j[jj + 1 < njobs ? jj + 1 : jj], CJ_DW, lane);      // consumed a job later
        if (jw_i(Wc, JW_OFF(EqdChainJob, type)) == 0) {
            if constexpr (RT == 1 && OCC == 1) {
                if (jj + 1 < njobs && jw_i(Wn, JW_OFF(EqdChainJob, type)) != 0 && jw_i(Wn, JW_OFF(EqdLinJob, M)) == 64) {
#define LJ(f) JW_OFF(EqdLinJob, f)
                    const int rows_n = jw_i(Wn, LJ(rows)), ldx_n = jw_i(Wn, LJ(s) + JW_OFF(EqdLinSrc, ldx));
                    const int l15 = lane & 15, g = lane >> 4, wave = (int)threadIdx.x >> 6;
                    const int rowi = row0 + l15;
                    yp_pre = *(const EQD_GAS f4v*)(jw_p<const float>(Wn, LJ(s) + JW_OFF(EqdLinSrc, X)) +
                                                   (size_t)(rowi < rows_n ? rowi : rows_n - 1) * ldx_n + 16 * wave + 4 * g);
#undef LJ
                    have_yp = true;
                }
            }
            const int nj = jw_i(Wc, JW_OFF(EqdChainJob, prefetch_next));   // next linear job whose first step may be fetched early, or -1
            JobW Wp = Wn;
            if (nj >= 0 && nj != jj + 1) Wp = jobw_load(&A.j[nj], CJ_DW, lane);
            linear_tile<RT, BF, OCC == 1>(C.lin, Wc, true, C.src_local, jw_i(Wc, JW_OFF(EqdChainJob, out_local)), sm, Lb, row0, RA, have,
                            nj >= 0, Wp, 210 + 4 * jj);
            have = nj >= 0;
        } else {
            bool fast = false;
            if constexpr (RT == 1) fast = jw_i(Wc, JW_OFF(EqdLinJob, M)) == 64;
            if constexpr (RT == 1) {
                if (fast) chain_lnbwd64(Wc, Lb, sm4, row0, have_yp, yp_pre);
            }
            if (!fast) chain_lnbwd<RT>(C, Lb, red, row0);
            have_yp = false;
        }
        Wc = Wn;
        __syncthreads();
        EQD_TR(201 + jj);      // job boundaries (phase-trace experiments only)
    }
    EQD_TR_WG_END();
}

#ifdef EQD_TRACE
__device__ long long eqd_trace_buf[1024];
extern "C" int eqd_trace_fetch(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(eqd_trace_buf), sizeof(long long) * 1024);
}
#endif

// the 16-byte loaders need >= 4 elements along the contiguous axis and one unit weight stride
static int lin_check_sources(const EqdLinJob& J) {
    for (int s = 0; s < J.nsrc; ++s) {
        const EqdLinSrc& S = J.s[s];
        if (!S.W) continue;   // LayerNorm-backward jobs carry no weights
        if (S.K < 4 || (S.w_cs != 1 && S.w_rs != 1)) {
            eqd_set_error("eqd_linear: source %d has K=%d w_rs=%d w_cs=%d (need K >= 4 and one unit stride)", s, S.K, S.w_rs,
                          S.w_cs);
            return EQD_ERR_SHAPE;
        }
    }
    return EQD_OK;
}

// ---- k_rowwave (eqd_rowwave_inl.h): which chains it takes ---------------------------------------------------------
// Which kernel.  k_rowchain / k_linear: four waves per 16-row tile, a step's weights staged per workgroup; k_rowwave:
// one wave per tile, every wave loads its own weights; k_rowres: persistent 8-wave workgroups, weights resident in LDS.
// A CU pulls ~10-12 B/clock from L2 whatever the pattern (profiles/r02_exp_trace_rowwave_*.txt), so with about one tile
// per CU (config B: 200 tiles) the four waves that share a tile's weights finish first (B 6 337 pairs/s against 5 011
// with k_rowwave and 4 422 with k_rowres), and with many tiles per CU the kernel that fetches the weights once per CU
// does (C fp32 7 107 -> 7 329 k_rowwave -> 7 607 k_rowres, C bf16 9 971 -> 10 537 -> 11 390, E 551 -> 552 -> 562).
// Default: k_rowres from 3 tiles per CU, k_rowchain / k_linear below; EQD_ROWWAVE = 0 / 1 / 2 forces k_rowchain +
// k_linear / k_rowwave / k_rowres for every eligible chain (tests, experiments).
static int rw_mode(int rows);
// 1 when chains over `rows` rows run on k_rowres: a wave walks a chain's jobs one after the other there, so the driver
// appends the next layer's five projection jobs to the node-update chain (they read its result from the LDS tile)
// instead of launching them on their own - below that size they run side by side on k_linear, which is faster
int eqd_rows_resident(int rows) { return rows > 0 && rw_mode(rows) == 2; }
static int rw_mode(int rows) {
    const char* f = eqd_tunable("EQD_ROWWAVE");
    if (f && f[0] >= '0' && f[0] <= '2' && f[1] == 0) return f[0] - '0';
    return (rows + 15) / 16 >= 3 * eqd_num_cus() ? 2 : 0;
}
static bool rw_eligible(const EqdChainJob* jobs, int njobs, int rows) {
    if (njobs <= 0 || rows <= 0 || rw_mode(rows) == 0) return false;
    for (int i = 0; i < njobs; ++i) {
        const EqdChainJob& C = jobs[i];
        const EqdLinJob& J = C.lin;
        if (J.M != 64 || J.rows != rows || C.out_local >= LIN_LOCALS) return false;
        if ((J.bf16 != 0) != (jobs[0].lin.bf16 != 0)) return false;
        // (no alignment condition: rows are read with 16-byte loads at 4-byte alignment - gfx950 runs global memory in
        // unaligned-access mode - which is what the 69-wide h0 rows need anyway)
        if (C.type != 0) {
            if (C.src_local[0] < 0 || C.src_local[0] >= LIN_LOCALS || !J.s[0].X || !J.ln_g || !C.aux) return false;
            continue;
        }
        if (J.nsrc <= 0 || J.nsrc > EQD_MAX_SRC) return false;
        if (J.ln_g && !J.ln_b) return false;
        const bool tp = J.s[0].w_cs != 1;
        for (int s = 0; s < J.nsrc; ++s) {
            const EqdLinSrc& S = J.s[s];
            const bool local = C.src_local[s] >= 0;
            if (!S.W || (S.w_cs != 1) != tp) return false;
            if (tp ? S.w_rs != 1 : S.w_cs != 1) return false;
            if (local && (C.src_local[s] >= LIN_LOCALS || S.K != 64 || S.mask)) return false;
            if (!local && !S.X) return false;
            if (S.K == 64) continue;
            if (S.K < 64 || S.K > 80 || local || S.mask) return false;
        }
    }
    // the kernels fetch a job's rows ahead (k_rowres: across job boundaries, and it keeps rows that several jobs share):
    // no global source may be data that an earlier job of the chain writes - such chains pass it on as an LDS tile
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].type != 0) continue;
        for (int s = 0; s < jobs[i].lin.nsrc; ++s) {
            if (jobs[i].src_local[s] >= 0) continue;
            const EqdLinSrc& S0 = jobs[i].lin.s[s];
            for (int m = 0; m < i; ++m) {
                const float* outs[2] = {jobs[m].lin.Y, jobs[m].lin.pre_ln};
                for (int q = 0; q < 2; ++q)
                    if (outs[q] && (outs[q] == S0.X || outs[q] == S0.mask)) return false;
            }
        }
    }
    return true;
}
// k_rowres80 (eqd_rowres80_inl.h): the 69-wide first layer's FORWARD jobs in bf16 mode at sizes where the row kernels keep
// their weights in LDS.  EQD_ROWRES80=0 keeps them on the four-wave kernels (A/B runs, tests of both forms).
int eqd_rowres80_on() {
    const char* f = eqd_tunable("EQD_ROWRES80");
    return !(f && f[0] == '0' && f[1] == 0);
}
static bool rw80_eligible(const EqdChainJob* jobs, int njobs, int rows) {
    if (njobs <= 0 || rows <= 0 || rw_mode(rows) != 2 || !eqd_rowres80_on()) return false;
    bool wide = false;
    if (jobs[0].type != 0) return false;      // (the kernel stages the first LINEAR job's weights in its prologue)
    for (int i = 0; i < njobs; ++i) {
        const EqdChainJob& C = jobs[i];
        const EqdLinJob& J = C.lin;
        if (!J.bf16 || J.rows != rows || J.M < 4 || J.M > 80 || C.out_local >= LIN_LOCALS) return false;
        wide = wide || J.M > 64;
        if (C.type != 0) {      // LeakyReLU -> LayerNorm backward: the wave's tile in, the wave's tile out
            if (C.src_local[0] < 0 || C.src_local[0] >= LIN_LOCALS || C.out_local < 0 || !J.s[0].X || !J.ln_g || !C.aux) return false;
            continue;
        }
        if (J.nsrc <= 0 || J.nsrc > EQD_MAX_SRC || (J.ln_g && !J.ln_b)) return false;
        if (J.Yb && ((J.ldyb & 3) != 0 || (((uintptr_t)J.Yb) & 7) != 0)) return false;
        for (int s = 0; s < J.nsrc; ++s) {
            const EqdLinSrc& S = J.s[s];
            const bool local = C.src_local[s] >= 0;
            if (!S.W || S.mask) return false;                       // no masked sources (none in the first layer's chains)
            if (S.w_cs != 1 && S.w_rs != 1) return false;           // one unit stride: either orientation is restaged as [m][k]
            if (local ? (C.src_local[s] >= LIN_LOCALS || S.K < 64 || S.K > 80) : (!S.X || S.K < 64 || S.K > 80)) return false;
            wide = wide || S.K > 64;
        }
    }
    if (!wide) return false;      // (plain 64-wide chains: k_rowres)
    // no global source that an earlier job of the chain writes (rows are fetched ahead), one live intermediate tile
    for (int i = 0; i < njobs; ++i)
        for (int s = 0; s < jobs[i].lin.nsrc; ++s) {
            if (jobs[i].src_local[s] >= 0) continue;
            for (int m = 0; m < i; ++m) {
                const float* outs[2] = {jobs[m].lin.Y, jobs[m].lin.pre_ln};
                for (int q = 0; q < 2; ++q)
                    if (outs[q] && outs[q] == jobs[i].lin.s[s].X) return false;
            }
        }
    int cur = -1;
    for (int i = 0; i < njobs; ++i) {
        const int ns = jobs[i].type != 0 ? 1 : jobs[i].lin.nsrc;
        for (int s = 0; s < ns; ++s)
            if (jobs[i].src_local[s] >= 0 && jobs[i].src_local[s] != cur) return false;
        if (jobs[i].out_local >= 0) cur = jobs[i].out_local;
    }
    return true;
}
static void chain_links(EqdChainArg& arg, const EqdChainJob* jobs, int njobs) {
    // prefetch_next[i]: the next linear job n whose first step may be loaded before job i's epilogue: its first
    // source is an LDS tile, or global data that none of the jobs i .. n-1 writes
    for (int i = 0; i < njobs; ++i) {
        arg.j[i].prefetch_next = -1;
        arg.j[i].next_lin = -1;
        if (jobs[i].type != 0) continue;
        int n = i + 1;
        while (n < njobs && jobs[n].type != 0) ++n;
        if (n >= njobs) continue;
        arg.j[i].next_lin = n;
        bool ok = true;
        if (jobs[n].src_local[0] < 0) {
            const EqdLinSrc& S0 = jobs[n].lin.s[0];
            for (int m = i; m < n && ok; ++m) {
                const float* outs[2] = {jobs[m].lin.Y, jobs[m].lin.pre_ln};
                for (int q = 0; q < 2; ++q)
                    if (outs[q] && (outs[q] == S0.X || outs[q] == S0.mask)) ok = false;
            }
        }
        if (ok) arg.j[i].prefetch_next = n;
    }
}
// k_rowres keeps ONE intermediate tile per (wave, tile slot): every LDS-resident source must be the tile written last
static bool rr_single_tile(const EqdChainJob* jobs, int njobs) {
    int cur = -1;
    for (int i = 0; i < njobs; ++i) {
        const EqdChainJob& C = jobs[i];
        const int ns = C.type != 0 ? 1 : C.lin.nsrc;
        for (int s = 0; s < ns; ++s)
            if (C.src_local[s] >= 0 && C.src_local[s] != cur) return false;
        if (C.out_local >= 0) cur = C.out_local;
    }
    return true;
}
static int rr_tiles_per_wg(int rows) {
    const char* f = eqd_tunable("EQD_ROWRES_TPS");      // tests: force the tiles per workgroup (1..16) to reach the two-slot paths
    if (f && f[0]) {
        const int v = atoi(f);
        if (v >= 1 && v <= RR_WAVES * RR_TMAX) return v;
    }
    const int nt = (rows + 15) / 16, cus = eqd_num_cus();
    int tps = (nt + cus - 1) / cus;
    tps = tps < 1 ? 1 : tps;
    return tps > RR_WAVES * RR_TMAX ? RR_WAVES * RR_TMAX : tps;
}
static int rr_blocks(int rows) {
    const int tps = rr_tiles_per_wg(rows);
    return ((rows + 15) / 16 + tps - 1) / tps;
}
static int rw_blocks(int rows) { return ((rows + 15) / 16 + RW_WAVES - 1) / RW_WAVES; }
// workgroups (= LayerNorm-backward partial rows) of the launch rw_launch would make
static int rw_launch_blocks(const EqdChainJob* jobs, int njobs, int rows) {
    return (rw_mode(rows) == 2 && rr_single_tile(jobs, njobs)) ? rr_blocks(rows) : rw_blocks(rows);
}
static int launch_rowwave(const EqdChainArg& arg, int rows, bool bf, hipStream_t st) {
    if (rw_mode(rows) == 2 && rr_single_tile(arg.j, arg.njobs)) {
        const int tps = rr_tiles_per_wg(rows);
        if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowres<true>), dim3(rr_blocks(rows)), dim3(64 * RR_WAVES), 0, st, arg, tps);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowres<false>), dim3(rr_blocks(rows)), dim3(64 * RR_WAVES), 0, st, arg, tps);
        return eqd_check_launch("k_rowres");
    }
    if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowwave<true>), dim3(rw_blocks(rows)), dim3(64 * RW_WAVES), 0, st, arg);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowwave<false>), dim3(rw_blocks(rows)), dim3(64 * RW_WAVES), 0, st, arg);
    return eqd_check_launch("k_rowwave");
}

int eqd_launch_rowchain(const EqdChainJob* jobs, int njobs, int rows, hipStream_t st, int* partial_rows) {
    if (njobs <= 0 || njobs > EQD_CHAIN_MAXJOBS) {
        eqd_set_error("eqd_launch_rowchain: %d jobs (1..%d)", njobs, EQD_CHAIN_MAXJOBS);
        return EQD_ERR_SHAPE;
    }
    if (rows <= 0) return EQD_OK;
    EqdChainArg arg;
    memset(&arg, 0, sizeof(arg));
    for (int i = 0; i < njobs; ++i) {
        arg.j[i] = jobs[i];
        const EqdLinJob& J = jobs[i].lin;
        if (J.M < 4 || J.M > 80 || J.rows != rows) {
            eqd_set_error("eqd_launch_rowchain: job %d has M=%d rows=%d (chain rows %d)", i, J.M, J.rows, rows);
            return EQD_ERR_SHAPE;
        }
        if (jobs[i].type == 0)
            if (int e = lin_check_sources(J)) return e;
        for (int s = 0; s < J.nsrc; ++s)
            if (jobs[i].src_local[s] >= 0 && (J.s[s].K > 80 || jobs[i].src_local[s] >= LIN_LOCALS)) {
                eqd_set_error("eqd_launch_rowchain: LDS-resident source wider than %d or tile index >= %d", 80, LIN_LOCALS);
                return EQD_ERR_SHAPE;
            }
    }
    arg.njobs = njobs;
    chain_links(arg, jobs, njobs);
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].out_local >= LIN_LOCALS) {
            eqd_set_error("eqd_launch_rowchain: LDS tile index %d >= %d", jobs[i].out_local, LIN_LOCALS);
            return EQD_ERR_SHAPE;
        }
    const bool bf = njobs > 0 && jobs[0].lin.bf16;     // one arithmetic mode per launch
    if (!rw_eligible(jobs, njobs, rows) && rw80_eligible(jobs, njobs, rows)) {      // the first layer's forward chain, bf16
        if (partial_rows) *partial_rows = rr_blocks(rows);      // (LayerNorm-backward partial sums: one row per workgroup)
        const int tps = rr_tiles_per_wg(rows);
        hipLaunchKernelGGL(k_rowres80, dim3(rr_blocks(rows)), dim3(64 * RR_WAVES), 0, st, arg, tps);
        return eqd_check_launch("k_rowres");
    }
    if (rw_eligible(jobs, njobs, rows)) {
        if (partial_rows) *partial_rows = rw_launch_blocks(jobs, njobs, rows);
        return launch_rowwave(arg, rows, bf, st);
    }
    if (partial_rows) *partial_rows = eqd_rowchain_blocks(rows);
    if (eqd_row_tiles(rows) == 2) {
        // (the two-tile forms need more than 256 registers: one workgroup per CU at every size)
        if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowchain<2, true, 1>), dim3(eqd_rowchain_blocks(rows)), dim3(EQD_BLOCK), 0, st, arg);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowchain<2, false, 1>), dim3(eqd_rowchain_blocks(rows)), dim3(EQD_BLOCK), 0, st, arg);
    } else {
        const char* oc = eqd_tunable("EQD_ROWCHAIN_OCC");      // experiments: 1 / 2 forces a register budget
        const bool one = oc && (oc[0] == '1' || oc[0] == '2') && oc[1] == 0 ? oc[0] == '1' : eqd_rowchain_blocks(rows) <= eqd_num_cus();
        if (bf && one) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowchain<1, true, 1>), dim3(eqd_rowchain_blocks(rows)), dim3(EQD_BLOCK), 0, st, arg);
        else if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowchain<1, true, 2>), dim3(eqd_rowchain_blocks(rows)), dim3(EQD_BLOCK), 0, st, arg);
        else if (one) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowchain<1, false, 1>), dim3(eqd_rowchain_blocks(rows)), dim3(EQD_BLOCK), 0, st, arg);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rowchain<1, false, 2>), dim3(eqd_rowchain_blocks(rows)), dim3(EQD_BLOCK), 0, st, arg);
    }
    return eqd_check_launch("k_rowchain");
}

extern "C" int eqd_linear(const EqdLinJob* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0) {
        eqd_set_error("eqd_linear: no jobs");
        return EQD_ERR_NULL;
    }
    hipStream_t st = (hipStream_t)stream;
    // jobs on the same rows that k_rowwave takes: one launch, a wave walks the jobs of its 16 rows one after the other
    // (k_linear runs them side by side on four-wave workgroups; per row tile the same MFMA work without the barriers)
    if (njobs <= EQD_CHAIN_MAXJOBS) {
        EqdChainArg carg;
        memset(&carg, 0, sizeof(carg));
        bool ok = true;
        for (int i = 0; i < njobs && ok; ++i) {
            EqdChainJob& C = carg.j[i];
            C.lin = jobs[i];
            for (int s = 0; s < EQD_MAX_SRC; ++s) C.src_local[s] = -1;
            C.out_local = -1;
            ok = (jobs[i].Y != nullptr || jobs[i].Yb != nullptr) && jobs[i].rows == jobs[0].rows;
        }
        if (ok && !rw_eligible(carg.j, njobs, jobs[0].rows) && rw80_eligible(carg.j, njobs, jobs[0].rows)) {
            for (int i = 0; i < njobs; ++i)
                if (int e = lin_check_sources(jobs[i])) return e;
            carg.njobs = njobs;
            const int rows80 = jobs[0].rows, tps = rr_tiles_per_wg(rows80);
            hipLaunchKernelGGL(k_rowres80, dim3(rr_blocks(rows80)), dim3(64 * RR_WAVES), 0, st, carg, tps);
            return eqd_check_launch("k_rowres");
        }
        if (ok && rw_eligible(carg.j, njobs, jobs[0].rows)) {
            for (int i = 0; i < njobs; ++i)
                if (int e = lin_check_sources(jobs[i])) return e;
            carg.njobs = njobs;
            EqdChainJob tmp[EQD_CHAIN_MAXJOBS];
            for (int i = 0; i < njobs; ++i) tmp[i] = carg.j[i];
            chain_links(carg, tmp, njobs);
            return launch_rowwave(carg, jobs[0].rows, jobs[0].bf16 != 0, st);
        }
    }
    for (int base = 0; base < njobs; base += LIN_MAXJOBS) {
        LinJobsArg arg;
        memset(&arg, 0, sizeof(arg));
        int n = njobs - base < LIN_MAXJOBS ? njobs - base : LIN_MAXJOBS;
        int maxrows = 0;
        for (int i = 0; i < n; ++i) {
            const EqdLinJob& J = jobs[base + i];
            if (J.M < 4 || J.M > 80 || J.nsrc <= 0 || J.nsrc > EQD_MAX_SRC || (!J.Y && !J.Yb)) {
                eqd_set_error("eqd_linear: job %d has M=%d nsrc=%d (need 4..80 outputs, 1..%d sources)", base + i, J.M,
                              J.nsrc, EQD_MAX_SRC);
                return EQD_ERR_SHAPE;
            }
            if (int e = lin_check_sources(J)) return e;
            if (J.ln_g && !J.ln_b) {
                eqd_set_error("eqd_linear: job %d has LayerNorm weight without bias", base + i);
                return EQD_ERR_NULL;
            }
            arg.j[i] = J;
            if (J.rows > maxrows) maxrows = J.rows;
        }
        if (maxrows == 0) continue;
        const int rt = linear_row_tiles(maxrows);
        dim3 grid((maxrows + 16 * rt - 1) / (16 * rt), n);
        const bool bf = jobs[base].bf16 != 0;       // one arithmetic mode per launch
        const int s80 = lin_simple80_mode();
        if (rt == 1 && lin_simple_eligible(arg.j, n)) {      // plain 64 x 64 projections: the small body, one round of workgroups
            if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear_simple<true>), grid, dim3(EQD_BLOCK), 0, st, arg);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear_simple<false>), grid, dim3(EQD_BLOCK), 0, st, arg);
        } else if (s80 >= (rt == 1 ? 1 : 2) && lin_simple80_eligible(arg.j, n)) {      // the first layer's 69-wide projections
            const dim3 grid16((maxrows + 15) / 16, n);
            if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear_simple80<true>), grid16, dim3(EQD_BLOCK), 0, st, arg);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear_simple80<false>), grid16, dim3(EQD_BLOCK), 0, st, arg);
        } else if (rt == 2) {
            if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<2, true>), grid, dim3(EQD_BLOCK), 0, st, arg);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<2, false>), grid, dim3(EQD_BLOCK), 0, st, arg);
        } else {
            if (bf) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<1, true>), grid, dim3(EQD_BLOCK), 0, st, arg);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<1, false>), grid, dim3(EQD_BLOCK), 0, st, arg);
        }
        int rc = eqd_check_launch("k_linear");
        if (rc) return rc;
    }
    return EQD_OK;
}

// ------------------------------------------------------------------------------------------
// k_atb: partial[unit][chunk][m][n] = sum_{rows in chunk} Xm[row][m] * Y[row][n0 + n]
//   wave tile: 80 (M axis, output rows m) x 64 (N axis, output cols n), K axis = graph rows.
//   then k_atb_reduce sums the chunks in a fixed order and accumulates into the gradient.
// ------------------------------------------------------------------------------------------
#define ATB_MAXUNITS 128  /* 128 x 112 B = 14 KB of kernel arguments: every weight-gradient GEMM of an 8-layer backward pass
                             (~100 units) in ONE launch + ONE reduction (36 units = 4 KB meant three of each: -25 us per
                             config-B step).  HIP passes kernel arguments of this size (checked with 12 KB on gfx950) */
#define ATB_TILE 5120  /* 80 x 64 */
#define ATB_PSTRIDE 5200
struct AtbUnit {
    EqdAtbJob job;
    int n0, nparts, nchunks;
    int fast;        // 0 general; 1 / 2: M == 64 with aligned X rows, Y block whole and aligned / through ld4u (atb_fast)
    long long poff;  // float offset of this unit's partials
};
static_assert(sizeof(AtbUnit) * ATB_MAXUNITS <= 16384, "AtbUnitsArg: keep the kernel-argument segment moderate");
struct AtbUnitsArg {
    AtbUnit u[ATB_MAXUNITS];
};

#define ATB_ROWS 64    /* graph rows per chunk */
#define ATB_LS 80      /* LDS row stride: 20 x 16 B; a fragment read (4 rows x 16 columns) covers the 64 banks once */
#define ATB_MAXBLOCKS 256
// Persistent workgroups: each walks the 64-row chunks c, c + nparts, ... of its unit.  Per chunk the
// X (<= 80 columns) and Y (64 columns) slabs are fetched with all loads in flight at once (36 per
// thread, 64-byte coalesced row segments) WHILE the previous chunk is multiplied, then written to LDS;
// wave w accumulates the output column block nb = w for every row block mb.  One partial tile per
// workgroup (no cross-wave reduction), summed later in a fixed order by k_atb_reduce.
// Four bf16 of a saved tensor (EqdAtbJob.y_bf16), split like ld4u_raw / ld4u_fix so that a batch of loads is issued
// without anything waiting on it: ld4_bf16_raw is ONE 8-byte load (p 8-byte aligned; n = valid elements at p, <= 0: the load
// goes to `safe`) whose two dwords are parked in the first two lanes of an f32x4; ld4_bf16_fix converts when the data is
// consumed (exact values: rounding them again when the MFMA operand is formed gives the same bits), zero beyond n.
// v_perm_b32: byte i of the result = byte sel[i] of the 8 bytes {hi : lo} (selector 0..3 = bytes of lo, 4..7 = bytes of hi)
__device__ __forceinline__ unsigned eqd_perm_b32(unsigned hi, unsigned lo, unsigned sel) {
#ifdef EQD_HOSTSIM
    const unsigned long long both = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((both >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
    return r;
#else
    return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}
__device__ __forceinline__ f32x4 ld4_bf16_raw(const unsigned short* __restrict__ p, int n, const unsigned short* __restrict__ safe) {
    const unsigned long long h = *(const EQD_GAS unsigned long long*)(n > 0 ? p : safe);      // (one global_load_dwordx2)
    f32x4 r;
    r[0] = __builtin_bit_cast(float, (unsigned)h);
    r[1] = __builtin_bit_cast(float, (unsigned)(h >> 32));
    r[2] = 0.f;
    r[3] = 0.f;
    return r;
}
__device__ __forceinline__ f32x4 ld4_bf16_fix(f32x4 raw, int n) {
    // (through scalar temporaries: __builtin_bit_cast applied to `raw[1]` directly returns element 0 with this clang)
    const float f0 = raw[0], f1 = raw[1];
    const unsigned lo = __builtin_bit_cast(unsigned, f0), hi = __builtin_bit_cast(unsigned, f1);
    f32x4 r;
    r[0] = n > 0 ? __builtin_bit_cast(float, lo << 16) : 0.f;
    r[1] = n > 1 ? __builtin_bit_cast(float, lo & 0xffff0000u) : 0.f;
    r[2] = n > 2 ? __builtin_bit_cast(float, hi << 16) : 0.f;
    r[3] = n > 3 ? __builtin_bit_cast(float, hi & 0xffff0000u) : 0.f;
    return r;
}
struct AtbRegs {
    f32x4 x[4][2], xm[4][2], y[4];   // raw 16-byte loads (see ld4u_raw / ld4u_fix)
};
__device__ __forceinline__ int atb_nx(const EqdAtbJob& J, int chunk, int t, int jr, int h) {
    const int row = chunk * ATB_ROWS + (t >> 4) + 16 * jr, tc = t & 15;
    return (row < J.rows && (h == 0 || tc < 4)) ? J.M - (4 * tc + 64 * h) : 0;
}
__device__ __forceinline__ int atb_ny(const EqdAtbJob& J, int n0, int chunk, int t, int jr) {
    const int row = chunk * ATB_ROWS + (t >> 4) + 16 * jr;
    return row < J.rows ? J.N - (n0 + 4 * (t & 15)) : 0;
}
__device__ __forceinline__ void atb_load(const EqdAtbJob& J, int n0, int chunk, int t, AtbRegs& R) {
    const int tr = t >> 4, tc = t & 15;
    const int r0 = chunk * ATB_ROWS;
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int row = r0 + tr + 16 * jr;
        const size_t ro = (size_t)(row < J.rows ? row : 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = atb_nx(J, chunk, t, jr, h);
            const size_t o = ro * J.ldx + 4 * tc + 64 * h;
            R.x[jr][h] = ld4u_raw(J.X + o, n, J.X);
            if (J.xmask) R.xm[jr][h] = ld4u_raw(J.xmask + o, n, J.xmask);
        }
        if (J.y_bf16)
            R.y[jr] = ld4_bf16_raw((const unsigned short*)J.Y + ro * J.ldy + n0 + 4 * tc, atb_ny(J, n0, chunk, t, jr),
                                   (const unsigned short*)J.Y);
        else
            R.y[jr] = ld4u_raw(J.Y + ro * J.ldy + n0 + 4 * tc, atb_ny(J, n0, chunk, t, jr), J.Y);
    }
}
__device__ __forceinline__ void atb_store(const EqdAtbJob& J, int n0, int chunk, int t, const AtbRegs& R,
                                          float* __restrict__ Xl, float* __restrict__ Yl) {
    const int tr = t >> 4, tc = t & 15;
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int row = tr + 16 * jr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && tc >= 4) continue;
            const int n = atb_nx(J, chunk, t, jr, h);
            float4 v = ld4u_fix(R.x[jr][h], n);
            if (J.xmask) {
                const float4 mk = ld4u_fix(R.xm[jr][h], n);
                v.x *= lrelu_grad(mk.x, J.slope); v.y *= lrelu_grad(mk.y, J.slope);
                v.z *= lrelu_grad(mk.z, J.slope); v.w *= lrelu_grad(mk.w, J.slope);
            }
            *(float4*)&Xl[row * ATB_LS + 4 * tc + 64 * h] = v;
        }
        if (J.y_bf16) *(f32x4*)&Yl[row * ATB_LS + 4 * tc] = ld4_bf16_fix(R.y[jr], atb_ny(J, n0, chunk, t, jr));
        else *(float4*)&Yl[row * ATB_LS + 4 * tc] = ld4u_fix(R.y[jr], atb_ny(J, n0, chunk, t, jr));
    }
}
// one 64-row chunk: acc[mb] += X[:, 16 mb ..]^T Y[:, 16 wave ..]; MBN row blocks, no predicates in the loop
// (4 k-steps per trip: 4 (1 + MBN) LDS reads in flight, then 4 MBN MFMAs)
template <int MBN, bool BF = false>
__device__ __forceinline__ void atb_mma(f32x4 (&acc)[5], const float* __restrict__ Xl, const float* __restrict__ Yl,
                                        int wave, int l15, int g) {
    if constexpr (BF) {      // bf16 mode: eight k-steps (rows 4 (ks + u) + g) as one 32-deep v_mfma_f32_16x16x32_bf16
        static_assert((ATB_ROWS / 4) % 8 == 0, "whole 32-deep chunks");
        for (int ks = 0; ks < ATB_ROWS / 4; ks += 8) {
            float b[8], a[8][MBN];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = 4 * (ks + u) + g;
                b[u] = Yl[row * ATB_LS + 16 * wave + l15];
#pragma unroll
                for (int mb = 0; mb < MBN; ++mb) a[u][mb] = Xl[row * ATB_LS + 16 * mb + l15];
            }
            const s16x8 bp = cat_bf(pack_bf4(b[0], b[1], b[2], b[3]), pack_bf4(b[4], b[5], b[6], b[7]));
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb)
                acc[mb] = mfma_bf32(cat_bf(pack_bf4(a[0][mb], a[1][mb], a[2][mb], a[3][mb]), pack_bf4(a[4][mb], a[5][mb], a[6][mb], a[7][mb])),
                                    bp, acc[mb]);
        }
        return;
    }
    for (int ks = 0; ks < ATB_ROWS / 4; ks += 4) {
        float b[4], a[4][MBN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 4 * (ks + u) + g;
            b[u] = Yl[row * ATB_LS + 16 * wave + l15];
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb) a[u][mb] = Xl[row * ATB_LS + 16 * mb + l15];
        }
        if constexpr (BF) {
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int mb = 0; mb < MBN; ++mb) acc[mb] = mfma4(a[u][mb], b[u], acc[mb]);
        }
    }
}

// The common shape (64 x 64 block, aligned): 12 plain 16-byte loads per thread and chunk (rows beyond the matrix are
// clamped and zeroed when written to LDS) instead of the general path's 20 shifted / fixed-up ones, 8 of which fetch
// nothing when M <= 64 - the general path is bound by its ~800 VALU instructions per chunk, not by memory (its time
// does not change between 256 and 1 536 workgroups).  The column sums (bias gradients) are taken by all four waves,
// 16 rows each, and only when a bias gradient is wanted.
template <bool MASKED>      // (a template flag, not a run-time one: see atb_fast_bf)
__device__ __forceinline__ void atb_fast(const AtbUnit& u, int c, float* __restrict__ partial, float* __restrict__ Xl,
                                         float* __restrict__ Yl) {
    const EqdAtbJob& J = u.job;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tr = t >> 4, tc = t & 15;
    const int rows = J.rows, ldx = J.ldx, ldy = J.ldy, nparts = u.nparts, nchunks = u.nchunks;
    constexpr bool masked = MASKED;
    const bool want_bias = J.bias_out != nullptr && u.n0 == 0;
    const float slope = J.slope;
    const EQD_GAS float* const X = (const EQD_GAS float*)J.X + 4 * tc;
    const EQD_GAS float* const Xm = (const EQD_GAS float*)(masked ? J.xmask : J.X) + 4 * tc;
    const EQD_GAS float* const Y = (const EQD_GAS float*)J.Y + u.n0 + 4 * tc;
    const int ny = J.N - (u.n0 + 4 * tc);      // columns of Y left at this thread's vector (a narrow last block: < 4, <= 0)
    const bool yfull = u.fast == 1;            // 1: a whole 64-column block of 16-byte aligned rows
    f32x4 acc[5];
#pragma unroll
    for (int mb = 0; mb < 5; ++mb) acc[mb] = f4zero();
    float bacc = 0.f;
    f32x4 rx[4], rm[4], ry[4];
    auto load = [&](int chunk) {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            int row = chunk * ATB_ROWS + tr + 16 * jr;
            row = row < rows ? row : rows - 1;
            rx[jr] = *(const EQD_GAS f4v*)(X + (size_t)row * ldx);
            if constexpr (masked) rm[jr] = *(const EQD_GAS f4v*)(Xm + (size_t)row * ldx);
            ry[jr] = yfull ? *(const EQD_GAS f4v*)(Y + (size_t)row * ldy)
                           : ld4u_raw((const float*)(Y + (size_t)row * ldy), ny, J.Y);
        }
    };
    load(c);
    for (int chunk = c; chunk < nchunks; chunk += nparts) {
        __syncthreads();             // the previous chunk's LDS reads are done
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const bool rvalid = chunk * ATB_ROWS + tr + 16 * jr < rows;
            f32x4 v = rx[jr], y = ry[jr];
            if (!yfull) {
                const float4 f = ld4u_fix(ry[jr], ny);
                y = f32x4{f.x, f.y, f.z, f.w};
            }
            if constexpr (masked) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(rm[jr][i], slope);
            }
            if (!rvalid) {
                v = f4zero();
                y = f4zero();
            }
            *(f32x4*)&Xl[(tr + 16 * jr) * ATB_LS + 4 * tc] = v;
            *(f32x4*)&Yl[(tr + 16 * jr) * ATB_LS + 4 * tc] = y;
        }
        __syncthreads();
        if (chunk + nparts < nchunks) load(chunk + nparts);
        if (J.bf16) atb_mma<4, true>(acc, Xl, Yl, wave, l15, g);
        else atb_mma<4, false>(acc, Xl, Yl, wave, l15, g);
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc += Xl[(16 * wave + i) * ATB_LS + lane];
        }
    }
    float* P = partial + u.poff + (long long)c * ATB_PSTRIDE;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * mb + 4 * g + r) * 64 + 16 * wave + l15] = acc[mb][r];
    if (want_bias) {                 // (without one the bias slots of the partial are never read)
        __syncthreads();
        Yl[64 * wave + lane] = bacc;
        __syncthreads();
        if (t < 64) P[ATB_TILE + t] = (Yl[t] + Yl[64 + t]) + (Yl[128 + t] + Yl[192 + t]);
    }
}

// atb_fast in bf16 mode: the chunk's operands are rounded ONCE, when they are written to LDS, and stored in the layout the
// bf16 MFMA reads - [column][k-group] of four bf16, the four rows of a group being the ones the loader thread holds anyway
// (rows tr, tr + 16, tr + 32, tr + 48 of the chunk: the order of a contraction is free as long as X and Y agree), so a thread
// packs its 4 x 4 block column by column and an MFMA operand is one ds_read_b64.  The fp32 tiles cost 20 ds_read_b32 + 5
// packs per 4 MFMAs (the LDS pipe, not the MFMA, set the kernel's time: MfmaUtil 7 %); at 8.7 KB per operand two chunks fit the
// same LDS, so the "previous chunk's reads are done" barrier goes as well (a chunk's stores only have to wait for the reads
// of two chunks back, which are behind the barrier in between).  Column sums (bias gradients) are taken from the fp32
// registers, as before from the fp32 tile.
#define ATB_KG 17      /* k-groups per column row: 16 + 1 (34 dwords: the 16 lanes of a b64 read phase hit distinct banks) */
// YBF: Y is a saved bf16 tensor (EqdAtbJob.y_bf16) - a template parameter, not a run-time branch: with the two load forms in
// one loop body the compiler's wait-count pass put an s_waitcnt vmcnt(0) behind every Y load at the branch merges (the loads
// of the next chunk, issued a whole chunk ahead, then ran one round trip after the other: 321 -> 376 us per pass)
// MASKED (X multiplied by LeakyReLU'(xmask): one unit per backward pass, the head's mlp_h_mean_ROT) likewise: as a run-time
// flag the mask rows were loaded for every unit (the compiler turned `if (masked) load` into an unconditional load of X again)
template <bool YBF, bool MASKED>
__device__ __forceinline__ void atb_fast_bf(const AtbUnit& u, int c, float* __restrict__ partial, float* __restrict__ Xl_,
                                            float* __restrict__ Yl_) {
    const EqdAtbJob& J = u.job;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tr = t >> 4, tc = t & 15;
    const int rows = J.rows, ldx = J.ldx, ldy = J.ldy, nparts = u.nparts, nchunks = u.nchunks;
    constexpr bool masked = MASKED;
    const bool want_bias = J.bias_out != nullptr && u.n0 == 0;
    const float slope = J.slope;
    const EQD_GAS float* const Y = (const EQD_GAS float*)J.Y + u.n0 + 4 * tc;
    const int ny = J.N - (u.n0 + 4 * tc);
    const bool yfull = u.fast == 1;
    constexpr bool ybf = YBF;            // Y is a saved bf16 tensor (uint16 rows): 8-byte loads
    static_assert(2 * 64 * ATB_KG * 8 <= ATB_ROWS * ATB_LS * 4, "two bf16 chunks must fit one fp32 tile");
    s16x4* const Xb = (s16x4*)Xl_;      // [2][64 columns][ATB_KG]
    s16x4* const Yb = (s16x4*)Yl_;
    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f4zero();
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 rx[4], rm[4], ry[4];
    unsigned long long ryh[4] = {0ull, 0ull, 0ull, 0ull};      // raw rows of a saved bf16 Y (whole column blocks)
    auto load = [&](int chunk) {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            int row = chunk * ATB_ROWS + tr + 16 * jr;
            row = row < rows ? row : rows - 1;
            // (wave-uniform base + 32-bit per-lane byte offset: no 64-bit address arithmetic per load; the host takes this
            //  body only for operands below 2 GB, atb_units)
            const unsigned xo = 4u * (unsigned)(row * ldx + 4 * tc);
            rx[jr] = *(const EQD_GAS f4v*)((const char*)J.X + xo);
            if constexpr (masked) rm[jr] = *(const EQD_GAS f4v*)((const char*)J.xmask + xo);
            if constexpr (ybf) {
                // ONE unconditional 8-byte load into registers nothing else writes (a ragged last block: the address is
                // clamped like ld4u_raw's; the columns beyond the matrix are masked when the block is converted)
                const unsigned yo = ny > 0 ? 2u * (unsigned)(row * ldy + u.n0 + 4 * tc) : 0u;
                ryh[jr] = *(const EQD_GAS unsigned long long*)((const char*)J.Y + yo);
            } else {
                ry[jr] = yfull ? *(const EQD_GAS f4v*)(Y + (size_t)row * ldy)
                               : ld4u_raw((const float*)(Y + (size_t)row * ldy), ny, J.Y);
            }
        }
    };
    load(c);
    int buf = 0;
    for (int chunk = c; chunk < nchunks; chunk += nparts) {
        f32x4 xv[4], yv[4];
        const bool tail = (chunk + 1) * ATB_ROWS > rows;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const bool rvalid = chunk * ATB_ROWS + tr + 16 * jr < rows;
            f32x4 v = rx[jr], y = ry[jr];
            if constexpr (ybf) {
                if (!yfull) {      // (a ragged last column block: converted, packed below)
                    f32x4 raw = f4zero();
                    raw[0] = __builtin_bit_cast(float, (unsigned)ryh[jr]);
                    raw[1] = __builtin_bit_cast(float, (unsigned)(ryh[jr] >> 32));
                    y = ld4_bf16_fix(raw, ny);
                }
            } else if (!yfull) {
                const float4 f = ld4u_fix(ry[jr], ny);
                y = f32x4{f.x, f.y, f.z, f.w};
            }
            if constexpr (masked) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(rm[jr][i], slope);
            }
            if (tail && !rvalid) {      // (only the matrix's last chunk has rows beyond it: a uniform test everywhere else)
                v = f4zero();
                y = f4zero();
            }
            xv[jr] = v;
            yv[jr] = y;
        }
        s16x4* const Xc = Xb + buf * 64 * ATB_KG;
        s16x4* const Yc = Yb + buf * 64 * ATB_KG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {      // column 4 tc + i, k-group tr = this thread's four rows
            Xc[(4 * tc + i) * ATB_KG + tr] = pack_bf4(xv[0][i], xv[1][i], xv[2][i], xv[3][i]);
            if (want_bias) bs[i] += (xv[0][i] + xv[1][i]) + (xv[2][i] + xv[3][i]);
        }
        if (ybf && yfull) {
            // a saved bf16 Y: the column-major groups are assembled from the raw halves (one byte permute per dword - a
            // conversion to fp32 and back costs twice the VALU work of the fp32 operand's packing, and this kernel's chunk
            // loop is bound by its VALU work: measured 321 -> 376 us per pass at 64 x (300, 300), profiles/r05_f_atb_ab.txt)
            unsigned lo[4], hi[4];
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {      // (rows beyond the matrix: zeros)
                const bool rvalid = !tail || chunk * ATB_ROWS + tr + 16 * jr < rows;
                lo[jr] = rvalid ? (unsigned)ryh[jr] : 0u;
                hi[jr] = rvalid ? (unsigned)(ryh[jr] >> 32) : 0u;
            }
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            auto col = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, bool high) {
                const unsigned sel = high ? 0x07060302u : 0x05040100u;
                const u32x2 d = {eqd_perm_b32(a1, a0, sel), eqd_perm_b32(a3, a2, sel)};
                return __builtin_bit_cast(s16x4, d);
            };
            Yc[(4 * tc + 0) * ATB_KG + tr] = col(lo[0], lo[1], lo[2], lo[3], false);
            Yc[(4 * tc + 1) * ATB_KG + tr] = col(lo[0], lo[1], lo[2], lo[3], true);
            Yc[(4 * tc + 2) * ATB_KG + tr] = col(hi[0], hi[1], hi[2], hi[3], false);
            Yc[(4 * tc + 3) * ATB_KG + tr] = col(hi[0], hi[1], hi[2], hi[3], true);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) Yc[(4 * tc + i) * ATB_KG + tr] = pack_bf4(yv[0][i], yv[1][i], yv[2][i], yv[3][i]);
        }
        __syncthreads();             // the chunk is in LDS (and every wave is past its reads of the chunk before the last)
        if (chunk + nparts < nchunks) load(chunk + nparts);
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {      // 64 rows = two 32-deep chunks of v_mfma_f32_16x16x32_bf16
            const s16x8 b = cat_bf(Yc[(16 * wave + l15) * ATB_KG + 8 * kp + g], Yc[(16 * wave + l15) * ATB_KG + 8 * kp + 4 + g]);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                acc[mb] = mfma_bf32(cat_bf(Xc[(16 * mb + l15) * ATB_KG + 8 * kp + g], Xc[(16 * mb + l15) * ATB_KG + 8 * kp + 4 + g]), b,
                                    acc[mb]);
        }
        buf ^= 1;
    }
    float* P = partial + u.poff + (long long)c * ATB_PSTRIDE;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * mb + 4 * g + r) * 64 + 16 * wave + l15] = acc[mb][r];
    if (want_bias) {                 // the wave's 4 row groups (lanes 16 apart), then the 4 waves through LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bs[i] = group_sum(bs[i]);
        }
        __syncthreads();
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Yl_[64 * wave + 4 * tc + i] = bs[i];
        }
        __syncthreads();
        if (t < 64) P[ATB_TILE + t] = (Yl_[t] + Yl_[64 + t]) + (Yl_[128 + t] + Yl_[192 + t]);
    }
}

// Tail work that rides in the two launches (DB5.5-sized batches, eqd_atb_with_tail): the end of the backward is the chain
// d h[0] (one linear job) -> embedding gradient (k_embed_bwd's body), 20 us of small launches that the weight-gradient GEMMs do
// not depend on.  As trailing workgroups of k_atb (the linear job: grid rows y >= y0) and of k_atb_reduce (the embedding
// partials) they run beside them - a second BRANCH of the step graph for the same two kernels measured slower (the fork /
// join of a replayed graph costs more than it hides).
#define EMB_ROWS 16      /* nodes per block of the embedding backward (k_embed_bwd) */
struct AtbLinTail {
    EqdLinJob job;
    int y0, nblk;      // first grid row of the tail, 16-row tiles of the job (0: no tail)
};
struct AtbEmbTail {
    const int32_t* res;
    const float *dh0, *dh0b;
    float* partial;
    int ld, n, d_emb, y0, nblk;      // nblk: blocks of EMB_ROWS nodes (0: no tail)
};
union alignas(16) AtbSmem {
    struct {
        float Xl[ATB_ROWS * ATB_LS];
        float Yl[ATB_ROWS * ATB_LS];
    } a;
    struct {
        LinSmem<1> sm;
        EqdLinJob J;
    } lin;
    __device__ AtbSmem() {}
};
__global__ __launch_bounds__(EQD_BLOCK) void k_atb(AtbUnitsArg U, float* __restrict__ partial, AtbLinTail T) {
    __shared__ AtbSmem S;
    float* const Xl = S.a.Xl;
    float* const Yl = S.a.Yl;
    if (T.nblk > 0 && (int)blockIdx.y >= T.y0) {      // the ride-along linear job: k_linear<1>'s body on tile b
        const int b = ((int)blockIdx.y - T.y0) * (int)gridDim.x + (int)blockIdx.x;
        if (b >= T.nblk) return;
#ifdef EQD_HOSTSIM
        for (int i = threadIdx.x; i < (int)(sizeof(EqdLinJob) / 4); i += blockDim.x) ((int*)&S.lin.J)[i] = ((const int*)&T.job)[i];
#else
        // (kernarg segment: U, the 8-byte `partial`, then T - copied with vector loads like every job descriptor)
        static_assert(alignof(AtbLinTail) == 8 && sizeof(AtbUnitsArg) % 8 == 0, "kernarg layout of k_atb");
        kernarg_to_lds(S.lin.J, EQD_KERNARG_PTR(U), (int)(sizeof(AtbUnitsArg) + sizeof(float*) + offsetof(AtbLinTail, job)));
#endif
        __syncthreads();
        LinRegs<1> RA;
        const JobW W = jobw_load(&S.lin.J, (int)(sizeof(EqdLinJob) / 4), threadIdx.x & 63);
        if (S.lin.J.bf16) linear_tile<1, true>(S.lin.J, W, false, nullptr, -1, S.lin.sm, nullptr, 16 * b, RA, false, false, W);
        else linear_tile<1, false>(S.lin.J, W, false, nullptr, -1, S.lin.sm, nullptr, 16 * b, RA, false, false, W);
        return;
    }
    const AtbUnit& u = U.u[blockIdx.y];
    const int c = blockIdx.x;
    if (c >= u.nparts) return;       // uniform per workgroup
    if (u.fast) {
        if (u.job.bf16) {
            const bool mk = u.job.xmask != nullptr;
            if (u.job.y_bf16) {
                if (mk) atb_fast_bf<true, true>(u, c, partial, Xl, Yl);
                else atb_fast_bf<true, false>(u, c, partial, Xl, Yl);
            } else {
                if (mk) atb_fast_bf<false, true>(u, c, partial, Xl, Yl);
                else atb_fast_bf<false, false>(u, c, partial, Xl, Yl);
            }
        } else {
            if (u.job.xmask) atb_fast<true>(u, c, partial, Xl, Yl);
            else atb_fast<false>(u, c, partial, Xl, Yl);
        }
        return;
    }
    const EqdAtbJob& J = u.job;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    (void)t;
    const int mbn = (J.M + 15) >> 4;
    f32x4 acc[5];
#pragma unroll
    for (int mb = 0; mb < 5; ++mb) acc[mb] = f4zero();
    float bacc = 0.f;
    AtbRegs R;
    atb_load(J, u.n0, c, t, R);
    for (int chunk = c; chunk < u.nchunks; chunk += u.nparts) {
        __syncthreads();             // the previous chunk's LDS reads are done
        atb_store(J, u.n0, chunk, t, R, Xl, Yl);
        __syncthreads();
        if (chunk + u.nparts < u.nchunks) atb_load(J, u.n0, chunk + u.nparts, t, R);
        if (J.bf16) {
            if (mbn > 4) atb_mma<5, true>(acc, Xl, Yl, wave, l15, g);
            else atb_mma<4, true>(acc, Xl, Yl, wave, l15, g);
        } else {
            if (mbn > 4) atb_mma<5, false>(acc, Xl, Yl, wave, l15, g);
            else atb_mma<4, false>(acc, Xl, Yl, wave, l15, g);
        }
        if (t < 80) {
#pragma unroll 8
            for (int row = 0; row < ATB_ROWS; ++row) bacc += Xl[row * ATB_LS + t];
        }
    }
    float* P = partial + u.poff + (long long)c * ATB_PSTRIDE;
#pragma unroll
    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * mb + 4 * g + r) * 64 + 16 * wave + l15] = acc[mb][r];
    if (t < 80) P[ATB_TILE + t] = bacc;
}

// Sums a unit's partial tiles in a fixed order and accumulates into the gradient.  Thread = (group of 4 elements, one
// of 16 part lanes): 16-byte loads, a workgroup covers 256 elements of the 5 200-element partial (80 x 64 tile + 80
// column sums); lane pl sums parts pl, pl + 16, .., then the 16 lanes are added in index order - the order of the
// earlier one-element-per-thread kernel, which issued 4 x the load instructions (26.9 us for the ~100 units of a
// config-B pass in one launch).
#define ATB_RED_ELEMS 256
__device__ __forceinline__ void embed_bwd_body(float* __restrict__ acc, const int32_t* __restrict__ res,
                                               const float* __restrict__ dh0, const float* __restrict__ dh0b, int ld, int n,
                                               int d_emb, float* __restrict__ partial, int blk);
__global__ __launch_bounds__(1024) void k_atb_reduce(AtbUnitsArg U, const float* __restrict__ partial, AtbEmbTail E) {
    __shared__ __attribute__((aligned(16))) float red[16][ATB_RED_ELEMS + 4];
    if (E.nblk > 0 && (int)blockIdx.y >= E.y0) {      // the ride-along embedding partials: k_embed_bwd's body on block b
        const int b = ((int)blockIdx.y - E.y0) * (int)gridDim.x + (int)blockIdx.x;
        if (b >= E.nblk || threadIdx.x >= 64) return;      // (one wave works; a finished wave is not waited for)
        embed_bwd_body(&red[0][0], E.res, E.dh0, E.dh0b, E.ld, E.n, E.d_emb, E.partial, b);
        return;
    }
    const AtbUnit& u = U.u[blockIdx.y];
    const EqdAtbJob& J = u.job;
    const int t = threadIdx.x, cg = t & 63, pl = t >> 6;       // element group cg (4 elements), part lane pl
    const int e0 = blockIdx.x * ATB_RED_ELEMS + 4 * cg;
    f32x4 acc = f4zero();
    if (e0 < ATB_PSTRIDE) {
        const float* P = partial + u.poff + e0;
#pragma unroll 4
        for (int p = pl; p < u.nparts; p += 16) acc += *(const f32x4*)(P + (long long)p * ATB_PSTRIDE);
    }
    *(f32x4*)&red[pl][4 * cg] = acc;
    __syncthreads();
    if (t < ATB_RED_ELEMS) {
        const int e = blockIdx.x * ATB_RED_ELEMS + t;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j][t];
        if (J.scale != 0.f) s *= J.scale;
        if (e < ATB_TILE) {
            const int m = e >> 6, n = u.n0 + (e & 63);
            if (m < J.M && n < J.N) J.out[(size_t)m * J.o_rs + (size_t)n * J.o_cs] += s;
        } else if (e < ATB_PSTRIDE) {
            const int m = e - ATB_TILE;
            if (J.bias_out && u.n0 == 0 && m < J.M) J.bias_out[m] += s;
        }
    }
}

// Host side.  Every (job, 64-column block) is a unit; units are packed into launches of at most ATB_MAXUNITS
// such that no two units of a launch accumulate into the same output (shared-weight layers), and each
// unit's rows are split over ~ATB_TARGET_WGS / units workgroups.
#define ATB_TARGET_WGS_DEFAULT 1024
static int atb_target_wgs() {
    const char* f = eqd_tunable("EQD_ATB_WGS");      // tuning experiments only
    const int v = f ? atoi(f) : 0;
    return v >= 64 && v <= 4096 ? v : ATB_TARGET_WGS_DEFAULT;
}
#define ATB_TARGET_WGS atb_target_wgs()
static int atb_units(const EqdAtbJob* jobs, int njobs, std::vector<AtbUnit>& units) {
    for (int i = 0; i < njobs; ++i) {
        const EqdAtbJob& J = jobs[i];
        if (J.M < 4 || J.M > 80 || J.N < 4 || J.rows < 0) {
            eqd_set_error("eqd_atb: job %d has M=%d N=%d rows=%d (need M in 4..80, N >= 4)", i, J.M, J.N, J.rows);
            return EQD_ERR_SHAPE;
        }
        if (J.y_bf16 && ((J.ldy & 3) != 0 || (((uintptr_t)J.Y) & 7) != 0)) {
            eqd_set_error("eqd_atb: job %d has a bf16 Y (y_bf16) with ldy = %d not a multiple of 4, or Y not 8-byte aligned", i,
                          J.ldy);
            return EQD_ERR_SHAPE;
        }
        const int nchunks = J.rows > 0 ? (J.rows + ATB_ROWS - 1) / ATB_ROWS : 0;
        for (int n0 = 0; n0 < J.N; n0 += 64) {
            AtbUnit u;
            memset(&u, 0, sizeof(u));
            u.job = J;
            u.n0 = n0;
            u.nchunks = nchunks;
            auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
            // 1: everything aligned and a whole 64-column block of Y; 2: X as in 1, Y through the shifting loader (the 5
            // columns left over of the 69-wide h0: a general-path unit costs twice a fast one for 1/13 of the columns)
            const bool xfast = J.M == 64 && J.rows > 0 && (J.ldx & 3) == 0 && al16(J.X) && (!J.xmask || al16(J.xmask));
            u.fast = !xfast ? 0 : (J.N - n0 >= 64 && (J.ldy & 3) == 0 && al16(J.Y)) ? 1 : (J.N - n0 >= 4 || n0 >= 4) ? 2 : 0;
            if (J.y_bf16 && u.fast && !J.bf16) u.fast = 0;      // (the fp32 fast body has no bf16 loader; not a shape the model makes)
            // (the bf16 fast body addresses its operands with 32-bit byte offsets)
            if (u.fast && J.bf16 && ((size_t)J.rows * J.ldx * 4 >= (1ull << 31) || (size_t)J.rows * J.ldy * 4 >= (1ull << 31))) u.fast = 0;
            units.push_back(u);
        }
    }
    return EQD_OK;
}
// [first, first + returned) is the next launch; fills nparts / poff of its units and the partial floats needed
static int atb_next_batch(std::vector<AtbUnit>& units, size_t first, long long* floats) {
    int n = 0;
    while (first + n < units.size() && n < ATB_MAXUNITS) {
        const AtbUnit& c = units[first + n];
        bool clash = false;
        for (int i = 0; i < n && !clash; ++i) {
            const AtbUnit& o = units[first + i];
            const bool same_tile = o.job.out == c.job.out && o.n0 == c.n0;
            const bool same_bias = c.job.bias_out && o.job.bias_out == c.job.bias_out && o.n0 == 0 && c.n0 == 0;
            clash = same_tile || same_bias;
        }
        if (clash) break;
        ++n;
    }
    // ~1024 workgroups per 36 units (the launch size this was tuned at); twice that once a unit has more 64-row chunks
    // than that leaves it (config C: +1 %); at most 4096 (eqd_atb_batch_partial_bytes)
    int target = ATB_TARGET_WGS * ((n + 35) / 36 > 0 ? (n + 35) / 36 : 1);
    if (!eqd_tunable("EQD_ATB_WGS") && n > 0 && units[first].nchunks > 8 * (target / n)) target *= 2;
    target = target > 4096 ? 4096 : target;
    int per = (target + n - 1) / (n > 0 ? n : 1);
    // a multiple of 8: the launch's grid is (parts, units), so part c of EVERY unit runs on XCD c % 8 - the units of a layer
    // that share an operand (h with six of them, dz with four: the caller lists them back to back) fetch a part's rows of
    // it into the same L2 at about the same time - and every XCD gets the same number of parts (a grid merely padded to a
    // multiple of 8 measured k_atb 443 -> 487 us in fp32 at 64 x (300, 300): three XCDs a part short per unit)
    const char* xa = eqd_tunable("EQD_ATB_XCD_ALIGN");      // experiments: 0 = any count
    if (!(xa && xa[0] == '0' && xa[1] == 0) && per >= 8) per = (per + 7) / 8 * 8;
    per = per > ATB_MAXBLOCKS ? ATB_MAXBLOCKS : per;
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        AtbUnit& u = units[first + i];
        u.nparts = u.nchunks < per ? u.nchunks : per;
        u.poff = off;
        off += (long long)u.nparts * ATB_PSTRIDE;
    }
    *floats = off;
    return n;
}

// upper bound of the partial workspace of ANY eqd_atb call (independent of the jobs)
size_t eqd_atb_batch_partial_bytes(int rows) {
    (void)rows;
    // (parts of a launch: units x per, per <= target / units + 1 rounded up to a multiple of 8, target <= 4096)
    return (size_t)(4096 + ATB_MAXBLOCKS + 8 * ATB_MAXUNITS) * ATB_PSTRIDE * sizeof(float) + 256;
}

extern "C" size_t eqd_atb_partial_bytes(const EqdAtbJob* jobs, int njobs) {
    std::vector<AtbUnit> units;
    if (atb_units(jobs, njobs, units) != EQD_OK) return 0;
    long long worst = 0;
    for (size_t first = 0; first < units.size();) {
        long long f = 0;
        const int n = atb_next_batch(units, first, &f);
        if (f > worst) worst = f;
        first += n;
    }
    return (size_t)worst * sizeof(float) + 256;
}

static int atb_launch_all(const EqdAtbJob* jobs, int njobs, void* partial, size_t partial_bytes, hipStream_t st,
                          const EqdLinJob* lin_tail, const AtbEmbTail* emb_tail) {
    std::vector<AtbUnit> units;
    int rc = atb_units(jobs, njobs, units);
    if (rc) return rc;
    bool tails_done = false;
    for (size_t first = 0; first < units.size();) {
        long long f = 0;
        const int n = atb_next_batch(units, first, &f);
        if ((size_t)f * sizeof(float) > partial_bytes || (!partial && f > 0)) {
            eqd_set_error("eqd_atb: partial workspace too small (%zu < %lld)", partial_bytes, f * 4LL);
            return EQD_ERR_WORKSPACE;
        }
        AtbUnitsArg arg;
        memset(&arg, 0, sizeof(arg));
        int maxparts = 0;
        for (int i = 0; i < n; ++i) {
            arg.u[i] = units[first + i];
            if (arg.u[i].nparts > maxparts) maxparts = arg.u[i].nparts;
        }
        // the tails ride in the FIRST batch's two launches (the linear job in k_atb, the embedding partials - which read its
        // output - in the k_atb_reduce behind it)
        AtbLinTail LT;
        AtbEmbTail ET;
        memset(&LT, 0, sizeof(LT));
        memset(&ET, 0, sizeof(ET));
        int ly = 0, ey = 0;
        const int redx = (ATB_PSTRIDE + ATB_RED_ELEMS - 1) / ATB_RED_ELEMS;
        if (!tails_done && maxparts > 0) {
            if (lin_tail) {
                LT.job = *lin_tail;
                LT.y0 = n;
                LT.nblk = (lin_tail->rows + 15) / 16;
                ly = (LT.nblk + maxparts - 1) / maxparts;
            }
            if (emb_tail) {
                ET = *emb_tail;
                ET.y0 = n;
                ey = (ET.nblk + redx - 1) / redx;
            }
            tails_done = true;
        }
        if (maxparts > 0) {
            hipLaunchKernelGGL(k_atb, dim3(maxparts, n + ly), dim3(EQD_BLOCK), 0, st, arg, (float*)partial, LT);
            rc = eqd_check_launch("k_atb");
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_atb_reduce, dim3(redx, n + ey), dim3(1024), 0, st, arg, (const float*)partial, ET);
        rc = eqd_check_launch("k_atb_reduce");
        if (rc) return rc;
        first += n;
    }
    if ((lin_tail || emb_tail) && !tails_done) {
        eqd_set_error("eqd_atb_with_tail: no weight-gradient launch to ride in");
        return EQD_ERR_SHAPE;
    }
    return EQD_OK;
}
extern "C" int eqd_atb(const EqdAtbJob* jobs, int njobs, void* partial, size_t partial_bytes, void* stream) {
    return atb_launch_all(jobs, njobs, partial, partial_bytes, (hipStream_t)stream, nullptr, nullptr);
}
// 1 when `dh0_job` (the gradient w.r.t. h[0], a general linear job) would run on k_linear's one-tile body anyway - the form
// that rides in k_atb: small batches (no LDS-resident-weights kernel, one row tile per workgroup)
int eqd_atb_tail_wanted(const EqdLinJob* dh0_job, int n_atb_jobs) {
    const char* f = eqd_tunable("EQD_ATB_TAIL");      // 0: keep the separate launches (A/B runs)
    if (f && f[0] == '0' && f[1] == 0) return 0;
    if (!dh0_job || n_atb_jobs <= 0 || dh0_job->rows <= 0) return 0;
    if (rw_mode(dh0_job->rows) != 0 || linear_row_tiles(dh0_job->rows) != 1) return 0;
    return dh0_job->M >= 4 && dh0_job->M <= 80 && dh0_job->nsrc >= 1 && dh0_job->nsrc <= EQD_MAX_SRC && dh0_job->Y != nullptr &&
           lin_check_sources(*dh0_job) == EQD_OK;
}
// eqd_atb + the end-of-backward chain riding in its launches: dh0_job in k_atb, then k_embed_bwd's work (inputs dh0acc / dh0b
// = dh0_job's output, partials -> `emb_partial`, reduced later through `defer` like eqd_launch_embed_bwd's)
int eqd_atb_with_tail(const EqdAtbJob* jobs, int njobs, void* partial, size_t partial_bytes, hipStream_t st,
                      const EqdLinJob* dh0_job, const EqdGraph* g, const float* dh0acc, const float* dh0b, int ld, int d_emb,
                      float* demb, float* emb_partial, EqdRedList* defer) {
    if (d_emb > 64 || !defer || defer->n + 1 > 512) {
        eqd_set_error("eqd_atb_with_tail: embedding width %d > 64, or no room on the deferred-reduction list", d_emb);
        return EQD_ERR_UNSUPPORTED;
    }
    AtbEmbTail ET;
    memset(&ET, 0, sizeof(ET));
    ET.res = g->res_id; ET.dh0 = dh0acc; ET.dh0b = dh0b; ET.partial = emb_partial;
    ET.ld = ld; ET.n = g->n_nodes; ET.d_emb = d_emb;
    ET.nblk = (g->n_nodes + EMB_ROWS - 1) / EMB_ROWS;
    int rc = atb_launch_all(jobs, njobs, partial, partial_bytes, st, dh0_job, &ET);
    if (rc) return rc;
    defer->seg[defer->n++] = EqdRedSeg{emb_partial, ET.nblk, 21 * d_emb, 21 * d_emb, demb, 0, 0, 0};
    return EQD_OK;
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
// One workgroup per (chain, block of 64 output columns); only blocks that have columns are launched (chain_blk0 is the
// prefix of the chains' block counts).  Thread = (group of 4 columns, one of 64 part lanes): a part's 64 columns are ONE
// 256-byte row segment fetched by 16 lanes with 16-byte loads, eight parts per lane in flight at once (a 4-byte-per-lane
// version of this kernel ran at 1 TB/s on the 90 MB of edge weight-gradient partials of a config-B pass, and chains of
// 2 000 short partials serialised 8 load round trips in one workgroup).  The summation order is fixed: parts
// pl, pl + 64, .. in each lane, segments of a chain in list order, then the 64 lanes as 4 x 16 in index order.
// PL part lanes (thread = 16 column groups x PL): 64 in k_reduce_segments, 16 when the block rides in k_node_gather
__global__ __launch_bounds__(1024) void k_reduce_segments(EqdRedArg A) {
    __shared__ __attribute__((aligned(16))) float red[64][68];
    __shared__ float red2[4][64];
    reduce_block<64>(A, (int)blockIdx.x, red, red2);
}
// segments -> launch descriptors: chains of segments with the same output (shared-weight layers) are summed by ONE
// workgroup column.  Fills as many whole chains as fit one EqdRedArg starting at chain *c0; returns the workgroup count.
struct RedPlan {
    int order[512], cfirst[512], clen[512], nc;
};
static int red_plan(const EqdRedSeg* segs, int nseg, RedPlan& P) {
    static thread_local bool used[512];
    if (nseg > 512) {
        eqd_set_error("eqd_launch_reduce_segments: too many segments");
        return EQD_ERR_SHAPE;
    }
    int no = 0;
    P.nc = 0;
    for (int i = 0; i < nseg; ++i) used[i] = false;
    for (int i = 0; i < nseg; ++i) {
        if (used[i]) continue;
        P.cfirst[P.nc] = no;
        for (int j = i; j < nseg; ++j)
            if (!used[j] && segs[j].out == segs[i].out && segs[j].n == segs[i].n) {
                used[j] = true;
                P.order[no++] = j;
            }
        P.clen[P.nc] = no - P.cfirst[P.nc];
        if (P.clen[P.nc] > EQD_RED_MAXSEG) {
            eqd_set_error("eqd_launch_reduce_segments: chain too long");
            return EQD_ERR_SHAPE;
        }
        ++P.nc;
    }
    return EQD_OK;
}
static int red_fill(const EqdRedSeg* segs, const RedPlan& P, int& c0, EqdRedArg& arg) {
    memset(&arg, 0, sizeof(arg));
    int ns = 0, nch = 0, nblk = 0;
    while (c0 < P.nc && ns + P.clen[c0] <= EQD_RED_MAXSEG) {
        const int n0 = segs[P.order[P.cfirst[c0]]].n;
        if (n0 > 0) {
            arg.chain_first[nch] = ns;
            arg.chain_len[nch] = P.clen[c0];
            arg.chain_blk0[nch] = nblk;
            nblk += (n0 + 63) / 64;
            for (int j = 0; j < P.clen[c0]; ++j) arg.s[ns++] = segs[P.order[P.cfirst[c0] + j]];
            ++nch;
        }
        ++c0;
    }
    arg.nchains = nch;
    return nblk;
}
int eqd_launch_reduce_segments(const EqdRedSeg* segs, int nseg, hipStream_t st) {
    static thread_local RedPlan P;
    if (int e = red_plan(segs, nseg, P)) return e;
    int c0 = 0;
    while (c0 < P.nc) {
        EqdRedArg arg;
        const int before = c0;
        const int nblk = red_fill(segs, P, c0, arg);
        if (c0 == before) {      // cannot happen while red_plan rejects chains longer than one descriptor; never spin on it
            eqd_set_error("eqd_launch_reduce_segments: reduction chain does not fit one descriptor");
            return EQD_ERR_SHAPE;
        }
        if (nblk == 0) continue;
        hipLaunchKernelGGL(k_reduce_segments, dim3(nblk), dim3(1024), 0, st, arg);
        int rc = eqd_check_launch("k_reduce_segments");
        if (rc) return rc;
    }
    return EQD_OK;
}
int eqd_launch_vec_reduce(const float* partial, int nparts, int pstride, int n, float* out, hipStream_t st) {
    EqdRedSeg s = {partial, nparts, pstride, n, out, 0, 0, 0};
    return eqd_launch_reduce_segments(&s, 1, st);
}

__global__ void k_fill(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_axpy(float* y, const float* x, float a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
int eqd_launch_axpy(float* y, const float* x, float a, size_t n, hipStream_t st) {
    if (n == 0) return EQD_OK;
    hipLaunchKernelGGL(k_axpy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, a, n);
    return eqd_check_launch("k_axpy");
}
int eqd_launch_fill(float* p, float v, size_t n, hipStream_t st) {
    if (n == 0) return EQD_OK;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
    return eqd_check_launch("k_fill");
}

// Embedding lookup + log(mu_r_norm) concat (rigid_docking_model.py:459-471)
__global__ void k_embed_fwd(const int32_t* __restrict__ res, const float* __restrict__ mu,
                            const float* __restrict__ emb, int n, int d_emb, int use_mu, float* __restrict__ h0,
                            int ld) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (i >= n) return;
    const int t = res[i];
    for (int cc = c; cc < d_emb; cc += 64) h0[(size_t)i * ld + cc] = emb[(size_t)t * d_emb + cc];
    if (use_mu && c < 5) h0[(size_t)i * ld + d_emb + c] = logf(mu[(size_t)i * 5 + c]);
}
int eqd_launch_embed_fwd(const EqdGraph* g, const float* emb, int d_emb, int use_mu, float* h0, int ld,
                         hipStream_t st) {
    if (g->n_nodes == 0) return EQD_OK;
    hipLaunchKernelGGL(k_embed_fwd, dim3((g->n_nodes + 3) / 4), dim3(256), 0, st, g->res_id, g->mu_r_norm, emb,
                       g->n_nodes, d_emb, use_mu, h0, ld);
    return eqd_check_launch("k_embed_fwd");
}

// Embedding backward: per block of 16 nodes, thread c owns column c -> deterministic per-type sums.  All
// loads of the block are in flight together; the (small) per-block tables are summed by k_reduce_segments.
__device__ __forceinline__ void embed_bwd_body(float* __restrict__ acc, const int32_t* __restrict__ res,
                                               const float* __restrict__ dh0, const float* __restrict__ dh0b, int ld, int n,
                                               int d_emb, float* __restrict__ partial, int blk) {
    const int c = threadIdx.x;  // 64 threads (one wave: its LDS accesses are ordered)
    for (int t = 0; t < 21; ++t) acc[t * 64 + c] = 0.f;
    const int i0 = blk * EMB_ROWS;
    int rid[EMB_ROWS];
    float v[EMB_ROWS];
#pragma unroll
    for (int r = 0; r < EMB_ROWS; ++r) {
        const int i = i0 + r;
        const bool ok = i < n && c < d_emb;
        rid[r] = i < n ? res[i] : 0;
        v[r] = ok ? dh0[(size_t)i * ld + c] + (dh0b ? dh0b[(size_t)i * ld + c] : 0.f) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < EMB_ROWS; ++r) acc[rid[r] * 64 + c] += v[r];
    for (int t = 0; t < 21; ++t)
        if (c < d_emb) partial[((size_t)blk * 21 + t) * d_emb + c] = acc[t * 64 + c];
}
__global__ __launch_bounds__(64) void k_embed_bwd(const int32_t* __restrict__ res, const float* __restrict__ dh0,
                                                  const float* __restrict__ dh0b, int ld, int n, int d_emb,
                                                  float* __restrict__ partial) {
    __shared__ float acc[21 * 64];
    embed_bwd_body(acc, res, dh0, dh0b, ld, n, d_emb, partial, (int)blockIdx.x);
}
size_t eqd_embed_bwd_partial_floats(const EqdGraph* g, int d_emb) {
    return (size_t)((g->n_nodes + EMB_ROWS - 1) / EMB_ROWS) * 21 * d_emb;
}
int eqd_launch_embed_bwd(const EqdGraph* g, const float* dh0, const float* dh0b, int ld, int d_emb, float* demb,
                         float* partial, hipStream_t st, EqdRedList* defer) {
    if (g->n_nodes == 0) return EQD_OK;
    if (d_emb > 64) {
        eqd_set_error("embedding width %d > 64 unsupported", d_emb);
        return EQD_ERR_UNSUPPORTED;
    }
    const int nb = (g->n_nodes + EMB_ROWS - 1) / EMB_ROWS;
    hipLaunchKernelGGL(k_embed_bwd, dim3(nb), dim3(64), 0, st, g->res_id, dh0, dh0b, ld, g->n_nodes, d_emb, partial);
    int rc = eqd_check_launch("k_embed_bwd");
    if (rc) return rc;
    if (defer && defer->n + 1 <= 512) {
        defer->seg[defer->n++] = EqdRedSeg{partial, nb, 21 * d_emb, 21 * d_emb, demb, 0, 0, 0};
        return EQD_OK;
    }
    return eqd_launch_vec_reduce(partial, nb, 21 * d_emb, 21 * d_emb, demb, st);
}

// Per-node sums of the per-edge backward outputs (no atomics):
//   dP[j] = sum over out-edges (CSC, edges with src == j) of dz[e]
//   dQ[j] = sum over in-edges  (CSR, edges with dst == j) of dz[e]
//   dx[j] = a * d_xnew[j] + sum_{src == j} dxrel[e] - sum_{dst == j} dxrel[e]      (x_rel = x[src] - x[dst])
// Workgroups beyond the gather's own (blockIdx.x >= ngather) sum pending partials (EqdRedArg): the layer's edge
// weight-gradient partials were written by the launch before, are read here while they are still in the memory-side
// cache, and the workgroups run on CUs the small gather leaves idle - instead of 90 MB of partials of a config-B
// pass being streamed from HBM by two launches at the end (68 us).
// BF: dz holds bf16 rows ([E][64] unsigned short; the edge backward's bf16 mode), sums are fp32
// A node is handled by 16 lanes with 16-byte vectors (lane c4 owns features 4 c4 .. 4 c4 + 3; four nodes per wave): a row
// of dz is ONE 256-byte request of 16 lanes.  With one lane per feature (a wave per node, 4-byte loads) the kernel issued
// 4 x as many load instructions for the same bytes, and the vector-memory path accepts instructions, not bytes, at a
// fixed rate (profiles/r02_exp_trace_rowwave_*.txt).  Sums per feature run over the edges in the same order as before.
template <bool BF>
__global__ __launch_bounds__(256) void k_node_gather(EqdGatherArgs GA, EqdRedArg RA) {
    if ((int)blockIdx.x >= GA.ngather) {
        __shared__ __attribute__((aligned(16))) float red[16][68];
        __shared__ float red2[4][64];
        reduce_block<16>(RA, (int)blockIdx.x - GA.ngather, red, red2);
        return;
    }
    node_gather_body<BF>(GA, (int)blockIdx.x);
}
// pending: reductions to run in the same launch (emptied on return); what does not fit one descriptor is launched on
// its own.  Two steps, so that the launch itself can be someone else's (eqd_launch_attention_bwd_gather):
//   eqd_gather_plan: fills the kernel's arguments (GA.ngather, the first reduction descriptor RA, its workgroup count nblk)
//   eqd_gather_rest: launches whatever of `pending` did not fit RA and empties the list
static thread_local RedPlan g_gather_plan;
static thread_local int g_gather_c0;
int eqd_gather_plan(const EqdGraph* g, const EqdGatherCall* c, EqdRedList* pending, EqdGatherArgs* GA, EqdRedArg* RA,
                    int* nblk) {
    *nblk = 0;
    g_gather_c0 = 0;
    memset(RA, 0, sizeof(*RA));
    if (pending && pending->n > 0) {
        if (int e = red_plan(pending->seg, pending->n, g_gather_plan)) return e;
        *nblk = red_fill(pending->seg, g_gather_plan, g_gather_c0, *RA);
    }
    GA->csc_ptr = g->csc_ptr; GA->csc_eid = g->csc_eid; GA->rowptr = g->rowptr; GA->n = g->n_nodes;
    GA->dz = c->dz; GA->dxrel = c->dxrel; GA->d_xnew = c->d_xnew; GA->a = c->a;
    GA->dP = c->dP; GA->dQ = c->dQ; GA->dx = c->dx;
    GA->ngather = (g->n_nodes + GATHER_NODES - 1) / GATHER_NODES;
    GA->bf16 = c->dz_bf16;
    return EQD_OK;
}
int eqd_gather_rest(EqdRedList* pending, hipStream_t st) {
    if (pending && pending->n > 0) {
        while (g_gather_c0 < g_gather_plan.nc) {         // (more than 64 segments or chains: not the case for one layer)
            EqdRedArg more;
            const int before = g_gather_c0;
            const int nb = red_fill(pending->seg, g_gather_plan, g_gather_c0, more);
            if (g_gather_c0 == before) {      // (see eqd_launch_reduce_segments)
                eqd_set_error("eqd_gather_rest: reduction chain does not fit one descriptor");
                return EQD_ERR_SHAPE;
            }
            if (nb == 0) continue;
            hipLaunchKernelGGL(k_reduce_segments, dim3(nb), dim3(1024), 0, st, more);
            if (int rc = eqd_check_launch("k_reduce_segments")) return rc;
        }
        pending->n = 0;
    }
    return EQD_OK;
}
int eqd_launch_node_gather(const EqdGraph* g, const float* dz, const float* dxrel, const float* d_xnew, float a,
                           float* dP, float* dQ, float* dx, hipStream_t st, EqdRedList* pending, bool dz_bf16) {
    const EqdGatherCall c = {dz, dxrel, d_xnew, a, dP, dQ, dx, dz_bf16 ? 1 : 0};
    static thread_local EqdRedArg arg;
    EqdGatherArgs GA;
    int nblk = 0;
    if (int e = eqd_gather_plan(g, &c, pending, &GA, &arg, &nblk)) return e;
    if (GA.ngather + nblk > 0) {
        if (dz_bf16)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_node_gather<true>), dim3(GA.ngather + nblk), dim3(256), 0, st, GA, arg);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_node_gather<false>), dim3(GA.ngather + nblk), dim3(256), 0, st, GA, arg);
        if (int rc = eqd_check_launch("k_node_gather")) return rc;
    }
    return eqd_gather_rest(pending, st);
}

// Backward of LeakyReLU -> LayerNorm (node_mlp.2/.3): y_act = LeakyReLU(z) is saved by the forward.
// One wave per row group; lane owns features lane and lane+64 (d <= 128).
#define LNB_ROWS_PER_BLOCK 16
__global__ __launch_bounds__(EQD_BLOCK) void k_ln_act_bwd(const float* __restrict__ y_act,
                                                          const float* __restrict__ d_out,
                                                          const float* __restrict__ gamma, int rows, int d, int ld,
                                                          float slope, float eps, float* __restrict__ dz,
                                                          float* __restrict__ partial) {
    __shared__ float red[EQD_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f0 = lane, f1 = lane + 64;
    const bool v0 = f0 < d, v1 = f1 < d;
    const float g0 = v0 ? gamma[f0] : 0.f, g1 = v1 ? gamma[f1] : 0.f;
    float dg0 = 0.f, dg1 = 0.f, db0 = 0.f, db1 = 0.f;
    const float invd = 1.f / (float)d;
    const int rbase = blockIdx.x * LNB_ROWS_PER_BLOCK;
    for (int rr = wave; rr < LNB_ROWS_PER_BLOCK; rr += EQD_WAVES) {
        const int row = rbase + rr;
        if (row >= rows) break;   // wave-uniform
        const size_t o = (size_t)row * ld;
        const float y0 = v0 ? y_act[o + f0] : 0.f, y1 = v1 ? y_act[o + f1] : 0.f;
        const float o0 = v0 ? d_out[o + f0] : 0.f, o1 = v1 ? d_out[o + f1] : 0.f;
        const float mean = wave_sum(y0 + y1) * invd;
        const float c0 = v0 ? y0 - mean : 0.f, c1 = v1 ? y1 - mean : 0.f;
        const float rstd = 1.f / sqrtf(wave_sum(c0 * c0 + c1 * c1) * invd + eps);
        const float xh0 = c0 * rstd, xh1 = c1 * rstd;
        const float dx0 = o0 * g0, dx1 = o1 * g1;
        const float s1 = wave_sum(dx0 + dx1) * invd;
        const float s2 = wave_sum(dx0 * xh0 + dx1 * xh1) * invd;
        if (v0) dz[o + f0] = rstd * (dx0 - s1 - xh0 * s2) * lrelu_grad(y0, slope);
        if (v1) dz[o + f1] = rstd * (dx1 - s1 - xh1 * s2) * lrelu_grad(y1, slope);
        dg0 += o0 * xh0;
        dg1 += o1 * xh1;
        db0 += o0;
        db1 += o1;
    }
    red[wave][lane] = dg0;
    red[wave][64 + lane] = dg1;
    red[wave][128 + lane] = db0;
    red[wave][192 + lane] = db1;
    __syncthreads();
    const int t = threadIdx.x;
    const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    // partial layout per block: [dgamma(128) | dbeta(128)]
    partial[(size_t)blockIdx.x * 256 + t] = s;
}
size_t eqd_ln_act_bwd_partial_floats(int rows, int d) {
    (void)d;
    return (size_t)((rows + LNB_ROWS_PER_BLOCK - 1) / LNB_ROWS_PER_BLOCK) * 256;
}
int eqd_launch_ln_act_bwd(const float* y_act, const float* d_out, const float* gamma, int rows, int d, int ld,
                          float slope, float eps, float* dz, float* dgamma, float* dbeta, float* partial,
                          hipStream_t st, EqdRedList* defer) {
    if (rows == 0) return EQD_OK;
    if (d > 128) {
        eqd_set_error("ln_act_bwd: width %d > 128 unsupported", d);
        return EQD_ERR_UNSUPPORTED;
    }
    const int nb = (rows + LNB_ROWS_PER_BLOCK - 1) / LNB_ROWS_PER_BLOCK;
    hipLaunchKernelGGL(k_ln_act_bwd, dim3(nb), dim3(EQD_BLOCK), 0, st, y_act, d_out, gamma, rows, d, ld, slope, eps,
                       dz, partial);
    int rc = eqd_check_launch("k_ln_act_bwd");
    if (rc) return rc;
    EqdRedSeg segs[2] = {{partial, nb, 256, d, dgamma, 0, 0, 0}, {partial + 128, nb, 256, d, dbeta, 0, 0, 0}};
    if (defer && defer->n + 2 <= 512) {
        defer->seg[defer->n++] = segs[0];
        defer->seg[defer->n++] = segs[1];
        return EQD_OK;
    }
    return eqd_launch_reduce_segments(segs, 2, st);
}
