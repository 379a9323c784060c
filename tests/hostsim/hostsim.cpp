// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/hostsim/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <vector>

hostsim_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

extern "C" void hostsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hostsim_switch
.type hostsim_switch,@function
hostsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace hostsim {

static const size_t kStack = 512 * 1024;
static const int kMaxThreads = 1024;

struct Fiber {
    void* sp;
    char* stack;
    bool done;
    hostsim_uint3 tid;
    int wave, lane;
    unsigned ncoll;      // wave collectives executed so far
};
struct Wave {
    int nlanes, arrived;
    unsigned gen;
    float fa[2][64], fb[2][64];
    int ia[2][64];
    unsigned short ha[2][64][4], hb[2][64][4];
};

static Fiber g_fib[kMaxThreads];
static Wave g_wave[kMaxThreads / 64];
static char* g_stacks = nullptr;
static void* g_sched_sp;
static Fiber* g_cur;
static const std::function<void()>* g_body;
static int g_nthreads, g_blk_arrived;
static int g_alive;      // threads of the block that have not finished: like s_barrier, __syncthreads only counts those
static unsigned g_blk_gen;
static unsigned long g_progress;

static void yield() {
    Fiber* f = g_cur;
    hostsim_switch(&f->sp, g_sched_sp);
}

static void fiber_main() {
    (*g_body)();
    g_cur->done = true;
    --g_alive;
    if (g_blk_arrived > 0 && g_blk_arrived == g_alive) {      // the threads waiting at a barrier were waiting for this one only
        g_blk_arrived = 0;
        g_blk_gen++;
    }
    g_progress++;
    yield();
    abort();
}

static void wave_sync() {
    Fiber* f = g_cur;
    Wave& w = g_wave[f->wave];
    unsigned my = w.gen;
    if (++w.arrived == w.nlanes) {
        w.arrived = 0;
        w.gen++;
        g_progress++;
    } else {
        while (w.gen == my) yield();
    }
}

int lane_id() { return g_cur->lane; }
void wave_barrier() { g_cur->ncoll++; wave_sync(); }

void sync_block() {
    unsigned my = g_blk_gen;
    if (++g_blk_arrived == g_alive) {
        g_blk_arrived = 0;
        g_blk_gen++;
        g_progress++;
    } else {
        while (g_blk_gen == my) yield();
    }
}

float shfl_f(float v, int src) {
    Fiber* f = g_cur;
    Wave& w = g_wave[f->wave];
    int p = (f->ncoll++) & 1;
    w.fa[p][f->lane] = v;
    wave_sync();
    src &= 63;
    return src < w.nlanes ? w.fa[p][src] : v;
}

int shfl_i(int v, int src) {
    Fiber* f = g_cur;
    Wave& w = g_wave[f->wave];
    int p = (f->ncoll++) & 1;
    w.ia[p][f->lane] = v;
    wave_sync();
    src &= 63;
    return src < w.nlanes ? w.ia[p][src] : v;
}

void mfma16x16x4(float a, float b, const float* c, float* d) {
    Fiber* f = g_cur;
    Wave& w = g_wave[f->wave];
    if (w.nlanes != 64) { fprintf(stderr, "hostsim: MFMA in a partial wave\n"); abort(); }
    int p = (f->ncoll++) & 1;
    w.fa[p][f->lane] = a;
    w.fb[p][f->lane] = b;
    wave_sync();
    int col = f->lane & 15, grp = f->lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * grp + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[p][row + 16 * k], w.fb[p][col + 16 * k], acc);
        d[r] = acc;
    }
}

static inline float bf16_to_f32(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// v_mfma_f32_16x16x16_bf16: lane (i = lane & 15, g = lane >> 4) brings A[i][4g..4g+3] and B[4g..4g+3][i] (as 4 bf16),
// receives D[4g + r][i]; products of bf16 values are exact in fp32, the 16-term sum is accumulated in fp32
void mfma16x16x16bf16(const unsigned short* a, const unsigned short* b, const float* c, float* d) {
    Fiber* f = g_cur;
    Wave& w = g_wave[f->wave];
    if (w.nlanes != 64) { fprintf(stderr, "hostsim: MFMA in a partial wave\n"); abort(); }
    int p = (f->ncoll++) & 1;
    for (int i = 0; i < 4; ++i) {
        w.ha[p][f->lane][i] = a[i];
        w.hb[p][f->lane][i] = b[i];
    }
    wave_sync();
    int col = f->lane & 15, grp = f->lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * grp + r;
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc = fmaf(bf16_to_f32(w.ha[p][row + 16 * (k >> 2)][k & 3]), bf16_to_f32(w.hb[p][col + 16 * (k >> 2)][k & 3]), acc);
        d[r] = acc;
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > kMaxThreads) { fprintf(stderr, "hostsim: bad block size %d\n", nt); abort(); }
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { perror("hostsim mmap"); abort(); }
    }
    const std::function<void()>* saved_body = g_body;
    g_body = &body;
    g_nthreads = nt;
    blockDim = block;
    gridDim = grid;
    int nwaves = (nt + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blk_arrived = 0;
        g_blk_gen = 0;
        g_alive = nt;
        for (int w = 0; w < nwaves; ++w) {
            g_wave[w].nlanes = (w == nwaves - 1) ? nt - 64 * w : 64;
            g_wave[w].arrived = 0;
            g_wave[w].gen = 0;
        }
        for (int t = 0; t < nt; ++t) {
            Fiber& f = g_fib[t];
            f.stack = g_stacks + kStack * t;
            f.done = false;
            f.ncoll = 0;
            f.wave = t / 64;
            f.lane = t % 64;
            f.tid.x = t % block.x;
            f.tid.y = (t / block.x) % block.y;
            f.tid.z = t / (block.x * block.y);
            // initial frame: 6 callee-saved slots + return address (slot address = 0 mod 16)
            uintptr_t top = ((uintptr_t)(f.stack + kStack) & ~(uintptr_t)15) - 64;
            void** sp = (void**)top;
            *sp = (void*)&fiber_main;
            sp -= 6;
            for (int i = 0; i < 6; ++i) sp[i] = nullptr;
            f.sp = sp;
        }
        int remaining = nt;
        while (remaining > 0) {
            unsigned long before = g_progress;
            for (int t = 0; t < nt; ++t) {
                Fiber& f = g_fib[t];
                if (f.done) continue;
                g_cur = &f;
                threadIdx = f.tid;
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                hostsim_switch(&g_sched_sp, f.sp);
                if (f.done) --remaining;
            }
            if (g_progress == before && remaining > 0) {
                fprintf(stderr, "hostsim: deadlock (divergent barrier/collective) in block %u,%u\n", bx, by);
                fprintf(stderr, "  block barrier: %d of %d threads arrived\n", g_blk_arrived, g_nthreads);
                for (int w = 0; w < (nt + 63) / 64; ++w) {
                    unsigned lo = ~0u, hi = 0;
                    int ndone = 0;
                    for (int t = 64 * w; t < nt && t < 64 * w + 64; ++t) {
                        lo = g_fib[t].ncoll < lo ? g_fib[t].ncoll : lo;
                        hi = g_fib[t].ncoll > hi ? g_fib[t].ncoll : hi;
                        ndone += g_fib[t].done;
                    }
                    fprintf(stderr, "  wave %d: %d lanes waiting at a wave collective, collectives executed %u..%u, %d lanes finished\n",
                            w, g_wave[w].arrived, lo, hi, ndone);
                }
                abort();
            }
        }
    }
    g_body = saved_body;
}

}  // namespace hostsim
