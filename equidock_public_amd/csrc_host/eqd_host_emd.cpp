// Exact optimal transport between two UNIFORM discrete measures on the host: the solver behind the pocket OT term of the
// reference's loss (src/utils/ot_utils.py:22-29 calls POT's `ot.emd(a, b, M, numItermax=10000)` with a = 1/n, b = 1/m on a
// (n_pocket x 50) cost matrix per pair per step, src/train.py:117-129).  POT 0.7.0 is a third-party C++ network simplex
// that is neither vendored in the reference nor installed here, so this is a from-scratch exact solver, pinned against an
// independent LP solution (oracle/ot_port.py, scipy HiGHS) - "POT parity unpinned" (DESIGN.md).
//
// Problem sizes are tiny (n <= a few hundred pocket residues, m = 50 keypoints): a GPU cannot beat ~0.1-0.5 ms of
// branchy scalar code per problem, so the exact plan is computed here, one problem per worker thread, while the cost
// matrix and the loss / gradient stay on the device (csrc/eqd_loss_kernels.hip: eqd_pocket_ot_*).
//
// Algorithm: successive shortest paths with node potentials on the bipartite transportation network, integer supplies
// (source i carries m / g units, sink j needs n / g, g = gcd(n, m)), dense Dijkstra on reduced costs with early exit
// at the first sink that still has demand.  Costs in double; exact up to floating-point ties, which only matter when
// the optimal plan is not unique.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

struct EmdScratch {
    std::vector<double> c, pu, pv, dist;
    std::vector<int64_t> f, supply, demand;
    std::vector<int> prev;
    std::vector<char> done;
};

// cost: [n][m] fp32 row-major -> plan [n][m] fp32 (mass, sums to 1), returns sum plan * cost (double)
double emd_uniform(int n, int m, const float* cost, float* plan, EmdScratch& S) {
    if (n <= 0 || m <= 0) return 0.0;
    const int g = std::gcd(n, m);
    const int64_t su = m / g, de = n / g;          // units per source / per sink; total = n m / g
    const double unit = (double)g / ((double)n * (double)m);
    const int V = n + m;
    S.c.assign((size_t)n * m, 0.0);
    for (size_t i = 0; i < (size_t)n * m; ++i) S.c[i] = (double)cost[i];
    S.f.assign((size_t)n * m, 0);
    S.pu.assign(n, 0.0);
    S.pv.assign(m, 0.0);
    S.supply.assign(n, su);
    S.demand.assign(m, de);
    S.dist.resize(V);
    S.prev.resize(V);
    S.done.resize(V);
    // potentials: reduced cost of arc i -> j is c_ij + pu_i - pv_j >= 0; start with pv_j = min_i c_ij (pu = 0)
    for (int j = 0; j < m; ++j) {
        double mn = std::numeric_limits<double>::infinity();
        for (int i = 0; i < n; ++i) mn = std::min(mn, S.c[(size_t)i * m + j]);
        S.pv[j] = mn;
    }
    const double INF = std::numeric_limits<double>::infinity();
    for (int s = 0; s < n; ++s) {
        while (S.supply[s] > 0) {
            // Dijkstra from source s over nodes {sources 0..n-1, sinks n..n+m-1}
            std::fill(S.dist.begin(), S.dist.end(), INF);
            std::fill(S.done.begin(), S.done.end(), 0);
            S.dist[s] = 0.0;
            S.prev[s] = -1;
            int target = -1;
            for (;;) {
                int u = -1;
                double du = INF;
                for (int v = 0; v < V; ++v)
                    if (!S.done[v] && S.dist[v] < du) {
                        du = S.dist[v];
                        u = v;
                    }
                if (u < 0) break;
                S.done[u] = 1;
                if (u >= n) {                          // a sink
                    const int j = u - n;
                    if (S.demand[j] > 0) {
                        target = u;
                        break;
                    }
                    // backward arcs j -> i where flow exists: reduced cost -(c_ij + pu_i - pv_j) (0 up to rounding)
                    for (int i = 0; i < n; ++i) {
                        if (S.done[i] || S.f[(size_t)i * m + j] <= 0) continue;
                        double rc = -(S.c[(size_t)i * m + j] + S.pu[i] - S.pv[j]);
                        if (rc < 0.0) rc = 0.0;
                        if (du + rc < S.dist[i]) {
                            S.dist[i] = du + rc;
                            S.prev[i] = u;
                        }
                    }
                } else {                               // a source: forward arcs to every sink
                    const int i = u;
                    const double* ci = &S.c[(size_t)i * m];
                    const double pui = S.pu[i];
                    for (int j = 0; j < m; ++j) {
                        if (S.done[n + j]) continue;
                        double rc = ci[j] + pui - S.pv[j];
                        if (rc < 0.0) rc = 0.0;
                        if (du + rc < S.dist[n + j]) {
                            S.dist[n + j] = du + rc;
                            S.prev[n + j] = u;
                        }
                    }
                }
            }
            if (target < 0) return std::numeric_limits<double>::quiet_NaN();   // cannot happen: total supply == demand
            const double dt = S.dist[target];
            // bottleneck along the path
            int64_t delta = std::min(S.supply[s], S.demand[target - n]);
            for (int v = target; S.prev[v] >= 0; v = S.prev[v]) {
                const int u = S.prev[v];
                if (u >= n) delta = std::min(delta, S.f[(size_t)v * m + (u - n)]);   // backward arc sink u -> source v
            }
            for (int v = target; S.prev[v] >= 0; v = S.prev[v]) {
                const int u = S.prev[v];
                if (u < n) S.f[(size_t)u * m + (v - n)] += delta;    // forward arc source u -> sink v
                else S.f[(size_t)v * m + (u - n)] -= delta;
            }
            S.supply[s] -= delta;
            S.demand[target - n] -= delta;
            // potentials: with rc(u -> v) = w(u, v) + pi_u - pi_v (pi_i = pu_i, pi_j = pv_j; forward w = c_ij, backward
            // w = -c_ij), pi_v += min(dist_v, dist_t) keeps every reduced cost >= 0 and makes the path's arcs tight
            for (int i = 0; i < n; ++i) S.pu[i] += std::min(S.dist[i], dt);
            for (int j = 0; j < m; ++j) S.pv[j] += std::min(S.dist[n + j], dt);
        }
    }
    double val = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            const double p = (double)S.f[(size_t)i * m + j] * unit;
            plan[(size_t)i * m + j] = (float)p;
            val += p * S.c[(size_t)i * m + j];
        }
    return val;
}

// Persistent worker threads (round 6): a training step calls the solver once, between two hipGraph replays, and the GPU
// idles while it runs - creating and joining 7 threads per call (~40 us each) was a third of that gap at 8 pairs.  Workers
// sleep on a condition variable between calls and take problem indices from a shared counter until none is left.  A
// forked child (DataLoader workers import this library too) starts its own pool: threads do not survive fork().
class EmdPool {
public:
    static EmdPool& get() {
        static EmdPool* pool = nullptr;
        static pid_t owner = 0;
        static std::mutex guard;
        std::lock_guard<std::mutex> lk(guard);
        if (!pool || owner != getpid()) {      // (after fork(): the parent's pool object is abandoned, never destroyed)
            pool = new EmdPool();
            owner = getpid();
        }
        return *pool;
    }
    // run job(t) for t = 0 .. n - 1, the caller takes t = 0; returns when all are done
    void run(int n, const std::function<void(int)>& job) {
        std::lock_guard<std::mutex> serial(call_);      // one batch at a time
        grow(n - 1);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            n_ = n;
            next_ = 1;
            pending_ = n - 1;
            ++gen_;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void grow(int workers) {
        while ((int)th_.size() < workers) {
            th_.emplace_back([this] { loop(); });
            th_.back().detach();
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            int t = -1;
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen && next_ < n_; });
                t = next_++;
                if (next_ >= n_) seen = gen_;
                job = job_;
            }
            (*job)(t);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex m_, call_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int)>* job_ = nullptr;
    int n_ = 0, next_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
};

}  // namespace

extern "C" {

int eqd_host_emd_abi(void) { return 1; }

// n_problems independent problems; problem p has n_src[p] sources (rows) and n_snk sinks (columns); cost / plan are the
// row blocks of all problems one after the other ([sum n_src][n_snk] fp32).  value[p] (optional) = sum plan * cost.
// n_threads <= 0: min(n_problems, hardware threads / 2, 64).  Returns 0, or 1 if a problem failed (NaN value).
int eqd_host_emd_uniform(int n_problems, const int32_t* n_src, int n_snk, const float* cost, float* plan, double* value,
                         int n_threads) {
    if (n_problems <= 0) return 0;
    std::vector<size_t> off(n_problems + 1, 0);
    for (int p = 0; p < n_problems; ++p) off[p + 1] = off[p] + (size_t)std::max(0, n_src[p]) * (size_t)n_snk;
    if (n_threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        n_threads = (int)std::min<unsigned>(std::max(1u, hw / 2), 64u);      // (the workers are persistent: see EmdPool)
    }
    n_threads = std::min(n_threads, n_problems);
    std::vector<double> val(n_problems, 0.0);
    auto work = [&](int t) {
        EmdScratch S;
        for (int p = t; p < n_problems; p += n_threads)
            val[p] = emd_uniform(n_src[p], n_snk, cost + off[p], plan + off[p], S);
    };
    if (n_threads == 1) {
        work(0);
    } else {
        const std::function<void(int)> job = work;
        EmdPool::get().run(n_threads, job);
    }
    int bad = 0;
    for (int p = 0; p < n_problems; ++p) {
        if (value) value[p] = val[p];
        if (std::isnan(val[p])) bad = 1;
    }
    return bad;
}

}  // extern "C"
