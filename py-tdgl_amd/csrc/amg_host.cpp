// Host-side AMG set-up loops (include/tdgl_host_amg.h): Lanczos estimate of rho(D^-1 A) and MIS(2) aggregation.
// Plain C++17 + std::thread.  A team of threads lives for the duration of one call (nothing survives it: safe
// across fork) and meets at a spinning barrier between the phases; every reduction is over fixed blocks of rows
// combined in block order, so the results do not depend on the number of threads.
#include "tdgl_host_amg.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

namespace {

constexpr int64_t BLOCK_ROWS = 8192;

// Thrown inside `barrier()` of the members that are still running once one member has failed.
struct TeamAborted {};

class Team {
  public:
    explicit Team(int threads) : n_(std::max(1, threads)) {}
    int size() const { return n_; }
    // all members call this; returns when every one has arrived (or leaves by exception when a member has failed:
    // nobody spins for a thread that will never come)
    void barrier() {
        if (abort_.load(std::memory_order_acquire)) throw TeamAborted{};
        const int gen = generation_.load(std::memory_order_acquire);
        if (arrived_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            arrived_.store(0, std::memory_order_relaxed);
            generation_.store(gen + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (generation_.load(std::memory_order_acquire) == gen) {
                if (abort_.load(std::memory_order_acquire)) throw TeamAborted{};
                if (++spins > 2000) std::this_thread::yield();
            }
        }
    }
    // the rows [begin, end) of member `tid` when `nblocks` blocks are dealt out in contiguous runs
    static void share(int tid, int members, int64_t nblocks, int64_t &b0, int64_t &b1) {
        b0 = nblocks * tid / members;
        b1 = nblocks * (tid + 1) / members;
    }
    // Runs body(0..n-1) on n threads.  A member that throws (std::bad_alloc in its scratch arrays), or a thread that
    // cannot be created (std::system_error under a pid limit), aborts the team: the others leave at their next
    // barrier, everything is joined, and std::runtime_error is thrown to the caller -- the extern "C" entry points
    // turn it into an error code, no exception crosses the C boundary.
    void run(const std::function<void(int)> &body) {
        std::atomic<bool> go{false};
        auto member = [&](int t) {
            while (!go.load(std::memory_order_acquire)) {
                if (abort_.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            try {
                body(t);
            } catch (...) {
                abort_.store(true, std::memory_order_release);
            }
        };
        std::vector<std::thread> others;
        try {
            others.reserve((size_t)n_ - 1);
            for (int t = 1; t < n_; ++t) others.emplace_back(member, t);
        } catch (...) {
            abort_.store(true, std::memory_order_release);
        }
        go.store(true, std::memory_order_release);
        member(0);
        for (auto &t : others) t.join();
        if (abort_.load(std::memory_order_acquire)) throw std::runtime_error("thread team aborted");
    }

  private:
    int n_;
    std::atomic<int> arrived_{0};
    std::atomic<int> generation_{0};
    std::atomic<bool> abort_{false};
};

constexpr int ERR_RESOURCES = -5;  // out of memory / threads inside a call (include/tdgl_host_amg.h)

int pick_threads(int requested, int64_t nblocks) {
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    int t = requested > 0 ? requested : std::min(16, hw);
    if (requested <= 0)
        if (const char *env = std::getenv("TDGL_HOST_THREADS")) {  // a shared node: cap the set-up's threads
            const int v = std::atoi(env);
            if (v > 0) t = v;
        }
    t = std::min(t, hw);
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, nblocks));
}

}  // namespace

extern "C" int tdgl_host_lanczos(int64_t n, const int32_t *indptr, const int32_t *indices, const double *data, const double *dinv,
                                 int iters, const double *v0, int threads, double *alpha, double *beta, int *steps,
                                 double *gershgorin) {
    if (n < 1 || !indptr || !indices || !data || !dinv || !v0 || !alpha || !beta || !steps || !gershgorin || iters < 1) return -1;
    try {
    const int64_t nblocks = (n + BLOCK_ROWS - 1) / BLOCK_ROWS;
    Team team(pick_threads(threads, nblocks));
    std::vector<double> sq((size_t)n), va(v0, v0 + n), vb((size_t)n, 0.0), vc((size_t)n);
    std::vector<double> part((size_t)nblocks), part_g((size_t)nblocks);
    double *v = va.data(), *v_prev = vb.data(), *w = vc.data();
    double b_prev = 0.0;
    int done = iters;
    bool stop = false;
    team.run([&](int tid) {
        int64_t b0, b1;
        Team::share(tid, team.size(), nblocks, b0, b1);
        const int64_t r0 = b0 * BLOCK_ROWS, r1 = std::min(n, b1 * BLOCK_ROWS);
        // sqrt(dinv), Gershgorin bound
        for (int64_t b = b0; b < b1; ++b) {
            double g = 0.0;
            for (int64_t i = b * BLOCK_ROWS; i < std::min(n, (b + 1) * BLOCK_ROWS); ++i) {
                sq[(size_t)i] = std::sqrt(dinv[i]);
                double s = 0.0;
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) s += std::fabs(data[k]);
                g = std::max(g, s * dinv[i]);
            }
            part_g[(size_t)b] = g;
        }
        team.barrier();
        if (tid == 0) {
            double g = 0.0;
            for (double x : part_g) g = std::max(g, x);
            *gershgorin = g;
        }
        for (int j = 0; j < iters; ++j) {
            // w = D^-1/2 A D^-1/2 v - b_prev v_prev;  alpha = w . v
            for (int64_t b = b0; b < b1; ++b) {
                double dot = 0.0;
                for (int64_t i = b * BLOCK_ROWS; i < std::min(n, (b + 1) * BLOCK_ROWS); ++i) {
                    double y = 0.0;
                    for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) y += data[k] * (sq[(size_t)indices[k]] * v[indices[k]]);
                    const double wi = sq[(size_t)i] * y - b_prev * v_prev[i];
                    w[i] = wi;
                    dot += wi * v[i];
                }
                part[(size_t)b] = dot;
            }
            team.barrier();
            double a = 0.0;  // every member sums the same blocks in the same order
            for (double x : part) a += x;
            team.barrier();
            // w -= alpha v;  beta = |w|
            for (int64_t b = b0; b < b1; ++b) {
                double ss = 0.0;
                for (int64_t i = b * BLOCK_ROWS; i < std::min(n, (b + 1) * BLOCK_ROWS); ++i) {
                    const double wi = w[i] - a * v[i];
                    w[i] = wi;
                    ss += wi * wi;
                }
                part[(size_t)b] = ss;
            }
            team.barrier();
            double ss = 0.0;
            for (double x : part) ss += x;
            const double bj = std::sqrt(ss);
            const bool broke = bj <= 1e-12 * std::max(std::fabs(a), 1.0);
            if (tid == 0) {
                alpha[j] = a;
                beta[j] = broke ? 0.0 : bj;
                if (broke) done = j + 1, stop = true;
            }
            if (broke) break;  // (the same decision in every member)
            // v_prev, v = v, w / beta
            for (int64_t i = r0; i < r1; ++i) w[i] = w[i] / bj;
            team.barrier();
            if (tid == 0) {
                double *old = v_prev;
                v_prev = v;
                v = w;
                w = old;
                b_prev = bj;
            }
            team.barrier();
        }
    });
    (void)stop;
    *steps = done;
    return 0;
    } catch (...) {
        return ERR_RESOURCES;
    }
}

extern "C" int tdgl_host_mis2_aggregate(int64_t n, const int32_t *indptr, const int32_t *indices, const double *weight,
                                        const int64_t *priority, int threads, int64_t *agg, int64_t *n_agg) {
    if (n < 1 || !indptr || !indices || !weight || !priority || !agg || !n_agg) return -1;
    try {
    const int64_t nblocks = (n + BLOCK_ROWS - 1) / BLOCK_ROWS;
    Team team(pick_threads(threads, nblocks));
    std::vector<int8_t> state((size_t)n, 0), root((size_t)n), near1((size_t)n);
    std::vector<int64_t> p((size_t)n), m1((size_t)n), next_agg((size_t)n);
    std::vector<int64_t> undecided((size_t)team.size());
    std::atomic<int> failed{0};
    int64_t roots = 0;
    team.run([&](int tid) {
        int64_t b0, b1;
        Team::share(tid, team.size(), nblocks, b0, b1);
        const int64_t r0 = b0 * BLOCK_ROWS, r1 = std::min(n, b1 * BLOCK_ROWS);
        for (int64_t i = r0; i < r1; ++i) state[(size_t)i] = indptr[i + 1] == indptr[i] ? 1 : 0;  // isolated: its own aggregate
        team.barrier();
        int round = 0;
        for (;; ++round) {
            int64_t und = 0;
            for (int64_t i = r0; i < r1; ++i) {
                const bool u = state[(size_t)i] == 0;
                p[(size_t)i] = u ? priority[i] : 0;
                und += u;
            }
            undecided[(size_t)tid] = und;
            team.barrier();
            int64_t total = 0;
            for (int64_t x : undecided) total += x;
            if (total == 0) break;
            if (round == 200) {
                failed.store(1);
                break;
            }
            for (int64_t i = r0; i < r1; ++i) {  // largest undecided priority within distance 1
                int64_t m = p[(size_t)i];
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) m = std::max(m, p[(size_t)indices[k]]);
                m1[(size_t)i] = m;
            }
            team.barrier();
            for (int64_t i = r0; i < r1; ++i) {  // ... within distance 2: the node itself if it is a new root
                int64_t m = m1[(size_t)i];
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) m = std::max(m, m1[(size_t)indices[k]]);
                root[(size_t)i] = state[(size_t)i] == 0 && p[(size_t)i] == m;
            }
            team.barrier();
            for (int64_t i = r0; i < r1; ++i) {
                if (root[(size_t)i]) state[(size_t)i] = 1;
                int8_t d = root[(size_t)i];
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) d = std::max(d, root[(size_t)indices[k]]);
                near1[(size_t)i] = d;
            }
            team.barrier();
            for (int64_t i = r0; i < r1; ++i) {
                if (state[(size_t)i] != 0) continue;
                int8_t d = near1[(size_t)i];
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) d = std::max(d, near1[(size_t)indices[k]]);
                if (d > 0) state[(size_t)i] = -1;
            }
            team.barrier();
        }
        team.barrier();
        if (failed.load()) return;
        if (tid == 0) {  // roots numbered in ascending node order
            int64_t c = 0;
            for (int64_t i = 0; i < n; ++i) agg[i] = state[(size_t)i] == 1 ? c++ : -1;
            roots = c;
        }
        team.barrier();
        for (int pass = 0; pass < 2; ++pass) {
            for (int64_t i = r0; i < r1; ++i) {
                int64_t a = agg[i];
                if (a < 0) {
                    double best = -1.0;
                    for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k)
                        if (agg[indices[k]] >= 0 && weight[k] > best) best = weight[k], a = agg[indices[k]];  // first best in row order
                }
                next_agg[(size_t)i] = a;
            }
            team.barrier();
            for (int64_t i = r0; i < r1; ++i) agg[i] = next_agg[(size_t)i];
            team.barrier();
        }
    });
    if (failed.load()) return -2;
    for (int64_t i = 0; i < n; ++i)
        if (agg[i] < 0) agg[i] = roots++;  // (cannot happen for a maximal MIS(2))
    *n_agg = roots;
    return 0;
    } catch (...) {
        return ERR_RESOURCES;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C = A B for CSR matrices (Gustavson, row blocks dealt to threads, one dense accumulator per thread).  Every entry
// is accumulated in the order (entries of A's row) x (entries of B's row), exact zeros are not stored, the columns
// of a row come out ascending: the result does not depend on the number of threads.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Product {
    int64_t rows = 0;
    std::vector<int64_t> row_nnz;               // per row
    std::vector<std::vector<int32_t>> indices;  // per thread, rows in order
    std::vector<std::vector<double>> data;
    std::vector<int64_t> first_row;             // per thread (+ end)
};
}  // namespace

extern "C" void *tdgl_host_spgemm(int64_t rows, int64_t cols, const int32_t *a_indptr, const int32_t *a_indices, const double *a_data,
                                  const int32_t *b_indptr, const int32_t *b_indices, const double *b_data, int threads, int64_t *nnz) {
    if (rows < 0 || cols < 1 || !a_indptr || !b_indptr || !nnz) return nullptr;
    if ((a_indptr[rows] > 0 && (!a_indices || !a_data)) || cols > (int64_t)2147483647) return nullptr;
    try {
    // accumulators: 12 bytes per column and thread, at most ~512 MB in all
    int t = pick_threads(threads, std::max<int64_t>(1, rows / 64));
    t = (int)std::max<int64_t>(1, std::min<int64_t>(t, ((int64_t)512 << 20) / (12 * cols)));
    Team team(t);
    std::unique_ptr<Product> out(new Product);  // (released to the caller only when the product is complete)
    out->rows = rows;
    out->row_nnz.assign((size_t)rows, 0);
    out->indices.resize((size_t)t);
    out->data.resize((size_t)t);
    out->first_row.resize((size_t)t + 1);
    for (int k = 0; k <= t; ++k) out->first_row[(size_t)k] = rows * k / t;
    team.run([&](int tid) {
        std::vector<double> sums((size_t)cols, 0.0);
        std::vector<uint8_t> seen((size_t)cols, 0);
        std::vector<int32_t> touched;
        auto &idx = out->indices[(size_t)tid];
        auto &val = out->data[(size_t)tid];
        for (int64_t i = out->first_row[(size_t)tid]; i < out->first_row[(size_t)tid + 1]; ++i) {
            touched.clear();
            for (int32_t jj = a_indptr[i]; jj < a_indptr[i + 1]; ++jj) {
                const int32_t j = a_indices[jj];
                const double v = a_data[jj];
                for (int32_t kk = b_indptr[j]; kk < b_indptr[j + 1]; ++kk) {
                    const int32_t k = b_indices[kk];
                    if (!seen[(size_t)k]) seen[(size_t)k] = 1, touched.push_back(k);
                    sums[(size_t)k] += v * b_data[kk];
                }
            }
            std::sort(touched.begin(), touched.end());
            int64_t count = 0;
            for (int32_t k : touched) {
                if (sums[(size_t)k] != 0.0) {
                    idx.push_back(k);
                    val.push_back(sums[(size_t)k]);
                    ++count;
                }
                sums[(size_t)k] = 0.0;
                seen[(size_t)k] = 0;
            }
            out->row_nnz[(size_t)i] = count;
        }
    });
    int64_t total = 0;
    for (const auto &v : out->indices) total += (int64_t)v.size();
    *nnz = total;
    return out.release();
    } catch (...) {
        return nullptr;
    }
}

extern "C" int tdgl_host_spgemm_take(void *handle, int64_t *indptr, int32_t *indices, double *data) {
    std::unique_ptr<Product> p(static_cast<Product *>(handle));
    if (!p) return -1;
    if (indptr) {
        indptr[0] = 0;
        for (int64_t i = 0; i < p->rows; ++i) indptr[i + 1] = indptr[i] + p->row_nnz[(size_t)i];
        int64_t at = 0;
        for (size_t t = 0; t < p->indices.size(); ++t) {
            if (indices) std::copy(p->indices[t].begin(), p->indices[t].end(), indices + at);
            if (data) std::copy(p->data[t].begin(), p->data[t].end(), data + at);
            at += (int64_t)p->indices[t].size();
        }
    }
    return 0;
}
