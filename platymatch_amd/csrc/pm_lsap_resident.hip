// pm_lsap_resident.hip — the whole device-resident assignment solve of ONE cost matrix in one native call (round 4).
// Reference call site: scipy.optimize.linear_sum_assignment(U_h), _dock_widget.py:604-611.
//
// platymatch_amd/lsap.py drives the scheme of pm_lsap_dev.hip / pm_lsap_core.cpp from Python: ~60 small calls per hypothesis
// (kernel launches through ctypes, torch allocations, blocking copies, NumPy glue) from four host threads that share the
// interpreter lock.  At 5 000 nuclei that glue is a third of the assignment stage (profiles/r03_lsap_phases.txt: 8.5-12.9 ms
// per hypothesis alone, 19-27 ms for four threads side by side).  This file is the same sequence — column reduction, core
// selection, eps-scaling auctions with pricing in between, shortest augmenting paths, pricing rounds until no entry of the
// dense matrix violates dual feasibility (pm_lsap_solve_resident), and the certificate with its uniqueness check
// (pm_lsap_certify_resident) — as host C++ around the same kernels and the same core solver: one foreign call per hypothesis,
// no interpreter in the loop, staging through a pinned buffer that lives with the calling thread.  Same duals, same
// assignment, same certificate (tests/test_gpu_lsap.py compares the two drivers).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pm_common.h"

namespace pm {

struct PinnedBuffer {                     // grow-only pinned staging area of the calling thread (host memory only)
    void *p = nullptr;
    size_t bytes = 0;
    ~PinnedBuffer() { if (p) (void)hipHostFree(p); }
    void *need(size_t n) {
        if (n <= bytes) return p;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return nullptr; }
        bytes = n;
        return p;
    }
};
static thread_local PinnedBuffer g_stage;

struct ResidentLayout {                   // device workspace of one solve (candidate arrays sized for RES_KMAX per row)
    size_t v, cols, costs, flag, colmin, u, c4r, summary, stats, tight, red, rows, total;
};
static ResidentLayout resident_layout(int nr, int nc, int kmax, int cap) {
    ResidentLayout L;
    size_t o = 0;
    auto take = [&](size_t b) { const size_t at = o; o = align_up(o + b, 256); return at; };
    L.v = take((size_t)nc * 8);
    L.cols = take((size_t)nr * kmax * 4);
    L.costs = take((size_t)nr * kmax * 8);
    L.flag = take(16);
    L.colmin = take(pm_lsap_col_min_workspace(nr, nc));
    L.u = take((size_t)nr * 8);
    L.c4r = take((size_t)nr * 4);
    L.summary = take(16);
    L.stats = take(16);
    L.tight = take((size_t)cap * 8);
    L.red = take((size_t)cap * 8);
    L.rows = take((size_t)nr * 16);
    L.total = o;
    return L;
}

constexpr int RES_KMAX = 64;               // most candidates per row a selection may return here (the product uses 16 and 8)
static inline int tight_cap(int nc) { return 8 * nc + 1024; }
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct CoreHandle {
    void *h;
    CoreHandle(int nr, int nc) : h(pm_lsap_core_create(nr, nc)) {}
    ~CoreHandle() { if (h) pm_lsap_core_destroy(h); }
};

#define PM_TRY(expr) do { const int rc_ = (expr); if (rc_ != PM_OK) return rc_; } while (0)
#define PM_HIP(expr) do { if ((expr) != hipSuccess) { (void)launch_status(); return PM_ERR_LAUNCH; } } while (0)

// row_select on the device + its [nr][k] candidates on the host (pinned staging), blocking
static int select_to_host(const double *U, int nr, int nc, size_t ld, const double *v_host, int k, char *dws, const ResidentLayout &L,
                          int32_t *cols_h, double *costs_h, int *nonfinite, hipStream_t s, double *t_dev) {
    const double t0 = now_s();
    double *v_d = (double *)(dws + L.v);
    if (v_host) PM_HIP(hipMemcpyAsync(v_d, v_host, (size_t)nc * 8, hipMemcpyHostToDevice, s));
    PM_TRY(pm_lsap_row_select(U, nr, nc, ld, v_host ? v_d : nullptr, k, (int32_t *)(dws + L.cols), (double *)(dws + L.costs),
                              (int32_t *)(dws + L.flag), s));
    int32_t flag = 0;
    PM_HIP(hipMemcpyAsync(cols_h, dws + L.cols, (size_t)nr * k * 4, hipMemcpyDeviceToHost, s));
    PM_HIP(hipMemcpyAsync(costs_h, dws + L.costs, (size_t)nr * k * 8, hipMemcpyDeviceToHost, s));
    PM_HIP(hipMemcpyAsync(&flag, dws + L.flag, 4, hipMemcpyDeviceToHost, s));
    PM_HIP(hipStreamSynchronize(s));
    *nonfinite = flag;
    *t_dev += now_s() - t0;
    return PM_OK;
}

}  // namespace pm

extern "C" {

size_t pm_lsap_resident_workspace(int nr, int nc) {
    if (nr <= 0 || nc < nr) return 0;
    return pm::resident_layout(nr, nc, pm::RES_KMAX, pm::tight_cap(nc)).total;
}

void pm_lsap_default_options(pm_lsap_options *o) {
    if (!o) return;
    o->core_edges = 16; o->price_edges = 8; o->max_pricing_rounds = 200;
    o->rel_delta = 1e-13; o->rel_eps_collect = 1e-7; o->rel_eps_floor = 1e-11; o->eps_safety = 16.0;
    o->column_reduction = 1;
    o->auction = 1; o->a_eps0 = 0.25; o->a_eps_min = 1e-6; o->a_factor = 5.0; o->a_rounds = 3; o->a_later_eps0 = 0.01;
    o->a_bids_per_row = 200; o->a_later_bids_per_row = 60; o->a_stop_below = 0.02; o->a_max_free_columns = 1.0;
    o->min_eps = 0.0;
}

int pm_lsap_certify_resident(const double *U, int nr, int nc, size_t ld, const pm_lsap_options *opt, const double *u, const double *v,
                             const int32_t *col4row, int32_t *tight_out, int tight_capacity, pm_lsap_report *rep, void *dev_ws,
                             size_t dev_ws_bytes, void *stream) {
    using namespace pm;
    if (!U || !opt || !u || !v || !col4row || !rep || nr <= 0 || nc < nr || ld < (size_t)nc) return PM_ERR_INVALID_ARG;
    if (!dev_ws || ((uintptr_t)dev_ws & 255) || dev_ws_bytes < pm_lsap_resident_workspace(nr, nc)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int cap = tight_cap(nc);
    const ResidentLayout L = resident_layout(nr, nc, RES_KMAX, cap);
    char *dws = (char *)dev_ws;
    const double t0 = now_s();
    double scale = 1e-300;
    for (int i = 0; i < nr; ++i) scale = std::max(scale, std::fabs(u[i]));
    for (int j = 0; j < nc; ++j) scale = std::max(scale, std::fabs(v[j]));
    const double delta = opt->rel_delta * scale;
    const double eps_collect = std::max(opt->rel_eps_collect * scale, 4.0 * opt->min_eps);
    rep->delta = delta;
    rep->optimal = rep->unique = 0;
    rep->tight_within_eps = -1;
    rep->n_tight = 0;
    PM_HIP(hipMemcpyAsync(dws + L.u, u, (size_t)nr * 8, hipMemcpyHostToDevice, s));
    PM_HIP(hipMemcpyAsync(dws + L.v, v, (size_t)nc * 8, hipMemcpyHostToDevice, s));
    PM_HIP(hipMemcpyAsync(dws + L.c4r, col4row, (size_t)nr * 4, hipMemcpyHostToDevice, s));
    PM_TRY(pm_lsap_certificate(U, nr, nc, ld, (const double *)(dws + L.u), (const double *)(dws + L.v), (const int32_t *)(dws + L.c4r), delta,
                               eps_collect, (int32_t *)(dws + L.summary), (double *)(dws + L.stats), (int32_t *)(dws + L.tight),
                               (double *)(dws + L.red), cap, (double *)(dws + L.rows), (double *)(dws + L.rows) + nr, s));
    char *stage = (char *)g_stage.need((size_t)nr * 16 + 64);
    if (!stage) return PM_ERR_WORKSPACE;
    int32_t *summary = (int32_t *)stage;
    double *rows_h = (double *)(stage + 64);
    PM_HIP(hipMemcpyAsync(summary, dws + L.summary, 16, hipMemcpyDeviceToHost, s));
    PM_HIP(hipMemcpyAsync(rows_h, dws + L.rows, (size_t)nr * 16, hipMemcpyDeviceToHost, s));
    PM_HIP(hipStreamSynchronize(s));
    const int viol = summary[0], n_tight = summary[1], loose = summary[2];
    double b0 = 0.0, b1 = 0.0;
    for (int i = 0; i < nr; ++i) { b0 += rows_h[i]; b1 += rows_h[nr + i]; }
    const double bound = b0 + b1;
    const double eps = std::max(std::max(opt->rel_eps_floor * scale, opt->eps_safety * bound), opt->min_eps + bound);
    rep->violations = viol;
    rep->loose = loose;
    rep->slack_bound = bound;
    rep->eps = eps;
    rep->seconds_certify = now_s() - t0;
    if (viol || loose) return PM_OK;
    double v_free = 0.0;
    if (nc > nr) {
        std::vector<char> held(nc, 0);
        for (int i = 0; i < nr; ++i) {
            if (col4row[i] < 0 || col4row[i] >= nc) return PM_ERR_INVALID_ARG;
            held[col4row[i]] = 1;
        }
        double vmax = -std::numeric_limits<double>::infinity();
        v_free = std::numeric_limits<double>::infinity();
        for (int j = 0; j < nc; ++j) {
            vmax = std::max(vmax, v[j]);
            if (!held[j]) v_free = std::min(v_free, v[j]);
        }
        if (vmax - v_free > delta) return PM_OK;          // a matched column priced above a free one: not a rectangular optimum
    }
    rep->optimal = 1;
    if (n_tight > cap || eps > eps_collect) return PM_OK;
    // the near-tight entries, filtered by eps
    std::vector<int32_t> tight((size_t)std::max(n_tight, 1) * 2);
    std::vector<double> red((size_t)std::max(n_tight, 1));
    if (n_tight > 0) {
        char *st2 = (char *)g_stage.need((size_t)n_tight * 16);
        if (!st2) return PM_ERR_WORKSPACE;
        PM_HIP(hipMemcpyAsync(st2, dws + L.tight, (size_t)n_tight * 8, hipMemcpyDeviceToHost, s));
        PM_HIP(hipMemcpyAsync(st2 + (size_t)n_tight * 8, dws + L.red, (size_t)n_tight * 8, hipMemcpyDeviceToHost, s));
        PM_HIP(hipStreamSynchronize(s));
        std::memcpy(tight.data(), st2, (size_t)n_tight * 8);
        std::memcpy(red.data(), st2 + (size_t)n_tight * 8, (size_t)n_tight * 8);
    }
    int kept = 0;
    for (int e = 0; e < n_tight; ++e)
        if (red[e] <= eps) { tight[2 * (size_t)kept] = tight[2 * (size_t)e]; tight[2 * (size_t)kept + 1] = tight[2 * (size_t)e + 1]; ++kept; }
    rep->tight_within_eps = kept;
    const int rc = pm_lsap_unique(nr, nc, col4row, v, v_free, eps, tight.data(), kept);
    if (rc < 0) return rc;
    rep->unique = rc == 1;
    if (tight_out && tight_capacity > 0) {
        const int m = std::min(kept, tight_capacity);
        std::memcpy(tight_out, tight.data(), (size_t)m * 8);
        rep->n_tight = kept <= tight_capacity ? kept : -1;             // -1: the caller's list is incomplete
    }
    rep->seconds_certify = now_s() - t0;
    return PM_OK;
}

int pm_lsap_solve_resident(const double *U, int nr, int nc, size_t ld, const pm_lsap_options *opt, double *u, double *v,
                           int32_t *col4row, pm_lsap_report *rep, void *dev_ws, size_t dev_ws_bytes, void *stream) {
    using namespace pm;
    if (!U || !opt || !u || !v || !col4row || !rep || nr <= 0 || nc < nr || ld < (size_t)nc) return PM_ERR_INVALID_ARG;
    const int k = opt->core_edges, kp = opt->price_edges;
    if (k <= 0 || kp <= 0 || k > RES_KMAX || kp > RES_KMAX || opt->max_pricing_rounds <= 0) return PM_ERR_INVALID_ARG;
    if (!dev_ws || ((uintptr_t)dev_ws & 255) || dev_ws_bytes < pm_lsap_resident_workspace(nr, nc)) return PM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const ResidentLayout L = resident_layout(nr, nc, RES_KMAX, tight_cap(nc));
    char *dws = (char *)dev_ws;
    std::memset(rep, 0, sizeof(*rep));
    rep->tight_within_eps = -1;
    rep->status = 2;
    const double t_begin = now_s();
    const int kk = std::max(k, kp);
    // pinned staging: candidates [nr][kk] (int32 + float64) and one vector of nc doubles
    const size_t st_cols = 0, st_costs = align_up((size_t)nr * kk * 4, 64), st_vec = st_costs + align_up((size_t)nr * kk * 8, 64);
    char *stage = (char *)g_stage.need(st_vec + (size_t)nc * 8 + (size_t)nr * 8);
    if (!stage) return PM_ERR_WORKSPACE;
    int32_t *cols = (int32_t *)(stage + st_cols);
    double *costs = (double *)(stage + st_costs);
    double *vec = (double *)(stage + st_vec);              // [nc] then [nr]
    std::vector<double> v0(nc, 0.0), safety(nr), dense_min(nr), uu(nr), vv(nc);
    std::vector<int32_t> c4r(nr), first_col(nr);
    double t_dev = 0.0;
    // column reduction (square problems): v = column minima
    const bool square = nr == nc && opt->column_reduction;
    if (square) {
        const double t0 = now_s();
        PM_TRY(pm_lsap_col_min(U, nr, nc, ld, (double *)(dws + L.v), dws + L.colmin, pm_lsap_col_min_workspace(nr, nc), s));
        PM_HIP(hipMemcpyAsync(vec, dws + L.v, (size_t)nc * 8, hipMemcpyDeviceToHost, s));
        PM_HIP(hipStreamSynchronize(s));
        std::memcpy(v0.data(), vec, (size_t)nc * 8);
        t_dev += now_s() - t0;
        for (int j = 0; j < nc; ++j)
            if (!std::isfinite(v0[j])) { rep->status = 3; return PM_OK; }
    }
    int bad = 0;
    PM_TRY(select_to_host(U, nr, nc, ld, square ? v0.data() : nullptr, k, dws, L, cols, costs, &bad, s, &t_dev));
    if (bad) { rep->status = 3; return PM_OK; }
    // (when v is NULL the kernel ranks raw costs: v0 = 0 gives the same reduced costs)
    {   // the diagonal: row i -> column i, so that the core always holds a perfect matching
        const double t0 = now_s();
        PM_TRY(pm_lsap_diagonal(U, nr, nc, ld, 0, (double *)(dws + L.u), s));
        PM_HIP(hipMemcpyAsync(vec, dws + L.u, (size_t)nr * 8, hipMemcpyDeviceToHost, s));
        PM_HIP(hipStreamSynchronize(s));
        std::memcpy(safety.data(), vec, (size_t)nr * 8);
        t_dev += now_s() - t0;
    }
    double scale = 1e-300;
    for (int i = 0; i < nr; ++i) {
        for (int t = 0; t < k; ++t)
            if (cols[(size_t)i * k + t] >= 0) scale = std::max(scale, std::fabs(costs[(size_t)i * k + t]));
        scale = std::max(scale, std::fabs(safety[i]));
    }
    for (int j = 0; j < nc; ++j) scale = std::max(scale, std::fabs(v0[j]));
    const double delta = opt->rel_delta * scale;
    for (int i = 0; i < nr; ++i) {
        first_col[i] = cols[(size_t)i * k];
        dense_min[i] = costs[(size_t)i * k] - v0[std::max(first_col[i], 0)];
    }
    CoreHandle core(nr, nc);
    if (!core.h) return PM_ERR_WORKSPACE;
    PM_TRY(pm_lsap_core_add(core.h, k, cols, costs));
    {
        std::vector<int32_t> dcol(nr);
        for (int i = 0; i < nr; ++i) dcol[i] = i;
        PM_TRY(pm_lsap_core_add(core.h, 1, dcol.data(), safety.data()));
    }
    if (square) PM_TRY(pm_lsap_core_init_duals(core.h, dense_min.data(), v0.data(), first_col.data()));
    long stats4[4] = {0, 0, 0, 0};
    if (opt->auction && (double)(nc - nr) <= opt->a_max_free_columns * nc) {
        // width of the core: mean spread in reduced cost between a row's first and last core entry
        double sum = 0.0;
        long have = 0;
        for (int i = 0; i < nr; ++i) {
            const int last = cols[(size_t)i * k + (k - 1)];
            if (last < 0) continue;
            sum += (costs[(size_t)i * k + (k - 1)] - v0[last]) - (costs[(size_t)i * k] - v0[std::max(first_col[i], 0)]);
            ++have;
        }
        const double width = have ? sum / (double)have : 0.0;
        if (width > 0.0 && std::isfinite(width)) {
            const double t_a = now_s();
            double eps0 = opt->a_eps0 * width;
            for (int a_round = 0; a_round < opt->a_rounds; ++a_round) {
                const long budget = (long)(a_round == 0 ? opt->a_bids_per_row : opt->a_later_bids_per_row) * nr;
                long bids = 0;
                PM_TRY(pm_lsap_core_auction(core.h, eps0, opt->a_eps_min * width, opt->a_factor, budget, &bids));
                rep->bids = bids;
                if (a_round + 1 == opt->a_rounds) break;
                PM_TRY(pm_lsap_core_get(core.h, uu.data(), vv.data(), c4r.data(), stats4));
                int nf = 0;
                PM_TRY(select_to_host(U, nr, nc, ld, vv.data(), kp, dws, L, cols, costs, &nf, s, &t_dev));
                int violated = 0;
                PM_TRY(pm_lsap_core_reprice(core.h, kp, cols, costs, delta, &violated));
                if (a_round < 8) rep->auction_violated[a_round] = violated;
                rep->n_auction_violated = std::min(a_round + 1, 8);
                if ((double)violated <= opt->a_stop_below * nr) break;
                eps0 = opt->a_later_eps0 * width;
            }
            rep->seconds_auction = now_s() - t_a;
        }
    }
    int rounds = 0;
    while (true) {
        const double t_s = now_s();
        const int rc = pm_lsap_core_solve(core.h);
        rep->seconds_core += now_s() - t_s;
        if (rc == PM_ERR_UNSUPPORTED) { rep->status = 4; return PM_OK; }       // infeasible
        PM_TRY(rc);
        PM_TRY(pm_lsap_core_get(core.h, uu.data(), vv.data(), c4r.data(), stats4));
        int nf = 0;
        PM_TRY(select_to_host(U, nr, nc, ld, vv.data(), kp, dws, L, cols, costs, &nf, s, &t_dev));
        int violated = 0;
        PM_TRY(pm_lsap_core_reprice(core.h, kp, cols, costs, delta, &violated));
        if (rounds < 32) rep->violated_per_round[rounds] = violated;
        ++rounds;
        if (violated == 0) break;
        if (rounds >= opt->max_pricing_rounds) { rep->rounds = rounds; return PM_OK; }
    }
    rep->rounds = rounds;
    rep->edges = stats4[0]; rep->steps = stats4[1]; rep->augmentations = stats4[2]; rep->dummy_scans = stats4[3];
    std::memcpy(u, uu.data(), (size_t)nr * 8);
    std::memcpy(v, vv.data(), (size_t)nc * 8);
    std::memcpy(col4row, c4r.data(), (size_t)nr * 4);
    rep->seconds_device = t_dev;
    rep->seconds_total = now_s() - t_begin;
    rep->status = 0;
    return PM_OK;
}

}  // extern "C"
