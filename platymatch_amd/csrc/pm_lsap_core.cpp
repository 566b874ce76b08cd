// pm_lsap_core.cpp — HOST side of the device-resident assignment solve: shortest augmenting paths on a SPARSE core of the
// cost matrix, with the dense matrix never leaving the GPU.
//
// The widget calls scipy.optimize.linear_sum_assignment on eight N x M float64 matrices (_dock_widget.py:604-611).  SciPy's
// solver (and its restatement pm_lsap_solve) scans a full cost row per Dijkstra step: at 20 000 nuclei that is 160 KB from
// DRAM per step and ~99 % of a registration's wall time, with the GPU idle.  An optimal assignment, however, lives almost
// entirely on each row's few cheapest entries.  So (platymatch_amd/lsap.py drives this):
//   1. the GPU selects ~k cheap entries per row (pm_lsap_row_select, one HBM pass) -> the core, a sparse bipartite graph;
//   2. this file solves the assignment problem on the core exactly (Dijkstra with a heap over k edges per row instead of M);
//   3. the GPU PRICES the dual solution against the whole matrix (again one pass): rows with an entry of negative reduced
//      cost hand their cheapest offenders back, the core grows, their duals are repaired and the rows re-augmented;
//   4. when no entry of the dense matrix violates dual feasibility the core's optimum IS the dense optimum (LP duality);
//      pm_lsap_certificate re-checks that on the device together with complementary slackness, and lists the entries whose
//      reduced cost is within eps of zero: if those admit no alternating cycle the optimum is unique with margin eps, hence
//      the assignment any exact solver — SciPy's included — returns.  Otherwise (ties: duplicate nuclei, symmetric clouds)
//      the caller falls back to pm_lsap_solve, SciPy's algorithm step for step.
// Rectangular problems (nr < nc) are squared with nc - nr dummy rows of zero cost, kept implicit (no edge storage): every
// column then ends matched, which is what makes step 3's "free the row and its column" repair valid.
//
// The algorithm is the textbook sparse Jonker-Volgenant / Hungarian augmentation (as in SciPy's solver, restricted to the
// core edges) — own code, no third-party source.  Plain C++, no GPU code; one instance per matrix, no global state.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <vector>

#include "../../include/platymatch_hip.h"

namespace {

struct Edge {
    int32_t col;
    double cost;
};

struct Core {
    int nr, nc;                                   // real rows <= columns; rows nr..nc-1 are implicit dummy rows (cost 0 to every column)
    std::vector<std::vector<Edge>> adj;           // real rows only
    struct Col {                                  // everything a relaxation touches about a column, on one cache line
        double v;                                 // dual
        double dist;                              // tentative distance of the current augmentation (valid if seen == stamp)
        int32_t seen, done;                       // stamps: reached / scanned in the current augmentation
    };
    std::vector<Col> col;                         // [nc]
    std::vector<double> u;                        // row duals [nc] (real + dummy rows)
    std::vector<int32_t> col4row, row4col;        // [nc] each, -1 = free
    std::vector<int32_t> free_rows;               // rows waiting for an augmentation
    // Dijkstra scratch, reset lazily through `stamp`
    std::vector<int32_t> pred;
    std::vector<int32_t> touched_rows, done_cols;
    int32_t stamp = 0;
    long edges = 0, steps = 0, dummy_scans = 0, augmentations = 0;

    Core(int nr_, int nc_) : nr(nr_), nc(nc_), adj(nr_), col(nc_, Col{0.0, 0.0, 0, 0}), u(nc_, 0.0), col4row(nc_, -1), row4col(nc_, -1),
                             pred(nc_) {
        free_rows.reserve(nc_);
        for (int i = 0; i < nc_; ++i) free_rows.push_back(i);
    }

    bool has_edge(int i, int j) const {
        for (const Edge &e : adj[i])
            if (e.col == j) return true;
        return false;
    }

    void add_edge(int i, int j, double c) {
        if (j < 0 || j >= nc || !(c < std::numeric_limits<double>::infinity())) return;
        if (has_edge(i, j)) return;
        adj[i].push_back({j, c});
        ++edges;
    }

    // One augmentation from free row `cur`.  Returns false if no free column is reachable (cannot happen once every real
    // row holds its safety edge and dummy rows see every column).
    bool augment(int cur) {
        using Item = std::pair<double, int32_t>;
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
        if (++stamp == std::numeric_limits<int32_t>::max()) {
            for (Col &c : col) c.seen = c.done = 0;
            stamp = 1;
        }
        touched_rows.clear();
        done_cols.clear();
        double min_val = 0.0;
        int i = cur, sink = -1;
        while (sink < 0) {
            touched_rows.push_back(i);
            const double base = min_val - u[i];
            if (i < nr) {
                for (const Edge &e : adj[i]) {
                    const int j = e.col;
                    Col &c = col[j];
                    if (c.done == stamp) continue;
                    const double d = (base + e.cost) - c.v;
                    if (c.seen != stamp || d < c.dist) {
                        c.seen = stamp;
                        c.dist = d;
                        pred[j] = i;
                        heap.push({d, j});
                    }
                }
            } else {                                  // dummy row: zero cost to every column
                ++dummy_scans;
                for (int j = 0; j < nc; ++j) {
                    Col &c = col[j];
                    if (c.done == stamp) continue;
                    const double d = base - c.v;
                    if (c.seen != stamp || d < c.dist) {
                        c.seen = stamp;
                        c.dist = d;
                        pred[j] = i;
                        heap.push({d, j});
                    }
                }
            }
            int j = -1;
            while (!heap.empty()) {
                const Item top = heap.top();
                heap.pop();
                if (col[top.second].done != stamp && top.first == col[top.second].dist) { j = top.second; min_val = top.first; break; }
            }
            if (j < 0) return false;
            ++steps;
            col[j].done = stamp;
            done_cols.push_back(j);
            if (row4col[j] < 0) sink = j;
            else i = row4col[j];
        }
        // dual update (the same as SciPy's: scanned rows rise, scanned columns fall, matched edges stay tight)
        u[cur] += min_val;
        for (int r : touched_rows)
            if (r != cur) u[r] += min_val - col[col4row[r]].dist;
        for (int j : done_cols) col[j].v -= min_val - col[j].dist;
        // flip the path
        int j = sink;
        while (true) {
            const int r = pred[j];
            row4col[j] = r;
            const int prev = col4row[r];
            col4row[r] = j;
            if (r == cur) break;
            j = prev;
        }
        ++augmentations;
        return true;
    }

    int solve() {
        // real rows first, dummy rows last; a dummy row facing a free column that was never scanned (v == 0, the largest
        // dual a column can have) takes it directly: that IS its shortest augmenting path, of length zero
        std::stable_sort(free_rows.begin(), free_rows.end());
        std::vector<int32_t> clean;
        bool clean_ready = false;
        for (size_t q = 0; q < free_rows.size(); ++q) {
            const int r = free_rows[q];
            if (col4row[r] >= 0) continue;
            if (r >= nr) {
                if (!clean_ready) {
                    for (int j = 0; j < nc; ++j)
                        if (row4col[j] < 0 && col[j].v == 0.0) clean.push_back(j);
                    clean_ready = true;
                }
                // valid only while the dummy's own dual is the untouched 0 and no column has v > 0 (v never rises)
                while (!clean.empty() && row4col[clean.back()] >= 0) clean.pop_back();
                if (!clean.empty() && u[r] == 0.0) {
                    const int j = clean.back();
                    clean.pop_back();
                    row4col[j] = r;
                    col4row[r] = j;
                    continue;
                }
            }
            if (!augment(r)) return PM_ERR_UNSUPPORTED;
        }
        free_rows.clear();
        return PM_OK;
    }

    // Pricing result for the real rows: cand_col/cand_cost [nr][k] are, per row, entries of the DENSE matrix among which
    // the one minimising cost - v[col] over the whole row is present (pm_lsap_row_select with the current v).  Rows whose
    // minimum reduced cost is below -delta are repaired: the offenders join the core, u[i] drops to the dense row minimum
    // (feasible again), the row and its column are freed for re-augmentation.
    int reprice(int k, const int32_t *cand_col, const double *cand_cost, double delta) {
        int violated = 0;
        for (int i = 0; i < nr; ++i) {
            double best = std::numeric_limits<double>::infinity();
            for (int t = 0; t < k; ++t) {
                const int j = cand_col[(size_t)i * k + t];
                if (j < 0) continue;
                const double red = cand_cost[(size_t)i * k + t] - col[j].v;
                if (red < best) best = red;
            }
            if (!(best - u[i] < -delta)) continue;
            ++violated;
            for (int t = 0; t < k; ++t) {
                const int j = cand_col[(size_t)i * k + t];
                if (j < 0) continue;
                if ((cand_cost[(size_t)i * k + t] - col[j].v) - u[i] < -delta) add_edge(i, j, cand_cost[(size_t)i * k + t]);
            }
            u[i] = best;
            const int j = col4row[i];
            if (j >= 0) { row4col[j] = -1; col4row[i] = -1; }
            free_rows.push_back(i);
        }
        return violated;
    }
};

}  // namespace

extern "C" {

void *pm_lsap_core_create(int nr, int nc) {
    if (nr <= 0 || nc < nr) return nullptr;
    try {
        return new Core(nr, nc);
    } catch (...) {
        return nullptr;
    }
}

void pm_lsap_core_destroy(void *h) { delete static_cast<Core *>(h); }

int pm_lsap_core_add(void *h, int k, const int32_t *cols, const double *costs) {
    Core *c = static_cast<Core *>(h);
    if (!c || k <= 0 || !cols || !costs) return PM_ERR_INVALID_ARG;
    try {
        for (int i = 0; i < c->nr; ++i)
            for (int t = 0; t < k; ++t) c->add_edge(i, cols[(size_t)i * k + t], costs[(size_t)i * k + t]);
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
    return PM_OK;
}

int pm_lsap_core_init_duals(void *h, const double *u, const double *v, const int32_t *argmin_col) {
    Core *c = static_cast<Core *>(h);
    if (!c || !u || !v || !argmin_col || c->nr != c->nc || c->augmentations != 0) return PM_ERR_INVALID_ARG;
    for (int j = 0; j < c->nc; ++j) c->col[j].v = v[j];
    for (int i = 0; i < c->nr; ++i) {
        c->u[i] = u[i];
        const int j = argmin_col[i];
        if (j >= 0 && j < c->nc && c->row4col[j] < 0 && c->has_edge(i, j)) {      // tight by construction: u[i] = cost - v[j]
            c->row4col[j] = i;
            c->col4row[i] = j;
        }
    }
    return PM_OK;
}

int pm_lsap_core_solve(void *h) {
    Core *c = static_cast<Core *>(h);
    if (!c) return PM_ERR_INVALID_ARG;
    try {
        return c->solve();
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
}

int pm_lsap_core_reprice(void *h, int k, const int32_t *cand_col, const double *cand_cost, double delta, int *n_violated) {
    Core *c = static_cast<Core *>(h);
    if (!c || k <= 0 || !cand_col || !cand_cost || !n_violated || !(delta >= 0.0)) return PM_ERR_INVALID_ARG;
    try {
        *n_violated = c->reprice(k, cand_col, cand_cost, delta);
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
    return PM_OK;
}

int pm_lsap_core_get(void *h, double *u, double *v, int32_t *col4row, long *stats4) {
    Core *c = static_cast<Core *>(h);
    if (!c || !u || !v || !col4row) return PM_ERR_INVALID_ARG;
    std::memcpy(u, c->u.data(), sizeof(double) * c->nr);
    for (int j = 0; j < c->nc; ++j) v[j] = c->col[j].v;
    std::memcpy(col4row, c->col4row.data(), sizeof(int32_t) * c->nr);
    if (stats4) { stats4[0] = c->edges; stats4[1] = c->steps; stats4[2] = c->augmentations; stats4[3] = c->dummy_scans; }
    return PM_OK;
}

// Uniqueness of a certified optimum: `tight` lists the non-matching entries (row, col) of the dense matrix whose reduced
// cost is within eps of zero (pm_lsap_certificate).  An alternative optimum within eps per edge exists only if those edges
// close an alternating cycle — or, for nr < nc, an alternating path that ends on a column no real row holds.  Digraph on
// the real rows plus one node F for "the free columns / dummy rows": i -> owner(col) for a tight (i, col), i -> F if col is
// free, F -> owner(col) if a dummy row is tight on col (v[col] >= v_free_level - eps, v_free_level = the dual the free
// columns carry).  Returns 1 if acyclic (unique), 0 if a cycle exists, < 0 on error.
int pm_lsap_unique(int nr, int nc, const int32_t *col4row, const double *v, double v_free_level, double eps, const int32_t *tight,
                   int n_tight) {
    if (nr <= 0 || nc < nr || !col4row || !v || n_tight < 0 || (n_tight > 0 && !tight)) return PM_ERR_INVALID_ARG;
    try {
        std::vector<int32_t> owner(nc, -1);
        for (int i = 0; i < nr; ++i) {
            if (col4row[i] < 0 || col4row[i] >= nc || owner[col4row[i]] >= 0) return PM_ERR_INVALID_ARG;
            owner[col4row[i]] = i;
        }
        const int F = nr, nodes = nr + 1;
        std::vector<std::vector<int32_t>> out(nodes);
        std::vector<int32_t> indeg(nodes, 0);
        auto link = [&](int a, int b) { out[a].push_back(b); ++indeg[b]; };
        for (int e = 0; e < n_tight; ++e) {
            const int i = tight[2 * e], j = tight[2 * e + 1];
            if (i < 0 || i >= nr || j < 0 || j >= nc) return PM_ERR_INVALID_ARG;
            if (owner[j] == i) continue;
            link(i, owner[j] >= 0 ? owner[j] : F);
        }
        if (nc > nr)                       // a dummy row (dual -v_free_level, cost 0) is tight on column j if v_free_level - v[j] <= eps
            for (int j = 0; j < nc; ++j)
                if (owner[j] >= 0 && v_free_level - v[j] <= eps) link(F, owner[j]);
        std::vector<int32_t> stack;
        for (int a = 0; a < nodes; ++a)
            if (indeg[a] == 0) stack.push_back(a);
        int removed = 0;
        while (!stack.empty()) {
            const int a = stack.back();
            stack.pop_back();
            ++removed;
            for (int b : out[a])
                if (--indeg[b] == 0) stack.push_back(b);
        }
        return removed == nodes ? 1 : 0;
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
}

}  // extern "C"
