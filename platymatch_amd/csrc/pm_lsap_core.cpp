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
// column then ends matched, which is what makes step 3's "free the row and its column" repair valid.  A free column carries
// v = 0 (the largest column dual) until a dummy row takes it; the pieces that keep this cheap: scan_row (a dummy row is
// scanned at most from one base per search), phase (never scans a spare column it does not match), reverse_augment (a column
// stranded below 0 is put right by a search from the column side).
// Step 2 starts from a warm start where the caller asks for one (auction): an eps-scaling forward auction over the core's
// edges, with reverse steps for spare columns, leaves duals close to the optimum's and most rows matched tight; the
// shortest-path search then has ~10 steps per row left instead of hundreds.  The auction proves nothing — exactness rests
// on the search that follows and on the certificate.
//
// The algorithm is the textbook sparse Jonker-Volgenant / Hungarian augmentation (as in SciPy's solver, restricted to the
// core edges) — own code, no third-party source.  Plain C++, no GPU code; one instance per matrix, no global state.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/platymatch_hip.h"

namespace {

struct Edge {
    int32_t col;
    double cost;
};

constexpr int PHASE_MIN_ROWS = 16;          // fewer free rows than this are augmented one by one ...
constexpr int PHASE_MAX_STEPS_PER_ROW = 384; // ... and so is the rest once a phase pays more Dijkstra steps than this per matched row

struct Core {
    int nr, nc;                                   // real rows <= columns; rows nr..nc-1 are implicit dummy rows (cost 0 to every column)
    std::vector<std::vector<Edge>> adj;           // real rows only
    std::vector<std::vector<Edge>> cadj;          // nr < nc: the same edges by column (Edge::col holds the ROW): reverse_augment
    struct Col {                                  // everything a relaxation touches about a column, on one cache line
        double v;                                 // dual
        double dist;                              // tentative distance of the current augmentation (valid if seen == stamp)
        int32_t seen, done;                       // stamps: reached / scanned in the current augmentation
        int32_t hpos;                             // position in the heap of the current augmentation (valid while seen == stamp and not done)
        int32_t pad_;
    };
    std::vector<Col> col;                         // [nc]
    std::vector<double> u;                        // row duals [nc] (real + dummy rows)
    std::vector<int32_t> col4row, row4col;        // [nc] each, -1 = free
    std::vector<int32_t> free_rows;               // rows waiting for an augmentation
    // Dijkstra scratch, reset lazily through `stamp`
    std::vector<int32_t> pred;
    std::vector<int32_t> touched_rows, done_cols, sinks;
    std::vector<int32_t> root_of_row, root_stamp;  // phase(): the tree a scanned row belongs to; per root, the stamp of the phase it found a sink in
    std::vector<double> row_dist;                  // phase(): the distance at which a row was reached (0 for the free rows)
    double dummy_base = 0.0;                       // lowest base a dummy row was scanned from in the current search
    int32_t dummy_base_stamp = -1;
    long phases = 0;
    bool cold = true;                              // no solve() has run yet
    int32_t stamp = 0;
    long edges = 0, steps = 0, dummy_scans = 0, augmentations = 0;

    Core(int nr_, int nc_) : nr(nr_), nc(nc_), adj(nr_), cadj(nc_ > nr_ ? nc_ : 0), col(nc_, Col{0.0, 0.0, 0, 0, 0, 0}), u(nc_, 0.0), col4row(nc_, -1), row4col(nc_, -1),
                             pred(nc_), root_of_row(nc_, -1), root_stamp(nc_, 0), row_dist(nc_, 0.0) {
        free_rows.reserve(nc_);
        for (int i = 0; i < nc_; ++i) free_rows.push_back(i);
        if (const char *e = std::getenv("PM_LSAP_COLUMN_REPAIR"))
            column_repair_mode = (e[0] == '0') ? 0 : (e[0] == 'f' ? 2 : 1);         // "0" | "force" | anything else: by the criterion
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
        if (nc > nr) cadj[j].push_back({i, c});
        ++edges;
    }

    // Indexed 4-ary min-heap of the reached, not yet scanned columns, keyed by their tentative distance: one entry per column,
    // an improvement moves the entry up (the searches here improve a column a dozen times before scanning it: a heap with
    // duplicates was the solver's largest cost).
    // (round 4: the key travels with the entry — a sift reads the four children's keys from ONE cache line of the heap array
    // instead of four scattered Col records: the searches here pop a column for every ~17 relaxations)
    struct HeapEntry { double d; int32_t j; int32_t pad_; };
    std::vector<HeapEntry> heap;

    void sift_up(int32_t at) {
        const HeapEntry e = heap[at];
        while (at > 0) {
            const int32_t parent = (at - 1) >> 2;
            if (!(e.d < heap[parent].d)) break;
            heap[at] = heap[parent];
            col[heap[at].j].hpos = at;
            at = parent;
        }
        heap[at] = e;
        col[e.j].hpos = at;
    }

    void heap_offer(int32_t j, bool fresh) {      // col[j].dist was just set (lower than before, or for the first time)
        if (fresh) {
            heap.push_back({col[j].dist, j, 0});
            col[j].hpos = (int32_t)heap.size() - 1;
        } else {
            heap[col[j].hpos].d = col[j].dist;
        }
        sift_up(col[j].hpos);
    }

    int32_t heap_pop() {                          // the closest reached column; -1 if none
        if (heap.empty()) return -1;
        const int32_t top = heap[0].j;
        const HeapEntry last = heap.back();
        heap.pop_back();
        const int32_t size = (int32_t)heap.size();
        if (size > 0) {
            int32_t at = 0;
            while (true) {
                const int32_t c0 = 4 * at + 1;
                if (c0 >= size) break;
                int32_t best = c0;
                double bd = heap[c0].d;
                const int32_t c1 = c0 + 4 < size ? c0 + 4 : size;
                for (int32_t c = c0 + 1; c < c1; ++c) {
                    const double cd = heap[c].d;
                    if (cd < bd) { bd = cd; best = c; }
                }
                if (!(bd < last.d)) break;
                heap[at] = heap[best];
                col[heap[at].j].hpos = at;
                at = best;
            }
            heap[at] = last;
            col[last.j].hpos = at;
        }
        return top;
    }

    // Relax every edge of row i from distance base + u[i] (base = the row's own distance - u[i]).
    void scan_row(int i, double base) {
        if (i < nr) {
            const std::vector<Edge> &row = adj[i];
            for (const Edge &e : row) __builtin_prefetch(&col[e.col], 1, 3);     // the columns are scattered: start every miss at once
            for (const Edge &e : row) {
                const int j = e.col;
                Col &c = col[j];
                if (c.done == stamp) continue;
                const double d = (base + e.cost) - c.v;
                const bool fresh = c.seen != stamp;
                if (fresh || d < c.dist) {
                    c.seen = stamp;
                    c.dist = d;
                    pred[j] = i;
                    heap_offer(j, fresh);
                }
            }
        } else {                                  // dummy row: zero cost to every column
            // Every dummy row offers every column the same thing, base - v[j]: one scanned from a base no lower than an
            // earlier one of this search cannot improve any column.  (Without this, a search that wanders over columns held
            // by dummy rows — all of them, once the real rows are placed — pays O(nc) per such column: measured 200 000 dense
            // scans for a 7 400 x 8 000 core, 5.7 s; with it 0.3 s.)
            if (dummy_base_stamp == stamp && !(base < dummy_base)) return;
            dummy_base_stamp = stamp;
            dummy_base = base;
            ++dummy_scans;
            for (int j = 0; j < nc; ++j) {
                Col &c = col[j];
                if (c.done == stamp) continue;
                const double d = base - c.v;
                const bool fresh = c.seen != stamp;
                if (fresh || d < c.dist) {
                    c.seen = stamp;
                    c.dist = d;
                    pred[j] = i;
                    heap_offer(j, fresh);
                }
            }
        }
    }

    void next_stamp() {
        heap.clear();
        if (++stamp == std::numeric_limits<int32_t>::max()) {
            for (Col &c : col) c.seen = c.done = 0;
            std::fill(root_stamp.begin(), root_stamp.end(), 0);
            dummy_base_stamp = -1;
            stamp = 1;
        }
    }

    // One PHASE: shortest paths from ALL the given free rows at once.  The search grows one tree per free row (a column
    // belongs to the tree of the row that gave it its distance; a matched row to the tree of its column), until every
    // reachable column is scanned or every tree has met a free column.  Then ONE dual update with the radius D reached
    // (scanned rows rise by D - their distance, scanned columns fall likewise: feasibility is kept, every edge of the forest
    // becomes tight, matched edges stay tight) and one augmentation per tree that met a free column — the trees are
    // vertex-disjoint, so the paths are.  Intermediate matchings need not be optimal for their size: feasible duals and
    // tight matched edges are all the final certificate asks for, and in the squared problem every column ends matched.
    // Late rows of a hard matrix each need a search of hundreds of columns when taken one by one (and visit the same columns
    // again and again); a phase visits each column once for all of them.  Returns the number of rows matched.
    int phase(const std::vector<int32_t> &sources) {
        next_stamp();
        touched_rows.clear();
        done_cols.clear();
        sinks.clear();
        for (int r : sources) {
            root_of_row[r] = r;
            row_dist[r] = 0.0;
            root_stamp[r] = 0;                     // no sink yet (root_stamp[r] == stamp marks "this tree has its sink")
            touched_rows.push_back(r);
            scan_row(r, 0.0 - u[r]);
        }
        int found = 0;
        double D = 0.0;
        const int want = (int)sources.size();
        while (found < want) {
            const int j = heap_pop();
            if (j < 0) break;
            const int root = root_of_row[pred[j]];
            // With more columns than rows some columns stay free for good, and a free column must keep the dual it has (v = 0,
            // the largest: what lets a dummy row take it without a search, and what the optimum of the rectangular problem
            // asks of it).  A second free column met by a tree that already has its sink would be scanned without being
            // matched and lose that dual in the update below: the phase ends just before it.
            if (nc > nr && row4col[j] < 0 && root_stamp[root] == stamp) break;
            D = col[j].dist;
            ++steps;
            col[j].done = stamp;
            done_cols.push_back(j);
            if (row4col[j] < 0) {                  // a free column: the first one a tree meets is its sink
                if (root_stamp[root] != stamp) {
                    root_stamp[root] = stamp;
                    sinks.push_back(j);
                    ++found;
                }
                continue;
            }
            const int i = row4col[j];
            root_of_row[i] = root;
            row_dist[i] = D;
            touched_rows.push_back(i);
            scan_row(i, D - u[i]);
        }
        if (found == 0) return 0;
        for (int r : touched_rows) u[r] += D - row_dist[r];
        for (int j : done_cols) col[j].v -= D - col[j].dist;
        for (int sink : sinks) {
            int j = sink;
            while (true) {
                const int r = pred[j];
                row4col[j] = r;
                const int prev = col4row[r];
                col4row[r] = j;
                if (prev < 0) break;               // reached the tree's free row
                j = prev;
            }
            ++augmentations;
        }
        ++phases;
        return found;
    }

    // One augmentation from free row `cur`.  Returns false if no free column is reachable (cannot happen once every real
    // row holds its safety edge and dummy rows see every column).
    bool augment(int cur) {
        next_stamp();
        touched_rows.clear();
        done_cols.clear();
        double min_val = 0.0;
        int i = cur, sink = -1;
        while (sink < 0) {
            touched_rows.push_back(i);
            scan_row(i, min_val - u[i]);
            const int j = heap_pop();
            if (j < 0) return false;
            min_val = col[j].dist;
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

    // Warm start for the cold solve: a forward auction with eps-scaling (Bertsekas) over the core's edges.
    // Rows bid for their cheapest column at current prices (value = cost + price, price = -v), raising its price by the
    // margin over their second choice plus eps and evicting its holder; eps shrinks by `factor` per round of the scaling,
    // prices carry over.  The auction itself proves nothing here: what it leaves behind are column duals close to the
    // optimum's, from which u = row minima of cost - v are FEASIBLE duals by construction, and an assignment of which every
    // pair whose column is the row's strict minimiser is tight and is kept; the rest (near-ties within eps, a few rows per
    // thousand) is left to the shortest-augmenting-path search, which from these duals needs a handful of steps per row where
    // the cold search needed hundreds.  With more columns than rows (nr < nc) a column nobody holds at the end of a round goes
    // back to price 0 — v = 0, what a free column carries in the optimum of the squared problem and what lets a dummy row take
    // it without a search (solve()); rows that then prefer it are freed and find it in one step.
    // max_bids (> 0) bounds the work: rows contesting near-equal columns raise prices by eps per bid (a "price war");
    // stopping early is harmless, the rows still unassigned simply stay free.  bids = work done.
    // RESUME (round 4; price_in / assigned_in not null): the auction continues from prices and a partial assignment produced
    // elsewhere — synchronous (Jacobi) rounds over the same core, which can do the wide part of an eps phase (thousands of
    // rows bidding at once) and leave the narrow, inherently sequential tails (eviction chains, price wars of a handful of
    // rows) to this loop; the library ships no such producer (a device form was simulated through this entry and not built,
    // DESIGN.md §4.5c).  The first phase here does NOT reset the assignment, only the rows still unassigned bid.  assigned_in[i] must be -1 or a column of row i's core edges, no column held twice (else INVALID_ARG).
    long bids = 0;
    int auction(double eps0, double eps_min, double factor, long max_bids, const double *price_in = nullptr,
                const int32_t *assigned_in = nullptr) {
        if (!cold || !(eps0 > 0.0) || !(eps_min > 0.0) || !(factor > 1.0)) return PM_ERR_INVALID_ARG;
        if ((price_in == nullptr) != (assigned_in == nullptr)) return PM_ERR_INVALID_ARG;
        const long stop_at = max_bids > 0 ? bids + max_bids : std::numeric_limits<long>::max();
        const int n = nr;
        for (int i = 0; i < n; ++i)
            if (adj[i].empty()) return PM_ERR_UNSUPPORTED;
        std::vector<double> price(nc);
        for (int j = 0; j < nc; ++j) price[j] = price_in ? price_in[j] : -col[j].v;
        std::vector<int32_t> owner(nc), assigned(n), queue(n);
        std::vector<double> held_cost(n, 0.0);            // cost of the edge a row holds
        bool resume = price_in != nullptr;
        if (resume) {
            std::fill(owner.begin(), owner.end(), -1);
            for (int i = 0; i < n; ++i) {
                const int a = assigned_in[i];
                assigned[i] = -1;
                if (a < 0) continue;
                if (a >= nc || owner[a] >= 0) return PM_ERR_INVALID_ARG;
                bool found = false;
                for (const Edge &e : adj[i])
                    if (e.col == a) { held_cost[i] = e.cost; found = true; break; }
                if (!found) return PM_ERR_INVALID_ARG;
                owner[a] = i;
                assigned[i] = a;
            }
        }
        const double lone = eps0 * 1e6;                    // a row with a single edge outbids everyone for it
        // the core by columns (for the reverse steps below and the final tightening)
        std::vector<int32_t> cstart(nc + 1, 0);
        for (int i = 0; i < n; ++i)
            for (const Edge &e : adj[i]) ++cstart[e.col + 1];
        for (int j = 0; j < nc; ++j) cstart[j + 1] += cstart[j];
        std::vector<int32_t> crow(cstart[nc]);
        std::vector<double> ccost(cstart[nc]);
        {
            std::vector<int32_t> fill(cstart.begin(), cstart.end() - 1);
            for (int i = 0; i < n; ++i)
                for (const Edge &e : adj[i]) { crow[fill[e.col]] = i; ccost[fill[e.col]++] = e.cost; }
        }
        // the core by rows as flat arrays (edges of a row contiguous, columns and costs apart): what the bid loop streams through
        std::vector<int32_t> rstart(n + 1, 0);
        for (int i = 0; i < n; ++i) rstart[i + 1] = rstart[i] + (int32_t)adj[i].size();
        std::vector<int32_t> rcol(rstart[n]);
        std::vector<double> rcost(rstart[n]);
        for (int i = 0; i < n; ++i) {
            int32_t t = rstart[i];
            for (const Edge &e : adj[i]) { rcol[t] = e.col; rcost[t++] = e.cost; }
        }
        std::vector<int32_t> stack;
        for (double eps = eps0;; eps = std::max(eps / factor, eps_min)) {
            size_t head = 0, count = 0;                    // ring buffer of the unassigned rows (first in, first out)
            if (resume) {                                  // the imported state: only the rows it left unassigned bid
                for (int i = 0; i < n; ++i)
                    if (assigned[i] < 0) queue[count++] = i;
                resume = false;
            } else {
                std::fill(owner.begin(), owner.end(), -1);
                std::fill(assigned.begin(), assigned.end(), -1);
                for (int i = 0; i < n; ++i) queue[i] = i;
                count = (size_t)n;
            }
            // A PRICE WAR — a handful of rows contesting near-equal columns, each bid worth eps — shows as a round whose tail, with
            // hardly any row still unassigned (<= 0.5 %), goes on for many bids per row of the problem (16 x n; PM_LSAP_WAR_TAIL
            // overrides, for experiments): the search settles those few rows in a fraction of the bids.  Measured, eight
            // assignments / one 100 000 x 100 000 hypothesis: no cut-off 1.12 s / 3.67 s, 64 x n 0.62 s / 1.99 s, 16 x n
            // 0.68 s / 1.34 s, 4 x n 0.46 s / 2.50 s.
            const size_t war_rows = std::max<size_t>(16, (size_t)n / 200);
            static const long war_tail = [] { const char *e = std::getenv("PM_LSAP_WAR_TAIL"); return e ? std::atol(e) : 16L; }();
            long tail_from = bids;                          // the bid count when more than war_rows rows were last unassigned
            bool war = false;
            while (count > 0 && bids < stop_at) {
                if (count > war_rows) tail_from = bids;
                else if (bids - tail_from > war_tail * n) { war = true; break; }
                const int i = queue[head];
                head = head + 1 == (size_t)n ? 0 : head + 1;
                --count;
                if (count > 0) {                            // the next bidder's edges: start their cache misses now
                    const int32_t nx = rstart[queue[head]];
                    __builtin_prefetch(&rcol[nx], 0, 3);
                    __builtin_prefetch(&rcost[nx], 0, 3);
                    __builtin_prefetch(&rcost[nx] + 8, 0, 3);
                }
                double best = std::numeric_limits<double>::infinity(), second = best, bc = 0.0;
                int bj = -1;
                // branch-free two-smallest scan (which edge wins is unpredictable: a mispredicted branch per edge cost more than
                // the arithmetic); first minimum on ties, as before
                for (int32_t t = rstart[i], te = rstart[i + 1]; t < te; ++t) {
                    const int32_t cj = rcol[t];
                    const double ct = rcost[t];
                    const double val = ct + price[cj];
                    const bool lt = val < best;
                    const double s2 = val < second ? val : second;
                    second = lt ? best : s2;
                    bj = lt ? cj : bj;
                    bc = lt ? ct : bc;
                    best = lt ? val : best;
                }
                const double margin = second < std::numeric_limits<double>::infinity() ? second - best : lone;
                price[bj] += margin + eps;
                ++bids;
                const int prev = owner[bj];
                owner[bj] = i;
                assigned[i] = bj;
                held_cost[i] = bc;
                if (prev >= 0) {
                    assigned[prev] = -1;
                    size_t tail = head + count;
                    if (tail >= (size_t)n) tail -= (size_t)n;
                    queue[tail] = prev;
                    ++count;
                }
            }
            // More columns than rows: a column nobody holds must end at price 0 (v = 0: what a free column carries in the optimum
            // and what lets a dummy row take it without a search).  One left with a higher price from an earlier round is
            // brought down by REVERSE steps: it goes to the row that gains most from it, at the price its runner-up would pay
            // (at least eps below the winner's indifference point, so every move is a strict gain and the loop ends); the
            // winner's old column is then without a holder and is treated the same way; a column nobody wants at any price >= 0
            // stays free at price 0.
            if (nc > nr && count == 0) {
                stack.clear();
                for (int j = 0; j < nc; ++j)
                    if (owner[j] < 0 && price[j] > 0.0) stack.push_back(j);
                while (!stack.empty() && bids < stop_at) {
                    const int j = stack.back();
                    stack.pop_back();
                    if (owner[j] >= 0 || !(price[j] > 0.0)) continue;
                    double g1 = -std::numeric_limits<double>::infinity(), g2 = g1, c1 = 0.0;
                    int k1 = -1;
                    for (int t = cstart[j]; t < cstart[j + 1]; ++t) {
                        const int k = crow[t];
                        const double g = (held_cost[k] + price[assigned[k]]) - ccost[t];     // the price at which row k is indifferent
                        if (g > g1) { g2 = g1; g1 = g; k1 = k; c1 = ccost[t]; }
                        else if (g > g2) g2 = g;
                    }
                    if (k1 < 0 || !(g1 > eps)) { price[j] = 0.0; continue; }
                    price[j] = std::max(0.0, std::min(g2, g1 - eps));
                    ++bids;
                    const int a = assigned[k1];
                    owner[a] = -1;
                    owner[j] = k1;
                    assigned[k1] = j;
                    held_cost[k1] = c1;
                    if (price[a] > 0.0) stack.push_back(a);
                }
                for (int j : stack)                         // (budget exhausted: the rest simply drops to 0; rows preferring them are freed below)
                    if (owner[j] < 0) price[j] = 0.0;
            } else if (nc > nr) {
                for (int j = 0; j < nc; ++j)
                    if (owner[j] < 0) price[j] = 0.0;
            }
            if (eps <= eps_min || bids >= stop_at || war) break;     // (out of budget — a price war among near-equal rows: the search takes over)
        }
        // Duals, and the assignment made tight to the bit.  u = row minima of cost - v: feasible.  A row whose column is its
        // strict minimiser is tight already.  One that holds a column within eps of its minimum (eps-complementary slackness
        // is all an auction promises) is made tight by RAISING that column's dual by the row's slack — allowed if no other row
        // of the core then sees a negative reduced cost on that column, which with a slack of at most eps_min is the rule;
        // the exceptions, and rows left unassigned by an exhausted budget, are freed for the search.  Freeing a row strands its
        // column with a dual below what a free column must carry (nr < nc: 0) — a search from a dummy row later, thousands of
        // steps — which is why slack is absorbed rather than rows freed.
        for (int j = 0; j < nc; ++j) { col[j].v = -price[j]; row4col[j] = -1; }
        for (int i = 0; i < n; ++i) {
            double best = std::numeric_limits<double>::infinity();
            for (const Edge &e : adj[i]) best = std::min(best, e.cost - col[e.col].v);
            u[i] = best;
        }
        free_rows.clear();
        for (int i = 0; i < n; ++i) {
            col4row[i] = -1;
            const int a = assigned[i];
            if (a < 0) { free_rows.push_back(i); continue; }
            const double cost_a = held_cost[i];
            double va = col[a].v;
            if (cost_a - va != u[i]) {                     // slack: raise v[a] until the pair is tight (a few ulps of search at most)
                va = cost_a - u[i];
                for (int guard = 0; guard < 8 && cost_a - va > u[i]; ++guard) va = std::nextafter(va, std::numeric_limits<double>::infinity());
                bool ok = cost_a - va <= u[i] && (nc == nr || va <= 0.0);
                for (int t = cstart[a]; ok && t < cstart[a + 1]; ++t)
                    if (crow[t] != i && ccost[t] - va < u[crow[t]]) ok = false;
                if (!ok) { free_rows.push_back(i); continue; }
                col[a].v = va;
                u[i] = cost_a - va;                        // (<= the row minimum it had: still feasible)
            }
            col4row[i] = a;
            row4col[a] = i;
        }
        for (int r = nr; r < nc; ++r) {                    // dummy rows: untouched
            u[r] = 0.0;
            col4row[r] = -1;
            free_rows.push_back(r);
        }
        return PM_OK;
    }

    // nr < nc: give the free dummy row `d` a column when no free column carries v = 0 any more, by a search from the COLUMN
    // side.  Free column `a` is stranded below 0 (its row was freed after the auction or by a pricing round and went elsewhere).
    // A search from the dummy row would first visit every column whose dual lies above v[a] — most of them, thousands of steps
    // (measured: 8.7e6 steps for a 47 000 x 50 000 core).  From the column the same shortest alternating path is found in the
    // transposed graph: a -> rows that have an edge into a -> the columns they hold -> ...; from every column j reached at
    // distance dist[j] the dummy row is one step of cost -v[j] away (its reduced cost on j), so the path's length is
    // D = min (dist[j] - v[j]) and only rows closer than the best D so far (at most -v[a], the direct step) are ever scanned:
    // a neighbourhood of a.  Dual update of the transposed search (scanned columns rise by D - dist, scanned rows fall
    // likewise), which keeps every dual feasible — v stays <= 0 precisely because D is that minimum — and makes the path tight;
    // the rows along it shift one column towards a, and the column j* at the far end, now at v = 0, goes to the dummy row.
    struct RowScratch { double dist; int32_t seen, done, from; };
    std::vector<RowScratch> rscr;
    std::vector<double> cdist;
    std::vector<int32_t> rev_rows, rev_cols;
    std::vector<std::pair<double, int32_t>> rheap;
    int32_t rstamp = 0;
    long reverse_searches = 0;

    bool reverse_augment(int a, int d) {
        if (rscr.empty()) { rscr.assign(nr, RowScratch{0.0, 0, 0, -1}); cdist.assign(nc, 0.0); }
        ++rstamp;
        ++reverse_searches;
        rev_rows.clear();
        rev_cols.clear();
        rheap.clear();
        auto worse = [](const std::pair<double, int32_t> &x, const std::pair<double, int32_t> &y) { return x.first > y.first; };
        double D = -col[a].v;                       // the direct step
        int jstar = a;
        auto scan_column = [&](int j, double dj) {
            for (const Edge &e : cadj[j]) {
                const int k = e.col;
                RowScratch &r = rscr[k];
                if (r.done == rstamp) continue;
                const double dk = dj + ((e.cost - u[k]) - col[j].v);
                if (r.seen != rstamp || dk < r.dist) {
                    r.seen = rstamp;
                    r.dist = dk;
                    r.from = j;
                    rheap.emplace_back(dk, k);
                    std::push_heap(rheap.begin(), rheap.end(), worse);
                }
            }
        };
        cdist[a] = 0.0;
        rev_cols.push_back(a);
        scan_column(a, 0.0);
        while (!rheap.empty()) {
            std::pop_heap(rheap.begin(), rheap.end(), worse);
            const double dk = rheap.back().first;
            const int k = rheap.back().second;
            rheap.pop_back();
            if (!(dk < D)) break;
            RowScratch &r = rscr[k];
            if (r.done == rstamp || dk > r.dist) continue;          // a stale entry
            r.done = rstamp;
            ++steps;
            const int b = col4row[k];
            if (b < 0) return false;                                  // (a free real row: not expected here, leave it to the forward search)
            rev_rows.push_back(k);
            cdist[b] = dk;                                            // the matched pair is tight: the column is as far as its row
            rev_cols.push_back(b);
            if (dk - col[b].v < D) { D = dk - col[b].v; jstar = b; }
            scan_column(b, dk);
        }
        for (int j : rev_cols) col[j].v += D - cdist[j];
        for (int k : rev_rows) u[k] -= D - rscr[k].dist;
        // shift the rows of the path one column towards a: the holder of j* moves to the column it was reached from, that
        // column's holder to the one before, ... until a is taken
        if (jstar != a) {
            int k = row4col[jstar];
            while (true) {
                const int to = rscr[k].from;
                const int holder = row4col[to];                       // (-1 for a)
                col4row[k] = to;
                row4col[to] = k;
                if (to == a) break;
                k = holder;
            }
        }
        col[jstar].v = 0.0;                                           // D - dist[j*] + v[j*] up to rounding
        row4col[jstar] = d;
        col4row[d] = jstar;
        u[d] = 0.0;
        ++augmentations;
        return true;
    }

    int solve() {
        // real rows first, dummy rows last; a dummy row facing a free column that was never scanned (v == 0, the largest
        // dual a column can have) takes it directly: that IS its shortest augmenting path, of length zero
        std::stable_sort(free_rows.begin(), free_rows.end());
        // The COLD solve takes the real rows in phases while there are many of them: late rows of a hard matrix each need a
        // search of hundreds of columns when taken one by one, a phase visits every column once for all of them (measured on
        // a 5 000 x 5 000 matrix: 0.29 s instead of 0.90 s).  Rows freed by a pricing round sit next to good duals and
        // re-augment in a handful of steps each, which a phase cannot beat (measured: 4x worse): those go one by one.
        std::vector<int32_t> sources;
        while (cold) {
            sources.clear();
            for (int r : free_rows)
                if (r < nr && col4row[r] < 0 && (sources.empty() || sources.back() != r)) sources.push_back(r);
            if ((int)sources.size() < PHASE_MIN_ROWS) break;
            const long before = steps;
            const int got = phase(sources);
            if (got == 0) return PM_ERR_UNSUPPORTED;
            if (steps - before > (long)PHASE_MAX_STEPS_PER_ROW * got) break;      // a lone search costs about this much per late row
            if (nc > nr && (size_t)got * 8 < sources.size()) break;               // cut short by spare free columns: every source is rescanned per phase
        }
        cold = false;
        // ... the last few, and the dummy rows, one at a time
        std::vector<int32_t> clean, stranded;
        bool clean_ready = false, stranded_ready = false;
        for (size_t q = 0; q < free_rows.size(); ++q) {
            const int r = free_rows[q];
            if (col4row[r] >= 0) continue;
            if (r >= nr) {
                if (!clean_ready) {
                    for (int j = 0; j < nc; ++j)
                        if (row4col[j] < 0 && col[j].v == 0.0) clean.push_back(j);
                    clean_ready = true;
                }
                // valid only while the dummy's own dual is the untouched 0 and no column has v > 0 (v never rises above 0)
                while (!clean.empty() && row4col[clean.back()] >= 0) clean.pop_back();
                if (!clean.empty() && u[r] == 0.0) {
                    const int j = clean.back();
                    clean.pop_back();
                    row4col[j] = r;
                    col4row[r] = j;
                    continue;
                }
                // no free column at v = 0 left: the free ones are stranded below it; search from one of them (reverse_augment)
                if (u[r] == 0.0) {
                    if (!stranded_ready) {
                        for (int j = 0; j < nc; ++j)
                            if (row4col[j] < 0 && col[j].v < 0.0) stranded.push_back(j);
                        stranded_ready = true;
                    }
                    while (!stranded.empty() && row4col[stranded.back()] >= 0) stranded.pop_back();
                    if (!stranded.empty()) {
                        const int a = stranded.back();
                        stranded.pop_back();
                        if (reverse_augment(a, r)) continue;
                    }
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
    //
    // COLUMN-SIDE repair (round 5).  The last augmentations of a solve can lift the duals of one huge alternating tree: every
    // row of the tree then undercuts the FEW columns outside it through dense entries the core never held — measured on a
    // 50 000 x 50 000 filtered solve: 48 179 of 50 000 rows violated after the first complete solve, all of them freed, the
    // second solve re-augmenting 48 k rows (0.12 s of a 0.49 s hypothesis).  Those violations meet on a handful of columns, and a
    // violation c - u[i] - v[j] < 0 is cured just as well from the column's side: v[j] drops by the column's worst violation
    // (every entry of column j only gains reduced cost: nothing else becomes infeasible, the dense matrix included), the
    // offenders join the core, and only the column's HOLDER loses its tight edge and is freed — one row per offending column
    // instead of one per offending row.  Taken when the matching is complete (so every column has a holder), the violated rows
    // are many and the offending columns held by real rows at most half as many; rows with an offender in any other column
    // (held by a dummy row of a rectangular problem) are repaired from the row side as before.  Either repair leaves feasible
    // duals and tight matched edges, which is all the searches and the final certificate ask for.
    int column_repair_mode = 1;                    // 0 never, 1 by the criterion, 2 whenever the matching is complete (PM_LSAP_COLUMN_REPAIR; tests)
    long column_repairs = 0;
    std::vector<double> col_viol;
    std::vector<int32_t> viol_rows, viol_cols;
    std::vector<unsigned char> row_by_column;

    int reprice(int k, const int32_t *cand_col, const double *cand_cost, double delta) {
        // pass 1: which rows are violated, and do their offenders all sit in columns a real row holds?
        viol_rows.clear();
        viol_cols.clear();
        row_by_column.assign((size_t)nr, 0);
        if (col_viol.size() != (size_t)nc) col_viol.assign((size_t)nc, 0.0);
        const bool complete = !cold && free_rows.empty() && column_repair_mode != 0;
        long by_column_rows = 0;
        for (int i = 0; i < nr; ++i) {
            double best = std::numeric_limits<double>::infinity();
            for (int t = 0; t < k; ++t) {
                const int j = cand_col[(size_t)i * k + t];
                if (j < 0) continue;
                const double red = cand_cost[(size_t)i * k + t] - col[j].v;
                if (red < best) best = red;
            }
            if (!(best - u[i] < -delta)) continue;
            viol_rows.push_back(i);
            if (!complete) continue;
            bool ok = true;
            for (int t = 0; t < k && ok; ++t) {
                const int j = cand_col[(size_t)i * k + t];
                if (j < 0) continue;
                if (!((cand_cost[(size_t)i * k + t] - col[j].v) - u[i] < -delta)) continue;
                const int holder = row4col[j];
                ok = holder >= 0 && holder < nr;
            }
            if (!ok) continue;
            row_by_column[i] = 1;
            ++by_column_rows;
            for (int t = 0; t < k; ++t) {
                const int j = cand_col[(size_t)i * k + t];
                if (j < 0) continue;
                const double red = (cand_cost[(size_t)i * k + t] - col[j].v) - u[i];
                if (!(red < -delta)) continue;
                if (col_viol[j] == 0.0) viol_cols.push_back(j);
                if (-red > col_viol[j]) col_viol[j] = -red;
            }
        }
        const bool by_column = complete && by_column_rows > 0 &&
                               (column_repair_mode == 2 || (by_column_rows >= 64 && 2 * (long)viol_cols.size() <= by_column_rows));
        // pass 2: the repairs
        for (int i : viol_rows) {
            const bool from_column = by_column && row_by_column[i];
            double best = std::numeric_limits<double>::infinity();
            for (int t = 0; t < k; ++t) {
                const int j = cand_col[(size_t)i * k + t];
                if (j < 0) continue;
                const double red = cand_cost[(size_t)i * k + t] - col[j].v;
                if (red < best) best = red;
                if (red - u[i] < -delta) add_edge(i, j, cand_cost[(size_t)i * k + t]);
            }
            if (from_column) continue;
            u[i] = best;
            const int j = col4row[i];
            if (j >= 0) { row4col[j] = -1; col4row[i] = -1; }
            free_rows.push_back(i);
        }
        if (by_column) {
            // (after the row-side repairs: they read v as the pricing pass saw it)
            for (int j : viol_cols) {
                col[j].v -= col_viol[j];
                const int holder = row4col[j];
                if (holder >= 0) { row4col[j] = -1; col4row[holder] = -1; free_rows.push_back(holder); }
            }
            ++column_repairs;
        }
        for (int j : viol_cols) col_viol[j] = 0.0;
        return (int)viol_rows.size();
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

int pm_lsap_core_init_state(void *h, const double *u, const double *v, const int32_t *col4row) {
    Core *c = static_cast<Core *>(h);
    if (!c || !u || !v || !col4row || c->augmentations != 0) return PM_ERR_INVALID_ARG;
    for (int j = 0; j < c->nc; ++j) { c->col[j].v = v[j]; c->row4col[j] = -1; }
    for (int i = 0; i < c->nr; ++i) {
        c->u[i] = u[i];
        c->col4row[i] = -1;
        const int j = col4row[i];
        if (j < 0) continue;
        if (j >= c->nc || c->row4col[j] >= 0 || !c->has_edge(i, j)) return PM_ERR_INVALID_ARG;
        c->row4col[j] = i;
        c->col4row[i] = j;
    }
    return PM_OK;
}

int pm_lsap_core_auction(void *h, double eps0, double eps_min, double factor, long max_bids, long *bids) {
    Core *c = static_cast<Core *>(h);
    if (!c) return PM_ERR_INVALID_ARG;
    try {
        const int rc = c->auction(eps0, eps_min, factor, max_bids);
        if (bids) *bids = c->bids;
        return rc;
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
}

int pm_lsap_core_auction_resume(void *h, const double *price, const int32_t *assigned, double eps0, double eps_min, double factor,
                                long max_bids, long *bids) {
    Core *c = static_cast<Core *>(h);
    if (!c || !price || !assigned) return PM_ERR_INVALID_ARG;
    try {
        const int rc = c->auction(eps0, eps_min, factor, max_bids, price, assigned);
        if (bids) *bids = c->bids;
        return rc;
    } catch (...) {
        return PM_ERR_WORKSPACE;
    }
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

long pm_lsap_core_column_repairs(void *h) {
    Core *c = static_cast<Core *>(h);
    return c ? c->column_repairs : -1;
}

int pm_lsap_core_get(void *h, double *u, double *v, int32_t *col4row, long *stats4) {
    Core *c = static_cast<Core *>(h);
    if (!c || !u || !v || !col4row) return PM_ERR_INVALID_ARG;
    std::memcpy(u, c->u.data(), sizeof(double) * c->nr);
    for (int j = 0; j < c->nc; ++j) v[j] = c->col[j].v;
    std::memcpy(col4row, c->col4row.data(), sizeof(int32_t) * c->nr);
    if (stats4) { stats4[0] = c->edges; stats4[1] = c->steps; stats4[2] = c->augmentations; stats4[3] = c->dummy_scans + 1000000 * c->phases; }
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
