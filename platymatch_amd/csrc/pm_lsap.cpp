// pm_lsap.cpp — HOST-side rectangular linear sum assignment, callable without the Python GIL.
//
// The widget solves eight assignment problems with scipy.optimize.linear_sum_assignment (_dock_widget.py:604-611);
// SciPy holds the GIL while it runs, so the eight solves are serial there.  This file restates SciPy's solver —
// third-party code absent from the reference tree: SciPy 1.15.3, scipy/optimize/rectangular_lsap/rectangular_lsap.cpp,
// the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment algorithms",
// IEEE Trans. Aerospace and Electronic Systems 52(4), 2016 — operation for operation (same scan order of the remaining
// columns, same tie rule preferring an unassigned column, same dual updates, same float64 expression
// ((minVal + c) - u) - v), so that the returned indices are the ones SciPy returns, ties included.  Parity is
// anchored on SciPy itself: tests/test_lsap.py compares on thousands of random, tied, rectangular and constant
// matrices and on the reference fixtures' assignment vectors.  Plain C++, no GPU code.
//
// Two implementations of the same algorithm, chosen at run time:
//  * the scalar one, SciPy's loops as they are (any x86-64; PM_LSAP_SCALAR=1 forces it);
//  * an AVX-512 one for hosts that have it (the MI355X boxes' EPYC 9575F do).  One Dijkstra step updates
//    shortest[] for ALL columns in column order with the same float64 expression per column -- scanned columns are
//    parked at shortest = +inf, v = -inf so that they neither update nor win -- and takes the minimum.  SciPy's tie
//    rule depends on the order of its `remaining` list; that order only matters when two columns attain the minimum
//    with equal values, so the step counts them and, only on such a tie, replays SciPy's scan over `remaining`
//    (rebuilt lazily from the log of removals).  Same column chosen in every step => same duals, same indices.
//
// The scalar solver below (augmenting_path, solve_scalar) is derived from SciPy's rectangular_lsap.cpp and is used under
// SciPy's BSD 3-Clause licence, whose notice follows; the AVX-512 path further down is this project's own code.
//
//   Copyright (c) 2001-2002 Enthought, Inc. 2003-2024, SciPy Developers.
//   All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the
//   following conditions are met:
//   1. Redistributions of source code must retain the above copyright notice, this list of conditions and the following
//      disclaimer.
//   2. Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the
//      following disclaimer in the documentation and/or other materials provided with the distribution.
//   3. Neither the name of the copyright holder nor the names of its contributors may be used to endorse or promote
//      products derived from this software without specific prior written permission.
//
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES,
//   INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
//   DISCLAIMED. IN NO EVENT SHALL THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL,
//   SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR
//   SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY,
//   WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE
//   OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
#include <cmath>
#include <cstdlib>
#include <immintrin.h>
#include <cstdint>
#include <numeric>
#include <algorithm>
#include <vector>

#include "../../include/platymatch_hip.h"

namespace {

intptr_t augmenting_path(intptr_t nc, const double *cost, std::vector<double> &u, std::vector<double> &v,
                         std::vector<intptr_t> &path, std::vector<intptr_t> &row4col, std::vector<double> &shortest,
                         intptr_t i, std::vector<char> &SR, std::vector<char> &SC, std::vector<intptr_t> &remaining,
                         double *p_min) {
    double min_val = 0;
    intptr_t num_remaining = nc;
    for (intptr_t it = 0; it < nc; it++) remaining[it] = nc - it - 1;   // reverse order: constant matrices give the identity
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(shortest.begin(), shortest.end(), INFINITY);
    intptr_t sink = -1;
    while (sink == -1) {
        intptr_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        const double *row = cost + i * nc;
        const double ui = u[i];
        for (intptr_t it = 0; it < num_remaining; it++) {
            const intptr_t j = remaining[it];
            const double r = min_val + row[j] - ui - v[j];
            if (r < shortest[j]) {
                path[j] = i;
                shortest[j] = r;
            }
            // among equal minima prefer a column that is still free (a new sink)
            if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
                lowest = shortest[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;   // infeasible
        const intptr_t j = remaining[index];
        if (row4col[j] == -1) sink = j;
        else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}


// The augmentation loop with SciPy's scalar scan.  nr <= nc, cost validated.
int solve_scalar(const double *cost, intptr_t nr, intptr_t nc, std::vector<intptr_t> &col4row) {
    std::vector<double> u(nr, 0), v(nc, 0), shortest(nc);
    std::vector<intptr_t> path(nc, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (intptr_t cur = 0; cur < nr; cur++) {
        double min_val;
        const intptr_t sink = augmenting_path(nc, cost, u, v, path, row4col, shortest, cur, SR, SC, remaining, &min_val);
        if (sink < 0) return PM_ERR_UNSUPPORTED;
        u[cur] += min_val;
        for (intptr_t i = 0; i < nr; i++)
            if (SR[i] && i != cur) u[i] += min_val - shortest[col4row[i]];
        for (intptr_t j = 0; j < nc; j++)
            if (SC[j]) v[j] -= min_val - shortest[j];
        intptr_t j = sink;
        while (true) {
            const intptr_t i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    return PM_OK;
}

// ---- AVX-512 path ---------------------------------------------------------------------------------------------------

struct Work {
    std::vector<double> sh, vv, fin;                 // shortest (working), v with scanned columns parked, shortest at scan time
    std::vector<int64_t> path, remaining, pos, removed;
    std::vector<intptr_t> rows_seen;
};

// sh[j] = min(sh[j], ((min_val + row[j]) - ui) - vv[j]), path[j] = i where it decreased; returns the minimum of sh,
// *count = number of columns attaining it, *where = the last of them.  sh, vv, path are padded to a multiple of 8.
__attribute__((target("avx512f"))) double step_avx512(const double *row, const double *vv, double *sh, int64_t *path, intptr_t nc,
                                                      double min_val, double ui, int64_t i, long *count, intptr_t *where) {
    const __m512d vmin = _mm512_set1_pd(min_val), vui = _mm512_set1_pd(ui);
    const __m512i vi = _mm512_set1_epi64(i);
    __m512d best = _mm512_set1_pd(INFINITY);
    const intptr_t full = nc & ~(intptr_t)7;
    intptr_t j = 0;
    for (; j < full; j += 8) {
        _mm_prefetch((const char *)(row + j + 128), _MM_HINT_T0);
        const __m512d s = _mm512_loadu_pd(sh + j);
        const __m512d r = _mm512_sub_pd(_mm512_sub_pd(_mm512_add_pd(vmin, _mm512_loadu_pd(row + j)), vui), _mm512_loadu_pd(vv + j));
        const __mmask8 k = _mm512_cmp_pd_mask(r, s, _CMP_LT_OQ);
        _mm512_mask_storeu_pd(sh + j, k, r);
        _mm512_mask_storeu_epi64(path + j, k, vi);
        best = _mm512_min_pd(best, _mm512_mask_mov_pd(s, k, r));
    }
    if (j < nc) {                       // the cost row itself is not padded: masked load (0), the padding of vv gives r = +inf
        const __mmask8 tail = (__mmask8)((1u << (nc - j)) - 1u);
        const __m512d s = _mm512_loadu_pd(sh + j);
        const __m512d r = _mm512_sub_pd(_mm512_sub_pd(_mm512_add_pd(vmin, _mm512_maskz_loadu_pd(tail, row + j)), vui), _mm512_loadu_pd(vv + j));
        const __mmask8 k = _mm512_cmp_pd_mask(r, s, _CMP_LT_OQ);
        _mm512_mask_storeu_pd(sh + j, k, r);
        _mm512_mask_storeu_epi64(path + j, k, vi);
        best = _mm512_min_pd(best, _mm512_mask_mov_pd(s, k, r));
    }
    const double m = _mm512_reduce_min_pd(best);
    const __m512d vm = _mm512_set1_pd(m);
    long cnt = 0;
    intptr_t w = -1;
    const intptr_t padded = (nc + 7) & ~(intptr_t)7;
    for (j = 0; j < padded; j += 8) {
        const unsigned k = _mm512_cmp_pd_mask(_mm512_loadu_pd(sh + j), vm, _CMP_EQ_OQ);
        if (k) {
            cnt += __builtin_popcount(k);
            w = j + 31 - __builtin_clz(k);
        }
    }
    *count = cnt;
    *where = w;
    return m;
}

// The augmentation loop of pm_lsap_solve with step_avx512 as the inner scan.  nr <= nc, cost validated.
int solve_avx512(const double *cost, intptr_t nr, intptr_t nc, std::vector<intptr_t> &col4row) {
    const intptr_t ncp = (nc + 7) & ~(intptr_t)7;
    std::vector<double> u(nr, 0), v(nc, 0);
    std::vector<intptr_t> row4col(nc, -1);
    Work W;
    W.sh.assign(ncp, INFINITY);
    W.vv.assign(ncp, -INFINITY);
    W.fin.resize(nc);
    W.path.assign(ncp, -1);
    W.remaining.resize(nc);
    W.pos.resize(nc);
    for (intptr_t cur = 0; cur < nr; cur++) {
        std::fill(W.sh.begin(), W.sh.begin() + nc, INFINITY);
        std::copy(v.begin(), v.end(), W.vv.begin());
        W.removed.clear();
        W.rows_seen.clear();
        bool have_order = false;
        intptr_t num_remaining = nc, i = cur, sink = -1;
        double min_val = 0;
        while (sink == -1) {
            W.rows_seen.push_back(i);
            long count;
            intptr_t j;
            double lowest = step_avx512(cost + i * nc, W.vv.data(), W.sh.data(), W.path.data(), nc, min_val, u[i], i, &count, &j);
            if (lowest == INFINITY) return PM_ERR_UNSUPPORTED;
            if (count > 1) {
                if (!have_order) {      // SciPy's scan order: columns reversed, then every removal so far as a swap with the last
                    for (intptr_t it = 0; it < nc; it++) {
                        W.remaining[it] = nc - it - 1;
                        W.pos[nc - it - 1] = it;
                    }
                    intptr_t nrem = nc;
                    for (int64_t jr : W.removed) {
                        const intptr_t p = W.pos[jr], last = W.remaining[--nrem];
                        W.remaining[p] = last;
                        W.pos[last] = p;
                    }
                    have_order = true;
                }
                intptr_t index = -1;
                double low = INFINITY;
                for (intptr_t it = 0; it < num_remaining; it++) {       // SciPy's comparison, on the updated values
                    const intptr_t jj = W.remaining[it];
                    const double s = W.sh[jj];
                    if (s < low || (s == low && row4col[jj] == -1)) {
                        low = s;
                        index = it;
                    }
                }
                j = W.remaining[index];
                lowest = low;
            } else {
                lowest = W.sh[j];
            }
            min_val = lowest;
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            W.fin[j] = W.sh[j];
            W.sh[j] = INFINITY;
            W.vv[j] = -INFINITY;
            W.removed.push_back(j);
            --num_remaining;
            if (have_order) {
                const intptr_t p = W.pos[j], last = W.remaining[num_remaining];
                W.remaining[p] = last;
                W.pos[last] = p;
            }
        }
        u[cur] += min_val;
        for (intptr_t r : W.rows_seen)
            if (r != cur) u[r] += min_val - W.fin[col4row[r]];
        for (int64_t j : W.removed) v[j] -= min_val - W.fin[j];
        intptr_t j = sink;
        while (true) {
            const intptr_t r = W.path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    return PM_OK;
}

bool use_avx512() {
    const char *e = std::getenv("PM_LSAP_SCALAR");
    if (e && e[0] == '1') return false;
    return __builtin_cpu_supports("avx512f");
}

}  // namespace

// cost: nr x nc row-major HOST array.  rows/cols: min(nr, nc) entries each, as scipy returns them (rows ascending).
// Returns PM_OK, PM_ERR_INVALID_ARG (NaN or -inf entry: SciPy raises "matrix contains invalid numeric entries")
// or PM_ERR_UNSUPPORTED (infeasible: SciPy raises "cost matrix is infeasible").
extern "C" int pm_lsap_solve(const double *cost_in, long nr_in, long nc_in, int64_t *rows, int64_t *cols) {
    if (!cost_in || !rows || !cols || nr_in < 0 || nc_in < 0) return PM_ERR_INVALID_ARG;
    intptr_t nr = nr_in, nc = nc_in;
    if (nr == 0 || nc == 0) return PM_OK;
    const bool transpose = nc < nr;
    std::vector<double> temp;
    const double *cost = cost_in;
    if (transpose) {
        temp.resize((size_t)nr * nc);
        for (intptr_t i = 0; i < nr; i++)
            for (intptr_t j = 0; j < nc; j++) temp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
        cost = temp.data();
    }
    for (size_t k = 0, n = (size_t)nr * nc; k < n; k++)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) return PM_ERR_INVALID_ARG;

    std::vector<intptr_t> col4row(nr, -1);
    const int rc = use_avx512() ? solve_avx512(cost, nr, nc, col4row) : solve_scalar(cost, nr, nc, col4row);
    if (rc != PM_OK) return rc;
    if (transpose) {
        std::vector<intptr_t> order(nr);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](intptr_t a, intptr_t b) { return col4row[a] < col4row[b]; });
        intptr_t k = 0;
        for (intptr_t idx : order) { rows[k] = col4row[idx]; cols[k] = idx; k++; }
    } else {
        for (intptr_t i = 0; i < nr; i++) { rows[i] = i; cols[i] = col4row[i]; }
    }
    return PM_OK;
}
