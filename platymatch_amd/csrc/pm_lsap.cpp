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
#include <cmath>
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

    std::vector<double> u(nr, 0), v(nc, 0), shortest(nc);
    std::vector<intptr_t> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
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
