// Host-side construction of the level-ordered monotonic-sweep plan.
//
// Input: the tables the reference binds into its operator
// (scarlet/operator.py:62-96): weights[n_off][n_pix], flat offsets[n_off] and
// the sweep order dist_idx (pixels by increasing radius, peak excluded).
// The reference loop (scarlet/operators_pybind11.cc:14-36) is a Gauss-Seidel
// pass: pixel p reads neighbours p+offsets[i] with weights[i][p] > 0, some of
// which were already updated.  Here every pixel gets the smallest level that is
//   > the level of every neighbour it reads that is updated earlier in the order
//     (it must see the new value), and
//   > the level of every earlier pixel that reads it (they must see the old one).
// Processing level by level, all pixels of a level at once, then yields exactly
// the sequential result for any weight table, not only the radial ones.
#include <algorithm>

#include "common.h"

namespace smi {

bool build_sweep_plan(int32_t n_pix, const double *weights, const int32_t *offsets,
                      int32_t n_off, const int32_t *dist_idx, int32_t n_idx,
                      SweepPlanHost *out) {
    if (n_pix <= 0 || n_off <= 0 || n_idx < 0 || !weights || !offsets ||
        (n_idx > 0 && !dist_idx)) {
        set_error("sweep plan: bad table sizes");
        return false;
    }
    std::vector<int32_t> order(n_pix, -1), level(n_pix, -1);
    for (int32_t d = 0; d < n_idx; ++d) {
        const int32_t p = dist_idx[d];
        if (p < 0 || p >= n_pix || order[p] >= 0) {
            set_error("sweep plan: dist_idx must hold distinct pixel indices");
            return false;
        }
        order[p] = d;
    }
    int32_t n_levels = 0, max_terms = 0;
    std::vector<int32_t> terms(n_idx, 0);
    for (int32_t d = 0; d < n_idx; ++d) {
        const int32_t p = dist_idx[d];
        int32_t lv = 0, cnt = 0;
        for (int32_t i = 0; i < n_off; ++i) {
            if (weights[(int64_t)i * n_pix + p] > 0) {
                const int64_t n = (int64_t)p + offsets[i];
                if (n < 0 || n >= n_pix) {
                    set_error("sweep plan: weighted neighbour outside the image");
                    return false;
                }
                ++cnt;
                if (order[n] >= 0 && order[n] < d) lv = std::max(lv, level[n] + 1);
            }
            // earlier pixels that read p must do so before p changes
            const int64_t r = (int64_t)p - offsets[i];
            if (r >= 0 && r < n_pix && weights[(int64_t)i * n_pix + r] > 0 &&
                order[r] >= 0 && order[r] < d)
                lv = std::max(lv, level[r] + 1);
        }
        level[p] = lv;
        terms[d] = cnt;
        n_levels = std::max(n_levels, lv + 1);
        max_terms = std::max(max_terms, cnt);
    }
    if (max_terms == 0) max_terms = 1;

    SweepPlanHost &pl = *out;
    pl.n_entries = n_idx;
    pl.max_terms = max_terms;
    pl.level_start.assign(n_levels + 1, 0);
    for (int32_t d = 0; d < n_idx; ++d) pl.level_start[level[dist_idx[d]] + 1]++;
    for (int32_t l = 0; l < n_levels; ++l) pl.level_start[l + 1] += pl.level_start[l];
    std::vector<int32_t> cursor(pl.level_start.begin(), pl.level_start.end() - 1);
    pl.pix.assign(n_idx, 0);
    pl.cnt.assign(n_idx, 0);
    pl.nbr.assign((size_t)max_terms * n_idx, 0);
    pl.wt.assign((size_t)max_terms * n_idx, 0.0);
    for (int32_t d = 0; d < n_idx; ++d) {
        const int32_t p = dist_idx[d];
        const int32_t e = cursor[level[p]]++;
        pl.pix[e] = p;
        int32_t j = 0;
        for (int32_t i = 0; i < n_off; ++i) {
            const double wgt = weights[(int64_t)i * n_pix + p];
            if (wgt > 0) {
                pl.nbr[(size_t)j * n_idx + e] = p + offsets[i];
                pl.wt[(size_t)j * n_idx + e] = wgt;
                ++j;
            }
        }
        pl.cnt[e] = j;
        for (; j < max_terms; ++j) pl.nbr[(size_t)j * n_idx + e] = p;  // harmless
    }
    return true;
}

}  // namespace smi
