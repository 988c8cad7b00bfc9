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

#include <cstring>

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

// ---------------------------------------------------------------------------
// Ring schedule (common.h).  The builder accepts any tables in which every weighted neighbour
// of a pixel is one of its roles A .. D and precedes it in the sweep order -- then the lane
// program reproduces the sequential loop bit for bit -- and says no otherwise.
// ---------------------------------------------------------------------------
namespace {
struct Octant {
    bool major_x;
    int smaj, smin;
};
// rows: octant pairs sharing the +x, +y, -x, -y axis; halves: sign along the minor axis
const Octant kOctants[8] = {{true, +1, -1}, {true, +1, +1}, {false, +1, +1}, {false, +1, -1},
                            {true, -1, +1}, {true, -1, -1}, {false, -1, -1}, {false, -1, +1}};
// (dy, dx) of the roles A, B, C, D
void octant_roles(const Octant &o, int (&dy)[4], int (&dx)[4]) {
    if (o.major_x) {
        const int sx = o.smaj, sy = o.smin;
        const int ry[4] = {-sy, 0, sy, -sy}, rx[4] = {-sx, -sx, -sx, 0};
        for (int k = 0; k < 4; ++k) dy[k] = ry[k], dx[k] = rx[k];
    } else {
        const int sy = o.smaj, sx = o.smin;
        const int ry[4] = {-sy, -sy, -sy, 0}, rx[4] = {-sx, 0, sx, -sx};
        for (int k = 0; k < 4; ++k) dy[k] = ry[k], dx[k] = rx[k];
    }
}
}  // namespace

bool build_ring_plan(int32_t h, int32_t w, const double *weights, const int32_t *offsets,
                     int32_t n_off, const int32_t *dist_idx, int32_t n_idx, RingPlanHost *out) {
    const int32_t n_pix = h * w;
    if (h <= 0 || w <= 0 || n_off != 8 || n_idx != n_pix - 1 || !weights || !offsets || !dist_idx)
        return false;
    if (16 + 4 * (int64_t)n_pix > 65535) return false;  // 16-bit LDS addresses
    static const int ndy[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, ndx[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    for (int i = 0; i < 8; ++i)
        if (offsets[i] != ndy[i] * w + ndx[i]) return false;
    std::vector<int32_t> order(n_pix, -1);
    for (int32_t d = 0; d < n_idx; ++d) {
        const int32_t p = dist_idx[d];
        if (p < 0 || p >= n_pix || order[p] >= 0) return false;
        order[p] = d;
    }
    int32_t centre = -1;
    for (int32_t p = 0; p < n_pix; ++p)
        if (order[p] < 0) centre = p;  // (exactly one: n_idx distinct entries)
    const int cy = centre / w, cx = centre % w;
    const int rmax = std::max(std::max(cy, h - 1 - cy), std::max(cx, w - 1 - cx));
    if (rmax < 1 || rmax > 8 * kRingMaxPlanes * 3 - 1) return false;
    // at most 8 P rings of an octant are active at a level of the plain schedule: ceil((L+1)/3)
    // .. (L+1)/2 <= rmax.  One plane carries rings up to 31 with late starts (common.h), two
    // planes rings up to 47 on the plain schedule.
    const int planes = rmax <= 31 ? 1 : 2;
    if (rmax > (planes == 1 ? 31 : 47)) return false;
    const int span = 8 * planes;
    std::vector<int> start(rmax + 1, 0), lam(rmax + 1, 0);
    int first_late = 0;
    for (int r = 1; r <= rmax; ++r) {
        start[r] = 2 * r - 1;
        if (planes == 1 && r > 8) start[r] = std::max(start[r], start[r - 8] + r - 7);
        if (r > 1) {
            start[r] = std::max(start[r], start[r - 1] + 2);
            lam[r] = start[r] - start[r - 1] - 2;
        }
        if (lam[r] > 1) return false;  // (cannot happen up to ring 31)
        if (lam[r] && !first_late) first_late = r;
    }
    if (first_late && 16 + 4 * (int64_t)n_pix > kRingAddrMask) return false;

    RingPlanHost &rp = *out;
    rp.planes = planes;
    rp.rmax = rmax;
    rp.centre = centre;
    rp.n_steps = start[rmax] + rmax;
    rp.n_pad = (rp.n_steps + kRingUnroll - 1) / kRingUnroll * kRingUnroll;
    // the flagged part of the stream starts at least four steps before the first late ring
    // (its lanes need four steps of history), on a multiple of the unrolling
    rp.n_nat = first_late ? (start[first_late] - 5) / kRingUnroll * kRingUnroll : rp.n_pad;
    const size_t lanes = (size_t)(rp.n_pad + kRingAhead) * planes * 64;
    rp.wts.assign(lanes * 4, 0.f);
    rp.addr.assign(lanes, 0);
    rp.perm = 0;
    int role_idx[8][4];
    for (int o = 0; o < 8; ++o) {
        int dy[4], dx[4];
        octant_roles(kOctants[o], dy, dx);
        for (int k = 0; k < 4; ++k) role_idx[o][k] = (dy[k] + 1) * 3 + (dx[k] + 1) - ((dy[k] + 1) * 3 + (dx[k] + 1) > 4);
        // order of the sum = ascending neighbour index: A, B, C must come out ascending or
        // descending, D wherever
        const bool asc = role_idx[o][0] < role_idx[o][1] && role_idx[o][1] < role_idx[o][2];
        const bool desc = role_idx[o][0] > role_idx[o][1] && role_idx[o][1] > role_idx[o][2];
        if (!asc && !desc) return false;
        int pd = 0;
        for (int k = 0; k < 3; ++k) pd += role_idx[o][k] < role_idx[o][3];
        rp.perm |= (uint32_t)((asc ? 1 : 0) | (pd << 1)) << (3 * o);
    }
    int32_t covered = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int Y = y - cy, X = x - cx;
            if (!Y && !X) continue;
            const int r = std::max(std::abs(Y), std::abs(X)), j = std::min(std::abs(Y), std::abs(X));
            const int L = start[r] + j;  // 1-based level; step index L - 1
            const int32_t pix = y * w + x;
            bool placed = false;
            for (int o = 0; o < 8; ++o) {
                const Octant &oc = kOctants[o];
                const int maj = oc.major_x ? X : Y, mn = oc.major_x ? Y : X;
                if (std::abs(maj) != r || (maj > 0) != (oc.smaj > 0) || std::abs(mn) != j) continue;
                if (mn != 0 && (mn > 0) != (oc.smin > 0)) continue;
                const int q = r % span;
                const size_t e = ((size_t)(L - 1) * planes + q / 8) * 64 + (o / 2) * 16 + (o & 1) * 8 + q % 8;
                if (rp.addr[e]) return false;  // (cannot happen for rmax within the planes)
                rp.addr[e] = (uint16_t)(16 + 4 * pix);
                if (L - 1 >= rp.n_nat)
                    rp.addr[e] |= (lam[r] ? kRingLate : 0) | (j == 0 ? kRingAxis : 0) | (j == r ? kRingDiag : 0);
                int hits = 0;
                for (int k = 0; k < 4; ++k) {
                    const double wgt = weights[(int64_t)role_idx[o][k] * n_pix + pix];
                    if (wgt > 0) {
                        rp.wts[e * 4 + k] = (float)wgt;
                        ++hits;
                        // the sequential loop must have updated this neighbour already
                        const int32_t nb = pix + offsets[role_idx[o][k]];
                        if (nb < 0 || nb >= n_pix || (nb != centre && order[nb] > order[pix])) return false;
                    }
                }
                int weighted = 0;
                for (int i = 0; i < 8; ++i) weighted += weights[(int64_t)i * n_pix + pix] > 0;
                if (weighted != hits) return false;  // a weighted neighbour that is no role
                placed = true;
            }
            if (!placed) return false;
            ++covered;
        }
    return covered == n_idx;
}

bool ring_device_stream(const RingPlanHost &rp, std::vector<uint8_t> *out) {
    const size_t P = (size_t)rp.planes, steps = (size_t)rp.n_pad + kRingAhead;
    if (rp.addr.size() != steps * P * 64) return false;
    out->assign(steps * P * (8 * 16 + 64 * 2), 0);
    float *wts = reinterpret_cast<float *>(out->data());
    uint16_t *addr = reinterpret_cast<uint16_t *>(out->data() + steps * P * 8 * 16);
    for (size_t sp = 0; sp < steps * P; ++sp)
        for (int m = 0; m < 8; ++m) {
            const size_t first = sp * 64 + m;  // octant 0
            for (int o = 0; o < 8; ++o) {
                const size_t e = sp * 64 + (size_t)o * 8 + m;
                if ((rp.addr[e] != 0) != (rp.addr[first] != 0)) return false;
                if (memcmp(&rp.wts[e * 4], &rp.wts[first * 4], 16) != 0) return false;
                addr[e] = rp.addr[e];
            }
            memcpy(&wts[(sp * 8 + m) * 4], &rp.wts[first * 4], 16);
        }
    return true;
}

}  // namespace smi
