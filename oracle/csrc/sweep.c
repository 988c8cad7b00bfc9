/*
 * Oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py): plain-C
 * restatement of the two native functions of the reference that sit on the
 * proximal-gradient path.  The reference implements them in C++ on Eigen
 * (scarlet/operators_pybind11.cc); Eigen is not available in this image, so
 * the reference extension is unbuildable here and these few loops restate it.
 *
 *   oracle_prox_weighted_monotonic_{f32,f64}  <- operators_pybind11.cc:14-36
 *   oracle_apply_filter_{f32,f64}             <- operators_pybind11.cc:39-56
 *
 * Compile with -ffp-contract=off so that a*b+c is two roundings, as in the
 * reference build (x86-64 baseline, no FMA contraction).
 */
#include <stdint.h>
#include <string.h>

#define DEFINE_SWEEP(NAME, T)                                                   \
    void NAME(T *img, const T *weights, const int32_t *offsets, int n_off,      \
              const int32_t *dist_idx, int n_idx, int n_pix, T min_gradient)    \
    {                                                                           \
        /* pixels in order of increasing radius; each is clipped to the      */ \
        /* weighted mean of its already-updated neighbours nearer the peak   */ \
        for (int d = 0; d < n_idx; ++d) {                                       \
            const int p = dist_idx[d];                                          \
            T ref = 0;                                                          \
            for (int i = 0; i < n_off; ++i) {                                   \
                const T w = weights[(long)i * n_pix + p];                       \
                if (w > 0) {                                                    \
                    ref += img[p + offsets[i]] * w;                             \
                }                                                               \
            }                                                                   \
            const T lim = ref * (1 - min_gradient);                             \
            if (lim < img[p]) img[p] = lim;                                     \
        }                                                                       \
    }

DEFINE_SWEEP(oracle_prox_weighted_monotonic_f32, float)
DEFINE_SWEEP(oracle_prox_weighted_monotonic_f64, double)

#define DEFINE_FILTER(NAME, T)                                                  \
    void NAME(const T *image, int H, int W, const T *values, int n_taps,        \
              const int32_t *y_start, const int32_t *y_end,                     \
              const int32_t *x_start, const int32_t *x_end, T *result)          \
    {                                                                           \
        memset(result, 0, sizeof(T) * (size_t)H * (size_t)W);                   \
        for (int n = 0; n < n_taps; ++n) {                                      \
            const int rows = H - y_start[n] - y_end[n];                         \
            const int cols = W - x_start[n] - x_end[n];                         \
            const T v = values[n];                                              \
            for (int r = 0; r < rows; ++r) {                                    \
                T *dst = result + (long)(y_start[n] + r) * W + x_start[n];      \
                const T *src = image + (long)(y_end[n] + r) * W + x_end[n];     \
                for (int c = 0; c < cols; ++c) dst[c] += v * src[c];            \
            }                                                                   \
        }                                                                       \
    }

DEFINE_FILTER(oracle_apply_filter_f32, float)
DEFINE_FILTER(oracle_apply_filter_f64, double)
