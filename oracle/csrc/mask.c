/*
 * Oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py): plain-C restatement of
 * the two monotonic-mask functions of the reference's native extension
 * (scarlet/operators_pybind11.cc; Eigen is absent, the extension is unbuildable here).
 *
 *   oracle_get_valid_monotonic_pixels_{f32,f64}        <- operators_pybind11.cc:61-121
 *   oracle_linear_interpolate_invalid_pixels_{f32,f64} <- operators_pybind11.cc:124-232
 *
 * Row-major images, bool maps as bytes, bounds = (min row, max row, min col, max col).
 * Restated with the reference's control flow: depth-first recursion in the order
 * down / up / left / right; the threshold only applies to the four neighbours of the
 * start pixel (the recursive calls omit it, so it falls back to 0); the
 * comma-operator conditions of the column branches (only the second operand counts);
 * the asymmetric index tests `i > 2`, `i < rows - 2`; the `else if` bound updates.
 * The non-recursive branch indexes i-1 / i+1 / j-1 / j+1 without bounds checks in the
 * reference (undefined behaviour at the border); here those accesses are guarded.
 */
#include <stdint.h>

#define AT(i, j) ((long)(i) * cols + (j))

#define DEFINE_VALID(NAME, T)                                                          \
    static void NAME##_rec(int i, int j, const T *image, int rows, int cols,           \
                           uint8_t *unchecked, uint8_t *orphans, double variance,      \
                           int32_t *bounds, double thresh)                             \
    {                                                                                  \
        if (i > 0 && unchecked[AT(i - 1, j)]) {                                        \
            if (image[AT(i - 1, j)] < image[AT(i, j)] + variance &&                    \
                image[AT(i - 1, j)] > thresh) {                                        \
                unchecked[AT(i - 1, j)] = 0;                                           \
                orphans[AT(i - 1, j)] = 0;                                             \
                if (i - 1 < bounds[0]) bounds[0] = i - 1;                              \
                NAME##_rec(i - 1, j, image, rows, cols, unchecked, orphans, variance,  \
                           bounds, 0);                                                 \
            } else {                                                                   \
                orphans[AT(i - 1, j)] = 1;                                             \
            }                                                                          \
        }                                                                              \
        if (i < rows - 1 && unchecked[AT(i + 1, j)]) {                                 \
            if (image[AT(i + 1, j)] < image[AT(i, j)] + variance &&                    \
                image[AT(i + 1, j)] > thresh) {                                        \
                unchecked[AT(i + 1, j)] = 0;                                           \
                orphans[AT(i + 1, j)] = 0;                                             \
                if (i + 1 > bounds[1]) bounds[1] = i + 1;                              \
                NAME##_rec(i + 1, j, image, rows, cols, unchecked, orphans, variance,  \
                           bounds, 0);                                                 \
            } else {                                                                   \
                orphans[AT(i + 1, j)] = 1;                                             \
            }                                                                          \
        }                                                                              \
        if (j > 0 && unchecked[AT(i, j - 1)]) {                                        \
            if (image[AT(i, j - 1)] < image[AT(i, j)] + variance &&                    \
                image[AT(i, j - 1)] > thresh) {                                        \
                unchecked[AT(i, j - 1)] = 0;                                           \
                orphans[AT(i, j - 1)] = 0;                                             \
                if (j - 1 < bounds[2]) bounds[2] = j - 1;                              \
                NAME##_rec(i, j - 1, image, rows, cols, unchecked, orphans, variance,  \
                           bounds, 0);                                                 \
            } else {                                                                   \
                orphans[AT(i, j - 1)] = 1;                                             \
            }                                                                          \
        }                                                                              \
        if (j < cols - 1 && unchecked[AT(i, j + 1)]) {                                 \
            if (image[AT(i, j + 1)] < image[AT(i, j)] + variance &&                    \
                image[AT(i, j + 1)] > thresh) {                                        \
                unchecked[AT(i, j + 1)] = 0;                                           \
                orphans[AT(i, j + 1)] = 0;                                             \
                if (j + 1 > bounds[3]) bounds[3] = j + 1;                              \
                NAME##_rec(i, j + 1, image, rows, cols, unchecked, orphans, variance,  \
                           bounds, 0);                                                 \
            } else {                                                                   \
                orphans[AT(i, j + 1)] = 1;                                             \
            }                                                                          \
        }                                                                              \
    }                                                                                  \
    void NAME(int i, int j, const T *image, int rows, int cols, uint8_t *unchecked,    \
              uint8_t *orphans, double variance, int32_t *bounds, double thresh)       \
    {                                                                                  \
        NAME##_rec(i, j, image, rows, cols, unchecked, orphans, variance, bounds,      \
                   thresh);                                                            \
    }

DEFINE_VALID(oracle_get_valid_monotonic_pixels_f32, float)
DEFINE_VALID(oracle_get_valid_monotonic_pixels_f64, double)

#define DEFINE_INTERP(NAME, VALID, T)                                                  \
    void NAME(const int32_t *row_indices, const int32_t *column_indices, int n_idx,    \
              uint8_t *unchecked, T *model, int rows, int cols, uint8_t *orphans,      \
              double variance, int recursive, int32_t *bounds)                         \
    {                                                                                  \
        for (int n = 0; n < n_idx; ++n) {                                              \
            const int i = row_indices[n], j = column_indices[n];                       \
            T total = 0;                                                               \
            int valid = 0, pending = 0;                                                \
            if (!unchecked[AT(i, j)]) continue;                                        \
            unchecked[AT(i, j)] = 0;                                                   \
            if (i < rows - 2 && model[AT(i + 2, j)] > model[AT(i + 1, j)]) {           \
                if (unchecked[AT(i + 2, j)] || unchecked[AT(i + 1, j)]) {              \
                    pending = 1;                                                       \
                } else {                                                               \
                    const T grad = model[AT(i + 2, j)] - model[AT(i + 1, j)];          \
                    total += model[AT(i + 1, j)] - grad;                               \
                    valid += 1;                                                        \
                }                                                                      \
            }                                                                          \
            if (i > 2 && model[AT(i - 2, j)] > model[AT(i - 1, j)]) {                  \
                if (unchecked[AT(i - 2, j)] || unchecked[AT(i - 1, j)]) {              \
                    pending = 1;                                                       \
                } else {                                                               \
                    const T grad = model[AT(i - 2, j)] - model[AT(i - 1, j)];          \
                    total += model[AT(i - 1, j)] - grad;                               \
                    valid += 1;                                                        \
                }                                                                      \
            }                                                                          \
            if (j < cols - 2 && model[AT(i, j + 2)] > model[AT(i, j + 1)]) {           \
                if (unchecked[AT(i, j + 1)]) { /* `a, b` evaluates to b */             \
                    pending = 1;                                                       \
                } else {                                                               \
                    const T grad = model[AT(i, j + 2)] - model[AT(i, j + 1)];          \
                    total += model[AT(i, j + 1)] - grad;                               \
                    valid += 1;                                                        \
                }                                                                      \
            }                                                                          \
            if (j > 2 && model[AT(i, j - 2)] > model[AT(i, j - 1)]) {                  \
                if (unchecked[AT(i, j - 1)]) {                                         \
                    pending = 1;                                                       \
                } else {                                                               \
                    const T grad = model[AT(i, j - 2)] - model[AT(i, j - 1)];          \
                    total += model[AT(i, j - 1)] - grad;                               \
                    valid += 1;                                                        \
                }                                                                      \
            }                                                                          \
            if (total > 0) {                                                           \
                model[AT(i, j)] = total / valid;                                       \
                orphans[AT(i, j)] = 0;                                                 \
                if (i < bounds[0]) bounds[0] = i;                                      \
                else if (i > bounds[1]) bounds[1] = i;                                 \
                if (j < bounds[2]) bounds[2] = j;                                      \
                else if (j > bounds[3]) bounds[3] = j;                                 \
                if (recursive) {                                                       \
                    VALID(i, j, model, rows, cols, unchecked, orphans, variance,       \
                          bounds, 0);                                                  \
                } else {                                                               \
                    if (i > 0 && unchecked[AT(i - 1, j)]) orphans[AT(i - 1, j)] = 1;   \
                    if (i < rows - 1 && unchecked[AT(i + 1, j)])                       \
                        orphans[AT(i + 1, j)] = 1;                                     \
                    if (j > 0 && unchecked[AT(i, j - 1)]) orphans[AT(i, j - 1)] = 1;   \
                    if (j < cols - 1 && unchecked[AT(i, j + 1)])                       \
                        orphans[AT(i, j + 1)] = 1;                                     \
                }                                                                      \
            } else if (pending) {                                                      \
                unchecked[AT(i, j)] = 0;                                               \
            } else {                                                                   \
                orphans[AT(i, j)] = 1;                                                 \
                model[AT(i, j)] = 0;                                                   \
            }                                                                          \
        }                                                                              \
    }

DEFINE_INTERP(oracle_linear_interpolate_invalid_pixels_f32,
              oracle_get_valid_monotonic_pixels_f32, float)
DEFINE_INTERP(oracle_linear_interpolate_invalid_pixels_f64,
              oracle_get_valid_monotonic_pixels_f64, double)
