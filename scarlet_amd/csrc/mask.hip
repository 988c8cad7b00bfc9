// Seam 1, monotonic mask operators of the reference's native extension
// (scarlet/operators_pybind11.cc:58-232), used by operator.prox_monotonic_mask
// (operator.py:131-176), MonotonicityConstraint(use_mask=True) and the lite
// initialisation with use_mask=True.
//
// get_valid_monotonic_pixels is a recursive 4-neighbour flood fill: a pixel is accepted
// from an accepted neighbour c if image(p) < image(c) + variance and image(p) > thresh
// (thresh only for the neighbours of the start pixel; the recursion drops it to 0).
// A failed test leaves the pixel unchecked, so it can still be accepted from another
// side: the accepted set is the set reachable from the start along such steps and does
// not depend on the visiting order.  Likewise the final `orphans` (unaccepted pixels next
// to an accepted one) and `bounds` (bounding box of the accepted pixels).  The kernel
// therefore relaxes the whole image in parallel until nothing changes -- bit-identical
// maps to the depth-first recursion.
//
// linear_interpolate_invalid_pixels walks its pixel list in order and every step sees
// the model / maps left by the previous ones (and starts a flood fill when it fills a
// pixel), so the list is processed sequentially by one lane while the workgroup joins
// in for the fills.  All of the reference's quirks are kept: the comma-operator
// conditions of the column branches, `i > 2` / `i < rows - 2`, the `else if` bound
// updates; the unguarded i+-1 / j+-1 accesses of the non-recursive branch are guarded.
#include "common.h"

namespace smi {
namespace {

constexpr int kT = 1024;

struct FillShared {
    int changed;
    int rmin, rmax, cmin, cmax;
};

template <typename T>
__device__ void flood_fill(int start, const T *image, int rows, int cols, uint8_t *unchecked,
                           uint8_t *orphans, int32_t *visited, int gen, double variance,
                           double thresh, int32_t *bounds, FillShared *sh) {
    const int tid = threadIdx.x, N = rows * cols;
    if (tid == 0) {
        visited[start] = gen;
        sh->rmin = bounds[0];
        sh->rmax = bounds[1];
        sh->cmin = bounds[2];
        sh->cmax = bounds[3];
    }
    __syncthreads();
    for (;;) {
        if (tid == 0) sh->changed = 0;
        __syncthreads();
        for (int p = tid; p < N; p += kT) {
            if (!unchecked[p] || visited[p] == gen) continue;
            const int r = p / cols, c = p - r * cols;
            const double val = (double)image[p];
            bool accept = false;
            const int nb[4] = {r > 0 ? p - cols : -1, r < rows - 1 ? p + cols : -1,
                               c > 0 ? p - 1 : -1, c < cols - 1 ? p + 1 : -1};
            for (int k = 0; k < 4; ++k) {
                const int q = nb[k];
                if (q < 0 || visited[q] != gen) continue;
                const double th = q == start ? thresh : 0.0;
                if (val < (double)image[q] + variance && val > th) accept = true;
            }
            if (accept) {
                visited[p] = gen;
                unchecked[p] = 0;
                sh->changed = 1;
            }
        }
        __syncthreads();
        const int again = sh->changed;
        __syncthreads();
        if (!again) break;
    }
    for (int p = tid; p < N; p += kT) {
        const int r = p / cols, c = p - r * cols;
        if (visited[p] == gen) {
            if (p != start) {
                orphans[p] = 0;
                atomicMin(&sh->rmin, r);
                atomicMax(&sh->rmax, r);
                atomicMin(&sh->cmin, c);
                atomicMax(&sh->cmax, c);
            }
        } else if (unchecked[p]) {
            const bool touched = (r > 0 && visited[p - cols] == gen) ||
                                 (r < rows - 1 && visited[p + cols] == gen) ||
                                 (c > 0 && visited[p - 1] == gen) ||
                                 (c < cols - 1 && visited[p + 1] == gen);
            if (touched) orphans[p] = 1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        bounds[0] = sh->rmin;
        bounds[1] = sh->rmax;
        bounds[2] = sh->cmin;
        bounds[3] = sh->cmax;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kT) void valid_pixels_kernel(int i, int j, const T *image, int rows,
                                                          int cols, uint8_t *unchecked,
                                                          uint8_t *orphans, int32_t *visited,
                                                          double variance, int32_t *bounds,
                                                          double thresh) {
    __shared__ FillShared sh;
    flood_fill<T>(i * cols + j, image, rows, cols, unchecked, orphans, visited, 1, variance, thresh,
                  bounds, &sh);
}

template <typename T>
__global__ __launch_bounds__(kT) void interpolate_kernel(const int32_t *row_idx,
                                                         const int32_t *col_idx, int n_idx,
                                                         uint8_t *unchecked, T *model, int rows,
                                                         int cols, uint8_t *orphans,
                                                         int32_t *visited, double variance,
                                                         int recursive, int32_t *bounds) {
    __shared__ FillShared sh;
    __shared__ int fill;
    int gen = 0;
#define AT(a, b) ((a) * cols + (b))
    for (int n = 0; n < n_idx; ++n) {
        const int i = row_idx[n], j = col_idx[n];
        if (threadIdx.x == 0) {
            fill = 0;
            if (unchecked[AT(i, j)]) {
                T total = 0;
                int valid = 0, pending = 0;
                unchecked[AT(i, j)] = 0;
                if (i < rows - 2 && model[AT(i + 2, j)] > model[AT(i + 1, j)]) {
                    if (unchecked[AT(i + 2, j)] || unchecked[AT(i + 1, j)]) {
                        pending = 1;
                    } else {
                        const T grad = model[AT(i + 2, j)] - model[AT(i + 1, j)];
                        total += model[AT(i + 1, j)] - grad;
                        valid += 1;
                    }
                }
                if (i > 2 && model[AT(i - 2, j)] > model[AT(i - 1, j)]) {
                    if (unchecked[AT(i - 2, j)] || unchecked[AT(i - 1, j)]) {
                        pending = 1;
                    } else {
                        const T grad = model[AT(i - 2, j)] - model[AT(i - 1, j)];
                        total += model[AT(i - 1, j)] - grad;
                        valid += 1;
                    }
                }
                if (j < cols - 2 && model[AT(i, j + 2)] > model[AT(i, j + 1)]) {
                    if (unchecked[AT(i, j + 1)]) {  // `unchecked(i,j+2), unchecked(i,j+1)`
                        pending = 1;
                    } else {
                        const T grad = model[AT(i, j + 2)] - model[AT(i, j + 1)];
                        total += model[AT(i, j + 1)] - grad;
                        valid += 1;
                    }
                }
                if (j > 2 && model[AT(i, j - 2)] > model[AT(i, j - 1)]) {
                    if (unchecked[AT(i, j - 1)]) {
                        pending = 1;
                    } else {
                        const T grad = model[AT(i, j - 2)] - model[AT(i, j - 1)];
                        total += model[AT(i, j - 1)] - grad;
                        valid += 1;
                    }
                }
                if (total > 0) {
                    model[AT(i, j)] = total / valid;
                    orphans[AT(i, j)] = 0;
                    if (i < bounds[0]) bounds[0] = i;
                    else if (i > bounds[1]) bounds[1] = i;
                    if (j < bounds[2]) bounds[2] = j;
                    else if (j > bounds[3]) bounds[3] = j;
                    if (recursive) {
                        fill = 1;
                    } else {
                        if (i > 0 && unchecked[AT(i - 1, j)]) orphans[AT(i - 1, j)] = 1;
                        if (i < rows - 1 && unchecked[AT(i + 1, j)]) orphans[AT(i + 1, j)] = 1;
                        if (j > 0 && unchecked[AT(i, j - 1)]) orphans[AT(i, j - 1)] = 1;
                        if (j < cols - 1 && unchecked[AT(i, j + 1)]) orphans[AT(i, j + 1)] = 1;
                    }
                } else if (!pending) {
                    orphans[AT(i, j)] = 1;
                    model[AT(i, j)] = 0;
                }
            }
            __threadfence_block();
        }
        __syncthreads();
        const int do_fill = fill;
        __syncthreads();
        if (do_fill)
            flood_fill<T>(AT(i, j), model, rows, cols, unchecked, orphans, visited, ++gen, variance,
                          0.0, bounds, &sh);
    }
#undef AT
}

template <typename T>
struct Buf {
    T *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)); }
};

}  // namespace

template <typename T>
int mask_valid_host_buffers(int32_t i, int32_t j, const T *image, int32_t rows, int32_t cols,
                            uint8_t *unchecked, uint8_t *orphans, double variance,
                            int32_t *bounds, double thresh) {
    const size_t N = (size_t)rows * cols;
    Buf<T> d_img;
    Buf<uint8_t> d_un, d_or;
    Buf<int32_t> d_vis, d_b;
    SMI_HIP(d_img.alloc(N));
    SMI_HIP(d_un.alloc(N));
    SMI_HIP(d_or.alloc(N));
    SMI_HIP(d_vis.alloc(N));
    SMI_HIP(d_b.alloc(4));
    SMI_HIP(hipMemcpy(d_img.p, image, N * sizeof(T), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(d_un.p, unchecked, N, hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(d_or.p, orphans, N, hipMemcpyHostToDevice));
    SMI_HIP(hipMemset(d_vis.p, 0, N * sizeof(int32_t)));
    SMI_HIP(hipMemcpy(d_b.p, bounds, 4 * sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(valid_pixels_kernel<T>, dim3(1), dim3(kT), 0, 0, i, j, d_img.p, rows, cols,
                       d_un.p, d_or.p, d_vis.p, variance, d_b.p, thresh);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipDeviceSynchronize());
    SMI_HIP(hipMemcpy(unchecked, d_un.p, N, hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(orphans, d_or.p, N, hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(bounds, d_b.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SMI_OK;
}

template <typename T>
int mask_interpolate_host_buffers(const int32_t *row_idx, const int32_t *col_idx, int32_t n_idx,
                                  uint8_t *unchecked, T *model, int32_t rows, int32_t cols,
                                  uint8_t *orphans, double variance, int32_t recursive,
                                  int32_t *bounds) {
    const size_t N = (size_t)rows * cols;
    Buf<T> d_model;
    Buf<uint8_t> d_un, d_or;
    Buf<int32_t> d_vis, d_b, d_r, d_c;
    SMI_HIP(d_model.alloc(N));
    SMI_HIP(d_un.alloc(N));
    SMI_HIP(d_or.alloc(N));
    SMI_HIP(d_vis.alloc(N));
    SMI_HIP(d_b.alloc(4));
    SMI_HIP(d_r.alloc(n_idx));
    SMI_HIP(d_c.alloc(n_idx));
    SMI_HIP(hipMemcpy(d_model.p, model, N * sizeof(T), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(d_un.p, unchecked, N, hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(d_or.p, orphans, N, hipMemcpyHostToDevice));
    SMI_HIP(hipMemset(d_vis.p, 0, N * sizeof(int32_t)));
    SMI_HIP(hipMemcpy(d_b.p, bounds, 4 * sizeof(int32_t), hipMemcpyHostToDevice));
    if (n_idx) {
        SMI_HIP(hipMemcpy(d_r.p, row_idx, n_idx * sizeof(int32_t), hipMemcpyHostToDevice));
        SMI_HIP(hipMemcpy(d_c.p, col_idx, n_idx * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(interpolate_kernel<T>, dim3(1), dim3(kT), 0, 0, d_r.p, d_c.p, n_idx, d_un.p,
                       d_model.p, rows, cols, d_or.p, d_vis.p, variance, recursive, d_b.p);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipDeviceSynchronize());
    SMI_HIP(hipMemcpy(model, d_model.p, N * sizeof(T), hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(unchecked, d_un.p, N, hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(orphans, d_or.p, N, hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(bounds, d_b.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SMI_OK;
}

template int mask_valid_host_buffers<float>(int32_t, int32_t, const float *, int32_t, int32_t,
                                            uint8_t *, uint8_t *, double, int32_t *, double);
template int mask_valid_host_buffers<double>(int32_t, int32_t, const double *, int32_t, int32_t,
                                             uint8_t *, uint8_t *, double, int32_t *, double);
template int mask_interpolate_host_buffers<float>(const int32_t *, const int32_t *, int32_t,
                                                  uint8_t *, float *, int32_t, int32_t, uint8_t *,
                                                  double, int32_t, int32_t *);
template int mask_interpolate_host_buffers<double>(const int32_t *, const int32_t *, int32_t,
                                                   uint8_t *, double *, int32_t, int32_t,
                                                   uint8_t *, double, int32_t, int32_t *);

}  // namespace smi
