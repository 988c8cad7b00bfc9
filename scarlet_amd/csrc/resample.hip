// Multi-resolution rendering (reference scarlet/renderer.py:262-547, ResolutionRenderer):
// a high-resolution model is convolved with the difference kernel and resampled onto a
// low-resolution pixel grid.  For unrotated grids the reference does this per band as
//
//   model_conv[y, x, b] = (model shifted along x by the sub-pixel position of LR column b)
//   out[a, b]           = sum_{y, x} op[a, y, x] * model_conv[y, x, b]
//
// with op[a] = difference kernel shifted along y to LR row a (renderer.py:478-545).  Both
// steps are linear in the model, so the host precomputes the two operators once
// (scarlet_amd/renderer.py) and a rendering is two matrix products per band:
//
//   B[(y, x), b] = sum_x' model[y, x'] * P[(x, b), x']      (F_y x F_x) . (F_x x F_x n_b)
//   out[a, b]    = sum_{(y, x)} A[a, (y, x)] * B[(y, x), b]  (n_a x F_y F_x) . (F_y F_x x n_b)
//
// Plain dense f32 products with f64 accumulation (the second one sums F_y F_x ~ 4e4
// terms); the long reduction is split over workgroups and reduced in a second kernel.
#include "common.h"

namespace smi {
namespace {

constexpr int kTile = 64, kTK = 16;

// C[M x N] (+)= A[M x K] . B[K x N], row major, one 64 x 64 tile per workgroup of 256
// threads (4 x 4 outputs per thread); blockIdx.z selects the K slice [z * kslice, ...),
// partial results go to Cpart[z][M][N] in double.
__global__ __launch_bounds__(256) void gemm_slices_kernel(const float *A, const float *B,
                                                          double *Cpart, int M, int N, int K,
                                                          int kslice) {
    __shared__ float As[kTK][kTile + 1], Bs[kTK][kTile + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * kTile, n0 = blockIdx.x * kTile;
    const int k_lo = blockIdx.z * kslice, k_hi = min(K, k_lo + kslice);
    double acc[4][4] = {};
    for (int k0 = k_lo; k0 < k_hi; k0 += kTK) {
        for (int i = threadIdx.x; i < kTile * kTK; i += 256) {
            const int m = i / kTK, kk = i % kTK;  // A tile: 64 rows x 16 k
            As[kk][m] = (m0 + m < M && k0 + kk < k_hi) ? A[(int64_t)(m0 + m) * K + k0 + kk] : 0.f;
            const int kb = i / kTile, n = i % kTile;  // B tile: 16 k x 64 columns
            Bs[kb][n] = (n0 + n < N && k0 + kb < k_hi) ? B[(int64_t)(k0 + kb) * N + n0 + n] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[kk][ty * 4 + i];
                b[i] = Bs[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += (double)a[i] * (double)b[j];
        }
        __syncthreads();
    }
    double *Cz = Cpart + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) Cz[(int64_t)m * N + n] = acc[i][j];
        }
}

__global__ void reduce_slices_kernel(const double *Cpart, float *C, int64_t MN, int n_slices) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) return;
    double t = 0.0;
    for (int z = 0; z < n_slices; ++z) t += Cpart[(int64_t)z * MN + i];
    C[i] = (float)t;
}

int gemm(const float *A, const float *B, float *C, double *scratch, size_t scratch_elems, int M,
         int N, int K, hipStream_t s) {
    const int tiles = ((M + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
    // enough K slices to occupy the chip when the output is small
    int n_slices = std::max(1, std::min((K + 255) / 256, 2048 / std::max(tiles, 1)));
    while ((size_t)n_slices * M * N > scratch_elems && n_slices > 1) --n_slices;
    SMI_REQUIRE((size_t)n_slices * M * N <= scratch_elems, "resampler scratch too small");
    int kslice = (K + n_slices - 1) / n_slices;
    kslice = (kslice + kTK - 1) / kTK * kTK;
    n_slices = (K + kslice - 1) / kslice;
    hipLaunchKernelGGL(gemm_slices_kernel, dim3((N + kTile - 1) / kTile, (M + kTile - 1) / kTile, n_slices),
                       dim3(256), 0, s, A, B, scratch, M, N, K, kslice);
    const int64_t MN = (int64_t)M * N;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, s,
                       scratch, C, MN, n_slices);
    return SMI_OK;
}

}  // namespace

struct Resampler {
    int C = 0, n_a = 0, n_b = 0, Fy = 0, Fx = 0;
    float *A = nullptr;    // [C][n_a][Fy * Fx]
    float *Pt = nullptr;   // [Fx][Fx * n_b]  (transposed shift operator, shared by the bands)
    float *model = nullptr, *B = nullptr, *out = nullptr;
    double *scratch = nullptr;
    size_t scratch_elems = 0;
};

int resampler_create(const float *A, const float *Pt, int C, int n_a, int n_b, int Fy, int Fx,
                     Resampler **out) {
    auto *r = new Resampler;
    r->C = C; r->n_a = n_a; r->n_b = n_b; r->Fy = Fy; r->Fx = Fx;
    const size_t nA = (size_t)C * n_a * Fy * Fx, nP = (size_t)Fx * Fx * n_b;
    const size_t nB = (size_t)Fy * Fx * n_b;
    r->scratch_elems = std::max(nB, (size_t)n_a * n_b * 2048);
    *out = r;
    SMI_HIP(hipMalloc((void **)&r->A, nA * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->Pt, nP * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->model, (size_t)C * Fy * Fx * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->B, nB * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->out, (size_t)C * n_a * n_b * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->scratch, r->scratch_elems * sizeof(double)));
    SMI_HIP(hipMemcpy(r->A, A, nA * sizeof(float), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(r->Pt, Pt, nP * sizeof(float), hipMemcpyHostToDevice));
    return SMI_OK;
}

void resampler_destroy(Resampler *r) {
    if (!r) return;
    for (void *p : {(void *)r->A, (void *)r->Pt, (void *)r->model, (void *)r->B, (void *)r->out,
                    (void *)r->scratch})
        if (p) (void)hipFree(p);
    delete r;
}

int resampler_render(Resampler *r, const float *model, float *out) {
    const size_t plane = (size_t)r->Fy * r->Fx;
    SMI_HIP(hipMemcpy(r->model, model, r->C * plane * sizeof(float), hipMemcpyHostToDevice));
    for (int c = 0; c < r->C; ++c) {
        int rc = gemm(r->model + c * plane, r->Pt, r->B, r->scratch, r->scratch_elems, r->Fy,
                      r->Fx * r->n_b, r->Fx, nullptr);
        if (rc) return rc;
        rc = gemm(r->A + (size_t)c * r->n_a * plane, r->B, r->out + (size_t)c * r->n_a * r->n_b,
                  r->scratch, r->scratch_elems, r->n_a, r->n_b, (int)plane, nullptr);
        if (rc) return rc;
    }
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipDeviceSynchronize());
    SMI_HIP(hipMemcpy(out, r->out, (size_t)r->C * r->n_a * r->n_b * sizeof(float),
                      hipMemcpyDeviceToHost));
    return SMI_OK;
}

}  // namespace smi
