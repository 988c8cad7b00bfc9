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
    // transposed operators for the adjoint (built when a fit attaches the resampler)
    float *At = nullptr;   // [C][Fy * Fx][n_a]
    float *P = nullptr;    // [Fx * n_b][Fx]
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
                    (void *)r->scratch, (void *)r->At, (void *)r->P})
        if (p) (void)hipFree(p);
    delete r;
}

// r->model (device, padded) -> r->out, on stream s
static int resampler_forward(Resampler *r, hipStream_t s) {
    const size_t plane = (size_t)r->Fy * r->Fx;
    for (int c = 0; c < r->C; ++c) {
        int rc = gemm(r->model + c * plane, r->Pt, r->B, r->scratch, r->scratch_elems, r->Fy,
                      r->Fx * r->n_b, r->Fx, s);
        if (rc) return rc;
        rc = gemm(r->A + (size_t)c * r->n_a * plane, r->B, r->out + (size_t)c * r->n_a * r->n_b,
                  r->scratch, r->scratch_elems, r->n_a, r->n_b, (int)plane, s);
        if (rc) return rc;
    }
    return SMI_OK;
}

int resampler_render(Resampler *r, const float *model, float *out) {
    const size_t plane = (size_t)r->Fy * r->Fx;
    SMI_HIP(hipMemcpy(r->model, model, r->C * plane * sizeof(float), hipMemcpyHostToDevice));
    int rc = resampler_forward(r, nullptr);
    if (rc) return rc;
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipDeviceSynchronize());
    SMI_HIP(hipMemcpy(out, r->out, (size_t)r->C * r->n_a * r->n_b * sizeof(float),
                      hipMemcpyDeviceToHost));
    return SMI_OK;
}

// ---------------------------------------------------------------------------------------
// The low-resolution observation as a term of the fit (Blend._loss_func sums the
// observations' log-likelihoods, blend.py:265-271): rendering as above, residual
// w (m - d) and chi^2, and the transpose of the two products back to the model frame,
//
//   T[(y, x), b] = sum_a A[a, (y, x)] r[a, b]           (F_y F_x x n_a) . (n_a x n_b)
//   g[y, x']     = sum_{(x, b)} T[y, (x, b)] P[(x, b), x']   (F_y x F_x n_b) . (F_x n_b x F_x)
//
// One blend only: every blend of a batch would need its own operators.
// ---------------------------------------------------------------------------------------
namespace {

__global__ void transpose_kernel(const float *in, float *out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8)
        if (r0 + j < rows && c < cols) tile[j][threadIdx.x] = in[(int64_t)(r0 + j) * cols + c];
    __syncthreads();
    const int r = r0 + threadIdx.x, c0 = blockIdx.x * 32;
    for (int j = threadIdx.y; j < 32; j += 8)
        if (c0 + j < cols && r < rows) out[(int64_t)(c0 + j) * rows + r] = tile[threadIdx.x][j];
}

void launch_transpose(const float *in, float *out, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0,
                       s, in, out, rows, cols);
}

// centred zero padding of the observed bands of the model cube (fft._pad, fft.py:82-113)
__global__ void lowres_pad_kernel(const float *P, int Py, int Px, const int32_t *channels, int H,
                                  int W, int y0, int x0, float *padded, int Fy, int Fx) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Fy * Fx) return;
    const int y = i / Fx - y0, x = i % Fx - x0;
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    padded[(int64_t)c * Fy * Fx + i] = in ? P[((int64_t)channels[c] * Py + y) * Px + x] : 0.f;
}

// resid = w (m - d); term = log_norm + chi^2 / 2 (observation.py:147-186)
__global__ __launch_bounds__(1024) void lowres_residual_kernel(const float *rendered,
                                                               const float *data,
                                                               const float *weights, float *resid,
                                                               int n, double log_norm,
                                                               double *term) {
    __shared__ double part[16];
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float d = rendered[i] - data[i], w = weights[i];
        resid[i] = w * d;
        t += (double)w * d * d;
    }
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int i = 0; i < 16; ++i) total += part[i];
        term[0] = log_norm + 0.5 * total;
    }
}

__global__ void lowres_add_kernel(const float *gpad, int Fy, int Fx, int y0, int x0,
                                  const int32_t *channels, int H, int W, float *Q, int Py,
                                  int Px) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    Q[((int64_t)channels[c] * Py + y) * Px + x] +=
        gpad[((int64_t)c * Fy + y + y0) * Fx + x + x0];
}

}  // namespace

struct LowRes {
    Resampler *r = nullptr;  // not owned
    int H = 0, W = 0, y0 = 0, x0 = 0;
    int32_t *channels = nullptr;
    float *data = nullptr, *weights = nullptr, *resid = nullptr, *gpad = nullptr;
    double *term = nullptr;  // slot of the batch's array of extra loss terms (not owned)
    double log_norm = 0.0;
};

void lowres_destroy(LowRes *l) {
    if (!l) return;
    for (void *p : {(void *)l->channels, (void *)l->data, (void *)l->weights, (void *)l->resid,
                    (void *)l->gpad})
        if (p) (void)hipFree(p);
    delete l;
}

int lowres_create(Resampler *r, const int32_t *channels, const float *data, const float *weights,
                  double log_norm, int H, int W, double *term_slot, LowRes **out) {
    SMI_REQUIRE(r->Fy >= H && r->Fx >= W, "resampler FFT shape smaller than the model frame");
    auto *l = new LowRes;
    *out = l;
    l->r = r; l->H = H; l->W = W; l->log_norm = log_norm;
    l->y0 = (r->Fy - H + 1) / 2;
    l->x0 = (r->Fx - W + 1) / 2;
    const size_t plane = (size_t)r->Fy * r->Fx, n = (size_t)r->C * r->n_a * r->n_b;
    SMI_HIP(hipMalloc((void **)&l->channels, r->C * sizeof(int32_t)));
    SMI_HIP(hipMalloc((void **)&l->data, n * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&l->weights, n * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&l->resid, n * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&l->gpad, r->C * plane * sizeof(float)));
    l->term = term_slot;
    SMI_HIP(hipMemcpy(l->channels, channels, r->C * sizeof(int32_t), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(l->data, data, n * sizeof(float), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(l->weights, weights, n * sizeof(float), hipMemcpyHostToDevice));
    SMI_HIP(hipMemset(l->term, 0, sizeof(double)));
    if (!r->At) {
        SMI_HIP(hipMalloc((void **)&r->At, (size_t)r->C * r->n_a * plane * sizeof(float)));
        SMI_HIP(hipMalloc((void **)&r->P, (size_t)r->Fx * r->Fx * r->n_b * sizeof(float)));
        for (int c = 0; c < r->C; ++c)
            launch_transpose(r->A + (size_t)c * r->n_a * plane, r->At + (size_t)c * r->n_a * plane,
                             r->n_a, (int)plane, nullptr);
        launch_transpose(r->Pt, r->P, r->Fx, r->Fx * r->n_b, nullptr);
        SMI_HIP(hipGetLastError());
        SMI_HIP(hipDeviceSynchronize());
    }
    return SMI_OK;
}

// model cube P [C_model][Py][Px] -> rendering (r->out), term, and with `backward` the
// gradient of chi^2 / 2 with respect to the padded observed bands (l->gpad)
int lowres_evaluate(LowRes *l, const float *P, int Py, int Px, int backward, hipStream_t s) {
    Resampler *r = l->r;
    const int plane = r->Fy * r->Fx, n = r->C * r->n_a * r->n_b;
    hipLaunchKernelGGL(lowres_pad_kernel, dim3((plane + 255) / 256, r->C), dim3(256), 0, s, P, Py,
                       Px, l->channels, l->H, l->W, l->y0, l->x0, r->model, r->Fy, r->Fx);
    int rc = resampler_forward(r, s);
    if (rc) return rc;
    hipLaunchKernelGGL(lowres_residual_kernel, dim3(1), dim3(1024), 0, s, r->out, l->data,
                       l->weights, l->resid, n, l->log_norm, l->term);
    if (!backward) return SMI_OK;
    for (int c = 0; c < r->C; ++c) {
        rc = gemm(r->At + (size_t)c * r->n_a * plane, l->resid + (size_t)c * r->n_a * r->n_b, r->B,
                  r->scratch, r->scratch_elems, plane, r->n_b, r->n_a, s);
        if (rc) return rc;
        rc = gemm(r->B, r->P, l->gpad + (size_t)c * plane, r->scratch, r->scratch_elems, r->Fy,
                  r->Fx, r->Fx * r->n_b, s);
        if (rc) return rc;
    }
    return SMI_OK;
}

void lowres_add_gradient(LowRes *l, float *Q, int Py, int Px, hipStream_t s) {
    Resampler *r = l->r;
    hipLaunchKernelGGL(lowres_add_kernel, dim3((l->H * l->W + 255) / 256, r->C), dim3(256), 0, s,
                       l->gpad, r->Fy, r->Fx, l->y0, l->x0, l->channels, l->H, l->W, Q, Py, Px);
}

int lowres_get_rendered(LowRes *l, float *out, hipStream_t s) {
    Resampler *r = l->r;
    SMI_HIP(hipMemcpyAsync(out, r->out, (size_t)r->C * r->n_a * r->n_b * sizeof(float),
                           hipMemcpyDeviceToHost, s));
    SMI_HIP(hipStreamSynchronize(s));
    return SMI_OK;
}

}  // namespace smi
