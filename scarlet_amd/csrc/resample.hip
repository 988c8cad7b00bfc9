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
// The products run on the matrix cores: v_mfma_f32_32x32x2_f32 tiles (f32 inputs, f32
// accumulation = an fmaf chain in k order), staged through LDS.  The reductions are long
// (the second product sums F_y F_x ~ 4e4..2e5 terms), so K is cut into slices of a few
// hundred terms: inside a slice the accumulation is float32, across slices the partial
// tiles are summed in double by a second kernel -- that keeps the renderings within
// ~1e-7 of the reference's float64 evaluation and gives a small output enough workgroups
// to fill the chip.  The bands of an observation are one batched launch.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "common.h"

namespace smi {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 16;          // k depth of a block tile
constexpr int kPad = 4;          // LDS row padding (floats)
constexpr int kSliceTerms = 320; // float32 accumulation length inside a slice

// C_b[M x N] = A_b[M x K] . B_b[K x N] for b < n_batch (row major; the batch strides may
// be 0).  blockIdx.z = batch * n_slices + slice; slice z covers K in [z kslice, ...).
// n_slices == 1: the tile goes straight to C (float); otherwise to Cpart[b][z][M][N]
// (double) for reduce_slices_kernel.
//
// Block tile 64 TM x 64 TN, four wavefronts as 2 x 2, each with TM x TN tiles of
// v_mfma_f32_32x32x2_f32.  The operators of a rendering come in very different shapes --
// 300 x 15 000 outputs over 300 terms, 50 x 50 outputs over 90 000 terms, 90 000 x 50 over
// 50 -- so a dimension of up to 96 takes one MFMA tile per wavefront (round 2's single
// 128 x 128 tile spent 85 % of the matrix-core time of a 50 x 50 product on padding), and a
// wavefront skips the MFMA tiles that lie outside the matrix altogether (300 rows = 2.5 x 128:
// the last 64 rows of the third tile row are not computed).  The LDS tiles are double
// buffered: global loads of tile k + 1 are in flight during the products of tile k, their
// LDS stores go to the other buffer, one barrier per tile.
template <int TM, int TN, int BK = kBK>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const float *A, int64_t strideA,
                                                        const float *B, int64_t strideB,
                                                        float *C, int64_t strideC, double *Cpart,
                                                        int M, int N, int K, int kslice,
                                                        int n_slices) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ float As[2][BK][BM + kPad], Bs[2][BK][BN + kPad];
    const int batch = blockIdx.z / n_slices, z = blockIdx.z - batch * n_slices;
    A += batch * strideA;
    B += batch * strideB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int k_lo = z * kslice, k_hi = min(K, k_lo + kslice);

    // global -> registers -> LDS.  A tile: BM rows x BK k, thread t takes k = t % BK of
    // the rows t / BK + (256 / BK) q (a row's BK floats are one 64- or 128-byte segment); B
    // tile: BK k x BN columns, thread t takes column t % BN of the rows t / BN + (256 / BN) q.
    constexpr int RA = 256 / BK, QA = BM / RA, RB = 256 / BN, QB = BK / RB;
    const int a_k = tid % BK, a_r = tid / BK, b_n = tid % BN, b_k = tid / BN;
    float ra[QA], rb[QB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int m = m0 + a_r + RA * q, k = k0 + a_k;
            ra[q] = (m < M && k < k_hi) ? A[(int64_t)m * K + k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int kb = k0 + b_k + RB * q, n = n0 + b_n;
            rb[q] = (kb < k_hi && n < N) ? B[(int64_t)kb * N + n] : 0.f;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < QA; ++q) As[buf][a_k][a_r + RA * q] = ra[q];
#pragma unroll
        for (int q = 0; q < QB; ++q) Bs[buf][b_k + RB * q][b_n] = rb[q];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // MFMA tiles of this wavefront that hold anything (wave-uniform)
    bool on[TM][TN];
    bool any = false, full = true;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            on[i][j] = m0 + wm + 32 * i < M && n0 + wn + 32 * j < N;
            any |= on[i][j];
            full &= on[i][j];
        }

    fetch(k_lo);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
        const bool more = k0 + BK < k_hi;
        if (more) fetch(k0 + BK);  // in flight during the products
        // operand layout of v_mfma_f32_32x32x2_f32: lane l holds A[i = l & 31][k = l >> 5]
        // and B[k = l >> 5][j = l & 31]
        if (any) {
            const int kk = lane >> 5, ij = lane & 31;
            auto products = [&](auto all_on) {
#pragma unroll
                for (int ks = 0; ks < BK; ks += 2) {
                    float a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = As[buf][ks + kk][wm + 32 * i + ij];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = Bs[buf][ks + kk][wn + 32 * j + ij];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            if (decltype(all_on)::value || on[i][j])
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            };
            // (straight-line code for a wavefront whose tiles are all inside: a branch per
            // MFMA keeps the scheduler from moving the LDS reads ahead of the products)
            if (full) products(std::true_type{});
            else products(std::false_type{});
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // accumulator layout of the 32 x 32 tile: lane l, register e -> row 8 (e / 4) +
    // 4 (l >> 5) + e % 4, column l & 31
    double *Cz = Cpart ? Cpart + ((int64_t)batch * n_slices + z) * M * N : nullptr;
    float *Cb = C + batch * strideC;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * i + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                const int n = n0 + wn + 32 * j + (lane & 31);
                if (m < M && n < N) {
                    if (Cz)
                        Cz[(int64_t)m * N + n] = (double)acc[i][j][e];
                    else
                        Cb[(int64_t)m * N + n] = acc[i][j][e];
                }
            }
}

// C_b = sum over the slices of Cpart[b][z] (double), b < n_batch.  A workgroup of sixteen
// wavefronts per 64 outputs: wavefront w sums the slices w, w + 16, ... four at a time (the
// loads of a group are independent: a wavefront that adds one slice after the other waits for
// every load in turn, which is what this kernel used to spend its 20 us on), the sixteen sums
// are combined in a fixed order.
__global__ __launch_bounds__(1024) void reduce_slices_kernel(const double *Cpart, float *C,
                                                            int64_t strideC, int64_t MN,
                                                            int n_slices) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    double t = 0.0;
    if (i < MN) {
        const double *p = Cpart + (int64_t)b * n_slices * MN + i;
        int z = w;
        for (; z + 48 < n_slices; z += 64) {
            const double v0 = p[(int64_t)z * MN], v1 = p[(int64_t)(z + 16) * MN];
            const double v2 = p[(int64_t)(z + 32) * MN], v3 = p[(int64_t)(z + 48) * MN];
            t += (v0 + v1) + (v2 + v3);
        }
        for (; z < n_slices; z += 16) t += p[(int64_t)z * MN];
    }
    part[w][lane] = t;
    __syncthreads();
    if (w == 0 && i < MN) {
        double total = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 4)
            total += (part[q][lane] + part[q + 1][lane]) + (part[q + 2][lane] + part[q + 3][lane]);
        C[b * strideC + i] = (float)total;
    }
}

int gemm(const float *A, int64_t strideA, const float *B, int64_t strideB, float *C,
         int64_t strideC, int n_batch, double *scratch, size_t scratch_elems, int M, int N, int K,
         hipStream_t s) {
    // one MFMA tile per wavefront along a dimension of up to 96 (see the kernel)
    int tm = M <= 96 ? 1 : 2, tn = N <= 96 ? 1 : 2;
    auto n_tiles = [&](int a, int b) {
        return ((M + 64 * a - 1) / (64 * a)) * ((N + 64 * b - 1) / (64 * b)) * n_batch;
    };
    // (a product of a few hundred rows and columns -- the transforms of the spectral path --
    // is 45 workgroups in 128 x 128 tiles and 125 in 64 x 64 tiles)
    const bool few = n_tiles(tm, tn) * std::max(1, (K + kSliceTerms - 1) / kSliceTerms) < 200;
    if (few) tm = tn = 1;
    const int bm = 64 * tm, bn = 64 * tn;
    const int tiles = n_tiles(tm, tn);
    // slices of ~kSliceTerms terms (float32 accumulation length); fewer when the tiles
    // alone fill the chip several times over and the partials would not fit the scratch.
    // (Round 4: fewer, longer slices -- 18 instead of 47 for the 300 x 300 x 15 000 product, 61
    // instead of 169 MB of partials -- were measured and are slower, 260 against 230 us: the
    // 2 115 workgroups of the fine slicing hide each other's load latency, 810 do not.)
    int n_slices = std::max(1, (K + kSliceTerms - 1) / kSliceTerms);
    const size_t MN = (size_t)M * N;
    while (n_slices > 1 && ((size_t)n_slices * n_batch * MN > scratch_elems ||
                            (tiles >= 2048 && K <= 4 * kSliceTerms)))
        --n_slices;
    int kslice = (K + n_slices - 1) / n_slices;
    kslice = (kslice + kBK - 1) / kBK * kBK;
    n_slices = (K + kslice - 1) / kslice;
    const dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm, n_slices * n_batch);
    double *part = n_slices > 1 ? scratch : nullptr;
#define SMI_GEMM(TM, TN)                                                                         \
    hipLaunchKernelGGL((gemm_mfma_kernel<TM, TN>), grid, dim3(256), 0, s, A, strideA, B, strideB, \
                       C, strideC, part, M, N, K, kslice, n_slices)
    // (a workgroup alone on its CU waits for every k-tile's loads: twice the depth per tile
    // where the grid does not fill the chip)
    if (few)
        hipLaunchKernelGGL((gemm_mfma_kernel<1, 1, 2 * kBK>), grid, dim3(256), 0, s, A, strideA, B,
                           strideB, C, strideC, part, M, N, K, kslice, n_slices);
    else if (tm == 1 && tn == 1) SMI_GEMM(1, 1);
    else if (tm == 1) SMI_GEMM(1, 2);
    else if (tn == 1) SMI_GEMM(2, 1);
    else SMI_GEMM(2, 2);
#undef SMI_GEMM
    if (n_slices > 1)
        hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((MN + 63) / 64), n_batch),
                           dim3(1024), 0, s, scratch, C, strideC, (int64_t)MN, n_slices);
    return SMI_OK;
}

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


// ---------------------------------------------------------------------------------------
// The spectral path.  The reference shifts the padded model along x by a phase ramp between
// a real transform and its inverse on the padded grid (renderer.py:414-476), so the shift
// operator of a low-resolution column b is a circulant matrix: P[x', x, b] = s_b[(x - x') mod
// F_x].  With the transforms along x of the model rows, of the operator rows and of s_b
// (hat = sum_x . exp(-2 pi i k x / F_x), k <= F_x / 2, w_k = 1 for k = 0 and the Nyquist
// term, else 2), Parseval's identity turns the two dense products into
//
//   Mh[y, k]   = hat(model[y, :])[k]                           F_y x F_x x (F_x + 2)
//   G[a, k]    = sum_y conj(E[a, y, k]) Mh[y, k]               E = hat(A[a, y, :]), read once
//   out[a, b]  = Re sum_k beta[b, k] G[a, k]                   beta = w_k hat(s_b)[k] / F_x
//
// -- the same linear map (exact in exact arithmetic; checked against the dense products on
// every fixture pair), 3.2 GFLOP -> 0.13 GFLOP per 300 x 300 band and 90 MB of E through HBM
// instead of 450 MB of shifted models.  The adjoint runs the transposed chain:
//
//   Gbar[a, k] = sum_b resid[a, b] conj(beta[b, k])
//   Mbar[y, k] = sum_a Gbar[a, k] E[a, y, k]
//   g[y, x]    = sum_k Re(Mbar[y, k]) cos(2 pi k x / F_x) - Im(Mbar[y, k]) sin(2 pi k x / F_x)
//
// E is made on the device at set-up from the float32 operator the caller hands over, in
// float64 (row_dft_kernel); the sums over y, a, b and k run in float64 as well, the two
// transforms along x on the matrix cores (gemm above, K = F_x or F_x + 2 terms in float32).
// An operator Pt that is not circulant keeps the dense products.
// ---------------------------------------------------------------------------------------

// out[row][k] = scale_k sum_x in[row][x] tw[(k x) mod Fx], k < Kx; tw[j] = exp(-2 pi i j / Fx)
// in float64.  hermitian != 0: scale_k = w_k / Fx (the beta table), else 1.
__global__ __launch_bounds__(256) void row_dft_kernel(const float *in, const double2 *tw,
                                                      float2 *out, int Fx, int Kx, int hermitian) {
    extern __shared__ double dft_sm[];
    double *row = dft_sm;
    double2 *t = reinterpret_cast<double2 *>(dft_sm + ((Fx + 1) & ~1));
    const int64_t r = blockIdx.x;
    for (int x = threadIdx.x; x < Fx; x += 256) {
        row[x] = (double)in[r * Fx + x];
        t[x] = tw[x];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < Kx; k += 256) {
        double re = 0.0, im = 0.0;
        int j = 0;
        for (int x = 0; x < Fx; ++x) {
            const double a = row[x];
            const double2 w = t[j];
            re = fma(a, w.x, re);
            im = fma(a, w.y, im);
            j += k;
            if (j >= Fx) j -= Fx;
        }
        if (hermitian) {
            const double sc = ((k == 0 || 2 * k == Fx) ? 1.0 : 2.0) / (double)Fx;
            re *= sc;
            im *= sc;
        }
        out[r * Kx + k] = make_float2((float)re, (float)im);
    }
}

// One workgroup per (low-resolution row a, band c): G[k] = sum_y conj(E[c][a][y][k]) Mh[c][y][k]
// (wavefront w takes the rows y = w, w + 16, ..., four loads of E in flight per lane; lanes
// along k, so a wavefront reads 512 contiguous bytes of E per row), the sixteen partial sums
// combined in a fixed order, then out[c][a][b] = Re sum_k beta[b][k] G[k] by sixteen lanes
// per column b.  E is read exactly once per rendering.
__global__ __launch_bounds__(1024) void spectral_forward_kernel(const float2 *E, const float2 *Mh,
                                                                const float2 *beta, float *out,
                                                                int n_a, int n_b, int Fy, int Kx) {
    extern __shared__ double2 fwd_sm[];
    double2 *red = fwd_sm;         // [16][64]
    double2 *G = fwd_sm + 16 * 64;  // [Kx]
    const int a = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float2 *Ea = E + ((int64_t)c * n_a + a) * Fy * Kx;
    const float2 *Mc = Mh + (int64_t)c * Fy * Kx;
    for (int k0 = 0; k0 < Kx; k0 += 64) {
        const int k = k0 + lane;
        double re = 0.0, im = 0.0;
        if (k < Kx) {
            auto term = [&](float2 e, float2 m) {
                re = fma((double)e.x, (double)m.x, re);
                re = fma((double)e.y, (double)m.y, re);
                im = fma((double)e.x, (double)m.y, im);
                im = fma(-(double)e.y, (double)m.x, im);
            };
            int y = w;
            for (; y + 48 < Fy; y += 64) {
                const float2 e0 = Ea[(int64_t)y * Kx + k], e1 = Ea[(int64_t)(y + 16) * Kx + k];
                const float2 e2 = Ea[(int64_t)(y + 32) * Kx + k], e3 = Ea[(int64_t)(y + 48) * Kx + k];
                const float2 m0 = Mc[(int64_t)y * Kx + k], m1 = Mc[(int64_t)(y + 16) * Kx + k];
                const float2 m2 = Mc[(int64_t)(y + 32) * Kx + k], m3 = Mc[(int64_t)(y + 48) * Kx + k];
                term(e0, m0);
                term(e1, m1);
                term(e2, m2);
                term(e3, m3);
            }
            for (; y < Fy; y += 16) term(Ea[(int64_t)y * Kx + k], Mc[(int64_t)y * Kx + k]);
        }
        red[w * 64 + lane] = make_double2(re, im);
        __syncthreads();
        if (w == 0 && k < Kx) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                sr += red[q * 64 + lane].x;
                si += red[q * 64 + lane].y;
            }
            G[k] = make_double2(sr, si);
        }
        __syncthreads();
    }
    const int part = tid & 15;
    for (int b0 = 0; b0 < n_b; b0 += 64) {
        const int b = b0 + (tid >> 4);
        double t = 0.0;
        if (b < n_b)
            for (int k = part; k < Kx; k += 16) {
                const float2 bt = beta[(int64_t)b * Kx + k];
                const double2 g = G[k];
                t = fma((double)bt.x, g.x, t);
                t = fma(-(double)bt.y, g.y, t);
            }
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (b < n_b && part == 0) out[((int64_t)c * n_a + a) * n_b + b] = (float)t;
    }
}

// Gbar[ca][k] = sum_b resid[ca][b] conj(beta[b][k]); one workgroup per (band, row a)
__global__ __launch_bounds__(256) void spectral_gbar_kernel(const float *resid, const float2 *beta,
                                                            float2 *Gbar, int n_b, int Kx) {
    extern __shared__ float gbar_sm[];
    const int64_t ca = blockIdx.x;
    for (int b = threadIdx.x; b < n_b; b += 256) gbar_sm[b] = resid[ca * n_b + b];
    __syncthreads();
    for (int k = threadIdx.x; k < Kx; k += 256) {
        double re = 0.0, im = 0.0;
        for (int b = 0; b < n_b; ++b) {
            const float2 bt = beta[(int64_t)b * Kx + k];
            const double r = (double)gbar_sm[b];
            re = fma(r, (double)bt.x, re);
            im = fma(-r, (double)bt.y, im);
        }
        Gbar[ca * Kx + k] = make_float2((float)re, (float)im);
    }
}

// Mbar[c][y][k] = sum_a Gbar[c][a][k] E[c][a][y][k]; one workgroup of four wavefronts per
// (model row y, band c), wavefront w takes a = w, w + 4, ...
__global__ __launch_bounds__(256) void spectral_adjoint_kernel(const float2 *E, const float2 *Gbar,
                                                               float2 *Mbar, int n_a, int Fy,
                                                               int Kx) {
    __shared__ double2 red[4][64];
    const int y = blockIdx.x, c = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float2 *Ec = E + ((int64_t)c * n_a * Fy + y) * Kx;
    const float2 *Gc = Gbar + (int64_t)c * n_a * Kx;
    const int64_t sa = (int64_t)Fy * Kx;
    for (int k0 = 0; k0 < Kx; k0 += 64) {
        const int k = k0 + lane;
        double re = 0.0, im = 0.0;
        if (k < Kx) {
            auto term = [&](float2 g, float2 e) {
                re = fma((double)g.x, (double)e.x, re);
                re = fma(-(double)g.y, (double)e.y, re);
                im = fma((double)g.x, (double)e.y, im);
                im = fma((double)g.y, (double)e.x, im);
            };
            int a = w;
            for (; a + 12 < n_a; a += 16) {
                const float2 e0 = Ec[a * sa + k], e1 = Ec[(a + 4) * sa + k];
                const float2 e2 = Ec[(a + 8) * sa + k], e3 = Ec[(a + 12) * sa + k];
                const float2 g0 = Gc[(int64_t)a * Kx + k], g1 = Gc[(int64_t)(a + 4) * Kx + k];
                const float2 g2 = Gc[(int64_t)(a + 8) * Kx + k], g3 = Gc[(int64_t)(a + 12) * Kx + k];
                term(g0, e0);
                term(g1, e1);
                term(g2, e2);
                term(g3, e3);
            }
            for (; a < n_a; a += 4) term(Gc[(int64_t)a * Kx + k], Ec[a * sa + k]);
        }
        red[w][lane] = make_double2(re, im);
        __syncthreads();
        if (w == 0 && k < Kx) {
            const double sr = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
            const double si = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
            Mbar[((int64_t)c * Fy + y) * Kx + k] = make_float2((float)sr, (float)si);
        }
        __syncthreads();
    }
}

}  // namespace

struct Resampler {
    int C = 0, n_a = 0, n_b = 0, Fy = 0, Fx = 0;
    float *A = nullptr;    // [C][n_a][Fy * Fx]
    float *Pt = nullptr;   // [Fx][Fx * n_b]  (transposed shift operator, shared by the bands)
    float *model = nullptr, *out = nullptr;
    // dense products (made on first use: an operator Pt that is not circulant, or asked for)
    float *B = nullptr;    // [C][Fy * Fx * n_b]
    double *scratch = nullptr;
    size_t scratch_elems = 0;
    float *At = nullptr;   // [C][Fy * Fx][n_a]: A transposed (left operand of the adjoint)
    float *P = nullptr;    // [Fx * n_b][Fx]
    // spectral path
    int path = 0;          // 0: dense products, 1: spectral
    bool spectral = false; // Pt is circulant and the tables below exist
    int Kx = 0;            // Fx / 2 + 1
    float2 *E = nullptr;    // [C][n_a][Fy][Kx]: transforms along x of the operator rows
    float2 *beta = nullptr; // [n_b][Kx]
    float *Wx = nullptr;    // [Fx][2 Kx]: cos, -sin interleaved (forward transform as a product)
    float *WxT = nullptr;   // [2 Kx][Fx]
    float2 *Mh = nullptr;   // [C][Fy][Kx]: transforms of the model rows / Mbar of the adjoint
    float2 *Gbar = nullptr; // [C][n_a][Kx]
    double *sscratch = nullptr;  // slice partials of the two transforms when Fx > kSliceTerms
    size_t sscratch_elems = 0;
};

namespace {

// P[x', x, b] == P[0, (x - x') mod Fx, b] for all entries (to float32 rounding of the largest)?
bool circulant_along_x(const float *Pt, int Fx, int n_b) {
    float top = 0.f;
    for (size_t i = 0; i < (size_t)Fx * n_b; ++i) top = std::max(top, std::fabs(Pt[i]));
    const float tol = 4e-7f * top;
    for (int xp = 1; xp < Fx; ++xp) {
        const float *row = Pt + (size_t)xp * Fx * n_b;
        for (int x = 0; x < Fx; ++x) {
            int d = x - xp;
            if (d < 0) d += Fx;
            const float *a = row + (size_t)x * n_b, *b = Pt + (size_t)d * n_b;
            for (int q = 0; q < n_b; ++q)
                if (!(std::fabs(a[q] - b[q]) <= tol)) return false;
        }
    }
    return true;
}

int build_dense(Resampler *r) {
    if (r->B) return SMI_OK;
    const size_t plane = (size_t)r->Fy * r->Fx;
    const size_t nA = (size_t)r->C * r->n_a * plane, nB = (size_t)r->C * plane * r->n_b;
    // double partials of the sliced products: out (n_a x n_b, thousands of slices) and
    // the model-sized results of the adjoint (F_y x F_x, ~100 slices), all bands at once
    const size_t slices_out = (plane + kSliceTerms - 1) / kSliceTerms + 1;
    const size_t slices_g = ((size_t)r->Fx * r->n_b + kSliceTerms - 1) / kSliceTerms + 1;
    r->scratch_elems = (size_t)r->C * std::max((size_t)r->n_a * r->n_b * slices_out,
                                               plane * slices_g);
    SMI_HIP(hipMalloc((void **)&r->B, nB * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->scratch, r->scratch_elems * sizeof(double)));
    // At[C][Fy Fx][n_a]: the left operand of the adjoint's first product; P: its second
    SMI_HIP(hipMalloc((void **)&r->At, nA * sizeof(float)));
    for (int c = 0; c < r->C; ++c)
        launch_transpose(r->A + (size_t)c * r->n_a * plane, r->At + (size_t)c * r->n_a * plane,
                         r->n_a, (int)plane, nullptr);
    SMI_HIP(hipMalloc((void **)&r->P, (size_t)r->Fx * r->Fx * r->n_b * sizeof(float)));
    launch_transpose(r->Pt, r->P, r->Fx, r->Fx * r->n_b, nullptr);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipDeviceSynchronize());
    return SMI_OK;
}

int build_spectral(Resampler *r, const float *Pt_host) {
    const int Fx = r->Fx, Fy = r->Fy, Kx = Fx / 2 + 1, n_a = r->n_a, n_b = r->n_b, C = r->C;
    r->Kx = Kx;
    const double step = 2.0 * 3.14159265358979323846 / (double)Fx;
    std::vector<double2> tw(Fx);
    for (int j = 0; j < Fx; ++j) tw[j] = make_double2(std::cos(step * j), -std::sin(step * j));
    std::vector<float> Wx((size_t)Fx * 2 * Kx), WxT((size_t)2 * Kx * Fx), S((size_t)n_b * Fx);
    for (int x = 0; x < Fx; ++x)
        for (int k = 0; k < Kx; ++k) {
            const double2 w = tw[(int)(((int64_t)k * x) % Fx)];
            Wx[(size_t)x * 2 * Kx + 2 * k] = WxT[(size_t)(2 * k) * Fx + x] = (float)w.x;
            Wx[(size_t)x * 2 * Kx + 2 * k + 1] = WxT[(size_t)(2 * k + 1) * Fx + x] = (float)w.y;
        }
    for (int b = 0; b < n_b; ++b)  // s_b[x] = P[x' = 0][x][b]
        for (int x = 0; x < Fx; ++x) S[(size_t)b * Fx + x] = Pt_host[(size_t)x * n_b + b];
    double2 *d_tw = nullptr;
    float *d_S = nullptr;
    SMI_HIP(hipMalloc((void **)&d_tw, Fx * sizeof(double2)));
    SMI_HIP(hipMalloc((void **)&d_S, S.size() * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->E, (size_t)C * n_a * Fy * Kx * sizeof(float2)));
    SMI_HIP(hipMalloc((void **)&r->beta, (size_t)n_b * Kx * sizeof(float2)));
    SMI_HIP(hipMalloc((void **)&r->Wx, Wx.size() * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->WxT, WxT.size() * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->Mh, (size_t)C * Fy * Kx * sizeof(float2)));
    SMI_HIP(hipMalloc((void **)&r->Gbar, (size_t)C * n_a * Kx * sizeof(float2)));
    const size_t slices = (size_t)(std::max(Fx, 2 * Kx) + kSliceTerms - 1) / kSliceTerms + 1;
    r->sscratch_elems = (size_t)C * Fy * std::max(Fx, 2 * Kx) * slices;
    SMI_HIP(hipMalloc((void **)&r->sscratch, r->sscratch_elems * sizeof(double)));
    SMI_HIP(hipMemcpy(d_tw, tw.data(), Fx * sizeof(double2), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(d_S, S.data(), S.size() * sizeof(float), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(r->Wx, Wx.data(), Wx.size() * sizeof(float), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(r->WxT, WxT.data(), WxT.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t lds = (size_t)((Fx + 1) & ~1) * sizeof(double) + (size_t)Fx * sizeof(double2);
    hipLaunchKernelGGL(row_dft_kernel, dim3((unsigned)((size_t)C * n_a * Fy)), dim3(256), lds,
                       nullptr, r->A, d_tw, r->E, Fx, Kx, 0);
    hipLaunchKernelGGL(row_dft_kernel, dim3(n_b), dim3(256), lds, nullptr, d_S, d_tw, r->beta, Fx,
                       Kx, 1);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipDeviceSynchronize());
    (void)hipFree(d_tw);
    (void)hipFree(d_S);
    r->spectral = true;
    return SMI_OK;
}

}  // namespace

int resampler_create(const float *A, const float *Pt, int C, int n_a, int n_b, int Fy, int Fx,
                     Resampler **out) {
    auto *r = new Resampler;
    r->C = C; r->n_a = n_a; r->n_b = n_b; r->Fy = Fy; r->Fx = Fx;
    const size_t nA = (size_t)C * n_a * Fy * Fx, nP = (size_t)Fx * Fx * n_b;
    *out = r;
    SMI_HIP(hipMalloc((void **)&r->A, nA * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->Pt, nP * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->model, (size_t)C * Fy * Fx * sizeof(float)));
    SMI_HIP(hipMalloc((void **)&r->out, (size_t)C * n_a * n_b * sizeof(float)));
    SMI_HIP(hipMemcpy(r->A, A, nA * sizeof(float), hipMemcpyHostToDevice));
    SMI_HIP(hipMemcpy(r->Pt, Pt, nP * sizeof(float), hipMemcpyHostToDevice));
    // (the transform tables of a row live in LDS: 24 bytes per column)
    if (Fx <= 2048 && circulant_along_x(Pt, Fx, n_b)) {
        const int rc = build_spectral(r, Pt);
        if (rc) return rc;
        r->path = 1;
        return SMI_OK;
    }
    return build_dense(r);
}

int resampler_get_path(const Resampler *r) { return r->path; }

int resampler_set_path(Resampler *r, int path) {
    SMI_REQUIRE(path == 0 || path == 1, "path is 0 (dense products) or 1 (spectral)");
    SMI_REQUIRE(path == 0 || r->spectral,
                "the shift operator of this resampler is not circulant: dense products only");
    if (path == 0) {
        const int rc = build_dense(r);
        if (rc) return rc;
    }
    r->path = path;
    return SMI_OK;
}

void resampler_destroy(Resampler *r) {
    if (!r) return;
    for (void *p : {(void *)r->A, (void *)r->Pt, (void *)r->model, (void *)r->B, (void *)r->out,
                    (void *)r->scratch, (void *)r->At, (void *)r->P, (void *)r->E,
                    (void *)r->beta, (void *)r->Wx, (void *)r->WxT, (void *)r->Mh,
                    (void *)r->Gbar, (void *)r->sscratch})
        if (p) (void)hipFree(p);
    delete r;
}

// r->model (device, padded) -> r->out, on stream s; all bands in one launch per product
static int resampler_forward(Resampler *r, hipStream_t s) {
    const int64_t plane = (int64_t)r->Fy * r->Fx;
    if (r->path == 1) {
        const int Kx = r->Kx;
        int rc = gemm(r->model, plane, r->Wx, 0, reinterpret_cast<float *>(r->Mh),
                      (int64_t)r->Fy * 2 * Kx, r->C, r->sscratch, r->sscratch_elems, r->Fy, 2 * Kx,
                      r->Fx, s);
        if (rc) return rc;
        const size_t lds = (size_t)(16 * 64 + Kx) * sizeof(double2);
        hipLaunchKernelGGL(spectral_forward_kernel, dim3(r->n_a, r->C), dim3(1024), lds, s, r->E,
                           r->Mh, r->beta, r->out, r->n_a, r->n_b, r->Fy, Kx);
        return SMI_OK;
    }
    int rc = gemm(r->model, plane, r->Pt, 0, r->B, plane * r->n_b, r->C, r->scratch,
                  r->scratch_elems, r->Fy, r->Fx * r->n_b, r->Fx, s);
    if (rc) return rc;
    return gemm(r->A, (int64_t)r->n_a * plane, r->B, plane * r->n_b, r->out,
                (int64_t)r->n_a * r->n_b, r->C, r->scratch, r->scratch_elems, r->n_a, r->n_b,
                (int)plane, s);
}

// gradient of the padded bands from the weighted residual [C][n_a][n_b] (the transposed chain)
static int resampler_adjoint(Resampler *r, const float *resid, float *gpad, hipStream_t s) {
    const int64_t plane = (int64_t)r->Fy * r->Fx;
    if (r->path == 1) {
        const int Kx = r->Kx;
        hipLaunchKernelGGL(spectral_gbar_kernel, dim3(r->C * r->n_a), dim3(256),
                           r->n_b * sizeof(float), s, resid, r->beta, r->Gbar, r->n_b, Kx);
        hipLaunchKernelGGL(spectral_adjoint_kernel, dim3(r->Fy, r->C), dim3(256), 0, s, r->E,
                           r->Gbar, r->Mh, r->n_a, r->Fy, Kx);
        return gemm(reinterpret_cast<float *>(r->Mh), (int64_t)r->Fy * 2 * Kx, r->WxT, 0, gpad,
                    plane, r->C, r->sscratch, r->sscratch_elems, r->Fy, r->Fx, 2 * Kx, s);
    }
    int rc = gemm(r->At, (int64_t)r->n_a * plane, resid, (int64_t)r->n_a * r->n_b, r->B,
                  plane * r->n_b, r->C, r->scratch, r->scratch_elems, (int)plane, r->n_b, r->n_a,
                  s);
    if (rc) return rc;
    return gemm(r->B, plane * r->n_b, r->P, 0, gpad, plane, r->C, r->scratch, r->scratch_elems,
                r->Fy, r->Fx, r->Fx * r->n_b, s);
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

// device time of `n_rep` renderings of the resident model (no host transfers): the
// figure the flops roofline of BASELINE config 5 is measured on
int resampler_time(Resampler *r, int n_rep, double *ms_per_render) {
    hipEvent_t e0, e1;
    SMI_HIP(hipEventCreate(&e0));
    SMI_HIP(hipEventCreate(&e1));
    int rc = resampler_forward(r, nullptr);  // warm-up
    if (rc) return rc;
    SMI_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < n_rep; ++i)
        if ((rc = resampler_forward(r, nullptr))) return rc;
    SMI_HIP(hipEventRecord(e1, nullptr));
    SMI_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SMI_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_render = (double)ms / n_rep;
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
    return resampler_adjoint(r, l->resid, l->gpad, s);
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
