// Free Fourier shifts of image morphologies: ExtendedSource(shifting=True).
//
// The reference moves the image by a phase ramp in Fourier space every time the model
// is built (ImageMorphology.get_model -> fft.shift, morphology.py:124-130,
// fft.py:399-428, ramps interpolation.py:341-375) and lets autograd differentiate
// through it.  Zero-padding, (i)fftshift, rfftn / irfftn and the centred crop cancel
// into a linear map of the (h, w) image that is separable up to one rank-one term:
//
//     shifted = Dr x Tx^T - Di x Hx^T,       M[n, n'] = v(n - n')  (Toeplitz)
//
//   y axis (fftfreq, length Fy):  Dr(d) = (1 + 2 A + cn) / Fy,  Di(d) = (-1)^d sin(pi s)/Fy
//   x axis (rfftfreq, length Fx): Tx(d) = (1 + 2 A + cn) / Fx,  Hx(d) = (2 B + sn) / Fx
//   A = sum_k cos(2 pi k (d - s)/F), B = sum_k sin(..), k = 1 .. (F-1)/2,
//   cn / sn = cos / sin(pi (d - s)) for even F (the Nyquist frequency; its imaginary
//   part along y is what Di carries, along x the C2R transform drops it), 0 for odd F.
//
// FFT lengths are the reference's (fft.py:116-167 with padding 10): the periodic
// interpolation kernel depends on them.  Everything is evaluated directly (no FFT):
// the maps are applied as small dense Toeplitz products -- out of LDS for boxes of up to ~100
// pixels per side, with the image-sized work arrays in global memory beyond (up to 240).
//
// shift_backward_kernel: gathers d(-logL)/d(shifted image) over the box, pulls it back
//   to the image (Dr^T g Tx - Di^T g Hx), forms d/d(shift) with the derivative vectors
//   and takes the unconstrained AMSGrad step of the shift (step 1e-1,
//   morphology.py:673-676); the image itself is then updated by the ordinary update
//   kernel, which reads this gradient instead of gathering.
// shift_forward_kernel: shifted image from the (new) image and shift -> the buffer the
//   render stage and the gathers read.
#include "common.h"

namespace smi {
namespace {

constexpr int kT = 256;

__device__ __forceinline__ double block_sum(double v, double *scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

struct AxisOut {
    float *r;    // Dr / Tx
    float *h;    // Hx (x axis only)
    float *dr;   // d/ds of r
    float *dh;   // d/ds of h
};

// Toeplitz vectors of one axis for d = j - (n - 1), j in [0, 2n - 1); returns
// (sin(pi s)/F, pi cos(pi s)/F) for even F -- the y-axis Nyquist leftovers -- else 0.
__device__ double2 axis_vectors(int F, int n, double s, bool is_x, bool deriv, AxisOut out,
                                double *ct, double *st, double *cb, double *sb) {
    const int tid = threadIdx.x;
    const int K = (F - 1) / 2;
    const bool even = (F % 2) == 0;
    __syncthreads();
    for (int j = tid; j < F; j += kT) sincospi(2.0 * (double)j / (double)F, &st[j], &ct[j]);
    for (int k = tid; k <= K; k += kT)
        sincospi(2.0 * (double)k * s / (double)F, &sb[k], &cb[k]);
    __syncthreads();
    double sps, cps;
    sincospi(s, &sps, &cps);
    for (int j = tid; j < 2 * n - 1; j += kT) {
        const int d = j - (n - 1);
        const int dm = ((d % F) + F) % F;
        double A = 0, B = 0, A1 = 0, B1 = 0;
        int jj = 0;
        for (int k = 1; k <= K; ++k) {
            jj += dm;
            if (jj >= F) jj -= F;  // (k d) mod F
            const double cth = ct[jj] * cb[k] + st[jj] * sb[k];
            const double sth = st[jj] * cb[k] - ct[jj] * sb[k];
            const double wk = 2.0 * 3.141592653589793 * (double)k / (double)F;
            A += cth;
            B += sth;
            A1 += wk * sth;
            B1 += wk * cth;
        }
        const double sg = (d & 1) ? -1.0 : 1.0;
        const double cn = even ? sg * cps : 0.0;   // cos(pi (d - s))
        const double sn = even ? -sg * sps : 0.0;  // sin(pi (d - s))
        out.r[j] = (float)((1.0 + 2.0 * A + cn) / F);
        if (deriv) out.dr[j] = (float)((2.0 * A1 + 3.141592653589793 * sn) / F);
        if (is_x) {
            out.h[j] = (float)((2.0 * B + sn) / F);
            if (deriv) out.dh[j] = (float)(-(2.0 * B1 + 3.141592653589793 * cn) / F);
        }
    }
    __syncthreads();
    return even ? make_double2(sps / F, 3.141592653589793 * cps / F) : make_double2(0.0, 0.0);
}

struct ShiftLds {
    float *g, *P, *Pd, *xs;               // [Np]
    float *dr, *ddr, *tx, *hx, *dtx, *dhx;  // [2 * 64 * ...] Toeplitz vectors
    float *a, *ah, *adh;                  // [w]
    double *ct, *st, *cb, *sb, *red;
};

// `big`: the four image-sized arrays in global memory instead (a workgroup's own region; the
// barriers between the phases order its accesses), only the vectors in LDS
__device__ __forceinline__ ShiftLds carve(unsigned char *base, int Np, int nvec, float *big = nullptr) {
    ShiftLds L;
    double *d = reinterpret_cast<double *>(base);
    L.ct = d;
    L.st = L.ct + 512;
    L.cb = L.st + 512;
    L.sb = L.cb + 256;
    L.red = L.sb + 256;
    float *f = reinterpret_cast<float *>(L.red + 8);
    L.g = big ? big : f;
    L.P = L.g + Np;
    L.Pd = L.P + Np;
    L.xs = L.Pd + Np;
    L.dr = big ? f : L.xs + Np;
    L.ddr = L.dr + nvec;
    L.tx = L.ddr + nvec;
    L.hx = L.tx + nvec;
    L.dtx = L.hx + nvec;
    L.dhx = L.dtx + nvec;
    L.a = L.dhx + nvec;
    L.ah = L.a + nvec;
    L.adh = L.ah + nvec;
    return L;
}

// Steps 3-5 of the pull-back, for the image L.xs, the upstream gradient L.g and the
// Toeplitz vectors in L: d/d(image) = Dr^T g Tx - Di^T g Hx into g_image (if given), and
// this thread's part of d/d(shift) = <image, dDr^T g Tx - dDi^T g Hx>, <image, Dr^T g dTx -
// Di^T g dHx> added to *gsy / *gsx.
__device__ void shift_pull_back(const ShiftLds &L, int h, int w, float beta, float dbeta,
                                float *g_image, double *gsy, double *gsx) {
    const int tid = threadIdx.x, N = h * w;
    __syncthreads();
    // P = Dr^T g, Pd = dDr^T g, a[m] = sum_n (-1)^n g[n, m]
    for (int o = tid; o < N; o += kT) {
        const int np = o / w, m = o - np * w;
        float accP = 0.f, accPd = 0.f;
        for (int n = 0; n < h; ++n) {
            const float gv = L.g[n * w + m];
            accP = fmaf(L.dr[n - np + h - 1], gv, accP);
            accPd = fmaf(L.ddr[n - np + h - 1], gv, accPd);
        }
        L.P[o] = accP;
        L.Pd[o] = accPd;
    }
    for (int m = tid; m < w; m += kT) {
        float acc = 0.f;
        for (int n = 0; n < h; ++n) acc += (n & 1) ? -L.g[n * w + m] : L.g[n * w + m];
        L.a[m] = acc;
    }
    __syncthreads();
    // (a * hx), (a * dhx)
    for (int mp = tid; mp < w; mp += kT) {
        float s0 = 0.f, s1 = 0.f;
        for (int m = 0; m < w; ++m) {
            s0 = fmaf(L.a[m], L.hx[m - mp + w - 1], s0);
            s1 = fmaf(L.a[m], L.dhx[m - mp + w - 1], s1);
        }
        L.ah[mp] = s0;
        L.adh[mp] = s1;
    }
    __syncthreads();
    // gradient w.r.t. the image and the two shift derivatives
    for (int o = tid; o < N; o += kT) {
        const int np = o / w, mp = o - np * w;
        float accT = 0.f, accDT = 0.f, accPdT = 0.f;
        for (int m = 0; m < w; ++m) {
            const float p = L.P[np * w + m], pd = L.Pd[np * w + m];
            const float t = L.tx[m - mp + w - 1];
            accT = fmaf(p, t, accT);
            accDT = fmaf(p, L.dtx[m - mp + w - 1], accDT);
            accPdT = fmaf(pd, t, accPdT);
        }
        const float sg = (np & 1) ? -1.f : 1.f;
        if (g_image) g_image[o] = accT - beta * sg * L.ah[mp];
        *gsx += (double)L.xs[o] * (double)(accDT - beta * sg * L.adh[mp]);
        *gsy += (double)L.xs[o] * (double)(accPdT - dbeta * sg * L.ah[mp]);
    }
}

// shifted = Dr x Tx^T - Di x Hx^T for the image L.xs; element (n, m) goes to
// out[n * stride + m].  Returns non-zero if a value is not finite.
__device__ int shift_apply(const ShiftLds &L, int h, int w, float beta, float *out, int stride) {
    const int tid = threadIdx.x, N = h * w;
    __syncthreads();
    // U = x Tx^T (stored in P), ax[m'] = sum_n' (-1)^n' x[n', m']
    for (int o = tid; o < N; o += kT) {
        const int np = o / w, m = o - np * w;
        float acc = 0.f;
        for (int mp = 0; mp < w; ++mp) acc = fmaf(L.xs[np * w + mp], L.tx[m - mp + w - 1], acc);
        L.P[o] = acc;
    }
    for (int mp = tid; mp < w; mp += kT) {
        float acc = 0.f;
        for (int np = 0; np < h; ++np) acc += (np & 1) ? -L.xs[np * w + mp] : L.xs[np * w + mp];
        L.a[mp] = acc;
    }
    __syncthreads();
    for (int m = tid; m < w; m += kT) {
        float acc = 0.f;
        for (int mp = 0; mp < w; ++mp) acc = fmaf(L.a[mp], L.hx[m - mp + w - 1], acc);
        L.ah[m] = acc;  // b[m]
    }
    __syncthreads();
    int bad = 0;
    for (int o = tid; o < N; o += kT) {
        const int n = o / w, m = o - n * w;
        float acc = 0.f;
        for (int np = 0; np < h; ++np) acc = fmaf(L.dr[n - np + h - 1], L.P[np * w + m], acc);
        const float y = acc - beta * ((n & 1) ? -1.f : 1.f) * L.ah[m];
        out[n * stride + m] = y;
        bad |= !isfinite(y);
    }
    return bad;
}

// bare AMSGrad step of a 2-vector without constraint (lite/parameters.py:274-291), in
// double like the reference's float64 parameter; st = {x, m, v, vhat} x (y, x)
__device__ int amsgrad_pair(double *st, double gsy, double gsx, int it, double b1, double b2,
                            double eps, double alpha) {
    int bad = 0;
    for (int a = 0; a < 2; ++a) {
        const double g = a ? gsx : gsy;
        const double m = (1.0 - b1) * g + b1 * st[2 + a];
        const double vv = (1.0 - b2) * g * g + b2 * st[4 + a];
        const double vh = it == 0 ? vv : fmax(st[6 + a], vv);
        double upd = alpha * m / sqrt(fmax(vh, eps));
        if (it == 0) upd /= 10.0;
        st[2 + a] = m;
        st[4 + a] = vv;
        st[6 + a] = vh;
        st[a] -= upd;
        bad |= !isfinite(st[a]);
    }
    return bad;
}

extern __shared__ __attribute__((aligned(16))) unsigned char shift_lds[];

__global__ __launch_bounds__(kT) void shift_backward_kernel(BatchView v, const float *G, int it,
                                                            double *g_shift_out, int grad_only) {
    const int k = blockIdx.x;
    if (!(v.c_flags[k] & SMI_COMPONENT_SHIFTING)) return;
    const int b = v.c_blend[k];
    if (!grad_only && v.state[b] >= 2) return;
    const int tid = threadIdx.x;
    const int C = v.C, h = v.c_h[k], w = v.c_w[k], N = h * w, oy = v.c_oy[k], ox = v.c_ox[k];
    const int64_t moff = v.c_moff[k];
    float *big = v.shift_scratch ? v.shift_scratch + 4 * moff + 16 * k : nullptr;
    const int Np = ((big ? N : v.max_box_pixels) + 3) & ~3;
    const ShiftLds L = carve(shift_lds, Np, 2 * v.max_box_side, big);
    const float *shifted = v.morph + moff;
    const float *sed = v.sed + (int64_t)k * C;
    double *pt = v.pt + (int64_t)k * 8;
    const double s_y = pt[0], s_x = pt[1];

    // 1. gradient w.r.t. the shifted image over the box (zero outside the frame,
    //    blend.py:30-46) and w.r.t. the spectrum (lite/models.py:206-216)
    for (int i = tid; i < N; i += kT) {
        L.g[i] = 0.f;
        L.xs[i] = v.morph_param[moff + i];
    }
    for (int c = 0; c < C; ++c) {
        const float s = sed[c];
        double part = 0.0;
        const float *Gc = G + ((int64_t)b * C + c) * v.Fy * v.Fx;
        for (int i = tid; i < N; i += kT) {
            const int y = i / w, x = i - y * w;
            const int fy = y + oy, fx = x + ox;
            const bool ok = (unsigned)fy < (unsigned)v.H && (unsigned)fx < (unsigned)v.W;
            const float gv = ok ? Gc[(int64_t)fy * v.Fx + fx] : 0.f;
            L.g[i] = fmaf(s, gv, L.g[i]);
            part += (double)gv * (double)shifted[i];
        }
        const double tot = block_sum(part, L.red);
        if (tid == 0) v.g_sed_buf[(int64_t)k * C + c] = (float)tot;
    }

    // 2. Toeplitz vectors and their derivatives at the current shift
    const double2 by = axis_vectors(v.c_shift_fft[2 * k], h, s_y, false, true,
                                    AxisOut{L.dr, nullptr, L.ddr, nullptr}, L.ct, L.st, L.cb, L.sb);
    axis_vectors(v.c_shift_fft[2 * k + 1], w, s_x, true, true, AxisOut{L.tx, L.hx, L.dtx, L.dhx},
                 L.ct, L.st, L.cb, L.sb);
    const float beta = (float)by.x, dbeta = (float)by.y;

    double gsy = 0.0, gsx = 0.0;
    shift_pull_back(L, h, w, beta, dbeta, v.g_morph_buf + moff, &gsy, &gsx);
    gsy = block_sum(gsy, L.red);
    gsx = block_sum(gsx, L.red);
    if (tid != 0) return;
    if (g_shift_out) {
        g_shift_out[2 * k] = gsy;
        g_shift_out[2 * k + 1] = gsx;
    }
    if (grad_only) return;
    // 6. the shift has no constraint: bare AMSGrad step
    // relative_step (parameter.py:126-129): max(minimum, factor * mean(shift)), on the shift
    // before the update
    const double alpha = fmax((double)v.c_shift_step[k], (double)v.c_shift_rel[k] * 0.5 * (pt[0] + pt[1]));
    const int bad = amsgrad_pair(pt, gsy, gsx, it, v.b1, v.b2, v.eps, alpha);
    if (bad) atomicExch(&v.state[b], v.fail_code);
}

__global__ __launch_bounds__(kT) void shift_forward_kernel(BatchView v, int respect_state) {
    const int k = blockIdx.x;
    if (!(v.c_flags[k] & SMI_COMPONENT_SHIFTING)) return;
    const int b = v.c_blend[k];
    if (respect_state && v.state[b] >= 2) return;
    const int tid = threadIdx.x;
    const int h = v.c_h[k], w = v.c_w[k], N = h * w;
    const int64_t moff = v.c_moff[k];
    float *big = v.shift_scratch ? v.shift_scratch + 4 * moff + 16 * k : nullptr;
    const int Np = ((big ? N : v.max_box_pixels) + 3) & ~3;
    const ShiftLds L = carve(shift_lds, Np, 2 * v.max_box_side, big);
    const double *pt = v.pt + (int64_t)k * 8;
    for (int i = tid; i < N; i += kT) L.xs[i] = v.morph_param[moff + i];
    const double2 by = axis_vectors(v.c_shift_fft[2 * k], h, pt[0], false, false,
                                    AxisOut{L.dr, nullptr, nullptr, nullptr}, L.ct, L.st, L.cb, L.sb);
    axis_vectors(v.c_shift_fft[2 * k + 1], w, pt[1], true, false,
                 AxisOut{L.tx, L.hx, nullptr, nullptr}, L.ct, L.st, L.cb, L.sb);
    const float beta = (float)by.x;
    const int bad = __syncthreads_or(shift_apply(L, h, w, beta, v.morph + moff, w));
    if (bad) atomicExch(&v.state[b], v.fail_code);
}


// ---- free shift of the difference kernel (ConvolutionRenderer(psf_shift=...)) ----------
// d(-logL)/d(shift) = sum_bands <dK/ds, G_K>,  G_K[u] = d(-logL)/dK[u] = sum_x r[x] model[x - (u - p/2)]
// with r = w (rendered - data) (lite/models.py:537-545) and the convolution convention of
// fft.convolve (centred 'same' convolution, fft.py:368-396); dK/ds is the derivative of the
// same Toeplitz maps that move image morphologies above.

__global__ __launch_bounds__(256) void psf_residual_kernel(BatchView v, float *R) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= v.H * v.W) return;
    const int64_t i = ((int64_t)b * v.C + c) * v.H * v.W + pix;
    R[i] = v.weights[i] * (R[i] - v.data[i]);
}

// partial sums of G_K over one band and one slab of frame rows: grid (stamp tiles,
// parts, kernel images)
__global__ __launch_bounds__(256) void psf_kernel_gradient_kernel(BatchView v, const float *R,
                                                                  const float *M,
                                                                  KernelShiftView ks) {
    const int img = blockIdx.z, set = img / ks.bands, j = img - set * ks.bands;
    const int n_slab = (v.H + ks.slab - 1) / ks.slab;
    const int part = blockIdx.y, cs = part / n_slab, sl = part - cs * n_slab;
    const int c = ks.bands == 1 ? cs : j;  // a kernel shared by the bands collects them all
    const int b = ks.per_blend ? set : 0;
    const int n0 = ks.h0 * ks.w0;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n0) return;
    const int uy = u / ks.w0, ux = u - uy * ks.w0;
    const int dy = uy + ks.oy - ks.ph / 2, dx = ux + ks.ox - ks.pw / 2;
    const int H = v.H, W = v.W;
    const float *Rc = R + ((int64_t)b * v.C + c) * H * W;
    const float *Mc = M + ((int64_t)b * v.C + c) * H * W;
    double acc = 0.0;
    const int y1 = min(H, (sl + 1) * ks.slab);
    for (int y = sl * ks.slab; y < y1; ++y) {
        const int my = y - dy;
        if ((unsigned)my >= (unsigned)H) continue;
        const float *rr = Rc + (int64_t)y * W;
        const float *mr = Mc + (int64_t)my * W - dx;
        const int x0 = max(0, dx), x1 = min(W, W + dx);
        for (int x = x0; x < x1; ++x) acc = fma((double)rr[x], (double)mr[x], acc);
    }
    ks.partial[((int64_t)img * ks.n_part + part) * n0 + u] = acc;
}

__device__ __forceinline__ ShiftLds carve_stamp(const KernelShiftView &ks) {
    return carve(shift_lds, (ks.h0 * ks.w0 + 3) & ~3, 2 * max(ks.h0, ks.w0));
}

// one workgroup per kernel set
__global__ __launch_bounds__(kT) void psf_shift_backward_kernel(BatchView v, KernelShiftView ks,
                                                                int it, int grad_only) {
    const int set = blockIdx.x, tid = threadIdx.x;
    const int b = ks.per_blend ? set : 0;
    if (!grad_only && v.state[b] >= 2) return;
    const int h = ks.h0, w = ks.w0, N = h * w;
    const ShiftLds L = carve_stamp(ks);
    double *st = ks.state + (int64_t)set * 10;
    const double2 by = axis_vectors(ks.Fy, h, st[0], false, true,
                                    AxisOut{L.dr, nullptr, L.ddr, nullptr}, L.ct, L.st, L.cb, L.sb);
    axis_vectors(ks.Fx, w, st[1], true, true, AxisOut{L.tx, L.hx, L.dtx, L.dhx}, L.ct, L.st,
                 L.cb, L.sb);
    const float beta = (float)by.x, dbeta = (float)by.y;
    double gsy = 0.0, gsx = 0.0;
    for (int j = 0; j < ks.bands; ++j) {
        const int img = set * ks.bands + j;
        __syncthreads();
        for (int i = tid; i < N; i += kT) {
            double g = 0.0;
            for (int p = 0; p < ks.n_part; ++p) g += ks.partial[((int64_t)img * ks.n_part + p) * N + i];
            L.g[i] = (float)g;
            L.xs[i] = ks.stamp[(int64_t)img * N + i];
        }
        shift_pull_back(L, h, w, beta, dbeta, nullptr, &gsy, &gsx);
    }
    gsy = block_sum(gsy, L.red);
    gsx = block_sum(gsx, L.red);
    if (tid != 0) return;
    st[8] = gsy;
    st[9] = gsx;
    if (grad_only) return;
    const double alpha = fmax(ks.step, ks.rel * 0.5 * (st[0] + st[1]));  // (relative_step)
    if (amsgrad_pair(st, gsy, gsx, it, v.b1, v.b2, v.eps, alpha)) atomicExch(&v.state[b], v.fail_code);
}

// grid (bands, kernel sets): the stamps at the current shift
__global__ __launch_bounds__(kT) void psf_shift_forward_kernel(BatchView v, KernelShiftView ks,
                                                               int respect_state) {
    const int j = blockIdx.x, set = blockIdx.y, tid = threadIdx.x;
    const int b = ks.per_blend ? set : 0;
    if (respect_state && v.state[b] >= 2) return;
    const int h = ks.h0, w = ks.w0, N = h * w, img = set * ks.bands + j;
    const ShiftLds L = carve_stamp(ks);
    const double *st = ks.state + (int64_t)set * 10;
    for (int i = tid; i < N; i += kT) L.xs[i] = ks.stamp[(int64_t)img * N + i];
    const double2 by = axis_vectors(ks.Fy, h, st[0], false, false,
                                    AxisOut{L.dr, nullptr, nullptr, nullptr}, L.ct, L.st, L.cb, L.sb);
    axis_vectors(ks.Fx, w, st[1], true, false, AxisOut{L.tx, L.hx, nullptr, nullptr}, L.ct, L.st,
                 L.cb, L.sb);
    float *out = ks.shifted + ((int64_t)img * ks.ph + ks.oy) * ks.pw + ks.ox;
    const int bad = __syncthreads_or(shift_apply(L, h, w, (float)by.x, out, ks.pw));
    if (bad && tid == 0) atomicExch(&v.state[b], v.fail_code);
}

size_t stamp_lds_bytes(const KernelShiftView &ks) {
    const size_t Np = (ks.h0 * ks.w0 + 3) & ~3;
    return (512 * 2 + 256 * 2 + 8) * sizeof(double) +
           (4 * Np + 9 * 2 * (size_t)std::max(ks.h0, ks.w0)) * sizeof(float);
}

size_t shift_lds_bytes(const BatchView &v) {
    const size_t Np = v.shift_scratch ? 0 : (v.max_box_pixels + 3) & ~3;
    return (512 * 2 + 256 * 2 + 8) * sizeof(double) + (4 * Np + 9 * 2 * (size_t)v.max_box_side) * sizeof(float);
}

}  // namespace

bool shift_needs_scratch(int max_box_pixels, int max_box_side) {
    const size_t Np = (max_box_pixels + 3) & ~3;
    return (512 * 2 + 256 * 2 + 8) * sizeof(double) +
               (4 * Np + 9 * 2 * (size_t)max_box_side) * sizeof(float) > 160 * 1024;
}

static int configure_shift_kernels(size_t lds) {
    static size_t cfg_backward[kMaxDevices] = {}, cfg_forward[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(shift_backward_kernel), lds,
                                    cfg_backward))
        return rc;
    return ensure_dynamic_lds(reinterpret_cast<const void *>(shift_forward_kernel), lds,
                              cfg_forward);
}

int launch_shift_backward(const BatchView &v, const float *G, int32_t it, double *g_shift_out,
                          int32_t grad_only, hipStream_t s) {
    if (v.n_shift == 0 || v.n_comp == 0) return SMI_OK;
    const size_t lds = shift_lds_bytes(v);
    SMI_REQUIRE(lds <= 160 * 1024, "shifting component box too large for the LDS");
    if (int rc = configure_shift_kernels(lds)) return rc;
    hipLaunchKernelGGL(shift_backward_kernel, dim3(v.n_comp), dim3(kT), lds, s, v, G, it,
                       g_shift_out, grad_only);
    return SMI_OK;
}

int launch_shift_forward(const BatchView &v, int32_t respect_state, hipStream_t s) {
    if (v.n_shift == 0 || v.n_comp == 0) return SMI_OK;
    const size_t lds = shift_lds_bytes(v);
    SMI_REQUIRE(lds <= 160 * 1024, "shifting component box too large for the LDS");
    if (int rc = configure_shift_kernels(lds)) return rc;
    hipLaunchKernelGGL(shift_forward_kernel, dim3(v.n_comp), dim3(kT), lds, s, v, respect_state);
    return SMI_OK;
}


int launch_psf_shift_backward(const BatchView &v, const KernelShiftView &ks, float *R,
                              const float *M, int32_t it, int32_t grad_only, hipStream_t s) {
    const size_t lds = stamp_lds_bytes(ks);
    SMI_REQUIRE(lds <= 160 * 1024 && ks.Fy <= 512 && ks.Fx <= 512,
                "kernel stamp too large for the device psf_shift");
    static size_t cfg[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(psf_shift_backward_kernel), lds, cfg))
        return rc;
    hipLaunchKernelGGL(psf_residual_kernel, dim3((v.H * v.W + 255) / 256, v.C, v.nb), dim3(256), 0,
                       s, v, R);
    hipLaunchKernelGGL(psf_kernel_gradient_kernel,
                       dim3((ks.h0 * ks.w0 + 255) / 256, ks.n_part, ks.n_sets * ks.bands),
                       dim3(256), 0, s, v, R, M, ks);
    hipLaunchKernelGGL(psf_shift_backward_kernel, dim3(ks.n_sets), dim3(kT), lds, s, v, ks, it,
                       grad_only);
    return SMI_OK;
}

int launch_psf_shift_forward(const BatchView &v, const KernelShiftView &ks, int32_t respect_state,
                             hipStream_t s) {
    const size_t lds = stamp_lds_bytes(ks);
    SMI_REQUIRE(lds <= 160 * 1024 && ks.Fy <= 512 && ks.Fx <= 512,
                "kernel stamp too large for the device psf_shift");
    static size_t cfg[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(psf_shift_forward_kernel), lds, cfg))
        return rc;
    hipLaunchKernelGGL(psf_shift_forward_kernel, dim3(ks.bands, ks.n_sets), dim3(kT), lds, s, v, ks,
                       respect_state);
    return SMI_OK;
}

}  // namespace smi
