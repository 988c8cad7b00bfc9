// HIP kernels (gfx950 / CDNA4, wave64) of the proximal-gradient loop.
//
// Everything here is HBM/LDS-bound element-wise, stencil and reduction work; no
// MFMA.  Conventions: one thread per frame pixel with the band loop inside (so a
// morphology value is fetched once for all bands and every global access is
// coalesced along x); one wavefront per component for the parameter update, with
// the morphology, the metric and the proximal iterate resident in LDS.
#include <math.h>

#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "common.h"

namespace smi {

namespace {

constexpr int kBandChunk = 8;
constexpr int kPixBlock = 256;

// float reductions over the wavefront without LDS traffic: four DPP steps reduce each
// row of 16 lanes (quad swaps, half-row mirror, row mirror), v_readlane combines the
// four rows.  The result is wave-uniform and deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// np.maximum semantics: a NaN in the data propagates (fmaxf would drop it)
__device__ __forceinline__ float max_nan(float a, float b) { return a != a ? a : fmaxf(a, b); }

__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------
// Blend.get_model (blend.py:200-244, component.py:160-164): scatter-add of the
// boxed outer products sed (x) morph into the model cube P[nb][C][Fy][Fx] (image at
// the origin; for the rocFFT path Fy x Fx is the zero-padded FFT input and the padding
// stays zero, for the fused path the cube is compact).
//
// Pixel-owner: one wavefront per four frame rows, lane l the columns l, l + 64 of a
// 128-column strip; every pixel accumulates its components in ascending order (the
// summation order of blend.py:30-46) in registers and is written once.  Component
// metadata sits one component per lane and is broadcast with v_readlane; a component
// whose box misses the wave's rows or strip is skipped with a wave-uniform test; box
// rows are contiguous in memory, so a wave's loads are coalesced.  The kernel stays
// below 64 VGPRs on purpose: that is what a SIMD has left beside the four wavefronts of
// fused_conv_kernel, so render waves of one range of blends run in the issue slots the
// convolution of another range leaves idle.
// ---------------------------------------------------------------------------
constexpr int kRenderRows = 4, kRenderWaves = 4;
// NB: bands per pass (the accumulators are registers: 4 rows x 2 columns x NB)
template <int NB>
__global__ __launch_bounds__(64 * kRenderWaves) void render_kernel(BatchView v, float *P) {
    const int b = blockIdx.y + v.blend0;
    if (v.state[b] >= 2) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int y0 = (blockIdx.x * kRenderWaves + wave) * kRenderRows;
    if (y0 >= v.H) return;
    const int cs = v.comp_start[b], ce = v.comp_start[b + 1];
    const int H = v.H, W = v.W;
    const int y1 = min(y0 + kRenderRows, H);
    for (int x0 = 0; x0 < W; x0 += 128) {
        const int x1 = min(x0 + 128, W);
        for (int c0 = 0; c0 < v.C; c0 += NB) {
            const int nc = min(NB, v.C - c0);
            float acc[kRenderRows][2][NB];
#pragma unroll
            for (int r = 0; r < kRenderRows; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[r][q][j] = 0.f;
            for (int kb = cs; kb < ce; kb += 64) {
                const int kk = kb + lane;
                const bool have = kk < ce;
                const int l_oy = have ? v.c_oy[kk] : 0, l_ox = have ? v.c_ox[kk] : 0;
                const int l_h = have ? v.c_h[kk] : 0, l_w = have ? v.c_w[kk] : 0;
                const int l_mo = have ? (int)v.c_moff[kk] : 0;  // packed offsets fit 31 bits
                // the components (one per lane) whose box meets the wave's rows and strip
                const bool hit = have && min(y1, l_oy + l_h) > max(y0, l_oy) &&
                                 min(x1, l_ox + l_w) > max(x0, l_ox);
                unsigned long long todo = __ballot(hit);
                while (todo) {  // ascending component order
                    const int kl = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int oy = __builtin_amdgcn_readlane(l_oy, kl);
                    const int hh = __builtin_amdgcn_readlane(l_h, kl);
                    const int ox = __builtin_amdgcn_readlane(l_ox, kl);
                    const int w = __builtin_amdgcn_readlane(l_w, kl);
                    const int r_lo = max(y0, oy), r_hi = min(y1, oy + hh);
                    const int x_lo = max(x0, ox), x_hi = min(x1, ox + w);
                    const float *mbase = v.morph + __builtin_amdgcn_readlane(l_mo, kl);
                    float mv[kRenderRows][2];
                    bool ok[kRenderRows][2];
#pragma unroll
                    for (int r = 0; r < kRenderRows; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int yy = y0 + r, xx = x0 + lane + 64 * q;
                            ok[r][q] = yy >= r_lo && yy < r_hi && xx >= x_lo && xx < x_hi;
                            mv[r][q] = ok[r][q] ? mbase[(yy - oy) * w + (xx - ox)] : 0.f;
                        }
                    const float *sed = v.sed + (int64_t)(kb + kl) * v.C + c0;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (j >= nc) break;
                        const float sd = sed[j];
#pragma unroll
                        for (int r = 0; r < kRenderRows; ++r)
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                if (ok[r][q])  // pixels outside the box stay untouched
                                    acc[r][q][j] = fmaf(sd, mv[r][q], acc[r][q][j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j >= nc) break;
#pragma unroll
                for (int r = 0; r < kRenderRows; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int yy = y0 + r, xx = x0 + lane + 64 * q;
                        if (yy < H && xx < W)
                            P[(((int64_t)b * v.C + c0 + j) * v.Fy + yy) * v.Fx + xx] = acc[r][q][j];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Observation.get_log_likelihood (observation.py:147-170) and the upstream
// gradient w (m - d) (lite/models.py:537-545), fused: reads the rendered cube
// Q, data and weights once, writes the residual into the zero-padded cube R and
// one partial sum of w (m-d)^2 per block.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPixBlock) void residual_kernel(BatchView v, const float *Q,
                                                             float *R) {
    const int b = blockIdx.y;
    if (v.state[b] >= 2) return;
    const int pix = blockIdx.x * kPixBlock + threadIdx.x;
    const int y = pix / v.W, x = pix - y * v.W;
    const bool inside = pix < v.H * v.W;
    double acc = 0.0;
    if (inside) {
        for (int c = 0; c < v.C; ++c) {
            const int64_t iF = (((int64_t)b * v.C + c) * v.Fy + y) * v.Fx + x;
            const int64_t iD = (((int64_t)b * v.C + c) * v.H + y) * v.W + x;
            const float diff = Q[iF] - v.data[iD];
            const float r = v.weights[iD] * diff;
            R[iF] = r;
            acc += (double)(r * diff);
        }
    }
    acc = wave_sum(acc);
    __shared__ double part[kPixBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kPixBlock / 64; ++i) t += part[i];
        v.loss_partial[(int64_t)b * v.n_partial + blockIdx.x] = t;
    }
}

// loss bookkeeping + convergence test of Blend._callback (blend.py:294-299) for blend b, by
// one wavefront.  Nothing here is read by the component updates of the same iteration (they
// only ask whether state >= 2, which advance_kernel sets afterwards), so the workgroups that
// run this may share a launch with them (update_kernel_reg); the non-finite flag an update
// may raise at the same time survives: 0 -> 1 is a compare-and-swap.  That flag carries the
// iteration it was raised in (state = 3 + it, BatchView::fail_code): the loss of iteration
// `it` is recorded whenever the blend was still iterating when `it` began, whether an update
// of the same launch has already failed or not -- as in the reference, where _callback
// appends the loss before the step is taken (blend.py:294-299) -- so the length of the loss
// history of a failed blend does not depend on how the workgroups were scheduled.
__device__ __forceinline__ void finalize_blend(const BatchView &v, int b, int it, float e_rel,
                                               int min_iter, int check) {
    const int st = v.state[b];
    if (st == 2 || (st >= 3 && st - 3 < it)) return;
    const int lane = threadIdx.x & 63;
    double t = 0.0;
    for (int i = lane; i < v.n_partial; i += 64)
        t += v.loss_partial[(int64_t)b * v.n_partial + i];
    t = wave_sum(t);
    if (lane == 0) {
        double loss = v.log_norm[b] + 0.5 * t;
        for (int i = 0; i < v.n_extra; ++i) loss += v.extra_term[i];
        const int n = v.n_loss[b];
        const double prev = v.last_loss[b];
        if (n < v.hist_cap) v.loss_hist[(int64_t)b * v.hist_cap + n] = loss;
        v.n_loss[b] = n + 1;
        v.last_loss[b] = loss;
        const bool stop = check && (n >= 1 || v.have_prev[b]) && v.local_it(b, it) > min_iter &&
                          fabs(loss - prev) < (double)e_rel * fabs(loss);
        if (stop && v.conv_flag) v.conv_flag[b] = 1;
        // (a blend that pauses at its resize hook or at the end of its budget: smi_batch_set_pause_at)
        if (stop || (v.pause_at && it >= v.pause_at[b]))
            atomicCAS(&v.state[b], 0, 1);  // this iteration's update is the last one
    }
}

__global__ __launch_bounds__(64) void finalize_kernel(BatchView v, int it, float e_rel,
                                                      int min_iter, int check) {
    finalize_blend(v, blockIdx.x + v.blend0, it, e_rel, min_iter, check);
}

__global__ void advance_kernel(int32_t *state, int nb) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb && state[b] == 1) state[b] = 2;
}

__global__ void count_active_kernel(const int32_t *state, int nb, int32_t *out) {
    // out[0] = blends still iterating, out[1] = first non-finite blend or -1
    int active = 0, err = 0x7fffffff;
    for (int b = threadIdx.x; b < nb; b += 64) {
        active += state[b] < 2;
        if (state[b] >= 3) err = min(err, b);
    }
    for (int o = 32; o > 0; o >>= 1) {
        active += __shfl_xor(active, o, 64);
        err = min(err, __shfl_xor(err, o, 64));
    }
    if (threadIdx.x == 0) {
        out[0] = active;
        out[1] = err == 0x7fffffff ? -1 : err;
    }
}

// S *= K  or  S *= conj(K), K broadcast over blends and/or bands
__global__ __launch_bounds__(256) void cmul_kernel(float2 *S, const float2 *K, int C,
                                                   int64_t plane, int k_bands,
                                                   int k_per_blend, int conj,
                                                   const int32_t *state) {
    const int tiles = (int)((plane + 255) / 256);
    const int bc = blockIdx.x / tiles;
    const int b = bc / C, c = bc - b * C;
    if (state[b] >= 2) return;
    const int64_t i = (int64_t)(blockIdx.x - bc * tiles) * 256 + threadIdx.x;
    if (i >= plane) return;
    const int kb = k_per_blend ? b : 0, kc = k_bands == 1 ? 0 : c;
    const float2 k = K[((int64_t)kb * k_bands + kc) * plane + i];
    float2 s = S[(int64_t)bc * plane + i];
    const float ki = conj ? -k.y : k.y;
    S[(int64_t)bc * plane + i] = make_float2(s.x * k.x - s.y * ki, s.x * ki + s.y * k.x);
}

// ---------------------------------------------------------------------------
// The threads of one component in the general update kernel: one wavefront, or four for
// boxes beyond the register-resident kernels (the element-wise passes and the sweep levels
// of a 100^2 box keep four waves busy).  With one wavefront every reduction is the plain
// 64-lane tree of the other kernels.
// ---------------------------------------------------------------------------
template <int T>
struct Team {
    static_assert(T == 64 || T == 256, "team size");
    static constexpr int kWaves = T / 64;
    static __device__ __forceinline__ float sum(float v) {
        v = wave_sum(v);
        if (T == 64) return v;
        __shared__ float red[kWaves];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < kWaves; i += 4) t += (red[i] + red[i + 1]) + (red[i + 2] + red[i + 3]);
        __syncthreads();
        return t;
    }
    static __device__ __forceinline__ float max(float v) {
        v = wave_max(v);
        if (T == 64) return v;
        __shared__ float red[kWaves];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = red[0];
#pragma unroll
        for (int i = 1; i < kWaves; ++i) t = fmaxf(t, red[i]);
        __syncthreads();
        return t;
    }
    static __device__ __forceinline__ int any(int v) {
        return T == 64 ? wave_or(v) : __syncthreads_or(v);
    }
};

// ---------------------------------------------------------------------------
// Monotonic sweep on an LDS-resident image, level by level.
// Separate multiply and add (no FMA contraction): bit-identical to the
// reference's sequential loop (operators_pybind11.cc:14-36).
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T mul_rn(T a, T b);
template <>
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
template <>
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
template <typename T>
__device__ __forceinline__ T add_rn(T a, T b);
template <>
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
template <>
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

template <typename T, typename W, int THREADS = 64>
__device__ __forceinline__ void sweep_levels(T *img, const int32_t *level_start, int n_levels,
                                             int E, const int32_t *pix, const int32_t *cnt,
                                             const int32_t *nbr, const W *wt,
                                             T one_minus_g, int lane) {
    for (int l = 0; l < n_levels; ++l) {
        const int s = level_start[l], e = level_start[l + 1];
        for (int q = s + lane; q < e; q += THREADS) {
            const int p = pix[q];
            const int n = cnt[q];
            T ref = 0;
            for (int j = 0; j < n; ++j)
                ref = add_rn(ref, mul_rn(img[nbr[(int64_t)j * E + q]], (T)wt[(int64_t)j * E + q]));
            const T lim = mul_rn(ref, one_minus_g);
            if (lim < img[p]) img[p] = lim;
        }
        __syncthreads();
    }
}

// The same sweep with the plan entries of the next level requested before the current
// level is applied: the entries come from global memory (L2), and a level is only a few
// dependent instructions long, so without this every level waits for a load round trip
// (a 150^2 box has about 440 levels).  One entry per thread and level in registers;
// further entries of a level wider than the team are loaded in place.
template <int THREADS>
__device__ __forceinline__ void sweep_levels_prefetch(float *img, const int32_t *level_start,
                                                      int n_levels, int E, int max_terms,
                                                      const int32_t *pix, const int32_t *cnt,
                                                      const int32_t *nbr, const float *wt,
                                                      float one_minus_g, int lane) {
    constexpr int kTerms = 8;  // offsets of operators_pybind11.cc:14-36
    struct Entry {
        int p, n;
        int nb[kTerms];
        float w[kTerms];
    };
    auto load = [&](int q, Entry &e) {
        e.p = pix[q];
        e.n = cnt[q];
#pragma unroll
        for (int j = 0; j < kTerms; ++j)
            if (j < max_terms) {
                e.nb[j] = nbr[(int64_t)j * E + q];
                e.w[j] = wt[(int64_t)j * E + q];
            }
    };
    auto apply = [&](const Entry &e) {
        float ref = 0.f;
#pragma unroll
        for (int j = 0; j < kTerms; ++j)
            if (j < e.n) ref = __fadd_rn(ref, __fmul_rn(img[e.nb[j]], e.w[j]));
        const float lim = __fmul_rn(ref, one_minus_g);
        if (lim < img[e.p]) img[e.p] = lim;
    };
    if (n_levels <= 0) return;
    int s = level_start[0], e = level_start[1];
    Entry cur{}, nxt{};
    bool have = s + lane < e;
    if (have) load(s + lane, cur);
    for (int l = 0; l < n_levels; ++l) {
        const int e_next = l + 1 < n_levels ? level_start[l + 2] : e;
        const bool have_next = l + 1 < n_levels && e + lane < e_next;
        if (have_next) load(e + lane, nxt);
        if (have) apply(cur);
        for (int q = s + lane + THREADS; q < e; q += THREADS) {
            Entry more{};
            load(q, more);
            apply(more);
        }
        __syncthreads();
        cur = nxt;
        have = have_next;
        s = e;
        e = e_next;
    }
}

// ---------------------------------------------------------------------------
// Per-component gradient, AMSGrad step and proximal sub-iterations: the body
// of proxmin.adaprox as mirrored at lite/parameters.py:274-305, for the two
// parameters (spectrum, morphology image) of one FactorizedComponent.
// One wavefront per component; G is the gradient image d(-logL)/d(model).
// ---------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) float lds_dyn[];

struct CompCtx {
    int k, b, C, h, w, N, oy, ox, lane;
    int64_t moff;
    const float *morph, *sed;
    float *morph_out;
    bool pre;  // gradient already gathered and pulled back through a Fourier shift
};

__device__ __forceinline__ CompCtx comp_ctx(const BatchView &v) {
    CompCtx c;
    c.k = blockIdx.x + v.comp0;
    c.lane = threadIdx.x;
    c.b = v.c_blend[c.k];
    c.C = v.C;
    c.h = v.c_h[c.k];
    c.w = v.c_w[c.k];
    c.N = c.h * c.w;
    c.oy = v.c_oy[c.k];
    c.ox = v.c_ox[c.k];
    c.moff = v.c_moff[c.k];
    c.pre = v.n_shift && (v.c_flags[c.k] & SMI_COMPONENT_SHIFTING);
    // a shifting component is updated on its image parameter; `morph` (what the model
    // uses) is rebuilt from it by shift_forward_kernel afterwards
    c.morph_out = (c.pre ? v.morph_param : v.morph) + c.moff;
    c.morph = c.morph_out;
    c.sed = v.sed + (int64_t)c.k * c.C;
    return c;
}

// slice G into the box (zero outside the frame, blend.py:30-46); us[i] = sum_c sed G,
// return (in lane c) sum_yx G[c] morph  (lite/models.py:206-216)
template <int T>
__device__ __forceinline__ float gather_gradient(const BatchView &v, const CompCtx &c,
                                                 const float *G, float *us) {
    float g_sed = 0.f;
    if (c.pre) {
        for (int i = c.lane; i < c.N; i += T) us[i] = v.g_morph_buf[c.moff + i];
        return c.lane < c.C ? v.g_sed_buf[(int64_t)c.k * c.C + c.lane] : 0.f;
    }
    const float inv_w = 1.0f / (float)c.w;
    for (int c0 = 0; c0 < c.C; c0 += kBandChunk) {
        const int nc = min(kBandChunk, c.C - c0);
        float acc[kBandChunk];
#pragma unroll
        for (int j = 0; j < kBandChunk; ++j) acc[j] = 0.f;
        // kGU pixels per lane in flight: all loads of a group are issued before the first
        // one is consumed (the gather is bound by memory latency, not by bandwidth)
#ifndef SMI_GU
#define SMI_GU 3
#endif
        constexpr int kGU = SMI_GU;
        for (int i0 = c.lane; i0 < c.N; i0 += T * kGU) {
            float gv[kGU][kBandChunk], mv[kGU];
            bool in_box[kGU];
#pragma unroll
            for (int u = 0; u < kGU; ++u) {
                const int i = i0 + T * u;
                // exact for i < 2^20: the float quotient is off by < 1e-6 relative
                const int y = (int)(((float)i + 0.5f) * inv_w);
                const int x = i - y * c.w;
                const int fy = y + c.oy, fx = x + c.ox;
                in_box[u] = i < c.N;
                const bool ok =
                    in_box[u] && (unsigned)fy < (unsigned)v.H && (unsigned)fx < (unsigned)v.W;
                mv[u] = ok ? c.morph[i] : 0.f;
                const float *g = G + (((int64_t)c.b * c.C + c0) * v.Fy + fy) * v.Fx + fx;
#pragma unroll
                for (int j = 0; j < kBandChunk; ++j)
                    gv[u][j] = (ok && j < nc) ? g[(int64_t)j * v.Fy * v.Fx] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kGU; ++u) {
                const int i = i0 + T * u;
                if (!in_box[u]) continue;
                float gm = c0 == 0 ? 0.f : us[i];
#pragma unroll
                for (int j = 0; j < kBandChunk; ++j)
                    if (j < nc) {
                        gm = fmaf(c.sed[c0 + j], gv[u][j], gm);
                        acc[j] = fmaf(gv[u][j], mv[u], acc[j]);
                    }
                us[i] = gm;
            }
        }
#pragma unroll
        for (int j = 0; j < kBandChunk; ++j)
            if (j < nc) {
                const float t = Team<T>::sum(acc[j]);
                if (c.lane == c0 + j) g_sed = t;
            }
    }
    return g_sed;
}

// Parameter(fixed=True): the optimizer sees a zero gradient (blend.py:107-115)
template <int T>
__device__ __forceinline__ float hold_fixed(const BatchView &v, const CompCtx &c, float g_sed,
                                            float *us) {
    const int flags = v.c_flags[c.k];
    if (flags & SMI_COMPONENT_FIXED_MORPH)
        for (int i = c.lane; i < c.N; i += T) us[i] = 0.f;
    return (flags & SMI_COMPONENT_FIXED_SED) ? 0.f : g_sed;
}

// spectrum (spectrum.py:54-56): relative step, AMSGrad, positivity at 1e-20.
// FISTA (lite/parameters.py:134-150): y = z - step / sum(morph^2) g, x' = max(y, 1e-20),
// z' = x + (1 + (t - 1) / t') (x' - x).  The new spectrum is also left in
// `sed_new[lane]` for the background threshold of the morphology.
// Returns non-zero if the new value is not finite.
__device__ __forceinline__ int update_spectrum(const BatchView &v, const CompCtx &c, float g_sed,
                                               int it, float e2, int prox_max_iter,
                                               float sum_morph2, float *sed_new) {
    const bool on = c.lane < c.C;
    const float s = on ? c.sed[c.lane] : 0.f;
    const int64_t idx = (int64_t)c.k * c.C + c.lane;
    if (v.scheme == SMI_SCHEME_FISTA) {
        double *tp = v.fista_t + 2 * (int64_t)c.k;
        const float t = (float)tp[0];
        const float tn = 0.5f * (1.f + sqrtf(1.f + 4.f * t * t));
        const float omega = 1.f + (t - 1.f) / tn;
        float xn = 0.f;
        if (on) {
            const float step = v.c_fista_step[c.k] / sum_morph2;
            xn = max_nan(v.m_sed[idx] - step * g_sed, 1e-20f);
            v.m_sed[idx] = s + omega * (xn - s);
            v.sed[idx] = xn;
            sed_new[c.lane] = xn;
        }
        if (c.lane == 0) tp[0] = (double)tn;
        return on && !isfinite(xn);
    }
    const float mean = wave_sum(s) / (float)c.C;
    float psi = 0.f, x = s;
    if (on) {
        const float alpha = fmaxf(v.c_sed_min_step[idx], v.c_sed_rel[c.k] * mean);
        const float m = (1.f - v.b1) * g_sed + v.b1 * v.m_sed[idx];
        const float vv = (1.f - v.b2) * g_sed * g_sed + v.b2 * v.v_sed[idx];
        const float vh = it == 0 ? vv : fmaxf(v.vh_sed[idx], vv);
        v.m_sed[idx] = m;
        v.v_sed[idx] = vv;
        v.vh_sed[idx] = vh;
        psi = sqrtf(fmaxf(vh, v.eps));
        float upd = alpha * m / psi;
        if (it == 0) upd /= 10.f;  // lite/parameters.py:288-291
        x = s - upd;
    }
    const float pmax = wave_max(psi);
    const float ratio = on ? psi / pmax : 0.f;
    float z = x;
    for (int tau = 0; tau < prox_max_iter; ++tau) {
        const float zn = on ? max_nan(z - ratio * (z - x), 1e-20f) : 0.f;
        const float d2 = wave_sum((zn - z) * (zn - z));
        const float z2 = wave_sum(z * z);
        z = zn;
        if (d2 <= e2 * z2) break;
    }
    if (on) {
        v.sed[idx] = z;
        sed_new[c.lane] = z;
    }
    return on && !isfinite(z);
}

// MonotonicityConstraint(fit_center_radius=1): index (0..8, row major) of the brightest
// pixel of the 3x3 block around the box centre, first maximum like np.argmax
// (operator.py:99-129); the block is clipped at index 0 only, like the slices there
__device__ __forceinline__ int fit_center_index(const float *us, const CompCtx &c) {
    const int cy = c.h / 2, cx = c.w / 2;
    int best = 4;
    float bv = -INFINITY;
    bool first = true;
    for (int j = 0; j < 9; ++j) {
        const int yy = cy + j / 3 - 1, xx = cx + j % 3 - 1;
        if (yy < 0 || xx < 0 || yy >= c.h || xx >= c.w) continue;
        const float val = us[yy * c.w + xx];
        if (first || val > bv) {
            bv = val;
            best = j;
            first = false;
        }
    }
    return best;
}

// LDS data that only one wave touches: LDS operations of one wave execute in order, so between a
// step's writes and the next step's reads only the LDS counter has to drain.  Unlike
// __syncthreads() this leaves the global loads of the plan prefetch in flight.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// the same between the passes of a component's threads: a barrier once they are several waves
template <int T>
__device__ __forceinline__ void team_fence() {
    if (T == 64)
        wave_lds_fence();
    else
        __syncthreads();
}

// the element-wise members of the chain that act on the LDS image `us` before the
// final positivity / centre / normalisation pass (constraint.py:262-273, 117-145)
template <int T = 64>
__device__ __forceinline__ void chain_symmetry_threshold(float *us, const CompCtx &c, int flags,
                                                         float lthresh, const float *sed_new,
                                                         const float *bg_level,
                                                         float strength = 1.f) {
    const int h = c.h, w = c.w, N = c.N, lane = c.lane;
    if (flags & SMI_PROX_BG_THRESH) {
        // lite/models.py:222-228: zero where the model stays below the background
        // level in every band (spectrum already updated)
        for (int i = lane; i < N; i += T) {
            const float u = us[i];
            bool below = true;
            for (int b = 0; b < c.C; ++b) below = below && (sed_new[b] * u < bg_level[b]);
            if (below) us[i] = 0.f;
        }
        team_fence<T>();  // a team only touches its own image
    }
    if (flags & SMI_PROX_SYMMETRY) {
        // prox_soft_symmetry (operator.py:274-293), x <- s/2 (x + rot180 x) + (1 - s) x: even axes are
        // padded by one trailing zero before the 180-degree rotation
        const int hp = h + !(h & 1), wp = w + !(w & 1);
        const float s = strength, keep = 1.f - strength;
        for (int i = lane; i < N; i += T) {
            const int y = i / w, x = i - y * w;
            const int py = hp - 1 - y, px = wp - 1 - x;
            const bool has = py < h && px < w;
            const int j = py * w + px;
            if (s == 1.f) {
                if (!has) {
                    us[i] = 0.5f * us[i];
                } else if (j >= i) {
                    const float a = 0.5f * (us[i] + us[j]);
                    us[i] = a;
                    us[j] = a;
                }
            } else if (!has) {
                // 0.5 s (x + 0) + (1 - s) x
                us[i] = 0.5f * s * us[i] + keep * us[i];
            } else if (j >= i) {
                const float xi = us[i], xj = us[j];
                us[i] = 0.5f * s * (xi + xj) + keep * xi;
                us[j] = 0.5f * s * (xj + xi) + keep * xj;
            }
        }
        team_fence<T>();  // a team only touches its own image
    }
    if (flags & (SMI_PROX_L1 | SMI_PROX_L0)) {
        for (int i = lane; i < N; i += T) {
            const float u = us[i];
            if (flags & SMI_PROX_L1)
                us[i] = copysignf(fmaxf(fabsf(u) - lthresh, 0.f), u);
            else if (fabsf(u) < lthresh)
                us[i] = 0.f;
        }
    }
}

// MonotonicityConstraint(use_mask=True) (constraint.py:228-232 -> operator.py:131-176 with
// center_radius = 0, variance = 0, max_iter = 0): the pixels that get_valid_monotonic_pixels
// (operators_pybind11.cc:61-125) accepts -- reachable from `start` over 4-neighbour steps
// onto a strictly smaller, positive value -- keep the value they had before the sweep.
// The accepted set is the closure of that relation, so it is relaxed in parallel until
// nothing changes (as mask.hip does); `ws` receives the image, `fl` the accepted flags.
template <int T>
__device__ __forceinline__ void monotonic_mask(const float *us, float *ws, uint8_t *fl,
                                               const CompCtx &c, int start) {
    const int N = c.N, w = c.w, h = c.h, lane = c.lane;
    for (int i = lane; i < N; i += T) {
        ws[i] = us[i];
        fl[i] = i == start;
    }
    team_fence<T>();
    for (;;) {
        int changed = 0;
        for (int i = lane; i < N; i += T) {
            if (fl[i]) continue;
            const float val = ws[i];
            if (!(val > 0.f)) continue;
            const int y = i / w, x = i - y * w;
            const bool accept = (y > 0 && fl[i - w] && val < ws[i - w]) ||
                                (y < h - 1 && fl[i + w] && val < ws[i + w]) ||
                                (x > 0 && fl[i - 1] && val < ws[i - 1]) ||
                                (x < w - 1 && fl[i + 1] && val < ws[i + 1]);
            if (accept) {
                fl[i] = 1;
                changed = 1;
            }
        }
        team_fence<T>();
        if (!Team<T>::any(changed)) break;
    }
}

// -- generic variant: everything in LDS, plans with any number of terms -----
__device__ __forceinline__ void sweep_slots(float *us, const SweepSlotEntry *slots, int n_slots,
                                            float one_minus_g, int lane);
// T threads per component (Team): 64, or 256 for boxes of more than 64 x 59 pixels
template <int T>
__global__ __launch_bounds__(T) void update_kernel(BatchView v, const float *G, int it,
                                                   float e_rel, int prox_max_iter,
                                                   float *g_sed_out, float *g_morph_out,
                                                   int grad_only) {
    const CompCtx c = comp_ctx(v);
    it = v.local_it(c.b, it);
    // the team is a property of the box, not of the batch: a component gets the same bits
    // whatever else is fitted with it (two launches when a batch holds both kinds)
    // (the same boundary as the register-resident classes: 64 x 59 pixels)
    if ((c.N > 64 * kUpdateNpl[kNumSmallClasses - 1]) != (T == 256)) return;
    if (!grad_only && v.state[c.b] >= 2) return;
    if (!grad_only && v.n_point && (v.c_flags[c.k] & SMI_COMPONENT_POINT_SOURCE)) return;
    const int lane = c.lane, N = c.N;
    const int npad = (v.max_box_pixels + 3) & ~3;
    // x after the gradient step, psi / max(psi), current proximal iterate: in LDS, or --
    // for boxes beyond ~100^2 pixels -- in a global scratch area (every lane only ever
    // touches its own elements of these three); the image being swept is always in LDS
    float *xs = v.scratch ? v.scratch + 3 * c.moff : lds_dyn;
    float *rs = xs + (v.scratch ? c.N : npad);
    float *zs = rs + (v.scratch ? c.N : npad);
    // candidate (and g_morph before that); + 4: the spare cell in front of the image that
    // the slot plans send their idle lanes to
    float *us = (v.scratch ? lds_dyn : zs + npad) + 4;
    int32_t *lvl = reinterpret_cast<int32_t *>(us + npad);  // level_start of the plan
    // SMI_PROX_MONO_MASK: image before the sweep and the flags of the accepted pixels
    float *ws = reinterpret_cast<float *>(lvl + ((v.max_levels + 2 + 3) & ~3));
    uint8_t *fl = reinterpret_cast<uint8_t *>(ws + npad);

    float g_sed = gather_gradient<T>(v, c, G, us);
    __syncthreads();
    if (grad_only) {
        if (lane < c.C) g_sed_out[(int64_t)c.k * c.C + lane] = g_sed;
        for (int i = lane; i < N; i += T) g_morph_out[c.moff + i] = us[i];
        return;
    }
    g_sed = hold_fixed<T>(v, c, g_sed, us);
    __syncthreads();
    const float e2 = e_rel * e_rel;
    __shared__ float sed_new[64];
    const bool fista = v.scheme == SMI_SCHEME_FISTA;
    if (fista) prox_max_iter = 1;  // FistaParameter applies the prox once
    float msum = 0.f, msum2 = 0.f;
    for (int i = lane; i < N; i += T) {
        const float mval = c.morph[i];
        msum += mval;
        msum2 += mval * mval;
    }
    msum = Team<T>::sum(msum);
    msum2 = Team<T>::sum(msum2);
    // sum of the squared *old* spectrum: FISTA step of the morphology (lite/models.py:249-254)
    const float so = lane < c.C ? c.sed[lane] : 0.f;
    const float ssum2 = Team<T>::sum(so * so);
    const float t_old = fista ? (float)v.fista_t[2 * (int64_t)c.k + 1] : 1.f;
    // the spectrum belongs to the first wavefront (one band per lane)
    int bad = 0;
    if (T == 64 || threadIdx.x < 64)
        bad = update_spectrum(v, c, g_sed, it, e2, prox_max_iter, msum2, sed_new);

    const int flags = v.c_flags[c.k];
    const int plan_id = v.c_plan[c.k];
    const float alpha = fmaxf(v.c_morph_step[c.k], v.c_morph_rel[c.k] * (msum / (float)N));
    float pmax = 0.f;
    if (fista) {
        const float step = v.c_fista_step[c.k] / ssum2;
        for (int i = lane; i < N; i += T) {
            const float y = v.m_morph[c.moff + i] - step * us[i];
            xs[i] = y;
            zs[i] = y;
            rs[i] = 0.f;
        }
        pmax = 1.f;
    } else {
        for (int i = lane; i < N; i += T) {
            const float g = us[i];
            const float m = (1.f - v.b1) * g + v.b1 * v.m_morph[c.moff + i];
            const float vv = (1.f - v.b2) * g * g + v.b2 * v.v_morph[c.moff + i];
            const float vh = it == 0 ? vv : fmaxf(v.vh_morph[c.moff + i], vv);
            v.m_morph[c.moff + i] = m;
            v.v_morph[c.moff + i] = vv;
            v.vh_morph[c.moff + i] = vh;
            const float psi = sqrtf(fmaxf(vh, v.eps));
            float upd = alpha * m / psi;
            if (it == 0) upd /= 10.f;
            const float x = c.morph[i] - upd;
            xs[i] = x;
            zs[i] = x;
            rs[i] = psi;
            pmax = fmaxf(pmax, psi);
        }
        pmax = Team<T>::max(pmax);
    }

    const bool monotonic = (flags & SMI_PROX_MONOTONIC) && plan_id >= 0;
    const bool fit_center = monotonic && (flags & SMI_PROX_FIT_CENTER);
    SweepPlanDev pl;
    if (monotonic && !fit_center) {
        pl = v.plans[plan_id];
        for (int i = lane; i <= pl.n_levels; i += T) lvl[i] = pl.level_start[i];
    }
    const float one_minus_g = 1.f - v.c_min_grad[c.k];
    const int ctr = (c.h / 2) * c.w + (c.w / 2);
    const float cfloor = v.c_center_floor[c.k];
    const float pfloor = v.c_pos_floor[c.k];  // PositivityConstraint(zero)
    // relative thresholds scale with the step of the sub-iteration, gamma = alpha / max(psi)
    const float lthresh =
        v.c_lthresh[c.k] * ((flags & SMI_PROX_L_RELATIVE) ? alpha / pmax : 1.f);
    const float *bg_level = v.c_bg_level ? v.c_bg_level + (int64_t)c.k * c.C : nullptr;
    const int repeat = v.c_chain_repeat ? v.c_chain_repeat[c.k] : 1;
    __syncthreads();
    {
        const float rpmax = 1.f / pmax;  // like the register-resident kernels
        for (int i = lane; i < N; i += T) rs[i] = rs[i] * rpmax;
    }

    for (int tau = 0; tau < prox_max_iter; ++tau) {
        for (int i = lane; i < N; i += T) us[i] = zs[i] - rs[i] * (zs[i] - xs[i]);
        __syncthreads();
        // ConstraintChain (constraint.py:76-80) in the order of morphology.py:644-670,
        // `repeat` times over (constraint.py:60-80); the last normalisation is folded into
        // the convergence pass below
        float div = 1.f;
        for (int rep = 0; rep < repeat; ++rep) {
            int start = ctr;
            if (fit_center) {
                const int j = fit_center_index(us, c);
                start = ctr + (j / 3 - 1) * c.w + (j % 3 - 1);
                pl = v.plans[plan_id + j];
                for (int i = lane; i <= pl.n_levels; i += T) lvl[i] = pl.level_start[i];
                __syncthreads();
            }
            const bool masked = monotonic && (flags & SMI_PROX_MONO_MASK);
            if (masked) monotonic_mask<T>(us, ws, fl, c, start);
            if (monotonic) {
                if (pl.slots) {
                    // plans of at most four terms (every weighting of the reference but
                    // "flat" away from the axes): the packed slot plan of the register-
                    // resident kernels, swept by the first wavefront without barriers
                    if (T == 64 || threadIdx.x < 64) {
                        sweep_slots(us, pl.slots, pl.n_slots, one_minus_g, lane);
                    }
                    __syncthreads();
                } else {
                    sweep_levels_prefetch<T>(us, lvl, pl.n_levels, pl.n_entries, pl.max_terms,
                                             pl.pix, pl.cnt, pl.nbr, pl.wt, one_minus_g, lane);
                }
            }
            if (masked) {
                __syncthreads();
                for (int i = lane; i < N; i += T)
                    if (fl[i]) us[i] = ws[i];
                __syncthreads();
            }
            chain_symmetry_threshold<T>(us, c, flags, lthresh, sed_new, bg_level,
                                     (flags & SMI_PROX_SYMMETRY) ? v.c_sym_strength[c.k] : 1.f);
            float mx = -INFINITY, sm = 0.f;
            for (int i = lane; i < N; i += T) {
                float u = us[i];
                if (flags & SMI_PROX_POSITIVE) u = max_nan(u, pfloor);
                if ((flags & SMI_PROX_CENTER_ON) && i == ctr) u = max_nan(u, cfloor);
                us[i] = u;
                mx = fmaxf(mx, u);
                sm += u;
            }
            div = 1.f;
            if (flags & SMI_PROX_NORM_MAX) div = Team<T>::max(mx);
            if (flags & SMI_PROX_NORM_SUM) div = Team<T>::sum(sm);
            if (rep + 1 < repeat) {
                if (flags & (SMI_PROX_NORM_MAX | SMI_PROX_NORM_SUM))
                    for (int i = lane; i < N; i += T) us[i] = us[i] / div;
                __syncthreads();
            }
        }
        // the last normalisation like the register-resident kernels (one reciprocal, the
        // maximum itself maps to exactly 1): a component gets the same bits in either kernel
        const float rdiv = 1.f / div;
        float d2 = 0.f, z2 = 0.f;
        for (int i = lane; i < N; i += T) {
            float u = us[i];
            if (flags & (SMI_PROX_NORM_MAX | SMI_PROX_NORM_SUM))
                u = (u == div && (flags & SMI_PROX_NORM_MAX)) ? 1.f : u * rdiv;
            const float z = zs[i];
            d2 += (u - z) * (u - z);
            z2 += z * z;
            zs[i] = u;
        }
        d2 = Team<T>::sum(d2);
        z2 = Team<T>::sum(z2);
        __syncthreads();
        if (d2 <= e2 * z2) break;
    }
    float omega = 0.f;
    if (fista) {
        const float tn = 0.5f * (1.f + sqrtf(1.f + 4.f * t_old * t_old));
        omega = 1.f + (t_old - 1.f) / tn;
        if (lane == 0) v.fista_t[2 * (int64_t)c.k + 1] = (double)tn;
    }
    for (int i = lane; i < N; i += T) {
        const float z = zs[i];
        if (fista) {
            const float xo = c.morph[i];
            v.m_morph[c.moff + i] = xo + omega * (z - xo);
        }
        c.morph_out[i] = z;
        bad |= !isfinite(z);
    }
    if (Team<T>::any(bad) && lane == 0) atomicExch(&v.state[c.b], v.fail_code);  // model.py:153-165
}

// -- point sources ---------------------------------------------------------------
// PointSource (source.py:92-128): spectrum x model PSF at a free sub-pixel centre.
// The PSF is the separable pixel-integrated Gaussian of GaussianPSF._f
// (psf.py:128-142), normalised to unit sum over the box (psf.py:126); its
// derivative w.r.t. the centre is the difference of the Gaussian at the two pixel
// edges.  The centre has no constraint, so its update is the bare AMSGrad step of
// lite/parameters.py:274-291.  All of this is evaluated in double like the
// reference's float64 centre parameter; the box has at most 63 x 63 pixels.
__device__ __forceinline__ double integrated_gaussian(double X, double sigma) {
    const double sqrt2 = 1.4142135623730951;
    return 1.2533141373155001 * sigma *
           (1.0 - erfc((0.5 - X) / (sqrt2 * sigma)) + 1.0 -
            erfc((2.0 * X + 1.0) / (2.0 * sqrt2 * sigma)));
}

__device__ __forceinline__ double integrated_gaussian_deriv(double X, double sigma) {
    const double s2 = 2.0 * sigma * sigma;
    return exp(-(X + 0.5) * (X + 0.5) / s2) - exp(-(X - 0.5) * (X - 0.5) / s2);
}

__global__ __launch_bounds__(64) void point_source_kernel(BatchView v, const float *G, int it,
                                                          float e_rel, int prox_max_iter,
                                                          float *g_sed_out, double *g_ctr_out,
                                                          int mode) {
    const CompCtx c = comp_ctx(v);
    if (!(v.c_flags[c.k] & SMI_COMPONENT_POINT_SOURCE)) return;
    if (mode == 0 && v.state[c.b] >= 2) return;
    it = v.local_it(c.b, it);
    const int lane = c.lane, N = c.N;
    __shared__ double fy[64], fx[64], dfy[64], dfx[64];
    float *us = lds_dyn;
    double *pt = v.pt + (int64_t)c.k * 8;
    const double sigma = (double)v.c_sigma[c.k];
    double off_y = pt[0], off_x = pt[1];
    const float inv_w = 1.0f / (float)c.w;
    int bad = 0;
    // MoffatPSF (psf.py:145-202): A = (1 + r^2 / alpha^2)^-beta at the pixel centres, not
    // separable; d A / d centre_y = 2 beta Y / alpha^2 (1 + r^2 / alpha^2)^(-beta - 1)
    const double beta = (double)v.c_beta[c.k];
    const bool moffat = beta > 0.0;
    const double inv_a2 = 1.0 / (sigma * sigma);
    auto moffat_at = [&](int i, double &A, double &dAy, double &dAx) {
        const int y = (int)(((float)i + 0.5f) * inv_w);
        const int x = i - y * c.w;
        const double Y = (double)(y - c.h / 2) - off_y, X = (double)(x - c.w / 2) - off_x;
        const double q = 1.0 + (X * X + Y * Y) * inv_a2;
        A = pow(q, -beta);
        const double d = 2.0 * beta * inv_a2 * A / q;
        dAy = d * Y;
        dAx = d * X;
    };

    auto profiles = [&](bool with_deriv) {
        // FunctionPSF grid (psf.py:60-66): pixel j sits at j - size // 2
        const double Y = (double)(lane - c.h / 2) - off_y;
        const double X = (double)(lane - c.w / 2) - off_x;
        fy[lane] = lane < c.h ? integrated_gaussian(Y, sigma) : 0.0;
        fx[lane] = lane < c.w ? integrated_gaussian(X, sigma) : 0.0;
        if (with_deriv) {
            // d f(Y_j - offset) / d centre = -f'(Y_j - offset)
            dfy[lane] = lane < c.h ? -integrated_gaussian_deriv(Y, sigma) : 0.0;
            dfx[lane] = lane < c.w ? -integrated_gaussian_deriv(X, sigma) : 0.0;
        }
        __syncthreads();
    };

    if (mode != 2) {
        const float g_sed = gather_gradient<64>(v, c, G, us);
        __syncthreads();
        double gy = 0.0, gx = 0.0;
        if (moffat) {
            // morph = A / S: g . d morph = (sum g dA) / S - (sum g A) (sum dA) / S^2
            double S = 0.0, dSy = 0.0, dSx = 0.0, gA = 0.0;
            for (int i = lane; i < N; i += 64) {
                double A, dAy, dAx;
                moffat_at(i, A, dAy, dAx);
                const double g = (double)us[i];
                S += A;
                dSy += dAy;
                dSx += dAx;
                gA += g * A;
                gy += g * dAy;
                gx += g * dAx;
            }
            S = wave_sum(S);
            dSy = wave_sum(dSy);
            dSx = wave_sum(dSx);
            gA = wave_sum(gA);
            gy = wave_sum(gy) / S - gA * dSy / (S * S);
            gx = wave_sum(gx) / S - gA * dSx / (S * S);
        } else {
            profiles(true);
            const double Sy = wave_sum(fy[lane]), Sx = wave_sum(fx[lane]);
            const double dSy = wave_sum(dfy[lane]), dSx = wave_sum(dfx[lane]);
            const double S = Sy * Sx;
            for (int i = lane; i < N; i += 64) {
                const int y = (int)(((float)i + 0.5f) * inv_w);
                const int x = i - y * c.w;
                const double A = fy[y] * fx[x];
                const double d_y = dfy[y] * fx[x] / S - A * (dSy * Sx) / (S * S);
                const double d_x = fy[y] * dfx[x] / S - A * (Sy * dSx) / (S * S);
                gy += (double)us[i] * d_y;
                gx += (double)us[i] * d_x;
            }
            gy = wave_sum(gy);
            gx = wave_sum(gx);
        }
        if (mode == 1) {
            if (lane < c.C) g_sed_out[(int64_t)c.k * c.C + lane] = g_sed;
            if (lane == 0) {
                g_ctr_out[2 * c.k] = gy;
                g_ctr_out[2 * c.k + 1] = gx;
            }
            return;
        }
        __shared__ float sed_new[64];
        const int fixed = v.c_flags[c.k];  // Parameter(fixed=True): zero gradient
        if (fixed & SMI_COMPONENT_FIXED_MORPH) gy = gx = 0.0;
        bad = update_spectrum(v, c, (fixed & SMI_COMPONENT_FIXED_SED) ? 0.f : g_sed, it,
                              e_rel * e_rel, prox_max_iter, 1.f, sed_new);
        // step of the centre (source.py:115, or relative_step, parameter.py:126-129, on the
        // centre in frame pixels = mean of the box bounds + offset)
        const double b1 = v.b1, b2 = v.b2, eps = v.eps;
        const double ctr_mean = 0.5 * ((c.oy + 0.5 * c.h + off_y) + (c.ox + 0.5 * c.w + off_x));
        const double alpha = fmax((double)v.c_morph_step[c.k], (double)v.c_morph_rel[c.k] * ctr_mean);
        double upd[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double g = a ? gx : gy;
            const double m = (1.0 - b1) * g + b1 * pt[2 + a];
            const double vv = (1.0 - b2) * g * g + b2 * pt[4 + a];
            const double vh = it == 0 ? vv : fmax(pt[6 + a], vv);
            upd[a] = alpha * m / sqrt(fmax(vh, eps));
            if (it == 0) upd[a] /= 10.0;
            __syncthreads();
            if (lane == 0) {
                pt[2 + a] = m;
                pt[4 + a] = vv;
                pt[6 + a] = vh;
            }
        }
        off_y -= upd[0];
        off_x -= upd[1];
        __syncthreads();
        if (lane == 0) {
            pt[0] = off_y;
            pt[1] = off_x;
        }
        bad |= !isfinite(off_y) || !isfinite(off_x);
    }
    // morphology at the (new) centre: outer product / its sum (psf.py:104-126), or the
    // Moffat profile / its sum (psf.py:176-202)
    if (!moffat) profiles(false);
    auto value = [&](int i) {
        if (moffat) {
            double A, dAy, dAx;
            moffat_at(i, A, dAy, dAx);
            return A;
        }
        const int y = (int)(((float)i + 0.5f) * inv_w);
        return fy[y] * fx[i - y * c.w];
    };
    double part = 0.0;
    for (int i = lane; i < N; i += 64) part += value(i);
    const double total = wave_sum(part);
    for (int i = lane; i < N; i += 64) {
        const float z = (float)(value(i) / total);
        v.morph[c.moff + i] = z;
        bad |= !isfinite(z);
    }
    if (wave_or(bad) && lane == 0) atomicExch(&v.state[c.b], v.fail_code);
}

// -- fast variant --------------------------------------------------------------
// Boxes of at most 64*NPL pixels and plans with at most 4 terms per pixel (the
// reference's 'flat' / 'angle' / 'nearest' tables): x, psi/max(psi) and the
// iterate z live in registers (lane l owns pixels l, l+64, ...); only the image
// being swept is in LDS, so ~16 waves fit a CU.  The sweep plan is stored as
// 64-wide slots (one 32-byte entry per lane, two coalesced dwordx4 loads) and the
// next slot is fetched while the current one is processed.
// One 64-wide step of the sweep per loop iteration.  Entry layout (32 bytes, see
// SweepSlotEntry): LDS byte addresses of the pixel and of its four neighbours, and the
// four weights.  Terms a pixel does not have carry weight 0 and point at the pixel
// itself ("+ 0 * own value", exact for finite values); idle lanes point at a spare
// cell behind the image that always holds 0, so the loop body has no branches:
// ~35 instructions per level instead of ~65 (the sweep is VALU-issue bound).
// -- buffer addressing ------------------------------------------------------------
// A component's slices of the packed arrays (morph, m, v, vhat), a band plane of the
// gradient image and the sweep plan are addressed through buffer descriptors built
// from wave-uniform values: lanes beyond the box / outside the frame read 0 and their
// stores are dropped by the bounds check, so the per-pixel loops are straight-line code
// and all loads of a group are in flight together (the phase is bound by memory latency).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store(rsrc_t r, uint32_t byte_off, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), r, byte_off, 0, 0);
}
constexpr uint32_t kOutOfRange = 0x80000000u;

__device__ __forceinline__ CompCtx comp_ctx(const BatchView &v, int k, int lane) {
    CompCtx c;
    c.k = k;
    c.lane = lane;
    c.b = v.c_blend[k];
    c.C = v.C;
    c.h = v.c_h[k];
    c.w = v.c_w[k];
    c.N = c.h * c.w;
    c.oy = v.c_oy[k];
    c.ox = v.c_ox[k];
    c.moff = v.c_moff[k];
    c.pre = v.n_shift && (v.c_flags[k] & SMI_COMPONENT_SHIFTING);
    c.morph_out = (c.pre ? v.morph_param : v.morph) + c.moff;
    c.morph = c.morph_out;
    c.sed = v.sed + (int64_t)k * c.C;
    return c;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

// Between two steps of the sweep.  The LDS executes the DS instructions of ONE wave in the
// order they were issued, so a step's reads see the previous step's writes without the
// wave waiting for those writes to retire: only the compiler has to keep the order.
#ifndef SMI_SWEEP_DRAIN
#define SMI_SWEEP_DRAIN 0
#endif
__device__ __forceinline__ void sweep_fence() {
#if SMI_SWEEP_DRAIN
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

// `base` = the component's LDS image - 16 bytes: plan addresses are 16 + 4 * pixel, address 0
// is the spare cell in front of the image.
__device__ __forceinline__ void sweep_step(char *base, const u32x4 a, const u32x4 wb,
                                           float one_minus_g) {
    float *pp = reinterpret_cast<float *>(base + (a.x & 0xffff));
    const float cur = *pp;
    const float u0 = *reinterpret_cast<float *>(base + (a.x >> 16));
    const float u1 = *reinterpret_cast<float *>(base + (a.y & 0xffff));
    const float u2 = *reinterpret_cast<float *>(base + (a.y >> 16));
    const float u3 = *reinterpret_cast<float *>(base + (a.z & 0xffff));
    float ref = __fadd_rn(0.f, __fmul_rn(u0, __uint_as_float(wb.x)));
    ref = __fadd_rn(ref, __fmul_rn(u1, __uint_as_float(wb.y)));
    ref = __fadd_rn(ref, __fmul_rn(u2, __uint_as_float(wb.z)));
    ref = __fadd_rn(ref, __fmul_rn(u3, __uint_as_float(wb.w)));
    const float lim = __fmul_rn(ref, one_minus_g);
    if (lim < cur) *pp = lim;
}

// `slots` must be wave-uniform: the plan is read through a buffer descriptor (lane l at
// byte 32 l of every 2-KB step, the step offset in a scalar register), which takes the
// address arithmetic of the prefetch out of the vector unit.  The 16 bytes in front of
// `us` must belong to the component (spare cell).
//
// The plan stream -- 2 KB per step and component out of L2, up to ten times per iteration --
// is what the sweep of a full batch is most sensitive to (a second 2 KB per step: + 71 %;
// ten more VALU operations per step: + 4 %).  A level of a 41 x 41 box has 28 pixels on
// average, in the first lanes of its step, so only those lanes fetch: the others get an
// out-of-range offset, for which the buffer load returns zeros without touching memory --
// an entry of zeros is an idle entry (every address = the spare cell, weights 0).  The lane
// count of step s + 3 rides in the fourth dword of step s's entries.
__device__ __forceinline__ void sweep_slots(float *us, const SweepSlotEntry *slots, int n_slots,
                                            float one_minus_g, int lane) {
    char *base = reinterpret_cast<char *>(us) - 16;
    // uniformity made explicit, or the compiler wraps every load in a waterfall loop
    const uint64_t sp = reinterpret_cast<uint64_t>(slots);
    // (the builtin returns a signed int: go through uint32_t, or the low half sign-extends)
    const uint32_t sp_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sp);
    const uint32_t sp_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sp >> 32));
    const uint64_t sp_u = (uint64_t)sp_lo | ((uint64_t)sp_hi << 32);
    n_slots = __builtin_amdgcn_readfirstlane(n_slots);
    const rsrc_t r = make_rsrc(reinterpret_cast<const void *>(sp_u), (uint32_t)(n_slots + 3) * 2048u);
    const uint32_t vo = (uint32_t)lane * 32u;
    auto meta = [&](int step, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, step * 2048, 0); };
    auto wts = [&](int step, uint32_t off) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u, step * 2048, 0);
    };
    // four steps per iteration, each plan entry requested three steps before it is used
    // (register ping-pong, no rotation moves).  The plan is padded to a multiple of four
    // steps plus three idle steps, so neither the prefetch nor the loop needs a bound
    // check inside an iteration.
    // issued in the order of the steady state (entry, weights, step by step): the wait
    // counter at the loop head is the minimum over both ways into the loop
    u32x4 a0 = meta(0, vo);
    u32x4 w0 = wts(0, vo);
    __builtin_amdgcn_sched_barrier(0);
    u32x4 a1 = meta(1, vo);
    u32x4 w1 = wts(1, vo);
    __builtin_amdgcn_sched_barrier(0);
    u32x4 a2 = meta(2, vo);
    u32x4 w2 = wts(2, vo);
    __builtin_amdgcn_sched_barrier(0);
    auto lanes = [&](const u32x4 &e) {  // offset of this lane's entry three steps on, or none
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w);
        return (uint32_t)lane < n ? vo : kOutOfRange;
    };
    for (int s = 0; s < n_slots; s += 4) {
        uint32_t off = lanes(a0);
        const u32x4 a3 = meta(s + 3, off);
        const u32x4 w3 = wts(s + 3, off);
        sweep_step(base, a0, w0, one_minus_g);
        sweep_fence();
        off = lanes(a1);
        a0 = meta(s + 4, off);
        w0 = wts(s + 4, off);
        sweep_step(base, a1, w1, one_minus_g);
        sweep_fence();
        off = lanes(a2);
        a1 = meta(s + 5, off);
        w1 = wts(s + 5, off);
        sweep_step(base, a2, w2, one_minus_g);
        sweep_fence();
        off = lanes(a3);
        a2 = meta(s + 6, off);
        w2 = wts(s + 6, off);
        sweep_step(base, a3, w3, one_minus_g);
        sweep_fence();
    }
}

// -- ring schedule of the radial tables (common.h: RingPlanHost) --------------------------
// One lane per (octant, ring mod 8) and plane: the operands of a step are the lane's own
// previous result and the last three results of the lane of the next inner ring, fetched by
// DPP row rotations (lane = row * 16 + half * 8 + m: the inner ring is lane - 1, or lane + 7
// -- of the other plane when there are two -- for m = 0; the same two rotations with the roles
// swapped reach the octant across the axis).  The image in LDS is touched once per pixel: read
// ahead of the step, written behind it.
__device__ __forceinline__ float dpp_ror1(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_ror9(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x129, 0xf, 0xf, false));
}
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The plan stream in LDS, staged there by the workgroup (common.h: weights once per ring, the
// eight lanes of a ring read one address; addresses per lane): weights one step ahead,
// addresses two.
template <int P>
struct RingLds {
    static constexpr int kWeightsAhead = 1, kAddrAhead = 2;
    const char *w, *a;  // this lane's entries of step 0, plane 0
    __device__ __forceinline__ RingLds(const char *plan, int n_pad, int lane) {
        w = plan + (lane & 7) * 16;
        a = plan + (n_pad + kRingAhead) * (128 * P) + lane * 2;
    }
    __device__ __forceinline__ uint32_t addr(int step, int p) const {
        return *reinterpret_cast<const uint16_t *>(a + (step * P + p) * 128);
    }
    __device__ __forceinline__ f32x4 weights(int step, int p) const {
        return *reinterpret_cast<const f32x4 *>(w + (step * P + p) * 128);
    }
};

// The same stream read from global memory (L2), for wavefronts that have no staged copy: 256 B
// per step instead of the slot plan's 2 KB.  Entries are requested four / five steps ahead (an
// L2 round trip is about three steps).
template <int P>
struct RingGlobal {
    static constexpr int kWeightsAhead = 4, kAddrAhead = 5;
    rsrc_t r;
    uint32_t wo, ao;  // this lane's byte offsets inside a step
    __device__ __forceinline__ RingGlobal(const void *stream, uint32_t bytes, int n_pad, int lane) {
        r = make_rsrc(stream, bytes);
        wo = (uint32_t)(lane & 7) * 16u;
        ao = (uint32_t)(n_pad + kRingAhead) * (128u * P) + (uint32_t)lane * 2u;
    }
    __device__ __forceinline__ uint32_t addr(int step, int p) const {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, ao, (step * P + p) * 128, 0);
    }
    __device__ __forceinline__ f32x4 weights(int step, int p) const {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, wo, (step * P + p) * 128, 0);
        return __builtin_bit_cast(f32x4, t);
    }
};

// The 16 bytes in front of `us` are the spare cell (idle lanes); n_pad, rmax, perm, centre:
// the plan's ring_* fields, wave-uniform.  P planes: the lane works on ring m + 8 p (mod 8 P)
// of its octant in every step, P independent chains.
//
// LATE (one plane, rings 24 .. 31 of boxes up to 63^2; common.h: RingPlanHost): from step n_nat
// on a ring may start later than level 2 r - 1 -- when the lane of ring r - 8 is free --, its
// operands A, B, C are then one level older (one more entry of the rotation history: c4, and
// the mirror lane's last but one result g2), and the address words of the stream say which
// entries are late and which are the axis / diagonal pixel of their ring.
template <int P, bool LATE = false, class Plan = RingLds<P>>
__device__ __forceinline__ void sweep_ring_loop(float *us, const Plan &plan, int n_pad,
                                                int n_nat, int rmax, uint32_t perm,
                                                int centre_pix, float one_minus_g, int lane) {
    static_assert(!LATE || P == 1, "late rings: one plane");
    constexpr int DW = Plan::kWeightsAhead, DA = Plan::kAddrAhead;
    // (address words of a stream with late rings carry flags, also while they are prefetched)
    constexpr uint32_t kMask = LATE ? (uint32_t)kRingAddrMask : 0xFFFFFFFFu;
    static_assert(kRingUnroll == 6 && DA < 6 && DW < DA && (P == 1 || P == 2),
                  "register rotation of the loop");
    char *base = reinterpret_cast<char *>(us) - 16;
    auto lds = [&](uint32_t a) { return reinterpret_cast<float *>(base + a); };

    // constants of the lane
    const int m = lane & 7;
    const bool inner = m >= 1;  // the lane of ring r - 1 sits right below
    const uint32_t code = (perm >> (3 * (lane >> 3))) & 7u;
    const bool asc = code & 1u;
    // position of D in the sum; the first two terms commute (0 + x + y), so 0 counts as 1
    const uint32_t pd = code >> 1;
    const bool pd01 = pd <= 1, pd2 = pd == 2, pd3 = pd == 3;
    // octant across the diagonal: lane + 8 for the second halves, lane - 8 for the first
    const int diag_addr = (((lane & 8) ? lane + 8 : lane - 8) & 63) * 4;

    // idle lanes read, keep and pass on the spare cell: it must hold a finite value (0 times it
    // is added to their neighbours' sums)
    *lds(0) = 0.f;
    const float centre = us[centre_pix];
    float out[P], c1[P], c2[P], c3[P], cur[P];
    uint32_t a[6][P];
    f32x4 w[6][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        out[p] = c1[p] = c2[p] = c3[p] = centre;
#pragma unroll
        for (int i = 0; i < DA; ++i) a[i][p] = plan.addr(i, p);
#pragma unroll
        for (int i = 0; i < DW; ++i) w[i][p] = plan.weights(i, p);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) cur[p] = *lds(a[0][p] & kMask);

    // one level; I = step within the unrolled iteration (levels L = s + I + 1: axis pixels at
    // odd L, diagonal pixels at L = 3 r - 1)
    auto step = [&](auto Ic, int s) {
        constexpr int I = decltype(Ic)::value;
        float cur_next[P], f1[P], f9[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            w[(I + DW) % 6][p] = plan.weights(s + I + DW, p);
            a[(I + DA) % 6][p] = plan.addr(s + I + DA, p);
            cur_next[p] = *lds(a[(I + 1) % 6][p] & kMask);
            f1[p] = dpp_ror1(out[p]);
            f9[p] = dpp_ror9(out[p]);
        }
        float y = 0.f;
        int md = 8, pdiag = 0;
        if (I % 3 == 1) {  // ring (L + 1) / 3 ends on the diagonal
            const int rd = (s + I + 2) / 3;
            md = rd & 7;
            pdiag = (rd >> 3) & (P - 1);
            const float src = (P == 2 && pdiag) ? out[P - 1] : out[0];
            y = __builtin_bit_cast(
                float, __builtin_amdgcn_ds_bpermute(diag_addr, __builtin_bit_cast(int, src)));
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            constexpr int below = P - 1;  // (p + P - 1) % P for P <= 2 is p ^ (P - 1)
            c3[p] = c2[p];
            c2[p] = c1[p];
            c1[p] = inner ? f1[p] : f9[p ^ below];
            float A = c3[p], B = c2[p];
            if (I % 2 == 0) {  // ring (L + 1) / 2 starts on the axis
                const int ra = (s + I + 2) >> 1;
                const int ma = (ra <= rmax && ((ra >> 3) & (P - 1)) == p) ? (ra & 7) : 8;
                A = m == ma ? (inner ? f9[p] : f1[p ^ below]) : c3[p];
            }
            if (I % 3 == 1) B = (m == md && pdiag == p) ? y : c2[p];
            const f32x4 &wn = w[I][p];
            const float pA = __fmul_rn(A, wn.x), pB = __fmul_rn(B, wn.y);
            const float pC = __fmul_rn(c1[p], wn.z), pD = __fmul_rn(out[p], wn.w);
            const float e0 = asc ? pA : pC, e2 = asc ? pC : pA;
            float ref = __fadd_rn(0.f, e0);
            ref = __fadd_rn(ref, pd01 ? pD : pB);
            ref = __fadd_rn(ref, pd01 ? pB : (pd2 ? pD : e2));
            ref = __fadd_rn(ref, pd3 ? pD : e2);
            const float lim = __fmul_rn(ref, one_minus_g);
            out[p] = lim < cur[p] ? lim : cur[p];
            *lds(a[I][p] & kMask) = out[p];
            cur[p] = cur_next[p];
        }
    };
    for (int s = 0; s < (LATE ? n_nat : n_pad); s += kRingUnroll) {
        step(std::integral_constant<int, 0>(), s);
        step(std::integral_constant<int, 1>(), s);
        step(std::integral_constant<int, 2>(), s);
        step(std::integral_constant<int, 3>(), s);
        step(std::integral_constant<int, 4>(), s);
        step(std::integral_constant<int, 5>(), s);
    }
    if constexpr (LATE) {
        // the flagged part of the stream (c4, g2 need no start values: the first late entry
        // comes at least four steps in)
        float c4 = c3[0], g2 = out[0];
        auto late_step = [&](auto Ic, int s) {
            constexpr int I = decltype(Ic)::value;
            w[(I + DW) % 6][0] = plan.weights(s + I + DW, 0);
            a[(I + DA) % 6][0] = plan.addr(s + I + DA, 0);
            const float cur_next = *lds(a[(I + 1) % 6][0] & kMask);
            const float f1 = dpp_ror1(out[0]), f9 = dpp_ror9(out[0]);
            const float y = __builtin_bit_cast(
                float, __builtin_amdgcn_ds_bpermute(diag_addr, __builtin_bit_cast(int, out[0])));
            const float f_mir = inner ? f9 : f1;
            c4 = c3[0];
            c3[0] = c2[0];
            c2[0] = c1[0];
            c1[0] = inner ? f1 : f9;
            const uint32_t word = a[I][0];
            const bool late = word & kRingLate, axis = word & kRingAxis, dg = word & kRingDiag;
            const float A = axis ? (late ? g2 : f_mir) : (late ? c4 : c3[0]);
            const float B = dg ? y : (late ? c3[0] : c2[0]);
            const float C = late ? c2[0] : c1[0];
            g2 = f_mir;
            const f32x4 &wn = w[I][0];
            const float pA = __fmul_rn(A, wn.x), pB = __fmul_rn(B, wn.y);
            const float pC = __fmul_rn(C, wn.z), pD = __fmul_rn(out[0], wn.w);
            const float e0 = asc ? pA : pC, e2 = asc ? pC : pA;
            float ref = __fadd_rn(0.f, e0);
            ref = __fadd_rn(ref, pd01 ? pD : pB);
            ref = __fadd_rn(ref, pd01 ? pB : (pd2 ? pD : e2));
            ref = __fadd_rn(ref, pd3 ? pD : e2);
            const float lim = __fmul_rn(ref, one_minus_g);
            out[0] = lim < cur[0] ? lim : cur[0];
            *lds(word & kMask) = out[0];
            cur[0] = cur_next;
        };
        for (int s = n_nat; s < n_pad; s += kRingUnroll) {
            late_step(std::integral_constant<int, 0>(), s);
            late_step(std::integral_constant<int, 1>(), s);
            late_step(std::integral_constant<int, 2>(), s);
            late_step(std::integral_constant<int, 3>(), s);
            late_step(std::integral_constant<int, 4>(), s);
            late_step(std::integral_constant<int, 5>(), s);
        }
    }
}

// `pl` wave-uniform, its stream at `plan_lds`; kMaxPlanes: what the caller can meet (the update
// kernels stage one-plane plans only, refresh_view in batch.hip says why)
// kLate: plans with late rings (rings 24 .. 31 on the one plane) can occur
template <int kMaxPlanes, bool kLate = false>
__device__ __forceinline__ void sweep_ring(float *us, const SweepPlanDev &pl, const char *plan_lds,
                                           float one_minus_g, int lane) {
    const int n_pad = __builtin_amdgcn_readfirstlane(pl.ring_pad);
    const int n_nat = __builtin_amdgcn_readfirstlane(pl.ring_nat);
    const int rmax = __builtin_amdgcn_readfirstlane(pl.ring_rmax);
    const uint32_t perm = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl.ring_perm);
    const int centre = __builtin_amdgcn_readfirstlane(pl.ring_centre);
    if (kMaxPlanes == 2 && __builtin_amdgcn_readfirstlane(pl.ring_planes) == 2)
        sweep_ring_loop<2>(us, RingLds<2>(plan_lds, n_pad, lane), n_pad, n_pad, rmax, perm, centre,
                           one_minus_g, lane);
    else if (kLate && n_nat < n_pad)
        sweep_ring_loop<1, true>(us, RingLds<1>(plan_lds, n_pad, lane), n_pad, n_nat, rmax, perm,
                                 centre, one_minus_g, lane);
    else
        sweep_ring_loop<1>(us, RingLds<1>(plan_lds, n_pad, lane), n_pad, n_pad, rmax, perm, centre,
                           one_minus_g, lane);
}

// a one-plane plan straight from global memory (no staged copy: update_kernel_mixed)
template <bool kLate>
__device__ __forceinline__ void sweep_ring_global(float *us, const SweepPlanDev &pl,
                                                  float one_minus_g, int lane) {
    const int n_pad = __builtin_amdgcn_readfirstlane(pl.ring_pad);
    const int n_nat = __builtin_amdgcn_readfirstlane(pl.ring_nat);
    const int rmax = __builtin_amdgcn_readfirstlane(pl.ring_rmax);
    const uint32_t perm = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl.ring_perm);
    const int centre = __builtin_amdgcn_readfirstlane(pl.ring_centre);
    const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl.ring_bytes);
    const uint64_t sp = reinterpret_cast<uint64_t>(pl.ring);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sp);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sp >> 32));
    const void *stream = reinterpret_cast<const void *>((uint64_t)lo | ((uint64_t)hi << 32));
    const RingGlobal<1> plan(stream, bytes, n_pad, lane);
    if (kLate && n_nat < n_pad)
        sweep_ring_loop<1, true, RingGlobal<1>>(us, plan, n_pad, n_nat, rmax, perm, centre,
                                                one_minus_g, lane);
    else
        sweep_ring_loop<1, false, RingGlobal<1>>(us, plan, n_pad, n_pad, rmax, perm, centre,
                                                 one_minus_g, lane);
}

// occupancy the register allocator has to reach (waves per SIMD): three arrays of NPL
// registers (one for the teams, UpdState) + the sweep's prefetch; without the cap the
// scheduler trades waves for ILP
#ifndef SMI_XP_TEAM
#define SMI_XP_TEAM 1
#endif
constexpr bool update_xp(int team) { return SMI_XP_TEAM && team > 64; }  // (UpdState)
constexpr int update_waves(int npl, int team) {
    return npl <= 16 ? 4 : npl <= 27 ? (update_xp(team) ? 4 : 3) : 2;
}
#ifndef SMI_WAVES
#define SMI_WAVES __attribute__((amdgpu_waves_per_eu(update_waves(NPL, T))))
#endif
// MODE 0: Blend.fit, 1: lite with AdaproxParameter, 2: lite with FistaParameter
//
// The update of one component, in phases (UpdState holds what lives across them; it is
// always inlined into registers).
// The components of a launch all belong to one size class (`work` lists them class by
// class, common.h), so the per-pixel loops carry no bounds for the first kFull slots and a
// small box never runs through a large box's loops.
// XP (the four-wavefront teams): the pre-prox image x and the denominators psi live in global
// memory between the AMSGrad step and the sub-iterations (BatchView::xp_tmp, read back a few
// pixels ahead of their use), so that a lane carries ONE array of NPL registers -- z -- through
// the sub-iterations instead of three: 128 instead of 168 registers for the 71^2 / 81^2 boxes,
// four workgroups per CU instead of three while three of a team's four wavefronts idle
// through every sweep (cfg 4: update 0.516 -> 0.46 ms).  Same bits.  The one-wavefront classes
// keep all three in registers: they are bound by throughput, not by residency, and the extra
// stores and the read-back cost them 12 % (cfg 3: update 0.334 -> 0.374 ms at three or four
// wavefronts per SIMD alike; NOTEBOOK round 6).
template <int NPL, bool XP>
struct UpdState {
    CompCtx c;
    int k, flags, plan_id, bad, ctr, n_slots;
    float xs[XP ? 1 : NPL], rs[XP ? 1 : NPL], zs[NPL];
    float rpmax;      // XP: psi is stored unscaled
    bool fista_prox;  // XP, FISTA: the candidate is z itself
    float alpha, pmax, t_old;
    bool monotonic, fit_center;
    const SweepSlotEntry *slots;
    const SweepPlanDev *ring;  // the plan when it has a ring schedule
    const SweepPlanDev *staged;  // the plan whose ring stream the workgroup holds in LDS ...
    const char *plan_lds;        // ... there (nullptr: none)
    float one_minus_g, lthresh, cfloor, pfloor;
    const float *bg_level;
    float *us, *sed_new;
};

// every box of this size class has more than T * kFull pixels (common.h)
template <int NPL>
struct UpdFull {
    static constexpr int value = NPL == 7 ? 0 : NPL == 16 ? 7 : NPL == 27 ? 16 : NPL == 42 ? 27 : 42;
};

// gradient gather, spectrum update, AMSGrad (or FISTA) step of the image, set-up of the
// proximal sub-iterations
template <int NPL, int MODE, int T>
__device__ __forceinline__ void upd_step(const BatchView &v, const float *G, int it, float e2,
                                         int prox_max_iter, UpdState<NPL, update_xp(T)> &S) {
    constexpr bool LITE = MODE != 0;
    constexpr bool fista = MODE == 2;
    constexpr bool XP = update_xp(T);
    constexpr int kFull = UpdFull<NPL>::value;
    const CompCtx &c = S.c;
    it = v.local_it(c.b, it);
    const int lane = c.lane, k = S.k, N = c.N;
    float *us = S.us;
    // (XP: xs holds the gradient, then x, and is dead after this phase)
    float xs_local[NPL], rs_local[NPL];
    auto &xs = [&]() -> float(&)[NPL] {
        if constexpr (XP) return xs_local;
        else return S.xs;
    }();
    auto &rs = [&]() -> float(&)[NPL] {  // (XP: never touched)
        if constexpr (XP) return rs_local;
        else return S.rs;
    }();
    float(&zs)[NPL] = S.zs;
    const int flags = S.flags = v.c_flags[k];
    const int plan_id = S.plan_id = v.c_plan[k];
    const uint32_t nbytes = (uint32_t)N * 4u;

    // ---- gradient: xs = sum_c sed_c G_c[box], g_sed (lane c) = sum_yx G_c morph
    // (lite/models.py:206-216; slice of G into the box, zero outside the frame: blend.py:30-46)
    {
        const rsrc_t r_morph = make_rsrc(c.morph, nbytes);
#pragma unroll
        for (int j = 0; j < NPL; ++j) zs[j] = buf_load(r_morph, (uint32_t)(lane + T * j) * 4u);
    }
    float g_sed = 0.f;
    if (c.pre) {
        const rsrc_t r_g = make_rsrc(v.g_morph_buf + c.moff, nbytes);
#pragma unroll
        for (int j = 0; j < NPL; ++j) xs[j] = buf_load(r_g, (uint32_t)(lane + T * j) * 4u);
        g_sed = lane < c.C ? v.g_sed_buf[(int64_t)k * c.C + lane] : 0.f;
    } else {
        // byte offset of each of the lane's pixels inside a band plane of G (or out of
        // range), parked in the LDS image, which is not needed before the prox loop
        uint32_t *goff = reinterpret_cast<uint32_t *>(us);
        const float inv_w = 1.0f / (float)c.w;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = lane + T * j;
            // exact for i < 2^20: the float quotient is off by < 1e-6 relative
            const int y = (int)(((float)i + 0.5f) * inv_w);
            const int x = i - y * c.w;
            const int fy = y + c.oy, fx = x + c.ox;
            const bool ok = (j < kFull || i < N) && (unsigned)fy < (unsigned)v.H &&
                            (unsigned)fx < (unsigned)v.W;
            goff[i] = ok ? (uint32_t)(fy * v.Fx + fx) * 4u : kOutOfRange;
            xs[j] = 0.f;
        }
        const int64_t plane = (int64_t)v.Fy * v.Fx;
        for (int cb = 0; cb < c.C; ++cb) {
            const rsrc_t r_g = make_rsrc(G + ((int64_t)c.b * c.C + cb) * plane, (uint32_t)plane * 4u);
            float g[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) g[j] = buf_load(r_g, goff[lane + T * j]);
            const float s = c.sed[cb];
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                xs[j] = fmaf(s, g[j], xs[j]);
                acc = fmaf(g[j], zs[j], acc);
            }
            const float t = Team<T>::sum(acc);
            if (lane == cb) g_sed = t;
        }
    }
    // Parameter(fixed=True): the optimizer sees a zero gradient (blend.py:107-115)
    if (flags & SMI_COMPONENT_FIXED_MORPH) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) xs[j] = 0.f;
    }
    if (flags & SMI_COMPONENT_FIXED_SED) g_sed = 0.f;

    float msum = 0.f, msum2 = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        msum += zs[j];
        msum2 = fmaf(zs[j], zs[j], msum2);
    }
    float ssum2 = 1.f;
    S.t_old = 1.f;
    if (fista) {
        // sums over the *old* parameters: FISTA steps (lite/parameters.py:138)
        msum2 = Team<T>::sum(msum2);
        const float so = lane < c.C ? c.sed[lane] : 0.f;
        ssum2 = Team<T>::sum(so * so);
        S.t_old = (float)v.fista_t[2 * (int64_t)k + 1];
    }
    // the spectrum belongs to the first wavefront (one band per lane)
    S.bad = 0;
    if (T == 64 || threadIdx.x < 64)
        S.bad = update_spectrum(v, c, g_sed, it, e2, prox_max_iter, msum2, S.sed_new);
    const float alpha = fmaxf(v.c_morph_step[k], v.c_morph_rel[k] * (Team<T>::sum(msum) / (float)N));
    float pmax = 0.f;
    const rsrc_t r_m = make_rsrc(v.m_morph + c.moff, nbytes);
    const rsrc_t r_x = make_rsrc(v.xp_tmp + c.moff, nbytes);
    const rsrc_t r_p = make_rsrc(v.xp_tmp + v.n_morph_total + c.moff, nbytes);
    S.fista_prox = fista;
    if (fista) {
        const float step = v.c_fista_step[k] / ssum2;
        if constexpr (XP) {
            // (candidate = z - 0 * (z - x) = z: nothing to keep beside z)
#pragma unroll
            for (int j = 0; j < NPL; ++j) zs[j] = buf_load(r_m, (uint32_t)(lane + T * j) * 4u);
#pragma unroll
            for (int j = 0; j < NPL; ++j) zs[j] = zs[j] - step * xs[j];
        } else {
#pragma unroll
            for (int j = 0; j < NPL; ++j) rs[j] = buf_load(r_m, (uint32_t)(lane + T * j) * 4u);
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                xs[j] = rs[j] - step * xs[j];
                rs[j] = 0.f;
                zs[j] = xs[j];
            }
        }
        pmax = 1.f;
    } else {
        // AMSGrad moments (lite/parameters.py:274-291), CH pixels per lane at a time; the
        // loads of the next group are issued before the stores of this one
        const rsrc_t r_v = make_rsrc(v.v_morph + c.moff, nbytes);
        const rsrc_t r_vh = make_rsrc(v.vh_morph + c.moff, nbytes);
        constexpr int CH = NPL <= 6 ? NPL : 6;
        const float b1 = v.b1, b2 = v.b2, eps = v.eps;
        float cm[CH], cv[CH], cvh[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const uint32_t off = (uint32_t)(lane + T * u) * 4u;
            cm[u] = buf_load(r_m, off);
            cv[u] = buf_load(r_v, off);
            cvh[u] = buf_load(r_vh, off);
        }
#pragma unroll
        for (int j0 = 0; j0 < NPL; j0 += CH) {
            float nm[CH], nv[CH], nvh[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                nm[u] = nv[u] = nvh[u] = 0.f;
                if (j0 + CH + u < NPL) {
                    const uint32_t off = (uint32_t)(lane + T * (j0 + CH + u)) * 4u;
                    nm[u] = buf_load(r_m, off);
                    nv[u] = buf_load(r_v, off);
                    nvh[u] = buf_load(r_vh, off);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = j0 + u;
                if (j < NPL) {
                    const uint32_t off = (uint32_t)(lane + T * j) * 4u;
                    const float g = xs[j];
                    const float m = (1.f - b1) * g + b1 * cm[u];
                    const float vv = (1.f - b2) * g * g + b2 * cv[u];
                    const float vh = it == 0 ? vv : fmaxf(cvh[u], vv);
                    buf_store(r_m, off, m);
                    buf_store(r_v, off, vv);
                    buf_store(r_vh, off, vh);
                    const float psi = sqrtf(fmaxf(vh, eps));
                    float upd = alpha * m / psi;
                    if (it == 0) upd /= 10.f;
                    xs[j] = zs[j] - upd;
                    zs[j] = xs[j];
                    // slots beyond the box: x = z = 0 (loads returned 0), psi must not count
                    const bool in_box = j < kFull || lane + T * j < N;
                    if constexpr (XP) {
                        const float rj = in_box ? psi : 0.f;
                        buf_store(r_x, off, xs[j]);
                        buf_store(r_p, off, rj);
                        pmax = fmaxf(pmax, rj);
                    } else {
                        rs[j] = in_box ? psi : 0.f;
                        pmax = fmaxf(pmax, rs[j]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                cm[u] = nm[u];
                cv[u] = nv[u];
                cvh[u] = nvh[u];
            }
            // keep the scheduler from hoisting every group's loads to the top (registers)
            __builtin_amdgcn_sched_barrier(0);
        }
        pmax = Team<T>::max(pmax);
    }
    const float rpmax = 1.f / pmax;
    S.rpmax = rpmax;
    if constexpr (!XP) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) rs[j] = rs[j] * rpmax;
    }
    S.alpha = alpha;
    S.pmax = pmax;

    S.monotonic = (flags & SMI_PROX_MONOTONIC) && plan_id >= 0;
    S.fit_center = LITE && S.monotonic && (flags & SMI_PROX_FIT_CENTER);
    S.slots = nullptr;
    S.ring = nullptr;
    S.n_slots = 0;
    if (S.monotonic && !S.fit_center) {
        S.slots = v.plans[plan_id].slots;
        S.n_slots = v.plans[plan_id].n_slots;
        S.ring = v.plans[plan_id].ring ? &v.plans[plan_id] : nullptr;
    }
    S.one_minus_g = 1.f - v.c_min_grad[k];
    S.ctr = (c.h / 2) * c.w + (c.w / 2);
    S.lthresh = v.c_lthresh[k] * ((flags & SMI_PROX_L_RELATIVE) ? alpha / pmax : 1.f);
    S.cfloor = v.c_center_floor[k];
    S.pfloor = v.c_pos_floor[k];  // PositivityConstraint(zero)
    S.bg_level = (LITE && v.c_bg_level) ? v.c_bg_level + (int64_t)k * c.C : nullptr;
}

// candidate of a proximal sub-iteration into the LDS image
template <int NPL, int T>
__device__ __forceinline__ void upd_prox_begin(const BatchView &v, UpdState<NPL, update_xp(T)> &S) {
    const int lane = S.c.lane;
    if constexpr (update_xp(T)) {
        if (S.fista_prox) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) S.us[lane + T * j] = S.zs[j];
            return;
        }
        // x and psi back from memory (written by this lane in upd_step), CH pixels per lane at a
        // time, the next group requested before this one is used
        const uint32_t nbytes = (uint32_t)S.c.N * 4u;
        const rsrc_t r_x = make_rsrc(v.xp_tmp + S.c.moff, nbytes);
        const rsrc_t r_p = make_rsrc(v.xp_tmp + v.n_morph_total + S.c.moff, nbytes);
        constexpr int CH = NPL <= 8 ? NPL : 8;
        const float rpmax = S.rpmax;
        float cx[CH], cp[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const uint32_t off = (uint32_t)(lane + T * u) * 4u;
            cx[u] = buf_load(r_x, off);
            cp[u] = buf_load(r_p, off);
        }
#pragma unroll
        for (int j0 = 0; j0 < NPL; j0 += CH) {
            float nx[CH], np[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                nx[u] = np[u] = 0.f;
                if (j0 + CH + u < NPL) {
                    const uint32_t off = (uint32_t)(lane + T * (j0 + CH + u)) * 4u;
                    nx[u] = buf_load(r_x, off);
                    np[u] = buf_load(r_p, off);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = j0 + u;
                if (j < NPL) {
                    const float r = cp[u] * rpmax;
                    S.us[lane + T * j] = S.zs[j] - r * (S.zs[j] - cx[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                cx[u] = nx[u];
                cp[u] = np[u];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j) S.us[lane + T * j] = S.zs[j] - S.rs[j] * (S.zs[j] - S.xs[j]);
    }
}

// MonotonicityConstraint(fit_center_radius=1): the plan of this sub-iteration
template <int NPL, bool XP>
__device__ __forceinline__ void upd_prox_plan(const BatchView &v, UpdState<NPL, XP> &S) {
    if (S.fit_center) {
        const int centre = __builtin_amdgcn_readfirstlane(fit_center_index(S.us, S.c));
        const SweepPlanDev &pl = v.plans[S.plan_id + centre];
        S.slots = pl.slots;
        S.n_slots = pl.n_slots;
        S.ring = pl.ring ? &pl : nullptr;
    }
}

// rest of the chain after the sweep, convergence test; true when the sub-iterations end
template <int NPL, int MODE, int T>
__device__ __forceinline__ bool upd_prox_end(const BatchView &v, float e2,
                                             UpdState<NPL, update_xp(T)> &S) {
    constexpr bool LITE = MODE != 0;
    constexpr int kFull = UpdFull<NPL>::value;
    const CompCtx &c = S.c;
    const int lane = c.lane, N = c.N, flags = S.flags, ctr = S.ctr;
    const float pfloor = S.pfloor, cfloor = S.cfloor;
    float *us = S.us;
    chain_symmetry_threshold<T>(us, c, LITE ? flags : (flags & ~SMI_PROX_BG_THRESH), S.lthresh,
                                S.sed_new, S.bg_level,
                                (flags & SMI_PROX_SYMMETRY) ? v.c_sym_strength[S.k] : 1.f);
    float mx = -INFINITY, sm = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int i = lane + T * j;
        float u = us[i];
        if (flags & SMI_PROX_POSITIVE) u = max_nan(u, pfloor);
        if ((flags & SMI_PROX_CENTER_ON) && i == ctr) u = max_nan(u, cfloor);
        mx = (j < kFull || i < N) ? fmaxf(mx, u) : mx;
        sm += (j < kFull || i < N) ? u : 0.f;
    }
    float div = 1.f;
    if (flags & SMI_PROX_NORM_MAX) div = Team<T>::max(mx);
    if (flags & SMI_PROX_NORM_SUM) div = Team<T>::sum(sm);
    // one correctly rounded reciprocal per sub-iteration instead of N divisions; the
    // maximum itself still maps to exactly 1 (x / x), everything else is within
    // 1 ulp of the quotient
    const float rdiv = 1.f / div;
    float d2 = 0.f, z2 = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int i = lane + T * j;
        float u = us[i];  // second read instead of NPL more registers
        if (flags & SMI_PROX_POSITIVE) u = max_nan(u, pfloor);
        if ((flags & SMI_PROX_CENTER_ON) && i == ctr) u = max_nan(u, cfloor);
        if (flags & (SMI_PROX_NORM_MAX | SMI_PROX_NORM_SUM))
            u = (u == div && (flags & SMI_PROX_NORM_MAX)) ? 1.f : u * rdiv;
        if (!(j < kFull || i < N)) u = 0.f;  // slots beyond the box stay zero
        d2 += (u - S.zs[j]) * (u - S.zs[j]);
        z2 += S.zs[j] * S.zs[j];
        S.zs[j] = u;
    }
    d2 = Team<T>::sum(d2);
    z2 = Team<T>::sum(z2);
    return d2 <= e2 * z2;
}

// parameters back to memory, finite check
template <int NPL, int MODE, int T>
__device__ __forceinline__ void upd_store(const BatchView &v, UpdState<NPL, update_xp(T)> &S) {
    constexpr bool fista = MODE == 2;
    const CompCtx &c = S.c;
    const int lane = c.lane;
    const uint32_t nbytes = (uint32_t)c.N * 4u;
    float omega = 0.f;
    if (fista) {
        const float t_old = S.t_old;
        const float tn = 0.5f * (1.f + sqrtf(1.f + 4.f * t_old * t_old));
        omega = 1.f + (t_old - 1.f) / tn;
        if (lane == 0) v.fista_t[2 * (int64_t)S.k + 1] = (double)tn;
    }
    const rsrc_t r_out = make_rsrc(c.morph_out, nbytes);
    if (fista) {
        const rsrc_t r_m = make_rsrc(v.m_morph + c.moff, nbytes);
        float xo[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) xo[j] = buf_load(r_out, (uint32_t)(lane + T * j) * 4u);
#pragma unroll
        for (int j = 0; j < NPL; ++j)
            buf_store(r_m, (uint32_t)(lane + T * j) * 4u, xo[j] + omega * (S.zs[j] - xo[j]));
    }
    int bad = S.bad;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        buf_store(r_out, (uint32_t)(lane + T * j) * 4u, S.zs[j]);
        bad |= !isfinite(S.zs[j]);
    }
    if (Team<T>::any(bad) && lane == 0) atomicExch(&v.state[c.b], v.fail_code);  // model.py:153-165
}

// kGlobalRing: components without a staged plan read the ring stream out of L2 (update_kernel_mixed)
template <int NPL, int MODE, int T = 64, bool kGlobalRing = false>
__device__ __forceinline__ void update_component(const BatchView &v, const float *G, int it,
                                                 float e_rel, int prox_max_iter, int k,
                                                 float *us, float *sed_new,
                                                 const SweepPlanDev *staged = nullptr,
                                                 const char *plan_lds = nullptr, int tid = -1) {
    constexpr bool fista = MODE == 2;
    UpdState<NPL, update_xp(T)> S;
    S.k = k;
    S.staged = staged;
    S.plan_lds = plan_lds;
    if (tid < 0) tid = (int)threadIdx.x;
    S.c = comp_ctx(v, k, T == 64 ? (tid & 63) : tid);
    if (v.state[S.c.b] >= 2) return;
    S.us = us;
    S.sed_new = sed_new;
    const float e2 = e_rel * e_rel;
    if (fista) prox_max_iter = 1;  // FistaParameter applies the prox once
    upd_step<NPL, MODE, T>(v, G, it, e2, prox_max_iter, S);
    team_fence<T>();  // the offsets parked in `us` have been consumed
    for (int tau = 0; tau < prox_max_iter; ++tau) {
        upd_prox_begin<NPL, T>(v, S);
        team_fence<T>();
        upd_prox_plan(v, S);
        if (S.monotonic) {
            // one wavefront sweeps (the steps are sequential and at most 64 pixels wide)
            if (T == 64 || threadIdx.x < 64) {
                if (S.ring && S.ring == S.staged)
                    // (boxes beyond 47^2 -- the classes beyond 27 pixels per lane -- have rings
                    // that start late)
                    sweep_ring<1, (NPL > 27)>(S.us, *S.ring, S.plan_lds, S.one_minus_g, S.c.lane);
                else if (kGlobalRing && T == 64 && S.ring &&
                         __builtin_amdgcn_readfirstlane(S.ring->ring_planes) == 1)
                    // (no staged copy -- update_kernel_mixed, small launches: the stream out of L2,
                    // 256 B a step: 61^2 21.5 k clocks per sweep, 41^2 12.0 k; slot plan 27.4 k, 15.4 k)
                    sweep_ring_global<(NPL > 27)>(S.us, *S.ring, S.one_minus_g, S.c.lane);
                else
                    sweep_slots(S.us, S.slots, S.n_slots, S.one_minus_g, S.c.lane);
            }
            if (T > 64) __syncthreads();
        }
        if (upd_prox_end<NPL, MODE, T>(v, e2, S)) break;
    }
    upd_store<NPL, MODE, T>(v, S);
}

// XCD-aware placement of consecutive work items (workgroup b runs on XCD b % 8): XCD x gets
// a contiguous share of the list, so the components of one blend -- neighbours in the work
// list -- read the blend's gradient image through one L2 instead of eight.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return x * q + min(x, r) + (b >> 3);
}

// Wavefronts per workgroup of the one-wavefront classes: with `pack` > 1 a workgroup holds
// the components of `pack` consecutive work items, one per wavefront, nothing shared but the
// CU.  What it is for: the hardware spreads workgroups over the chip, so a launch of a few
// hundred one-wavefront workgroups (one GPU's shard of a multi-GPU job: 128 blends = 427
// components per range) leaves one or two of them on EVERY CU for its whole latency-bound
// ~0.1 ms -- and a CU that holds one cannot take a workgroup of the convolution kernel
// (157 KB of LDS, the whole register file), so the other ranges' convolutions queue up
// behind it.  Packed three to a SIMD the same components occupy 36 CUs and the rest of the
// chip stays free.  Every wavefront runs exactly what it would run alone.  Measured with four
// ranges on eight hardware queues (k blend-iterations/s, one wavefront per workgroup -> packed):
// 128 blends 553 -> 613, 256 blends 749 -> 772, 512 blends (1280 components per range)
// 872 -> 810: hence kUpdatePackLimit = 1024 components.
constexpr int update_pack_max(int npl, int team) {
    return team != 64 ? 1 : 4 * update_waves(npl, team);
}
// With `stage_plan` >= 0 the workgroup first copies that plan's ring stream into the LDS behind
// its images (common.h: RingPlanHost): the sweeps of the components that use the plan -- in a
// batch of standard sources all of them -- read the stream from there (one ds_read_b128 and
// one ds_read_u16 per step) instead of pulling 0.6 KB per step and wavefront through the
// vector memory pipe, which is what bounded a full batch.  With `persistent` the workgroup owns
// a contiguous share of the work list and its wavefronts take component after component from
// it (a counter in LDS) until it is used up -- a workgroup that holds 150 KB of LDS must not
// idle behind its slowest component.  (One global counter per XCD, measured: 10 240 atomic
// additions on eight addresses take 0.25 ms, longer than the updates themselves.)
template <int NPL, int MODE, int T>
__global__ __launch_bounds__(T * update_pack_max(NPL, T)) SMI_WAVES void update_kernel_reg(
    BatchView v, const float *G, int it, float e_rel, int prox_max_iter, int n_items,
    int n_finalize, int min_iter, int check, int stage_plan, int persistent) {
    constexpr int kPack = update_pack_max(NPL, T);
    __shared__ float sed_new[kPack][64];
    __shared__ int next_item;
    // (wave-uniform, and said so: the work item must stay in scalar registers)
    const int wave = T == 64 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int pack = T == 64 ? (int)(blockDim.x >> 6) : 1;
    const int n_blocks = (int)gridDim.x - n_finalize;
    // the last n_finalize workgroups do the loss bookkeeping of the range's blends (one
    // each): independent of the updates, so it rides along instead of being a launch of its
    // own between the convolution and the updates (finalize_blend)
    if ((int)blockIdx.x >= n_blocks) {
        if (threadIdx.x < 64)
            finalize_blend(v, (int)blockIdx.x - n_blocks + v.blend0, it, e_rel, min_iter, check);
        return;
    }
    float *us = lds_dyn + wave * (T * NPL + 4) + 4;
    const SweepPlanDev *staged = nullptr;
    const char *plan_lds = nullptr;
    if (T == 64 && (stage_plan >= 0 || persistent)) {
        if (threadIdx.x == 0) next_item = 0;
        if (stage_plan >= 0) {
            staged = &v.plans[stage_plan];
            u32x4 *dst = reinterpret_cast<u32x4 *>(lds_dyn + pack * (T * NPL + 4));
            const u32x4 *src = static_cast<const u32x4 *>(staged->ring);
            const uint32_t n16 = staged->ring_bytes >> 4;
            for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
            plan_lds = reinterpret_cast<const char *>(dst);
        }
        __syncthreads();
    }
    // persistent: share blockIdx.x of the list (contiguous: the components of a blend stay
    // together, in one L2 and mostly in one CU); otherwise one item per wavefront
    const bool loop = T == 64 && persistent;
    // (shares in the order of xcd_contiguous: the blend that straddles two shares is read
    // through one L2 by both)
    const int q = n_items / n_blocks, r = n_items % n_blocks;
    const int b = loop ? xcd_contiguous((int)blockIdx.x, n_blocks) : (int)blockIdx.x;
    const int lo = loop ? b * q + (b < r ? b : r) : 0;
    const int hi = loop ? lo + q + (b < r) : n_items;
    const int lane = (int)(threadIdx.x & 63);
    for (;;) {
        int k = xcd_contiguous(blockIdx.x, n_blocks) * pack + wave;
        if (loop) {
            if (lane == 0) k = atomicAdd(&next_item, 1);
            k = __builtin_amdgcn_readfirstlane(k);
        }
        if (lo + k >= hi) break;
        // (the thread index goes in through an opaque copy: what a component's phases derive
        // from it must not be hoisted out of this loop and live across all of them)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        update_component<NPL, MODE, T>(v, G, it, e_rel, prox_max_iter, v.work[lo + k + v.work0],
                                       us, sed_new[wave], staged, plan_lds, tid);
        if (!loop) break;
    }
}

// Components of several size classes (a blend with boxes of 21^2 .. 61^2 pixels): a launch
// per class runs the classes one after the other, each of them bounded below by the serial
// chain of one wave (gather, up to ten sub-iterations of 30 .. 90 sweep steps).  Here one
// launch holds every component of the range and each wavefront takes the code of its own
// class, so the chains of the classes overlap; the registers are those of the largest class
// (two waves per SIMD).  Measured on batches of the quickstart blend (k blend-it/s, launch
// per class -> this kernel): 16 blends 50 -> 103, 128: 340 -> 567, 512: 857 -> 1003, 768:
// 1027 -> 1062, 1024: 1159 -> 1145 (three ranges of 3413 components: the occupancy of the
// small classes starts to count), hence kMixedUpdateLimit.
// WAVES = wavefronts per SIMD the register allocator aims at.  2: 256 registers, 244 - 268 B of
// scratch.  1 (round 6): the whole register file of a SIMD for one wavefront -- what does not
// fit the 256 architectural registers is parked in accumulation registers (110 - 119 of them,
// v_accvgpr moves) instead of scratch memory: no spill traffic on the chain of a lone blend.
// Quickstart Blend.fit(100, 1e-4) 18.5 -> 17.2 ms, 16 quickstart blends 124 k -> 135 k
// blend-it/s, 128 blends 765 k -> 782 k; 256 blends 1 224 k -> 1 040 k: a batch of more than
// ~1 000 components needs the second wavefront per SIMD (launch_update picks).
template <int MODE, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void update_kernel_mixed(
    BatchView v, const float *G, int it, float e_rel, int prox_max_iter, int n_finalize,
    int min_iter, int check) {
    __shared__ float sed_new[64];
    if ((int)blockIdx.x >= v.n_comp) {  // loss bookkeeping riding along (update_kernel_reg)
        finalize_blend(v, (int)blockIdx.x - v.n_comp + v.blend0, it, e_rel, min_iter, check);
        return;
    }
    const int k = v.comp0 + blockIdx.x;
    if (v.c_flags[k] & SMI_COMPONENT_POINT_SOURCE) return;
    const int n = v.c_h[k] * v.c_w[k];  // uniform over the wavefront
    float *us = lds_dyn + 4;  // (spare cell of the sweep in front)
    if (n <= 64 * kUpdateNpl[0])
        update_component<kUpdateNpl[0], MODE, 64, true>(v, G, it, e_rel, prox_max_iter, k, us, sed_new);
    else if (n <= 64 * kUpdateNpl[1])
        update_component<kUpdateNpl[1], MODE, 64, true>(v, G, it, e_rel, prox_max_iter, k, us, sed_new);
    else if (n <= 64 * kUpdateNpl[2])
        update_component<kUpdateNpl[2], MODE, 64, true>(v, G, it, e_rel, prox_max_iter, k, us, sed_new);
    else if (n <= 64 * kUpdateNpl[3])
        update_component<kUpdateNpl[3], MODE, 64, true>(v, G, it, e_rel, prox_max_iter, k, us, sed_new);
    else
        update_component<kUpdateNpl[4], MODE, 64, true>(v, G, it, e_rel, prox_max_iter, k, us, sed_new);
}

// development aid: shader clocks of `n_rep` sweeps of one plan per wavefront, every wavefront
// on an image of its own in LDS (mode 0: slot plan, 2: ring schedule with the plan stream staged
// behind the images); the images come back for comparison
__global__ void sweep_timing_kernel(const SweepPlanDev *plans, int plan_id, int mode, int n_rep,
                                    float one_minus_g, long long *cycles, float *images) {
    const SweepPlanDev &pl = plans[plan_id];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = pl.h * pl.w, stride = ((n + 3) & ~3) + 4;
    float *us = lds_dyn + wave * stride + 4;
    const int gw = blockIdx.x * (blockDim.x >> 6) + wave;
    us[-4 + (lane & 3)] = 0.f;
    // mode 2: the plan stream staged behind the images
    char *plan_lds = reinterpret_cast<char *>(lds_dyn + (blockDim.x >> 6) * stride);
    if (mode == 2) {
        const uint32_t *src = static_cast<const uint32_t *>(pl.ring);
        for (uint32_t i = threadIdx.x; i < pl.ring_bytes / 4; i += blockDim.x)
            reinterpret_cast<uint32_t *>(plan_lds)[i] = src[i];
        __syncthreads();
    }
    long long total = 0;
    for (int rep = 0; rep < n_rep; ++rep) {
        for (int i = lane; i < n; i += 64) {
            uint32_t hsh = (uint32_t)(i * 2654435761u) ^ (uint32_t)(gw * 40503u + rep * 977u);
            hsh ^= hsh >> 13;
            hsh *= 0x5bd1e995u;
            hsh ^= hsh >> 15;
            us[i] = (float)(hsh & 0xffff) * (1.f / 65536.f);
        }
        wave_lds_fence();
        const long long t0 = __builtin_readcyclecounter();
        if (mode == 3)
            sweep_ring_global<true>(us, pl, one_minus_g, lane);
        else if (mode == 2)
            sweep_ring<2, true>(us, pl, plan_lds, one_minus_g, lane);
        else
            sweep_slots(us, pl.slots, pl.n_slots, one_minus_g, lane);
        wave_lds_fence();
        total += __builtin_readcyclecounter() - t0;
    }
    if (lane == 0) cycles[gw] = total;
    for (int i = lane; i < n; i += 64) images[(int64_t)gw * n + i] = us[i];
}

// ---------------------------------------------------------------------------
// Box resizing of image morphologies (morphology.py:132-207), the device's share: the two
// reductions ImageMorphology.update() decides on, for every component of the batch, so that
// only blends in which a box really changes have to visit the host.
//   margin[k] = width of the frame of pixels <= 0 around the image (shrink_box: min over the
//               pixels > 0 of their distance to the nearest edge; INT32_MAX if none is > 0)
//   pull[k]   = largest of the four edge means of  -m / sqrt(sqrt(v)) * step * (image > 0),
//               pixels with v == 0 left out as the masked array leaves them out
//               (-inf if every edge pixel is left out).  Double precision like the host, but
//               summed in another order and with the float32 step: a filter -- the caller
//               asks the host for the verdict on everything near the threshold.
// One wavefront per component; point sources and shifted images report (-1, nan).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void resize_test_kernel(BatchView v, int32_t *margin, double *pull) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const int flags = v.c_flags[k];
    if (flags & (SMI_COMPONENT_POINT_SOURCE | SMI_COMPONENT_SHIFTING)) {
        if (lane == 0) {
            margin[k] = -1;
            pull[k] = (double)NAN;
        }
        return;
    }
    const int h = v.c_h[k], w = v.c_w[k], N = h * w;
    const int64_t off = v.c_moff[k];
    const float *img = v.morph + off, *m = v.m_morph + off, *vv = v.v_morph + off;
    const double step = (double)v.c_morph_step[k];
    int mn = 0x7fffffff;
    double sum[4] = {0, 0, 0, 0};
    int cnt[4] = {0, 0, 0, 0};
    for (int i = lane; i < N; i += 64) {
        const int y = i / w, x = i - y * w;
        const float val = img[i];
        if (val > 0.f) mn = min(mn, min(min(y, h - 1 - y), min(x, w - 1 - x)));
        const bool edge[4] = {x == 0, x == w - 1, y == 0, y == h - 1};
        if ((edge[0] || edge[1] || edge[2] || edge[3]) && vv[i] != 0.f) {
            const double g = -(double)m[i] / sqrt(sqrt((double)vv[i])) * step * (val > 0.f ? 1.0 : 0.0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (edge[e]) {
                    sum[e] += g;
                    cnt[e]++;
                }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));
    double best = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double t = wave_sum(sum[e]);
        int c = cnt[e];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (c > 0) best = fmax(best, t / (double)c);
    }
    if (lane == 0) {
        margin[k] = mn;
        pull[k] = best;
    }
}

// State of a component as one record of floats: [sed C][m C][v C][vhat C] then
// [image N][m N][v N][vhat N] (N = h w): what a resize on the host needs and hands back.
// gather: component sel[j] -> staging + off[j]; scatter: staging + off[k] -> component k for
// every k with off[k] >= 0.  carry: the four pixel arrays of the components with keep[k] != 0
// from their old offsets to the new ones (the boxes, hence the packing, changed around them).
__global__ __launch_bounds__(256) void gather_states_kernel(BatchView v, const int32_t *sel,
                                                            const int64_t *off, float *staging) {
    const int k = sel[blockIdx.x], C = v.C, N = v.c_h[k] * v.c_w[k];
    float *dst = staging + off[blockIdx.x];
    const float *small[4] = {v.sed, v.m_sed, v.v_sed, v.vh_sed};
    const float *px[4] = {v.morph, v.m_morph, v.v_morph, v.vh_morph};
    const int64_t moff = v.c_moff[k];
    for (int i = threadIdx.x; i < 4 * C; i += 256) dst[i] = small[i / C][(int64_t)k * C + i % C];
    for (int a = 0; a < 4; ++a)
        for (int i = threadIdx.x; i < N; i += 256) dst[4 * C + (int64_t)a * N + i] = px[a][moff + i];
}

__global__ __launch_bounds__(256) void scatter_states_kernel(BatchView v, const int64_t *off,
                                                             const float *staging) {
    const int k = blockIdx.x;
    if (off[k] < 0) return;
    const int C = v.C, N = v.c_h[k] * v.c_w[k];
    const float *src = staging + off[k];
    float *small[4] = {v.sed, v.m_sed, v.v_sed, v.vh_sed};
    float *px[4] = {v.morph, v.m_morph, v.v_morph, v.vh_morph};
    const int64_t moff = v.c_moff[k];
    for (int i = threadIdx.x; i < 4 * C; i += 256) small[i / C][(int64_t)k * C + i % C] = src[i];
    for (int a = 0; a < 4; ++a)
        for (int i = threadIdx.x; i < N; i += 256) px[a][moff + i] = src[4 * C + (int64_t)a * N + i];
}

struct PixelArrays {
    float *p[4];
};
// keep[k]: 1 the box stays -- the four pixel arrays move to their new offsets; 2 / 3 the box is
// resized about its centre the way ImageMorphology.update does it (morphology.py:132-207):
// square boxes, old side ow, new side nw, pad = (nw - ow) / 2.
//   pad < 0 (shrink_box): the centred slice of image and moments;
//   pad > 0: the moments are padded with zeros, the image with np.pad(mode="linear_ramp") --
//   axis 0 over the original columns, then axis 1 over all rows, each ramp
//   np.linspace(0, edge, pad, endpoint=False) evaluated in float64: i * (edge / pad), or
//   (i / pad) * edge as soon as one entry of that edge line is 0 (numpy's any_step_zero branch),
//   rounded to float32.  keep = 3: the host image is a float64 array, so the rows added along
//   axis 0 enter the ramps along axis 1 unrounded.
__device__ __forceinline__ double linear_ramp(int i, double edge, int pad, bool any_zero) {
    return any_zero ? ((double)i / (double)pad) * edge : (double)i * (edge / (double)pad);
}
__global__ __launch_bounds__(256) void carry_states_kernel(const int32_t *keep, const int64_t *old_moff,
                                                           const int64_t *new_moff, PixelArrays from,
                                                           PixelArrays to) {
    const int k = blockIdx.x, code = keep[k], tid = threadIdx.x;
    if (!code) return;
    const int64_t src = old_moff[k], dst = new_moff[k];
    const int N = (int)(old_moff[k + 1] - src);
    if (code == 1) {
        for (int a = 0; a < 4; ++a)
            for (int i = tid; i < N; i += 256) to.p[a][dst + i] = from.p[a][src + i];
        return;
    }
    const int M = (int)(new_moff[k + 1] - dst);
    const int ow = (int)(sqrtf((float)N) + 0.5f), nw = (int)(sqrtf((float)M) + 0.5f);
    const int pad = (nw - ow) / 2;
    for (int a = pad > 0 ? 1 : 0; a < 4; ++a)
        for (int i = tid; i < M; i += 256) {
            const int y = i / nw - pad, x = i % nw - pad;
            const bool inside = y >= 0 && y < ow && x >= 0 && x < ow;
            to.p[a][dst + i] = inside ? from.p[a][src + y * ow + x] : 0.f;
        }
    if (pad <= 0) return;
    __shared__ double edge_l[1024], edge_r[1024];
    const bool wide = code == 3;
    const float *f = from.p[0] + src;
    float *t = to.p[0] + dst;
    int zt = 0, zb = 0;
    for (int x = tid; x < ow; x += 256) {
        zt |= f[x] == 0.f;
        zb |= f[(ow - 1) * ow + x] == 0.f;
    }
    zt = __syncthreads_or(zt);
    zb = __syncthreads_or(zb);
    for (int i = tid; i < nw * ow; i += 256) {
        const int y = i / ow, x = i - y * ow;
        double val;
        if (y < pad)
            val = linear_ramp(y, (double)f[x], pad, zt);
        else if (y >= pad + ow)
            val = linear_ramp(nw - 1 - y, (double)f[(ow - 1) * ow + x], pad, zb);
        else
            val = (double)f[(y - pad) * ow + x];
        const float r = (float)val;
        t[y * nw + pad + x] = r;
        if (x == 0) edge_l[y] = wide ? val : (double)r;
        if (x == ow - 1) edge_r[y] = wide ? val : (double)r;
    }
    __syncthreads();
    int zl = 0, zr = 0;
    for (int y = tid; y < nw; y += 256) {
        zl |= edge_l[y] == 0.0;
        zr |= edge_r[y] == 0.0;
    }
    zl = __syncthreads_or(zl);
    zr = __syncthreads_or(zr);
    for (int i = tid; i < nw * 2 * pad; i += 256) {
        const int y = i / (2 * pad), j = i - y * 2 * pad;
        if (j < pad)
            t[y * nw + j] = (float)linear_ramp(j, edge_l[y], pad, zl);
        else
            t[y * nw + nw - 1 - (j - pad)] = (float)linear_ramp(j - pad, edge_r[y], pad, zr);
    }
}

// data and weights by row pairs for the fused convolution kernel (BatchView::dw); grid.y = plane
__global__ __launch_bounds__(256) void interleave_obs_kernel(const float *data, const float *weights,
                                                            float4 *dw, int H, int W) {
    const int np = (H + 1) / 2;
    const int64_t plane = blockIdx.y;
    const float *d = data + plane * H * W, *w = weights + plane * H * W;
    float4 *o = dw + plane * np * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < np * W; i += gridDim.x * 256) {
        const int j = i / W, x = i - j * W;
        const bool two = 2 * j + 1 < H;
        o[i] = make_float4(d[2 * j * W + x], two ? d[(2 * j + 1) * W + x] : 0.f, w[2 * j * W + x],
                           two ? w[(2 * j + 1) * W + x] : 0.f);
    }
}

// log_norm of Observation (observation.py:172-186): D/2 ln(2 pi) + sum ln(1/sqrt(w))
__global__ __launch_bounds__(256) void log_norm_kernel(const float *weights, double *out,
                                                       int64_t n) {
    const int b = blockIdx.x;
    double cnt = 0.0, sl = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float wv = weights[(int64_t)b * n + i];
        if (wv != 0.f) {
            cnt += 1.0;
            sl += log((double)wv);
        }
    }
    cnt = wave_sum(cnt);
    sl = wave_sum(sl);
    __shared__ double pc[4], ps[4];
    if ((threadIdx.x & 63) == 0) {
        pc[threadIdx.x >> 6] = cnt;
        ps[threadIdx.x >> 6] = sl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double c = pc[0] + pc[1] + pc[2] + pc[3];
        const double s = ps[0] + ps[1] + ps[2] + ps[3];
        out[b] = 0.5 * c * 1.8378770664093453 - 0.5 * s;  // ln(2 pi)
    }
}

// kernel stamp -> FFT input with the stamp centre (ph/2, pw/2) at index (0,0),
// wrapped around (the layout the reference reaches with _pad + ifftshift,
// fft.py:255-273), scaled by 1/(Fy Fx) for the unnormalised inverse transform
__global__ void wrap_kernel_kernel(const float *kern, float *out, int ph, int pw, int Fy,
                                   int Fx, float scale) {
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ph * pw) return;
    const int ky = i / pw, kx = i - ky * pw;
    int ty = ky - ph / 2, tx = kx - pw / 2;
    if (ty < 0) ty += Fy;
    if (tx < 0) tx += Fx;
    out[((int64_t)img * Fy + ty) * Fx + tx] = kern[(int64_t)img * ph * pw + i] * scale;
}

__global__ void crop_kernel(const float *P, float *out, int H, int W, int Fy, int Fx) {
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    out[(int64_t)img * H * W + i] = P[((int64_t)img * Fy + y) * Fx + x];
}

// ---- seam 1 kernels --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void sweep_kernel(T *img, int n_pix, const int32_t *level_start,
                                                   int n_levels, int E, const int32_t *pix,
                                                   const int32_t *cnt, const int32_t *nbr,
                                                   const T *wt, T one_minus_g) {
    T *buf = reinterpret_cast<T *>(lds_dyn);
    const int lane = threadIdx.x;
    for (int i = lane; i < n_pix; i += 64) buf[i] = img[i];
    __syncthreads();
    sweep_levels<T, T>(buf, level_start, n_levels, E, pix, cnt, nbr, wt, one_minus_g, lane);
    for (int i = lane; i < n_pix; i += 64) img[i] = buf[i];
}

// One workgroup per image, every image with a plan of its own inside the concatenated plan
// arrays (smi_prox_weighted_monotonic_many_*: the detection images of a scene's sources).
struct SweepManyDesc {
    int64_t level_off, entry_off, term_off;  // into level_start / pix, cnt / nbr, wt
    int32_t n_levels, n_entries;
};
template <typename T>
__global__ __launch_bounds__(256) void sweep_many_kernel(T *images, int n_pix,
                                                         const SweepManyDesc *desc,
                                                         const int32_t *level_start,
                                                         const int32_t *pix, const int32_t *cnt,
                                                         const int32_t *nbr, const T *wt,
                                                         T one_minus_g) {
    const SweepManyDesc d = desc[blockIdx.x];
    T *buf = reinterpret_cast<T *>(lds_dyn);
    T *img = images + (int64_t)blockIdx.x * n_pix;
    const int t = threadIdx.x;
    for (int i = t; i < n_pix; i += 256) buf[i] = img[i];
    __syncthreads();
    sweep_levels<T, T, 256>(buf, level_start + d.level_off, d.n_levels, d.n_entries,
                            pix + d.entry_off, cnt + d.entry_off, nbr + d.term_off,
                            wt + d.term_off, one_minus_g, t);
    for (int i = t; i < n_pix; i += 256) img[i] = buf[i];
}

// The same sweep for images beyond the LDS (initialisation runs it on the whole detection
// image, e.g. 282 x 282 doubles): one workgroup of 1024 threads working in global memory,
// level by level; a workgroup-scope fence + barrier makes a level's writes visible to the
// next one (one CU, one L1).
template <typename T>
__global__ __launch_bounds__(1024) void sweep_global_kernel(T *image, const int32_t *level_start,
                                                            int n_levels, int E, int max_terms,
                                                            const int32_t *pix, const int32_t *cnt,
                                                            const int32_t *nbr, const T *wt,
                                                            T one_minus_g) {
    // Round 6: the plan entries of level l + 1 are requested before level l is applied (one
    // entry per thread and level in registers; a level of a 282 x 282 image has at most ~800
    // pixels), so a level costs the image's own round trip through L2 and the fence, not the
    // plan's on top: 2.9 -> 1.x ms for the 282 x 282 doubles of the multi-resolution set-up.
    constexpr int kTerms = 8;  // offsets of operators_pybind11.cc:14-36
    volatile T *img = image;
    struct Entry {
        int p, n;
        int nb[kTerms];
        T w[kTerms];
    };
    auto load = [&](int q, Entry &e) {
        e.p = pix[q];
        e.n = cnt[q];
#pragma unroll
        for (int j = 0; j < kTerms; ++j)
            if (j < max_terms) {
                e.nb[j] = nbr[(int64_t)j * E + q];
                e.w[j] = wt[(int64_t)j * E + q];
            }
    };
    auto apply = [&](const Entry &e) {
        T ref = 0;
#pragma unroll
        for (int j = 0; j < kTerms; ++j)
            if (j < e.n) ref = add_rn(ref, mul_rn((T)img[e.nb[j]], e.w[j]));
        const T lim = mul_rn(ref, one_minus_g);
        if (lim < img[e.p]) img[e.p] = lim;
    };
    if (n_levels <= 0) return;
    const int t = (int)threadIdx.x;
    if (max_terms > kTerms) {  // tables with more than eight offsets: entry by entry
        for (int l = 0; l < n_levels; ++l) {
            for (int q = level_start[l] + t; q < level_start[l + 1]; q += 1024) {
                T ref = 0;
                for (int j = 0; j < cnt[q]; ++j)
                    ref = add_rn(ref, mul_rn((T)img[nbr[(int64_t)j * E + q]], wt[(int64_t)j * E + q]));
                const T lim = mul_rn(ref, one_minus_g);
                if (lim < img[pix[q]]) img[pix[q]] = lim;
            }
            __threadfence_block();
            __syncthreads();
        }
        return;
    }
    int s = level_start[0], e = level_start[1];
    Entry cur{}, nxt{};
    bool have = s + t < e;
    if (have) load(s + t, cur);
    for (int l = 0; l < n_levels; ++l) {
        const int e_next = l + 1 < n_levels ? level_start[l + 2] : e;
        const bool have_next = l + 1 < n_levels && e + t < e_next;
        if (have_next) load(e + t, nxt);
        if (have) apply(cur);
        for (int q = s + t + 1024; q < e; q += 1024) {
            Entry more{};
            load(q, more);
            apply(more);
        }
        __threadfence_block();
        __syncthreads();
        cur = nxt;
        have = have_next;
        s = e;
        e = e_next;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void apply_filter_kernel(const T *image, int H, int W,
                                                           const T *values, int n_taps,
                                                           const int32_t *ys, const int32_t *ye,
                                                           const int32_t *xs, const int32_t *xe,
                                                           T *result) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    T acc = 0;
    for (int n = 0; n < n_taps; ++n) {
        const int r = y - ys[n], c = x - xs[n];
        if (r >= 0 && c >= 0 && r < H - ys[n] - ye[n] && c < W - xs[n] - xe[n])
            acc = add_rn(acc, mul_rn(values[n], image[(int64_t)(r + ye[n]) * W + c + xe[n]]));
    }
    result[i] = acc;
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline int pix_blocks(const BatchView &v) {
    return (v.H * v.W + kPixBlock - 1) / kPixBlock;
}

void launch_render(const BatchView &v, float *P, hipStream_t s) {
    const int rows_per_block = kRenderRows * kRenderWaves;
    const dim3 grid((v.H + rows_per_block - 1) / rows_per_block, v.nb), block(64 * kRenderWaves);
    // bands per pass: all of them when they fit the registers (C <= 6), else chunks of 4
    switch (v.C <= 6 ? v.C : 4) {
        case 1: hipLaunchKernelGGL(render_kernel<1>, grid, block, 0, s, v, P); break;
        case 2: hipLaunchKernelGGL(render_kernel<2>, grid, block, 0, s, v, P); break;
        case 3: hipLaunchKernelGGL(render_kernel<3>, grid, block, 0, s, v, P); break;
        case 4: hipLaunchKernelGGL(render_kernel<4>, grid, block, 0, s, v, P); break;
        case 5: hipLaunchKernelGGL(render_kernel<5>, grid, block, 0, s, v, P); break;
        default: hipLaunchKernelGGL(render_kernel<6>, grid, block, 0, s, v, P); break;
    }
}

void launch_residual(const BatchView &v, const float *Q, float *R, hipStream_t s) {
    hipLaunchKernelGGL(residual_kernel, dim3(pix_blocks(v), v.nb), dim3(kPixBlock), 0, s, v, Q, R);
}

void launch_finalize(const BatchView &v, int32_t it, float e_rel, int32_t min_iter,
                     int32_t check, hipStream_t s) {
    hipLaunchKernelGGL(finalize_kernel, dim3(v.nb), dim3(64), 0, s, v, it, e_rel, min_iter,
                       check);
}

void launch_advance(const BatchView &v, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3((v.nb + 255) / 256), dim3(256), 0, s,
                       v.state + v.blend0, v.nb);
}

void launch_count_active(const int32_t *state, int32_t nb, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(count_active_kernel, dim3(1), dim3(64), 0, s, state, nb, out);
}

void launch_cmul(float2 *S, const float2 *K, int32_t nb, int32_t C, int64_t plane,
                 int32_t k_bands, int32_t k_per_blend, int32_t conj, const int32_t *state,
                 hipStream_t s) {
    const unsigned tiles = (unsigned)((plane + 255) / 256);
    hipLaunchKernelGGL(cmul_kernel, dim3(tiles * (unsigned)(nb * C)), dim3(256), 0, s, S, K, C,
                       plane, k_bands, k_per_blend, conj, state);
}

static size_t update_lds_bytes(const BatchView &v) {
    const size_t npad = (v.max_box_pixels + 3) & ~3;
    size_t bytes = ((v.scratch ? 1 : 4) * npad + 4) * sizeof(float);
    if (v.mono_mask)  // image before the sweep + flags of the accepted pixels
        return bytes + (size_t)((v.max_levels + 2 + 3) & ~3) * sizeof(int32_t) + npad * 5;
    return bytes + (size_t)(v.max_levels + 2) * sizeof(int32_t);
}

// loss bookkeeping waiting for a launch to ride along with (launch_update_finalize)
struct PendingFinalize {
    bool on = false;
    int32_t min_iter = 0, check = 0;
};
static thread_local PendingFinalize pending_finalize;

static int cu_count() {
    static int count[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    if (!count[dev]) {
        hipDeviceProp_t prop;
        count[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return count[dev];
}

template <int NPL, int T>
static int launch_update_reg(const BatchView &v, const float *G, int32_t it, float e_rel,
                             int32_t prox_max_iter, int32_t cls, int32_t item0, int32_t n_items,
                             hipStream_t s) {
    BatchView vi = v;
    vi.work0 = item0;
    const PendingFinalize fin = pending_finalize;
    pending_finalize.on = false;
    const int n_fin = fin.on ? v.nb : 0;
    // small launches are packed (update_kernel_reg); a launch that fills the chip anyway keeps
    // one wavefront per workgroup, whose slots free up one by one.  SMI_UPDATE_PACK (development
    // aid) overrides the number of wavefronts per workgroup.
    static const int forced = [] {
        const char *e = getenv("SMI_UPDATE_PACK");
        return e ? atoi(e) : 0;
    }();
    // SMI_STAGE_PLAN=0 (development aid): never stage the ring plan in LDS
    static const bool may_stage = [] {
        const char *e = getenv("SMI_STAGE_PLAN");
        return !e || atoi(e) != 0;
    }();
    constexpr int kPack = update_pack_max(NPL, T);
    int pack = n_items <= kUpdatePackLimit ? kPack : 1;
    if (forced > 0) pack = std::min(forced, kPack);
    // the ring plan of the class staged in LDS: full workgroups, persistent when the list is
    // longer than the chip is wide
    int stage = -1, persistent = 0;
    int n_blocks = (n_items + pack - 1) / pack;
    size_t lds = (size_t)pack * (T * NPL + 4) * sizeof(float);
    if (T == 64 && may_stage && v.stage_plan[cls] >= 0) {
        // as many wavefronts as fit the LDS beside the stream (61^2 boxes: seven images of 15 KB
        // and 49 KB of plan)
        const size_t per_wave = (size_t)(T * NPL + 4) * sizeof(float);
        const size_t fixed = (size_t)kPack * 64 * sizeof(float) + 64 + v.stage_bytes[cls];
        const int fit = fixed < 160 * 1024 ? (int)((160 * 1024 - fixed) / per_wave) : 0;
        if (fit >= 4) {
            stage = v.stage_plan[cls];
            pack = std::min(forced > 0 ? pack : kPack, fit);
            lds = (size_t)pack * per_wave + v.stage_bytes[cls];
            n_blocks = (n_items + pack - 1) / pack;
            const int resident = cu_count() * std::max(1, (int)((160 * 1024) / (lds + kPack * 256 + 64)));
            static const bool may_persist = [] {  // development aid
                const char *e = getenv("SMI_PERSIST");
                return !e || atoi(e) != 0;
            }();
            static const int forced_blocks = [] {  // development aid
                const char *e = getenv("SMI_PERSIST_BLOCKS");
                return e ? atoi(e) : 0;
            }();
            if (forced_blocks > 0 && n_blocks > forced_blocks) {
                n_blocks = forced_blocks;
                persistent = 1;
            } else if (n_blocks > resident && may_persist) {
                n_blocks = resident;
                persistent = 1;
            }
        }
    }
    const dim3 grid(n_blocks + n_fin), block(T * pack);
#define SMI_LAUNCH(MODE)                                                                        \
    {                                                                                           \
        static size_t configured[kMaxDevices] = {};                                             \
        auto kern = update_kernel_reg<NPL, MODE, T>;                                            \
        if (lds > 48 * 1024)                                                                    \
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) \
                return rc;                                                                      \
        hipLaunchKernelGGL(kern, grid, block, lds, s, vi, G, it, e_rel, prox_max_iter, n_items, \
                           n_fin, fin.min_iter, fin.check, stage, persistent);                         \
    }
    if (v.scheme == SMI_SCHEME_FISTA) SMI_LAUNCH(2)
    else if (v.lite) SMI_LAUNCH(1)
    else SMI_LAUNCH(0)
#undef SMI_LAUNCH
    return SMI_OK;
}

// finalize + update of one iteration: the loss bookkeeping shares the first launch of a
// register-resident size class; where no such launch exists (mixed / general kernels, a
// range without components) it is the kernel of its own it used to be
int launch_update_finalize(const BatchView &v, const float *G, int32_t it, float e_rel,
                           int32_t min_iter, int32_t check, int32_t prox_max_iter,
                           hipStream_t s) {
    static const int mode = [] {  // development aid: 0 never, 1 always, default: small launches
        const char *e = getenv("SMI_FOLD_FINALIZE");
        return e ? atoi(e) : 2;
    }();
    if (mode == 0 || (mode == 2 && v.n_comp > kUpdatePackLimit)) {
        launch_finalize(v, it, e_rel, min_iter, check, s);
        return launch_update(v, G, it, e_rel, prox_max_iter, nullptr, nullptr, 0, s);
    }
    pending_finalize.on = true;
    pending_finalize.min_iter = min_iter;
    pending_finalize.check = check;
    const int rc = launch_update(v, G, it, e_rel, prox_max_iter, nullptr, nullptr, 0, s);
    const bool left = pending_finalize.on;
    pending_finalize.on = false;
    if (rc) return rc;
    if (left) launch_finalize(v, it, e_rel, min_iter, check, s);
    return SMI_OK;
}

int launch_update(const BatchView &v_in, const float *G, int32_t it, float e_rel,
                  int32_t prox_max_iter, float *g_sed_out, float *g_morph_out,
                  int32_t grad_only, hipStream_t s) {
    if (v_in.n_comp == 0) return SMI_OK;
    BatchView v = v_in;
    v.fail_code = 3 + std::max(it, 0);  // (finalize_blend)
    // chains that repeat take the general kernel (the register-resident ones apply it once)
    if (!grad_only && v.fast_plans && v.max_box_pixels <= kMaxRegisterBox && !v.c_chain_repeat &&
        !v.mono_mask) {
        // latency regime with several size classes: one launch for all of them
        int classes = 0, large = 0;
        for (int cls = 0; cls < kNumUpdateClasses; ++cls) {
            const int32_t *start = v.work_start + (size_t)cls * (v.nb_total + 1);
            const int present = start[v.blend0 + v.nb] > start[v.blend0];
            classes += present;
            large += present && cls >= kNumSmallClasses;
        }
        static const int mixed_limit = [] {  // development aid
            const char *e = getenv("SMI_MIXED_LIMIT");
            return e ? atoi(e) : kMixedUpdateLimit;
        }();
        if (classes > 1 && !large && v.n_comp <= mixed_limit) {
            const size_t lds = (size_t)(64 * kUpdateNpl[kNumUpdateClasses - 1] + 4) * sizeof(float);
            const PendingFinalize fin = pending_finalize;
            pending_finalize.on = false;
            const int n_fin = fin.on ? v.nb : 0;
            const dim3 grid(v.n_comp + n_fin);
            static const int one_wave_limit = [] {  // development aid
                const char *e = getenv("SMI_MIXED_ONE_WAVE");
                return e ? atoi(e) : 1024;
            }();
            const bool lone = v.n_comp_total <= one_wave_limit;  // (one wavefront per SIMD suffices)
#define SMI_MIXED(MODE)                                                                          \
    if (lone)                                                                                    \
        hipLaunchKernelGGL((update_kernel_mixed<MODE, 1>), grid, dim3(64), lds, s, v, G, it,     \
                           e_rel, prox_max_iter, n_fin, fin.min_iter, fin.check);                \
    else                                                                                         \
        hipLaunchKernelGGL((update_kernel_mixed<MODE, 2>), grid, dim3(64), lds, s, v, G, it,     \
                           e_rel, prox_max_iter, n_fin, fin.min_iter, fin.check)
            if (v.scheme == SMI_SCHEME_FISTA) {
                SMI_MIXED(2);
            } else if (v.lite) {
                SMI_MIXED(1);
            } else {
                SMI_MIXED(0);
            }
#undef SMI_MIXED
            return SMI_OK;
        }
        // one launch per size class that has components in this range of blends, the
        // largest boxes (longest sweeps) first.  Few components of several classes, four-
        // wavefront classes among them (update_kernel_mixed does not cover those): every
        // launch is one latency-bound chain per component, so the classes go to streams of
        // their own, forked from and joined to `s` (multi-resolution tutorial scene: three
        // launches of 0.23 + 0.13 + 0.10 ms in a row).
        // SMI_CLASS_STREAMS (development aid): 0 never, 1 always
        static const int class_streams = [] {
            const char *e = getenv("SMI_CLASS_STREAMS");
            return e ? atoi(e) : -1;
        }();
        const bool side_by_side =
            classes > 1 && (class_streams < 0 ? (v.n_comp <= kMixedUpdateLimit || v.class_streams)
                                              : class_streams > 0);
        // (ONE set of side streams per host thread, shared by the concurrent ranges of blends: the
        // launches of a class from two ranges queue up behind each other on that class's stream.
        // Side streams per range -- eight streams busy at once -- were measured: 1 616 k -> 1 300 k
        // blend-it/s on the 1024-blend quickstart batch.  SMI_SIDE_PER_RANGE=1 brings them back.)
        struct Side {
            hipStream_t stream[kNumUpdateClasses] = {};
            hipEvent_t fork = nullptr, join[kNumUpdateClasses] = {};
            int device = -1;
        };
        constexpr int kSideSlots = 4;
        static thread_local Side sides[kSideSlots];
        static const bool per_range = [] {  // development aid
            const char *e = getenv("SMI_SIDE_PER_RANGE");
            return e && atoi(e) != 0;
        }();
        Side &side = sides[per_range && v.range_slot >= 0 && v.range_slot < kSideSlots ? v.range_slot : 0];
        int n_side = 0;
        if (side_by_side) {
            int dev = 0;
            SMI_HIP(hipGetDevice(&dev));
            if (side.device != dev) {  // (first use on this device by this host thread)
                side = Side();
                side.device = dev;
                SMI_HIP(hipEventCreateWithFlags(&side.fork, hipEventDisableTiming));
            }
            SMI_HIP(hipEventRecord(side.fork, s));
        }
        for (int cls = kNumUpdateClasses - 1; cls >= 0; --cls) {
            const int32_t *start = v.work_start + (size_t)cls * (v.nb_total + 1);
            const int lo = start[v.blend0], hi = start[v.blend0 + v.nb];
            if (hi <= lo) continue;
            hipStream_t sc = s;
            if (side_by_side && n_side > 0) {  // the first class stays on `s`
                if (!side.stream[n_side]) {
                    SMI_HIP(hipStreamCreateWithFlags(&side.stream[n_side], hipStreamNonBlocking));
                    SMI_HIP(hipEventCreateWithFlags(&side.join[n_side], hipEventDisableTiming));
                }
                sc = side.stream[n_side];
                SMI_HIP(hipStreamWaitEvent(sc, side.fork, 0));
            }
#define SMI_CLASS(i) \
    case i: if (int rc = launch_update_reg<kUpdateNpl[i], kUpdateTeam[i]>(v, G, it, e_rel, prox_max_iter, i, lo, hi - lo, sc)) return rc; break;
            switch (cls) {
                SMI_CLASS(0) SMI_CLASS(1) SMI_CLASS(2) SMI_CLASS(3) SMI_CLASS(4)
                SMI_CLASS(5) SMI_CLASS(6) SMI_CLASS(7) SMI_CLASS(8)
            }
#undef SMI_CLASS
            if (sc != s) {
                SMI_HIP(hipEventRecord(side.join[n_side], sc));
                SMI_HIP(hipStreamWaitEvent(s, side.join[n_side], 0));
            }
            ++n_side;
        }
        return SMI_OK;
    }
    const size_t lds = update_lds_bytes(v);
    SMI_REQUIRE(lds <= 160 * 1024, "component box too large for the LDS-resident update");
    // four waves per component for boxes beyond 64 x 59 pixels (multi-resolution tutorial
    // fit, ms per iteration with 1 / 4 / 16 waves: 2.89 / 2.20 / 2.37); every workgroup of
    // the other launch returns at once
    if (v.max_box_pixels > 64 * kUpdateNpl[kNumSmallClasses - 1]) {
        static size_t configured[kMaxDevices] = {};
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(update_kernel<256>), lds, configured))
            return rc;
        hipLaunchKernelGGL(update_kernel<256>, dim3(v.n_comp), dim3(256), lds, s, v, G, it, e_rel,
                           prox_max_iter, g_sed_out, g_morph_out, grad_only);
    }
    static size_t configured[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(update_kernel<64>), lds, configured))
        return rc;
    hipLaunchKernelGGL(update_kernel<64>, dim3(v.n_comp), dim3(64), lds, s, v, G, it, e_rel,
                       prox_max_iter, g_sed_out, g_morph_out, grad_only);
    return SMI_OK;
}

int launch_point_sources(const BatchView &v_in, const float *G, int32_t it, float e_rel,
                         int32_t prox_max_iter, float *g_sed_out, double *g_center_out,
                         int32_t mode, hipStream_t s) {
    if (v_in.n_point == 0 || v_in.n_comp == 0) return SMI_OK;
    BatchView v = v_in;
    v.fail_code = 3 + std::max(it, 0);
    const size_t lds = (size_t)(((v.max_box_pixels + 3) & ~3) + 4) * sizeof(float);
    SMI_REQUIRE(lds <= 64 * 1024, "component box too large for the point-source kernel");
    hipLaunchKernelGGL(point_source_kernel, dim3(v.n_comp), dim3(64), lds, s, v, G, it, e_rel,
                       prox_max_iter, g_sed_out, g_center_out, mode);
    return SMI_OK;
}

int launch_sweep_timing(const SweepPlanDev *d_plans, const SweepPlanDev &host_plan, int plan_id,
                        int mode, int n_rep, float one_minus_g, int waves, int groups,
                        long long *cycles, float *images, hipStream_t s) {
    const int n = host_plan.h * host_plan.w;
    const size_t lds = (size_t)waves * (((n + 3) & ~3) + 4) * sizeof(float) +
                       (mode == 2 ? host_plan.ring_bytes : 0);
    SMI_REQUIRE(lds <= 160 * 1024 && waves >= 1 && waves <= 16, "images do not fit the LDS");
    SMI_REQUIRE(mode == 0 || mode == 2 || (mode == 3 && host_plan.ring_planes == 1),
                "mode: 0 slot plan, 2 ring schedule, 3 ring schedule with the stream in L2 (one plane)");
    SMI_REQUIRE(mode == 0 ? host_plan.slots != nullptr : host_plan.ring != nullptr, "plan has no such schedule");
    static size_t configured[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sweep_timing_kernel), lds, configured))
        return rc;
    hipLaunchKernelGGL(sweep_timing_kernel, dim3(groups), dim3(64 * waves), lds, s, d_plans, plan_id,
                       mode, n_rep, one_minus_g, cycles, images);
    return SMI_OK;
}

void launch_resize_test(const BatchView &v, int32_t *margin, double *pull, hipStream_t s) {
    if (v.n_comp)
        hipLaunchKernelGGL(resize_test_kernel, dim3(v.n_comp), dim3(64), 0, s, v, margin, pull);
}

void launch_gather_states(const BatchView &v, const int32_t *sel, const int64_t *off, int32_t n_sel,
                          float *staging, hipStream_t s) {
    if (n_sel)
        hipLaunchKernelGGL(gather_states_kernel, dim3(n_sel), dim3(256), 0, s, v, sel, off, staging);
}

void launch_scatter_states(const BatchView &v, const int64_t *off, const float *staging,
                           hipStream_t s) {
    if (v.n_comp)
        hipLaunchKernelGGL(scatter_states_kernel, dim3(v.n_comp), dim3(256), 0, s, v, off, staging);
}

void launch_carry_states(const int32_t *keep, const int64_t *old_moff, const int64_t *new_moff,
                         int32_t n, float *const from[4], float *const to[4], hipStream_t s) {
    PixelArrays a, b;
    for (int i = 0; i < 4; ++i) {
        a.p[i] = from[i];
        b.p[i] = to[i];
    }
    if (n)
        hipLaunchKernelGGL(carry_states_kernel, dim3(n), dim3(256), 0, s, keep, old_moff, new_moff, a, b);
}

void launch_interleave_obs(const float *data, const float *weights, float4 *dw, int64_t planes,
                           int32_t H, int32_t W, hipStream_t s) {
    const int blocks = std::max(1, std::min(((H + 1) / 2 * W + 255) / 256, 64));
    for (int64_t p0 = 0; p0 < planes; p0 += 65535) {  // (grid.y limit)
        const int np = (int)std::min<int64_t>(planes - p0, 65535);
        hipLaunchKernelGGL(interleave_obs_kernel, dim3(blocks, np), dim3(256), 0, s,
                           data + p0 * H * W, weights + p0 * H * W,
                           dw + p0 * ((H + 1) / 2) * W, H, W);
    }
}

void launch_log_norm(const float *weights, double *log_norm, int32_t nb, int64_t n,
                     hipStream_t s) {
    hipLaunchKernelGGL(log_norm_kernel, dim3(nb), dim3(256), 0, s, weights, log_norm, n);
}

void launch_wrap_kernel(const float *kern, float *out, int32_t n_img, int32_t ph, int32_t pw,
                        int32_t Fy, int32_t Fx, float scale, hipStream_t s) {
    hipLaunchKernelGGL(wrap_kernel_kernel, dim3((ph * pw + 255) / 256, n_img), dim3(256), 0, s,
                       kern, out, ph, pw, Fy, Fx, scale);
}

void launch_crop(const float *P, float *out, int32_t n_img, int32_t H, int32_t W, int32_t Fy,
                 int32_t Fx, hipStream_t s) {
    hipLaunchKernelGGL(crop_kernel, dim3((H * W + 255) / 256, n_img), dim3(256), 0, s, P, out, H,
                       W, Fy, Fx);
}

// ---- seam 1: host-buffer entry points --------------------------------------
// Device buffers of the host-buffer entry points (seam 1).  A call of the reference's sweep
// needs seven of them for a few hundred microseconds; hipMalloc / hipFree cost more than the
// kernel (a 128 x 128 image: 2.1 ms per call, most of it the allocator), so freed blocks are
// kept -- per host thread and device, by power-of-two size, up to 512 MB -- and handed out again.
// The entry points are synchronous: a block is idle when its call has returned.
struct PoolBlock {
    void *p;
    size_t cap;
    int dev;
};
static thread_local std::vector<PoolBlock> seam_pool;
static thread_local size_t seam_pool_bytes = 0;
constexpr size_t kSeamPoolLimit = (size_t)512 << 20;

static hipError_t pool_alloc(void **p, size_t bytes, size_t *cap) {
    size_t want = 256;
    while (want < bytes) want <<= 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (size_t i = 0; i < seam_pool.size(); ++i)
        if (seam_pool[i].cap == want && seam_pool[i].dev == dev) {
            *p = seam_pool[i].p;
            *cap = want;
            seam_pool_bytes -= want;
            seam_pool.erase(seam_pool.begin() + (long)i);
            return hipSuccess;
        }
    *cap = want;
    return hipMalloc(p, want);
}

static void pool_free(void *p, size_t cap) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (cap > kSeamPoolLimit / 4 || seam_pool_bytes + cap > kSeamPoolLimit) {
        (void)hipFree(p);
        return;
    }
    seam_pool.push_back({p, cap, dev});
    seam_pool_bytes += cap;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    ~DevBuf() {
        if (p) pool_free(p, cap);
    }
    hipError_t alloc(size_t n) {
        return pool_alloc(reinterpret_cast<void **>(&p), n * sizeof(T), &cap);
    }
    hipError_t upload(const T *h, size_t n) {
        hipError_t e = alloc(n > 0 ? n : 1);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice);
    }
};

template <typename T>
int sweep_host_buffers(T *flat_img, int32_t n_pix, const SweepPlanHost &plan, T min_gradient) {
    const size_t lds = (size_t)n_pix * sizeof(T);
    if (plan.n_entries == 0) return SMI_OK;
    DevBuf<T> d_img, d_wt;
    DevBuf<int32_t> d_ls, d_pix, d_cnt, d_nbr;
    std::vector<T> wt(plan.wt.size());
    for (size_t i = 0; i < wt.size(); ++i) wt[i] = (T)plan.wt[i];
    SMI_HIP(d_img.upload(flat_img, n_pix));
    SMI_HIP(d_wt.upload(wt.data(), wt.size()));
    SMI_HIP(d_ls.upload(plan.level_start.data(), plan.level_start.size()));
    SMI_HIP(d_pix.upload(plan.pix.data(), plan.pix.size()));
    SMI_HIP(d_cnt.upload(plan.cnt.data(), plan.cnt.size()));
    SMI_HIP(d_nbr.upload(plan.nbr.data(), plan.nbr.size()));
    const T omg = (T)1 - min_gradient;
    const int n_levels = (int)plan.level_start.size() - 1;
    if (lds <= 160 * 1024) {
        auto kern = sweep_kernel<T>;
        SMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), lds, 0, d_img.p, n_pix, d_ls.p, n_levels,
                           plan.n_entries, d_pix.p, d_cnt.p, d_nbr.p, d_wt.p, omg);
    } else {
        hipLaunchKernelGGL(sweep_global_kernel<T>, dim3(1), dim3(1024), 0, 0, d_img.p, d_ls.p,
                           n_levels, plan.n_entries, plan.max_terms, d_pix.p, d_cnt.p, d_nbr.p,
                           d_wt.p, omg);
    }
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipMemcpy(flat_img, d_img.p, (size_t)n_pix * sizeof(T), hipMemcpyDeviceToHost));
    return SMI_OK;
}
template int sweep_host_buffers<float>(float *, int32_t, const SweepPlanHost &, float);
template int sweep_host_buffers<double>(double *, int32_t, const SweepPlanHost &, double);

template <typename T>
int sweep_many_host_buffers(T *images, int32_t n_img, int32_t n_pix,
                            const std::vector<SweepPlanHost> &plans, T min_gradient) {
    const size_t lds = (size_t)n_pix * sizeof(T);
    if (lds > 160 * 1024) {  // beyond the LDS: image by image in global memory
        for (int i = 0; i < n_img; ++i)
            if (int rc = sweep_host_buffers<T>(images + (size_t)i * n_pix, n_pix, plans[i], min_gradient))
                return rc;
        return SMI_OK;
    }
    std::vector<SweepManyDesc> desc(n_img);
    std::vector<int32_t> ls, pix, cnt, nbr;
    std::vector<T> wt;
    for (int i = 0; i < n_img; ++i) {
        const SweepPlanHost &p = plans[i];
        desc[i].level_off = (int64_t)ls.size();
        desc[i].entry_off = (int64_t)pix.size();
        desc[i].term_off = (int64_t)nbr.size();
        desc[i].n_levels = p.n_entries ? (int32_t)p.level_start.size() - 1 : 0;
        desc[i].n_entries = p.n_entries;
        if (!p.n_entries) continue;
        ls.insert(ls.end(), p.level_start.begin(), p.level_start.end());
        pix.insert(pix.end(), p.pix.begin(), p.pix.end());
        cnt.insert(cnt.end(), p.cnt.begin(), p.cnt.end());
        nbr.insert(nbr.end(), p.nbr.begin(), p.nbr.end());
        for (double w : p.wt) wt.push_back((T)w);
    }
    if (pix.empty()) return SMI_OK;
    DevBuf<T> d_img, d_wt;
    DevBuf<int32_t> d_ls, d_pix, d_cnt, d_nbr;
    DevBuf<SweepManyDesc> d_desc;
    SMI_HIP(d_img.upload(images, (size_t)n_img * n_pix));
    SMI_HIP(d_wt.upload(wt.data(), wt.size()));
    SMI_HIP(d_ls.upload(ls.data(), ls.size()));
    SMI_HIP(d_pix.upload(pix.data(), pix.size()));
    SMI_HIP(d_cnt.upload(cnt.data(), cnt.size()));
    SMI_HIP(d_nbr.upload(nbr.data(), nbr.size()));
    SMI_HIP(d_desc.upload(desc.data(), desc.size()));
    auto kern = sweep_many_kernel<T>;
    SMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_img), dim3(256), lds, 0, d_img.p, n_pix, d_desc.p, d_ls.p,
                       d_pix.p, d_cnt.p, d_nbr.p, d_wt.p, (T)1 - min_gradient);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipMemcpy(images, d_img.p, (size_t)n_img * n_pix * sizeof(T), hipMemcpyDeviceToHost));
    return SMI_OK;
}
template int sweep_many_host_buffers<float>(float *, int32_t, int32_t,
                                            const std::vector<SweepPlanHost> &, float);
template int sweep_many_host_buffers<double>(double *, int32_t, int32_t,
                                             const std::vector<SweepPlanHost> &, double);

template <typename T>
int apply_filter_host_buffers(const T *image, int32_t H, int32_t W, const T *values,
                              int32_t n_taps, const int32_t *ys, const int32_t *ye,
                              const int32_t *xs, const int32_t *xe, T *result) {
    DevBuf<T> d_img, d_val, d_out;
    DevBuf<int32_t> d_ys, d_ye, d_xs, d_xe;
    const size_t n = (size_t)H * W;
    SMI_HIP(d_img.upload(image, n));
    SMI_HIP(d_val.upload(values, n_taps));
    SMI_HIP(d_ys.upload(ys, n_taps));
    SMI_HIP(d_ye.upload(ye, n_taps));
    SMI_HIP(d_xs.upload(xs, n_taps));
    SMI_HIP(d_xe.upload(xe, n_taps));
    SMI_HIP(d_out.alloc(n));
    hipLaunchKernelGGL(apply_filter_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0,
                       d_img.p, H, W, d_val.p, n_taps, d_ys.p, d_ye.p, d_xs.p, d_xe.p, d_out.p);
    SMI_HIP(hipGetLastError());
    SMI_HIP(hipMemcpy(result, d_out.p, n * sizeof(T), hipMemcpyDeviceToHost));
    return SMI_OK;
}
template int apply_filter_host_buffers<float>(const float *, int32_t, int32_t, const float *,
                                              int32_t, const int32_t *, const int32_t *,
                                              const int32_t *, const int32_t *, float *);
template int apply_filter_host_buffers<double>(const double *, int32_t, int32_t, const double *,
                                               int32_t, const int32_t *, const int32_t *,
                                               const int32_t *, const int32_t *, double *);

}  // namespace smi
