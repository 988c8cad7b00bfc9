// LDS-resident PSF convolution + likelihood for one (blend, band) per workgroup.
//
// Replaces, for frames whose padded band fits the 160 KiB LDS of a CU, the chain
//   rocFFT R2C -> x K^ -> rocFFT C2R -> residual/loss -> rocFFT R2C -> x conj(K^) ->
//   rocFFT C2R
// (renderer.py:247-259 / fft.py:368-396 forward, its transpose backward,
// observation.py:147-170 in between) by ONE kernel that keeps the half-spectrum of
// the band in LDS from the first row transform to the last: per blend-iteration the
// only HBM traffic left is the model cube, data and weights (read once) and the
// gradient image (written once).  The model cube itself comes from render_kernel
// (kernels.hip): rendering inside this kernel made every 64-row chunk wait, at a
// workgroup barrier, for the wavefront whose rows intersect the most component boxes
// (22 % of the kernel's time); plain rows are fetched one chunk ahead instead.
//
// Layout: T[kx][y], kx in [0, FX/2]; element y of a column sits at y + y / 16 and the
// column stride SY is 16 mod 32 complex, so that the 16 lanes that own a column and the
// four columns of a wavefront hit different banks in every pass.
// 1-D transforms of length F = F1 * 16 are two in-place passes (radix F1 over stride
// 16, radix 16 over contiguous blocks); the forward transform leaves the spectrum in
// the digit-swapped order pos(k1 + F1 k2) = 16 k1 + k2, the inverse starts from it,
// so no transposition pass is ever needed; the kernel spectrum K^ is stored in the
// same order.  Rows are transformed two at a time (row 2j + i row 2j+1) and
// separated by Hermitian symmetry; zero padding is never stored for the columns.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "fft_regs.h"

namespace smi {

using fftk::cf;
using fftk::cmul;
using fftk::cmulc;
using fftk::ld;
using fftk::st;

namespace {

// Workgroup size and row pairs per chunk.  This file is compiled twice: as it is
// (1024 threads) and through fused_conv_short.hip with 512 threads for transforms with
// short rows (SMI_CONV_SHORT_ROWS: only launch_fused_conv_short is emitted there).
#ifndef SMI_CONV_THREADS
#define SMI_CONV_THREADS 1024
#define SMI_CONV_PAIRS 32
#endif
constexpr int kThreads = SMI_CONV_THREADS;
constexpr int kF2 = 16;      // second radix of every 1-D transform
constexpr int kPairs = SMI_CONV_PAIRS;   // row pairs per chunk (64 rows)

template <int FY1, int FX1>
struct Cfg {
    static constexpr int FY = FY1 * kF2, FX = FX1 * kF2;
    static constexpr int NKX = FX / 2 + 1;
    // column stride of T (complex).  A column stores element y at y + y / 16 (one pad
    // per radix-16 block, so that the 10 lanes that each own a block hit different
    // banks); the stride is 16 mod 32 so that the four columns a wavefront works on
    // alternate between the two halves of the 64 banks
    static constexpr int SY = ((FY + FY / kF2 + 15) / 32) * 32 + 16;
    static constexpr int SX = FX + 1;  // row-pair stride of the scratch (complex), odd
    // + twiddle tables tw[k1 * 16 + n2] = exp(-2 pi i n2 k1 / F) for both axes
    static_assert(SY >= FY + FY / kF2 && SY % 32 == 16, "column stride");
    static constexpr size_t lds_bytes =
        sizeof(float2) * ((size_t)NKX * SY + (size_t)kPairs * SX + FY + FX);
};

// index of element y of a column of T
__device__ __forceinline__ int sk(int y) { return y + (y >> 4); }

// the same pass on a column of T (element 16 n1 + n2 lives at 17 n1 + n2)
template <int F1, bool INV>
__device__ __forceinline__ void pass_stride_col(float2 *a, int n2, const float2 *tw, int valid) {
    cf v[F1];
#pragma unroll
    for (int n1 = 0; n1 < F1; ++n1) {
        const int idx = kF2 * n1 + n2;
        v[n1] = (INV || idx < valid) ? ld(a[(kF2 + 1) * n1 + n2]) : cf{0.f, 0.f};
        if (INV && n1 > 0) v[n1] = cmulc(v[n1], ld(tw[kF2 * n1 + n2]));
    }
    fftk::Dft<F1, INV>::run(v);
#pragma unroll
    for (int k1 = 0; k1 < F1; ++k1) {
        if (!INV && k1 > 0) v[k1] = cmul(v[k1], ld(tw[kF2 * k1 + n2]));
        a[(kF2 + 1) * k1 + n2] = st(v[k1]);
    }
}

// Single wavefront: LDS operations of one wave execute in order, so between passes
// that exchange data only among the lanes of a wave the LDS counter has to drain, no
// workgroup barrier is needed.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- radix-16 pass over the contiguous block a[16 k1 .. 16 k1 + 15] ---------------
template <bool INV>
__device__ __forceinline__ void pass_block(float2 *a, int k1) {
    cf v[kF2];
#pragma unroll
    for (int j = 0; j < kF2; ++j) v[j] = ld(a[kF2 * k1 + j]);
    fftk::Dft<kF2, INV>::run(v);
#pragma unroll
    for (int j = 0; j < kF2; ++j) a[kF2 * k1 + j] = st(v[j]);
}

// position of natural frequency k in the digit-swapped order of a length F1*16 transform
template <int F1>
__device__ __forceinline__ int pos(int k) {
    return kF2 * (k % F1) + k / F1;
}

__device__ __forceinline__ double block_sum(double v, double *part) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) t += part[i];
    return t;
}

// workgroup barrier that orders LDS traffic only (no vector-memory drain)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One band plane [H][W] of the model cube / data / weights through a buffer descriptor:
// rows beyond H are beyond the descriptor's range and read 0, columns beyond W get an
// out-of-range offset -- guarded loads without a branch or a 64-bit address per element.
using plane_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ plane_t band_plane(const float *p, int n_elements) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, n_elements * 4, 0x00020000);
}
__device__ __forceinline__ void plane_store(plane_t r, int y, int x, int W, float value) {
    const uint32_t off = x < W ? (uint32_t)(y * W + x) * 4u : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, value), r, off, 0, 0);
}
__device__ __forceinline__ float plane_load(plane_t r, int y, int x, int W) {
    const uint32_t off = x < W ? (uint32_t)(y * W + x) * 4u : 0x80000000u;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

template <int FY1, int FX1>
struct Conv {
    using C = Cfg<FY1, FX1>;
    float2 *T, *Z, *twy, *twx;
    int tid, bt;
    // The row transforms alternate between stride passes (kPairs x 16 work items) and radix-16
    // passes (kPairs x FX1).  A workgroup of 1024 threads gives each kind to one half of its
    // wavefronts, and the two halves run separate code between the same barriers: the
    // registers of one role (the prefetched data / weights of the stride role, the sixteen
    // points of the radix-16 role) are not live in the other's code.
    static constexpr bool kSplit = kThreads >= 2 * kPairs * kF2;
    static constexpr int kBlockThreads = kSplit ? kThreads - kPairs * kF2 : kThreads;
    static constexpr int kChunkStep = 2 * kPairs + 2 * kPairs / kF2;  // sk(y + 64) - sk(y)
    static_assert((2 * kPairs) % kF2 == 0, "chunk rows");

    // column transforms fused with the spectral product:
    //   T <- IFFT_y( FFT_y(T) * K )   for every column kx, K in digit-swapped order.
    // A column belongs to 16 lanes of one wavefront from start to end: lane g does the
    // radix-FY1 butterfly n2 = g of the first pass, the radix-16 block k1 = g of the
    // second (lanes g < FY1; forward, x K, inverse in registers), and n2 = g of the last.
    // The three passes of a column only exchange data inside that group, so no
    // workgroup barrier separates them and the wavefronts drift through the stage
    // independently (the barriers cost 43 % of this stage before).
    __device__ __forceinline__ void columns(const float2 *Kt, int H, bool conj) {
        const int g = tid & (kF2 - 1);
        for (int kx = tid >> 4; kx < C::NKX; kx += kThreads / kF2) {
            float2 *a = T + kx * C::SY;
            pass_stride_col<FY1, false>(a, g, twy, H);
            wave_lds_fence();
            if (g < FY1) {
                float2 *blk = a + (kF2 + 1) * g;
                const float2 *kp = Kt + (int64_t)kx * C::FY + kF2 * g;
                cf v[kF2], kv[kF2];
#pragma unroll
                for (int j = 0; j < kF2; ++j) kv[j] = ld(kp[j]);
#pragma unroll
                for (int j = 0; j < kF2; ++j) v[j] = ld(blk[j]);
                fftk::Dft<kF2, false>::run(v);
#pragma unroll
                for (int j = 0; j < kF2; ++j) v[j] = conj ? cmulc(v[j], kv[j]) : cmul(v[j], kv[j]);
                fftk::Dft<kF2, true>::run(v);
#pragma unroll
                for (int j = 0; j < kF2; ++j) blk[j] = st(v[j]);
            }
            wave_lds_fence();
            pass_stride_col<FY1, true>(a, g, twy, C::FY);
        }
        lds_barrier();
    }

    // The radix-16 pass of the row transforms is fused with the Hermitian separation /
    // recombination of the row pairs.  Block k1 of a row holds the frequencies k1 + FX1 k2
    // (k2 = 0 .. 15); the mirror frequency FX - k sits in block FX1 - k1 at 15 - k2 (block 0
    // mirrors onto itself at (16 - k2) % 16, block FX1 / 2 of an even FX1 at 15 - k2).  Work
    // item (row pair j = lane % 32, slot): the slots are ordered 1, FX1 - 1, 2, FX1 - 2, ...,
    // 0, FX1 / 2, so that the two slots of a wavefront are a block and its mirror block:
    // after its transform a lane gets the mirror values from lane ^ 32 and separates the
    // eight frequencies <= FX / 2 of its block.
    static_assert(kPairs == 32, "a wavefront = two slots of 32 row pairs");
    static constexpr int kDouble = (FX1 - 1) / 2;  // blocks 1 .. kDouble have a distinct mirror block
    static __device__ __forceinline__ int slot_block(int slot) {
        if (slot < 2 * kDouble) return (slot & 1) ? FX1 - (slot / 2 + 1) : slot / 2 + 1;
        return slot == 2 * kDouble ? 0 : FX1 / 2;
    }
    static __device__ __forceinline__ cf from_partner(cf v) {
        return cf{__shfl_xor(v.x, 32, 64), __shfl_xor(v.y, 32, 64)};
    }
    // Xa = za + conj(zb), Xb = -i (za - conj(zb))   (the 1/2 lives in K^)
    static __device__ __forceinline__ void put(float2 *t, bool ok, cf za, cf zb) {
        if (ok) {
            t[0] = make_float2(za.x + zb.x, za.y - zb.y);
            t[1] = make_float2(za.y + zb.y, zb.x - za.x);
        }
    }
    // (no row guard: a row pair beyond FY reads other elements of the LDS array and leaves
    // values in its own rows of Z that nothing consumes -- the pairs never mix, the residual
    // and the stores look at rows < H only, put() is guarded; a branch or a select per load
    // would expose every LDS latency)
    static __device__ __forceinline__ void get(const float2 *t, cf &plus, cf &minus) {
        const float2 xa = t[0], xb = t[1];
        plus = cf{xa.x - xb.y, xa.y + xb.x};   // Xa + i Xb
        minus = cf{xa.x + xb.y, xb.x - xa.y};  // conj(Xa) + i conj(Xb)
    }

    // forward: radix-16 pass of the chunk's rows in Z and separation into
    // T[kx][y0 + 2j], T[kx][y0 + 2j + 1]
    __device__ __forceinline__ void blocks_forward(int ch) {
        for (int b = bt; b < kPairs * FX1; b += kBlockThreads) {
            const int j = b & 31, slot = b >> 5, k1 = slot_block(slot);
            const float2 *z = Z + j * C::SX + kF2 * k1;
            const bool ok = ch * 2 * kPairs + 2 * j + 1 < C::FY;
            float2 *t = T + sk(2 * j) + ch * kChunkStep + k1 * C::SY;
            cf va[kF2];
#pragma unroll
            for (int i = 0; i < kF2; ++i) va[i] = ld(z[i]);
            fftk::Dft<kF2, false>::run(va);
            if (slot < 2 * kDouble) {  // uniform over the wavefront
#pragma unroll
                for (int k2 = 0; k2 < 8; k2 += 2) {
                    // two frequencies at a time (registers: the scheduler would otherwise
                    // start all sixteen exchanges at once)
                    __builtin_amdgcn_sched_barrier(0);
                    const cf m0 = from_partner(va[15 - k2]), m1 = from_partner(va[14 - k2]);
                    put(t + FX1 * k2 * C::SY, ok, va[k2], m0);
                    put(t + FX1 * (k2 + 1) * C::SY, ok, va[k2 + 1], m1);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (k1 == 0) {
                // (two branches: a select between va[a] and va[b] would be compiled into a
                // select of the index, i.e. a dynamically indexed register array)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2)
                    put(t + FX1 * k2 * C::SY, ok, va[k2], va[(16 - k2) & 15]);
                put(t + FX1 * 8 * C::SY, ok, va[8], va[8]);
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) put(t + FX1 * k2 * C::SY, ok, va[k2], va[15 - k2]);
            }
        }
    }

    // inverse: Z[pair] <- radix-16 pass of (Xa + i Xb) rebuilt from T
    __device__ __forceinline__ void blocks_inverse(int ch) {
        for (int b = bt; b < kPairs * FX1; b += kBlockThreads) {
            const int j = b & 31, slot = b >> 5, k1 = slot_block(slot);
            float2 *z = Z + j * C::SX + kF2 * k1;
            const float2 *t = T + sk(2 * j) + ch * kChunkStep;
            cf va[kF2], spare;
            if (slot < 2 * kDouble) {
                const int kb = FX1 - k1;  // the mirror block's columns give the upper half
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    get(t + (k1 + FX1 * k2) * C::SY, va[k2], spare);
                    get(t + (kb + FX1 * k2) * C::SY, spare, va[15 - k2]);
                }
            } else if (k1 == 0) {
                get(t, va[0], spare);
                get(t + FX1 * 8 * C::SY, va[8], spare);
#pragma unroll
                for (int k2 = 1; k2 < 8; ++k2) get(t + FX1 * k2 * C::SY, va[k2], va[16 - k2]);
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2)
                    get(t + (k1 + FX1 * k2) * C::SY, va[k2], va[15 - k2]);
            }
            fftk::Dft<kF2, true>::run(va);
#pragma unroll
            for (int i = 0; i < kF2; ++i) z[i] = st(va[i]);
        }
    }

    // ---- stride passes of the row transforms, fed from / drained into registers ----------
    // Work item = (row pair j, residue n2): the radix-FX1 butterfly over the elements
    // 16 n1 + n2 of the pair.  These are the passes next to global memory -- the model rows
    // come in, data / weights meet the rendered rows, the gradient rows go out -- and the
    // butterfly's FX1 points are exactly what a thread needs of its two rows, so the rows
    // never pass through the scratch on their own: loads feed the forward butterfly, the
    // inverse butterfly feeds the residual and the residual the next forward butterfly in
    // registers (round 2 wrote the rows to Z, met a barrier and read them back: 4 of the 16
    // LDS round trips of a band and 4 of its 14 barriers per chunk).
    // The sixteen residues of a pair sit in sixteen neighbouring lanes: a wavefront's global
    // accesses are 64-byte runs of four row pairs, and its scratch accesses are conflict-free
    // because the two pairs of a half-wave are 16 apart (16 SX complex = 32 banks mod 64).
    static constexpr int kStrideItems = kPairs * kF2;
    static_assert(kStrideItems == 512 && (kSplit || kThreads == kStrideItems), "stride items");
    __device__ __forceinline__ void stride_item(int &j, int &n2) const {
        const int l = tid & 63, w = (tid >> 6) & (kStrideItems / 64 - 1), q = l >> 4;
        n2 = l & 15;
        j = 2 * w + (q >> 1) + 16 * (q & 1);
    }
    // forward butterfly of v (v[n1] = element 16 n1 + n2 of the pair) into Z
    __device__ __forceinline__ void stride_forward(cf *v, int j, int n2) {
        float2 *a = Z + j * C::SX;
        fftk::Dft<FX1, false>::run(v);
#pragma unroll
        for (int k1 = 0; k1 < FX1; ++k1) {
            if (k1 > 0) v[k1] = cmul(v[k1], ld(twx[kF2 * k1 + n2]));
            a[kF2 * k1 + n2] = st(v[k1]);
        }
    }
    // inverse butterfly out of Z: v[k1] = element 16 k1 + n2 of the pair's rows
    __device__ __forceinline__ void stride_inverse(cf *v, int j, int n2) {
        const float2 *a = Z + j * C::SX;
#pragma unroll
        for (int n1 = 0; n1 < FX1; ++n1) {
            v[n1] = ld(a[kF2 * n1 + n2]);
            if (n1 > 0) v[n1] = cmulc(v[n1], ld(twx[kF2 * n1 + n2]));
        }
        fftk::Dft<FX1, true>::run(v);
    }
};


extern __shared__ __attribute__((aligned(16))) float2 lds_conv[];

// model: the model cube [nb][C][H][W] (render_kernel)
// mode 0: full (writes the gradient image G[nb][C][H][W] and the loss partial)
// mode 1: forward only (writes the rendered cube instead and the loss partial)
template <int FY1, int FX1>
__global__ __launch_bounds__(kThreads) void fused_conv_kernel(BatchView v, const float *model,
                                                              const float2 *Kt, int k_bands,
                                                              int k_per_blend, float *out,
                                                              int mode, long long *dbg) {
    using C = Cfg<FY1, FX1>;
    // XCD-aware placement: consecutive logical ids (the bands of one blend) share an
    // XCD and therefore its L2 (morphologies, data of neighbouring bands)
    const int total = gridDim.x;
    int lid = blockIdx.x;
    if (total % 8 == 0) lid = (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8;
    const int b = lid / v.C + v.blend0, c = lid % v.C;
    if (v.state[b] >= 2) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = v.H, W = v.W;

    Conv<FY1, FX1> cv;
    cv.T = lds_conv;
    cv.Z = cv.T + C::NKX * C::SY;
    cv.twy = cv.Z + kPairs * C::SX;
    cv.twx = cv.twy + C::FY;
    cv.tid = tid;
    cv.bt = Conv<FY1, FX1>::kSplit ? tid - Conv<FY1, FX1>::kStrideItems : tid;
    for (int j = tid; j < C::FY; j += kThreads) {  // j = 16 k1 + n2
        float s, co;
        sincospif(2.0f * (float)((j / kF2) * (j % kF2)) / (float)C::FY, &s, &co);
        cv.twy[j] = make_float2(co, -s);
    }
    for (int j = tid; j < C::FX; j += kThreads) {
        float s, co;
        sincospif(2.0f * (float)((j / kF2) * (j % kF2)) / (float)C::FX, &s, &co);
        cv.twx[j] = make_float2(co, -s);
    }
    const float2 *K = Kt + ((int64_t)(k_per_blend ? b : 0) * k_bands + (k_bands == 1 ? 0 : c)) *
                               C::FY * C::NKX;
    const int n_chunks = (H + 2 * kPairs - 1) / (2 * kPairs);
    __syncthreads();
// stage stamps of one workgroup (tools/stage_cycles.py): workgroup 0 runs in the first of a
// launch's rounds, where every CU bursts its loads at the same moment; a development build
// with -DSMI_DBG_BLOCK=<n> stamps a workgroup of the steady state instead
#ifndef SMI_DBG_BLOCK
#define SMI_DBG_BLOCK 0
#endif
#define SMI_STAMP(i) if (dbg && tid == 0 && blockIdx.x == SMI_DBG_BLOCK) dbg[i] = clock64()
    SMI_STAMP(0);

    // ---- A: model rows (blend.py:200-244, rendered by render_kernel) and their forward
    // row transforms.  The rows of the next chunk are requested before this chunk's
    // transforms, so that their latency hides behind them.
    int sj, sn2;
    cv.stride_item(sj, sn2);
    using std::integral_constant;
    // roles(f): f(stride role, radix-16 role) on the code path of this wavefront
    auto roles = [&](auto &&f) {
        if constexpr (Conv<FY1, FX1>::kSplit) {
            if (wave < Conv<FY1, FX1>::kStrideItems / 64)
                f(integral_constant<bool, true>{}, integral_constant<bool, false>{});
            else
                f(integral_constant<bool, false>{}, integral_constant<bool, true>{});
        } else {
            f(integral_constant<bool, true>{}, integral_constant<bool, true>{});
        }
    };
    const int64_t band = ((int64_t)b * v.C + c) * H * W;
    roles([&](auto stride_role, auto block_role) {
        constexpr bool S = decltype(stride_role)::value, B = decltype(block_role)::value;
        const plane_t r_model = band_plane(model + band, H * W);
        cf mrow[FX1];
        auto fetch = [&](int y0) {
#pragma unroll
            for (int n1 = 0; n1 < FX1; ++n1) {
                const int x = kF2 * n1 + sn2, y = y0 + 2 * sj;
                mrow[n1] = cf{plane_load(r_model, y, x, W), plane_load(r_model, y + 1, x, W)};
            }
        };
        if (S) fetch(0);
        SMI_STAMP(6);
        for (int ch = 0; ch < n_chunks; ++ch) {
            if (S) {
                cf cur[FX1];
#pragma unroll
                for (int n1 = 0; n1 < FX1; ++n1) cur[n1] = mrow[n1];
                if (ch + 1 < n_chunks) fetch((ch + 1) * 2 * kPairs);
                cv.stride_forward(cur, sj, sn2);
            }
            lds_barrier();
            if (ch == 0) SMI_STAMP(8);
            if (B) cv.blocks_forward(ch);
            lds_barrier();
            if (ch == 0) SMI_STAMP(9);
        }
    });
    SMI_STAMP(1);
    // ---- B: columns, x K^ --------------------------------------------------------
    cv.columns(K, H, false);
    SMI_STAMP(2);
    // ---- C: rendered rows -> residual, loss, forward rows of the residual ------------
    // (observation.py:147-170)  One pass per chunk between the two radix-16 passes: inverse
    // butterfly, w (m - d) and the loss on the thread's 2 x FX1 pixels, forward butterfly.
    double loss = 0.0;
    roles([&](auto stride_role, auto block_role) {
        constexpr bool S = decltype(stride_role)::value, B = decltype(block_role)::value;
        const plane_t r_data = band_plane(v.data + band, H * W);
        const plane_t r_weights = band_plane(v.weights + band, H * W);
        const plane_t r_rendered = band_plane(out + band, H * W);
        // data / weights of the chunk are fetched before the inverse radix-16 pass so that
        // their HBM latency hides behind it; columns >= 16 kPre are loaded late
        constexpr int kPre = FX1 > 8 ? 8 : FX1;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int y = ch * 2 * kPairs + 2 * sj;
            cf dpre[kPre], wpre[kPre];
            if (S) {
#pragma unroll
                for (int n1 = 0; n1 < kPre; ++n1) {
                    const int x = kF2 * n1 + sn2;
                    dpre[n1] = cf{plane_load(r_data, y, x, W), plane_load(r_data, y + 1, x, W)};
                    wpre[n1] = cf{plane_load(r_weights, y, x, W), plane_load(r_weights, y + 1, x, W)};
                }
            }
            if (ch == 0) SMI_STAMP(10);
            if (B) cv.blocks_inverse(ch);
            lds_barrier();
            if (ch == 0) SMI_STAMP(11);
            if (S) {
                cf m[FX1];
                cv.stride_inverse(m, sj, sn2);
#pragma unroll
                for (int n1 = 0; n1 < FX1; ++n1) {
                    const int x = kF2 * n1 + sn2;
                    cf dv, wv;
                    if (n1 < kPre) {
                        dv = dpre[n1 < kPre ? n1 : 0];
                        wv = wpre[n1 < kPre ? n1 : 0];
                    } else {
                        dv = cf{plane_load(r_data, y, x, W), plane_load(r_data, y + 1, x, W)};
                        wv = cf{plane_load(r_weights, y, x, W), plane_load(r_weights, y + 1, x, W)};
                    }
                    if (mode == 1) {
                        plane_store(r_rendered, y, x, W, m[n1].x);
                        plane_store(r_rendered, y + 1, x, W, m[n1].y);
                    }
                    // pixels outside the frame have weight 0 (the descriptor returns zeros)
                    const bool in0 = x < W && y < H, in1 = x < W && y + 1 < H;
                    const float d0 = m[n1].x - dv.x, d1 = m[n1].y - dv.y;
                    const float r0 = in0 ? wv.x * d0 : 0.f, r1 = in1 ? wv.y * d1 : 0.f;
                    // (rows beyond the frame may hold anything, NaN included: keep them out)
                    loss += (double)(in0 ? r0 * d0 : 0.f);
                    loss += (double)(in1 ? r1 * d1 : 0.f);
                    m[n1] = cf{r0, r1};
                }
                cv.stride_forward(m, sj, sn2);
            }
            lds_barrier();
            if (ch == 0) SMI_STAMP(13);
            if (B) cv.blocks_forward(ch);
            lds_barrier();
            if (ch == 0) SMI_STAMP(14);
        }
    });
    {
        double *part = reinterpret_cast<double *>(cv.Z);  // Z is free between stages
        const double t = block_sum(loss, part);
        if (tid == 0) v.loss_partial[(int64_t)b * v.n_partial + c] = t;
        __syncthreads();
    }
    SMI_STAMP(3);
    if (mode == 1) return;
    // ---- B': columns, x conj(K^) ---------------------------------------------------
    cv.columns(K, H, true);
    SMI_STAMP(4);
    // ---- D: gradient image rows -----------------------------------------------------
    roles([&](auto stride_role, auto block_role) {
        constexpr bool S = decltype(stride_role)::value, B = decltype(block_role)::value;
        // (stores beyond row H - 1 or column W - 1 are dropped by the descriptor's range check)
        const plane_t r_out = band_plane(out + band, H * W);
        for (int ch = 0; ch < n_chunks; ++ch) {
            if (B) cv.blocks_inverse(ch);
            lds_barrier();
            if (ch == 0) SMI_STAMP(12);
            if (S) {
                const int y = ch * 2 * kPairs + 2 * sj;
                cf g[FX1];
                cv.stride_inverse(g, sj, sn2);
#pragma unroll
                for (int k1 = 0; k1 < FX1; ++k1) {
                    plane_store(r_out, y, kF2 * k1 + sn2, W, g[k1].x);
                    plane_store(r_out, y + 1, kF2 * k1 + sn2, W, g[k1].y);
                }
            }
            lds_barrier();  // the scratch is free again; the stores may still be in flight
            if (ch == 0) SMI_STAMP(7);
        }
    });
    SMI_STAMP(5);
#undef SMI_STAMP
}

#ifndef SMI_CONV_SHORT_ROWS
// natural-order spectrum (rocFFT, [img][ky][kx]) -> [img][pos_y(ky)][kx], scaled
template <int FY1>
__global__ void permute_kernel_spectrum(const float2 *Khat, float2 *Kt, int NKX, float scale) {
    constexpr int FY = FY1 * kF2;
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FY * NKX) return;
    const int ky = i / NKX, kx = i - ky * NKX;
    const float2 k = Khat[(int64_t)img * FY * NKX + i];
    Kt[((int64_t)img * NKX + kx) * FY + pos<FY1>(ky)] = make_float2(k.x * scale, k.y * scale);
}

// Kernel spectrum for the fused path without rocFFT (whose plan creation costs ~0.6 s of
// run-time compilation per batch): a separable direct DFT of the small stamp,
//   A[dy][kx]  = sum_dx K[dy][dx] exp(-2 pi i kx (dx - pw/2) / Fx)
//   K^[ky][kx] = sum_dy A[dy][kx] exp(-2 pi i ky (dy - ph/2) / Fy)
// i.e. the stamp centre sits on index 0 (the layout the reference reaches with _pad +
// ifftshift, fft.py:255-273); written in the digit-swapped ky order, scaled.
__global__ void stamp_dft_x(const float *kern, double2 *A, int ph, int pw, int Fx, int NKX) {
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ph * NKX) return;
    const int dy = i / NKX, kx = i - dy * NKX;
    double re = 0.0, im = 0.0;  // set-up code: double accumulation, exact angle reduction
    for (int dx = 0; dx < pw; ++dx) {
        int t = (kx * (dx - pw / 2)) % Fx;
        if (t < 0) t += Fx;
        double s, c;
        sincospi(2.0 * (double)t / (double)Fx, &s, &c);
        const double k = (double)kern[((int64_t)img * ph + dy) * pw + dx];
        re += k * c;
        im -= k * s;
    }
    A[(int64_t)img * ph * NKX + i] = make_double2(re, im);
}

template <int FY1>
__global__ void stamp_dft_y(const double2 *A, float2 *Kt, int ph, int NKX, double scale) {
    constexpr int FY = FY1 * kF2;
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FY * NKX) return;
    const int ky = i / NKX, kx = i - ky * NKX;
    double re = 0.0, im = 0.0;
    for (int dy = 0; dy < ph; ++dy) {
        int t = (ky * (dy - ph / 2)) % FY;
        if (t < 0) t += FY;
        double s, c;
        sincospi(2.0 * (double)t / (double)FY, &s, &c);
        const double2 a = A[((int64_t)img * ph + dy) * NKX + kx];
        re += a.x * c + a.y * s;   // a * (c - i s)
        im += a.y * c - a.x * s;
    }
    Kt[((int64_t)img * NKX + kx) * FY + pos<FY1>(ky)] =
        make_float2((float)(re * scale), (float)(im * scale));
}

#endif  // SMI_CONV_SHORT_ROWS

template <int FY1, int FX1>
int launch_impl(const BatchView &v, const float *model, const float2 *Kt, int k_bands,
                int k_per_blend, float *out, int mode, long long *dbg, hipStream_t s) {
    using C = Cfg<FY1, FX1>;
    auto kern = fused_conv_kernel<FY1, FX1>;
    static size_t configured[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), C::lds_bytes, configured))
        return rc;
    hipLaunchKernelGGL(kern, dim3(v.nb * v.C), dim3(kThreads), C::lds_bytes, s, v, model, Kt,
                       k_bands, k_per_blend, out, mode, dbg);
    return SMI_OK;
}

}  // namespace

#ifndef SMI_CONV_SHORT_ROWS
// supported (FY, FX): multiples of 16 with first radix in {4,5,6,8,10}; LDS must fit
bool fused_conv_supported(int Fy, int Fx) {
    auto ok = [](int f) { return f == 64 || f == 80 || f == 96 || f == 128 || f == 160; };
    if (!ok(Fy) || !ok(Fx)) return false;
    const size_t sy = (size_t)((Fy + Fy / kF2 + 15) / 32) * 32 + 16;
    const size_t lds = sizeof(float2) * ((size_t)(Fx / 2 + 1) * sy +
                                         (size_t)kPairs * (Fx + 1) + Fy + Fx);
    return lds <= 160 * 1024;
}

// smallest supported length >= n, or 0
int fused_conv_length(int n) {
    for (int f : {64, 80, 96, 128, 160})
        if (f >= n) return f;
    return 0;
}

#endif

#define SMI_FUSED_DISPATCH(FN, ...)                                             \
    switch (Fy / 16 * 100 + Fx / 16) {                                          \
        case 404: return FN<4, 4>(__VA_ARGS__);                  \
        case 405: return FN<4, 5>(__VA_ARGS__);                  \
        case 406: return FN<4, 6>(__VA_ARGS__);                  \
        case 408: return FN<4, 8>(__VA_ARGS__);                  \
        case 410: return FN<4, 10>(__VA_ARGS__);                 \
        case 504: return FN<5, 4>(__VA_ARGS__);                  \
        case 505: return FN<5, 5>(__VA_ARGS__);                  \
        case 506: return FN<5, 6>(__VA_ARGS__);                  \
        case 508: return FN<5, 8>(__VA_ARGS__);                  \
        case 510: return FN<5, 10>(__VA_ARGS__);                 \
        case 604: return FN<6, 4>(__VA_ARGS__);                  \
        case 605: return FN<6, 5>(__VA_ARGS__);                  \
        case 606: return FN<6, 6>(__VA_ARGS__);                  \
        case 608: return FN<6, 8>(__VA_ARGS__);                  \
        case 610: return FN<6, 10>(__VA_ARGS__);                 \
        case 804: return FN<8, 4>(__VA_ARGS__);                  \
        case 805: return FN<8, 5>(__VA_ARGS__);                  \
        case 806: return FN<8, 6>(__VA_ARGS__);                  \
        case 808: return FN<8, 8>(__VA_ARGS__);                  \
        case 810: return FN<8, 10>(__VA_ARGS__);                 \
        case 1004: return FN<10, 4>(__VA_ARGS__);                \
        case 1005: return FN<10, 5>(__VA_ARGS__);                \
        case 1006: return FN<10, 6>(__VA_ARGS__);                \
        case 1008: return FN<10, 8>(__VA_ARGS__);                \
        case 1010: return FN<10, 10>(__VA_ARGS__);               \
        default: break;                                                         \
    }

#ifdef SMI_CONV_SHORT_ROWS
int launch_fused_conv_short(const BatchView &v, int Fy, int Fx, const float *model,
                            const float2 *Kt, int k_bands, int k_per_blend, float *out, int mode,
                            long long *dbg, hipStream_t s) {
    SMI_FUSED_DISPATCH(launch_impl, v, model, Kt, k_bands, k_per_blend, out, mode, dbg, s)
    set_error("fused convolution: FFT shape not instantiated");
    return SMI_ERR_INVALID;
}
#else
// every (Fy, Fx) in {64,80,96,128,160}^2 is instantiated; the LDS bound decides
bool fused_conv_instantiated(int Fy, int Fx) { return fused_conv_supported(Fy, Fx); }

// smallest-area supported shape with Fy >= ny and Fx >= nx
bool fused_conv_choose(int ny, int nx, int *Fy, int *Fx) {
    long best = 0;
    for (int fy : {64, 80, 96, 128, 160})
        for (int fx : {64, 80, 96, 128, 160}) {
            if (fy < ny || fx < nx || !fused_conv_supported(fy, fx)) continue;
            if (!best || (long)fy * fx < best) {
                best = (long)fy * fx;
                *Fy = fy;
                *Fx = fx;
            }
        }
    return best != 0;
}

int launch_fused_conv(const BatchView &v, int Fy, int Fx, const float *model, const float2 *Kt,
                      int k_bands, int k_per_blend, float *out, int mode, long long *dbg,
                      hipStream_t s) {
    // small transforms: the row passes of a chunk have 32 x 16 and 32 x Fx/16 work items, which
    // leave most of 1024 threads idle behind the barriers; 512 threads are faster there
    // (tools/conv_sizes.py, ms per 512 blends x 5 bands, 1024 -> 512 threads: 64^2 0.127 ->
    // 0.086, 80^2 0.145 -> 0.107, 96^2 0.233 -> 0.178, 64 x 128 0.175 -> 0.134, 160 x 64 0.211
    // -> 0.152, 160 x 80 0.234 -> 0.181; but 96 x 128 0.267 -> 0.285, 128 x 96 0.249 -> 0.278,
    // 80 x 160 0.228 -> 0.259, 160 x 96 0.266 -> 0.298, 128^2 0.283 -> 0.302, 160^2 0.383 -> 0.431)
    static const char *force = getenv("SMI_CONV_WORKGROUP");  // development aid: "512" / "1024"
    if (force ? force[0] == '5' : (Fx <= 80 || Fy * Fx <= 96 * 96))
        return launch_fused_conv_short(v, Fy, Fx, model, Kt, k_bands, k_per_blend, out, mode, dbg, s);
    SMI_FUSED_DISPATCH(launch_impl, v, model, Kt, k_bands, k_per_blend, out, mode, dbg, s)
    set_error("fused convolution: FFT shape not instantiated");
    return SMI_ERR_INVALID;
}

// d_kern: [n_img][ph][pw] on the device; Kt: [n_img][Fy][Fx/2+1]
int launch_stamp_spectrum(const float *d_kern, double2 *d_tmp, float2 *Kt, int n_img, int ph,
                          int pw, int Fy, int Fx, double scale, hipStream_t s) {
    const int NKX = Fx / 2 + 1;
    hipLaunchKernelGGL(stamp_dft_x, dim3((ph * NKX + 255) / 256, n_img), dim3(256), 0, s, d_kern,
                       d_tmp, ph, pw, Fx, NKX);
    const dim3 grid((Fy * NKX + 255) / 256, n_img);
    switch (Fy / 16) {
        case 4: hipLaunchKernelGGL(stamp_dft_y<4>, grid, dim3(256), 0, s, d_tmp, Kt, ph, NKX, scale); break;
        case 5: hipLaunchKernelGGL(stamp_dft_y<5>, grid, dim3(256), 0, s, d_tmp, Kt, ph, NKX, scale); break;
        case 6: hipLaunchKernelGGL(stamp_dft_y<6>, grid, dim3(256), 0, s, d_tmp, Kt, ph, NKX, scale); break;
        case 8: hipLaunchKernelGGL(stamp_dft_y<8>, grid, dim3(256), 0, s, d_tmp, Kt, ph, NKX, scale); break;
        case 10: hipLaunchKernelGGL(stamp_dft_y<10>, grid, dim3(256), 0, s, d_tmp, Kt, ph, NKX, scale); break;
        default:
            set_error("stamp spectrum: unsupported FFT height");
            return SMI_ERR_INVALID;
    }
    return SMI_OK;
}

int launch_permute_kernel_spectrum(const float2 *Khat, float2 *Kt, int n_img, int Fy, int Fx,
                                   float scale, hipStream_t s) {
    const int NKX = Fx / 2 + 1;
    const dim3 grid((Fy * NKX + 255) / 256, n_img);
    switch (Fy / 16) {
        case 4: hipLaunchKernelGGL(permute_kernel_spectrum<4>, grid, dim3(256), 0, s, Khat, Kt, NKX, scale); break;
        case 5: hipLaunchKernelGGL(permute_kernel_spectrum<5>, grid, dim3(256), 0, s, Khat, Kt, NKX, scale); break;
        case 6: hipLaunchKernelGGL(permute_kernel_spectrum<6>, grid, dim3(256), 0, s, Khat, Kt, NKX, scale); break;
        case 8: hipLaunchKernelGGL(permute_kernel_spectrum<8>, grid, dim3(256), 0, s, Khat, Kt, NKX, scale); break;
        case 10: hipLaunchKernelGGL(permute_kernel_spectrum<10>, grid, dim3(256), 0, s, Khat, Kt, NKX, scale); break;
        default:
            set_error("permute_kernel_spectrum: unsupported FFT height");
            return SMI_ERR_INVALID;
    }
    return SMI_OK;
}

#endif  // SMI_CONV_SHORT_ROWS

}  // namespace smi
