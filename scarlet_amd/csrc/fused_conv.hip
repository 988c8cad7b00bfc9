// LDS-resident PSF convolution + likelihood for one (blend, band) per workgroup.
//
// Replaces, for frames whose padded band fits the 160 KiB LDS of a CU, the chain
//   rocFFT R2C -> x K^ -> rocFFT C2R -> residual/loss -> rocFFT R2C -> x conj(K^) ->
//   rocFFT C2R
// (renderer.py:247-259 / fft.py:368-396 forward, its transpose backward,
// observation.py:147-170 in between) by ONE kernel that keeps the half-spectrum of
// the band in LDS from the first row transform to the last: per blend-iteration the
// only HBM traffic left is the model, data and weights (read once) and the gradient
// image (written once).  The model comes as a cube from render_kernel (kernels.hip) or,
// for a batch of nothing but factorized components, is rendered here row by row from the
// components' spectra and morphologies (ModelGather below): no cube, no launch.
//
// Layout: ONE array T[kx][y], kx in [0, FX/2) (the Nyquist frequency rides in the imaginary
// part of column 0); element y of a column sits at y + y / 16
// and the column stride SY is 4 mod 8 complex.  1-D transforms of length F = F1 * 16 are
// two in-place passes (radix F1 over stride 16, radix 16 over contiguous blocks); the
// forward transform leaves the spectrum in the digit-swapped order pos(k1 + F1 k2) =
// 16 k1 + k2, the inverse starts from it, so no transposition pass is ever needed; the
// kernel spectrum K^ is stored in the same order.  Rows are transformed two at a time
// (row 2j + i row 2j+1) and separated by Hermitian symmetry; zero padding is never stored
// for the columns.
//
// Round 3: the row transforms run IN the rows of T.  The complex sequence z of the row
// pair j (FX elements) lives in the 2 x (FX/2 + 1) slots that the pair's two rows own:
// element p at T[p / 2][2j + p % 2].  Rounds 1 and 2 transformed the pairs in a scratch
// Z of 32 pairs next to T -- all the LDS left -- so every row pass had 512 / 320 work
// items for 1024 threads, twice per stage, each followed by a barrier.  Now all pairs of
// the band go through a pass together (1024 / 640 work items for 128 rows), the
// scratch, the copies into and out of it and half of the barriers are gone.  The
// Hermitian separation after the radix-16 pass is an in-place permutation of the pair's
// slots (every work item reads its sixteen points, the workgroup meets a barrier, every
// work item writes its separated frequencies).
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

// Workgroup size.  This file is compiled twice: as it is (1024 threads) and through
// fused_conv_short.hip with 512 threads for small transforms (SMI_CONV_SHORT_ROWS: only
// launch_fused_conv_short is emitted there).
#ifndef SMI_CONV_THREADS
#define SMI_CONV_THREADS 1024
#endif
constexpr int kThreads = SMI_CONV_THREADS;
constexpr int kF2 = 16;      // second radix of every 1-D transform

constexpr int conv_column_stride(int fy) { return ((fy + fy / kF2 + 3) / 8) * 8 + 4; }
constexpr size_t conv_lds_bytes(int fy, int fx) {
    return sizeof(float2) * ((size_t)(fx / 2) * conv_column_stride(fy) + 2 * fy + fx) +
           sizeof(double) * (kThreads / 64);
}

// ZB: trailing 16-element blocks of BOTH axes that hold nothing but zero padding (frame rows
// < 16 (FY1 - ZB), frame columns < 16 (FX1 - ZB); chosen at launch, 0 .. 2).  The radix-F1
// passes over stride 16 own exactly one element of every block, so a padding block is a
// literal zero input of the forward butterflies (no load, and the butterfly is pruned by
// constant folding) and a dead output of the inverse ones (no store).
template <int FY1, int FX1, int ZB>
struct Cfg {
    static constexpr int FY = FY1 * kF2, FX = FX1 * kF2;
    static constexpr int NY1 = FY1 - ZB, NX1 = FX1 - ZB;  // blocks that can hold frame rows / columns
    static_assert(ZB >= 0 && NY1 >= 1 && NX1 >= 1, "padding blocks");
    // columns of K^ (kx = 0 .. FX / 2) and of T: the DC and the Nyquist frequency of a real row
    // are both real, so they share column 0 of T as its real and imaginary part (Conv::columns)
    static constexpr int NKX = FX / 2 + 1, NT = FX / 2;
    // column stride of T (complex).  A column stores element y at y + y / 16 (one pad
    // per radix-16 block, so that the lanes that each own a block hit different banks).
    // SY = 4 (mod 8): the eight columns 8 k1 + a that hold the elements 16 k1 + n2 of a
    // row pair then start 4, 8, .. 28 (mod 32) slots apart in some order, and the four
    // slots in between take the two rows of two neighbouring pairs -- a half-wave of a
    // stride pass (16 residues x 2 pairs) touches every bank once.  Columns four apart
    // start 16 (mod 32) slots apart: the two columns of a half-wave in the column stage
    // are chosen that way.
    static constexpr int SY = conv_column_stride(FY);
    static_assert(SY >= FY + FY / kF2 && SY % 8 == 4, "column stride");
    // + twiddle tables tw[k1 * 16 + n2] = exp(-2 pi i n2 k1 / F) for both axes, + column FX / 2
    // of the kernel spectrum (Conv::columns, column 0), + the partial sums of the loss
    static constexpr size_t lds_bytes = conv_lds_bytes(FY, FX);
};

// index of element y of a column of T
__device__ __forceinline__ int sk(int y) { return y + (y >> 4); }

// radix-F1 pass on a column of T (element 16 n1 + n2 lives at 17 n1 + n2).
// forward: A[k1] = w_F^(n2 k1) DFT_F1(x)[k1] written to a[16 k1 + n2]; the blocks n1 >= LIVE are
// zero padding (literal zeros), the rows below 16 LIVE that lie beyond the frame HOLD zeros
// (Conv::zero_tail), so no load is guarded.  inverse: the inputs B[k1] are first multiplied by
// w_F^(-n2 k1) (both directions carry their twiddles in this pass, which has the most
// work items, so the radix-16 pass stays short), then inverse DFT_F1; only the blocks
// k1 < LIVE are stored.
// `tw`: the lane's twiddles w_F^(n2 k1), k1 = 0 .. F1 - 1, in registers (the column stage reads
// them from the table once for all its passes)
template <int F1, bool INV, int LIVE>
__device__ __forceinline__ void pass_stride_col(float2 *a, int n2, const cf (&tw)[F1]) {
    cf v[F1];
#pragma unroll
    for (int n1 = 0; n1 < F1; ++n1) {
        v[n1] = (INV || n1 < LIVE) ? ld(a[(kF2 + 1) * n1 + n2]) : cf{0.f, 0.f};
        if (INV && n1 > 0) v[n1] = cmulc(v[n1], tw[n1]);
    }
    fftk::Dft<F1, INV>::run(v);
#pragma unroll
    for (int k1 = 0; k1 < F1; ++k1) {
        if (INV && k1 >= LIVE) continue;
        if (!INV && k1 > 0) v[k1] = cmul(v[k1], tw[k1]);
        a[(kF2 + 1) * k1 + n2] = st(v[k1]);
    }
}

// Single wavefront: LDS operations of one wave execute in order, so between passes
// that exchange data only among the lanes of a wave the LDS counter has to drain, no
// workgroup barrier is needed.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// column of T (and of K^) that holds the frequency kx of the row transforms (Conv, radix-16
// passes): frequency k1 + FX1 k2 in column 8 k1 + k2, the Nyquist frequency in column FX / 2
__host__ __device__ __forceinline__ int column_of(int kx, int Fx) {
    const int fx1 = Fx / kF2;
    return 2 * kx == Fx ? kx : 8 * (kx % fx1) + kx / fx1;
}

// position of natural frequency k in the digit-swapped order of a length F1*16 transform
template <int F1>
__device__ __forceinline__ int pos(int k) {
    return kF2 * (k % F1) + k / F1;
}

// sum over the workgroup in two halves around a barrier the caller has anyway: every
// wavefront leaves its partial sum (wave_partial, before the barrier), one thread adds them
// up (sum_partials, after it)
__device__ __forceinline__ void wave_partial(double v, double *part) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
}
__device__ __forceinline__ double sum_partials(const double *part) {
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
// Both rows y, y + 1 of a pair at the columns 16 n1 + n2 of a stride-pass work item: two
// byte offsets per work item, the column step is an immediate of the load (an offset
// register per load would cost the butterfly its registers).  Rows beyond H read 0 (range
// check); a column beyond W reads whatever follows the row -- the caller discards it.
struct PairRows {
    uint32_t o0, o1;
};
__device__ __forceinline__ PairRows pair_rows(int y, int n2, int W) {
    PairRows a;
    a.o0 = (uint32_t)(y * W + n2) * 4u;
    a.o1 = a.o0 + (uint32_t)W * 4u;
    // (opaque: hoisted out of a loop, offset + constant would become a register per load)
    asm volatile("" : "+v"(a.o0), "+v"(a.o1));
    return a;
}
// data and weights of the row pair j at the columns 16 N1 + n2, from the plane of float4 elements
// of BatchView::dw: d = (data[2j], data[2j+1]), w = (weights[2j], weights[2j+1])
// (a global load with a scalar base: offset register + immediate, like the buffer loads)
template <int N1>
__device__ __forceinline__ void dw_load(const char *plane, uint32_t o, cf &d, cf &w) {
    const float4 t = *reinterpret_cast<const float4 *>(plane + (size_t)o + 16u * kF2 * N1);
    d = cf{t.x, t.y};
    w = cf{t.z, t.w};
}
template <int N1>
__device__ __forceinline__ cf pair_load(plane_t r, PairRows a) {
    return cf{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, a.o0 + 4u * kF2 * N1, 0, 0)),
              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, a.o1 + 4u * kF2 * N1, 0, 0))};
}

template <int FY1, int FX1, int ZB>
struct Conv {
    using C = Cfg<FY1, FX1, ZB>;
    static constexpr int NX1 = C::NX1;
    float2 *T, *twy, *twx, *qtab;
    int tid, n_pairs;

    // The rows 2 n_pairs .. 16 NY1 - 1 of a column lie beyond the frame but inside a block
    // that the forward column pass reads: they hold zeros whenever that pass starts.  The
    // row passes never touch them; the inverse column pass leaves its (unused) outputs
    // there, so the wavefront that ran it clears them again.  (Nothing to do for a frame
    // that fills its blocks, e.g. 128 rows.)
    __device__ __forceinline__ void zero_tail(float2 *a, int g) const {
        for (int y = 2 * n_pairs + g; y < kF2 * C::NY1; y += kF2) a[sk(y)] = make_float2(0.f, 0.f);
    }

    // column transforms fused with the spectral product:
    //   T <- IFFT_y( FFT_y(T) * K )   for every column kx, K in digit-swapped order.
    // A column belongs to 16 lanes of one wavefront from start to end: lane g does the
    // radix-FY1 butterfly n2 = g of the first pass, the radix-16 block k1 = g of the
    // second (lanes g < FY1; forward, x K, inverse in registers), and n2 = g of the last.
    // The three passes of a column only exchange data inside that group, so no
    // workgroup barrier separates them and the wavefronts drift through the stage
    // independently.  The two columns of a half-wave are four apart (16 slots mod 32).
    __device__ __forceinline__ void columns(const float2 *Kt, bool conj) {
        const int g = tid & (kF2 - 1), G = tid >> 4;
        const int col = 8 * (G >> 3) + ((G >> 1) & 3) + 4 * (G & 1);
        constexpr int kStep = kThreads / kF2;
        cf tw[FY1];
#pragma unroll
        for (int k1 = 0; k1 < FY1; ++k1) tw[k1] = ld(twy[kF2 * k1 + g]);
        for (int kt = col; kt < C::NT; kt += kStep) {
            // (columns 0 .. 15 and 16 .. 31 change places: column 0, which costs its wavefront
            // a few hundred instructions more, goes to a wavefront that has no column of the
            // second trip on top)
            const int kx = kt < 32 ? kt ^ 16 : kt;
            float2 *a = T + kx * C::SY;
            pass_stride_col<FY1, false, C::NY1>(a, g, tw);
            wave_lds_fence();
            if (g < FY1) {
                float2 *blk = a + (kF2 + 1) * g;
                // (requesting K^ before the first pass was measured: slower, 12.9 -> 13.9 k cycles;
                // so was requesting the next trip's behind the product: 11.5 -> 13.0 k)
                const float2 *kp = Kt + (int64_t)kx * C::FY + kF2 * g;
                cf v[kF2], kv[kF2];
#pragma unroll
                for (int j = 0; j < kF2; ++j) kv[j] = ld(kp[j]);
#pragma unroll
                for (int j = 0; j < kF2; ++j) v[j] = ld(blk[j]);
                fftk::Dft<kF2, false>::run(v);
                // Column 0 carries two real columns, c = dc + i nyquist (both spectra of real
                // rows are real there), so its transform is C = A + i B with A, B Hermitian, and
                // what has to come out is K0 A + i KN B (K0, KN: K^ at kx = 0 and FX / 2,
                // Hermitian too).  With C~[ky] = conj(C[-ky]), A = (C + C~) / 2 and i B =
                // (C - C~) / 2, that is P C + Q C~, P = (K0 + KN) / 2, Q = (K0 - KN) / 2 -- the
                // kernel spectrum holds P in column 0 and Q in column FX / 2 (pack_dc_nyquist;
                // Q staged in LDS: `qtab`); for the adjoint conj(P) C + conj(Q) C~.  Frequency
                // -ky of entry j of block g: entry 15 - j of block FY1 - g, for block 0 its own
                // entry (16 - j) % 16.  The transformed blocks go back to the column, the common
                // product P C runs, then every lane of the column adds Q C~ entry by entry from
                // its mirror entries there (in place: the two sides of this lane-dependent
                // branch must not meet in different registers).
                if (kx == 0) {
#pragma unroll
                    for (int j = 0; j < kF2; ++j) blk[j] = st(v[j]);
                }
#pragma unroll
                for (int j = 0; j < kF2; ++j) v[j] = conj ? cmulc(v[j], kv[j]) : cmul(v[j], kv[j]);
                if (kx == 0) {
                    wave_lds_fence();
                    // (mirror entry of j: base[15 - j], base = the partner block, or for block 0
                    // its own entries from 1 on -- where entry 0 is its own mirror)
                    const float2 *base = g == 0 ? a + 1 : a + (kF2 + 1) * (FY1 - g);
                    const float2 *base0 = g == 0 ? a - 15 : base;
                    const float2 *qb = qtab + kF2 * g;
#pragma unroll
                    for (int j = 0; j < kF2; ++j) {
                        const cf m = ld((j == 0 ? base0 : base)[15 - j]), q = ld(qb[j]);
                        v[j] += conj ? cmul(cf{q.x, -q.y}, cf{m.x, -m.y}) : cmulc(q, m);
                        // (four entries at a time: all thirty-two loads up front would not
                        // fit the registers)
                        if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    wave_lds_fence();
                }
                fftk::Dft<kF2, true>::run(v);
#pragma unroll
                for (int j = 0; j < kF2; ++j) blk[j] = st(v[j]);
            }
            wave_lds_fence();
            pass_stride_col<FY1, true, C::NY1>(a, g, tw);
            zero_tail(a, g);
        }
        lds_barrier();
    }

    // ---- stride passes of the row transforms -------------------------------------------
    // Work item = (row pair j, residue n2): the radix-FX1 butterfly over the elements
    // 16 n1 + n2 of the pair.  These are the passes next to global memory -- the model rows
    // come in, data / weights meet the rendered rows, the gradient rows go out -- and the
    // butterfly's FX1 points are exactly what a thread needs of its two rows, so the rows
    // themselves never touch the LDS: loads feed the forward butterfly, the inverse butterfly
    // feeds the residual and the residual the next forward butterfly in registers.
    // Item -> lanes: the sixteen residues of a pair in sixteen neighbouring lanes (a wave's
    // global accesses are 64-byte runs of four row pairs), then the pairs.
    struct StrideItem {
        int j, n2, zb;  // zb: slot of element n2 of the pair; element 16 k1 + n2 is 8 SY k1 further
        bool on;
    };
    __device__ __forceinline__ StrideItem stride_item(int it) const {
        StrideItem s;
        s.n2 = it & 15;
        s.j = it >> 4;
        s.on = s.j < n_pairs;
        s.zb = (s.n2 >> 1) * C::SY + sk(2 * s.j + (s.n2 & 1));
        return s;
    }
    __device__ __forceinline__ int stride_items() const { return n_pairs * kF2; }
    // forward butterfly of v (v[n1] = element 16 n1 + n2 of the pair; the padding blocks
    // n1 >= NX1 are set to zero here) into the pair's slots
    // the item's twiddles w_FX^(n2 k1) from the table (keeping them in registers between the
    // inverse and the forward butterfly of stage C was measured: no change)
    __device__ __forceinline__ void stride_twiddles(cf (&tw)[FX1], const StrideItem &s) const {
#pragma unroll
        for (int k1 = 1; k1 < FX1; ++k1) tw[k1] = ld(twx[kF2 * k1 + s.n2]);
    }
    __device__ __forceinline__ void stride_forward(cf *v, const StrideItem &s, const cf (&tw)[FX1]) {
        float2 *a = T + s.zb;
#pragma unroll
        for (int n1 = NX1; n1 < FX1; ++n1) v[n1] = cf{0.f, 0.f};
        fftk::Dft<FX1, false>::run(v);
#pragma unroll
        for (int k1 = 0; k1 < FX1; ++k1) {
            if (k1 > 0) v[k1] = cmul(v[k1], tw[k1]);
            a[8 * C::SY * k1] = st(v[k1]);
        }
    }
    __device__ __forceinline__ void stride_forward(cf *v, const StrideItem &s) {
        cf tw[FX1];
        stride_twiddles(tw, s);
        stride_forward(v, s, tw);
    }
    // inverse butterfly: v[k1] = element 16 k1 + n2 of the pair's rows (the caller uses
    // k1 < NX1 only; the rest is dead code)
    __device__ __forceinline__ void stride_inverse(cf *v, const StrideItem &s, const cf (&tw)[FX1]) {
        const float2 *a = T + s.zb;
#pragma unroll
        for (int n1 = 0; n1 < FX1; ++n1) {
            v[n1] = ld(a[8 * C::SY * n1]);
            if (n1 > 0) v[n1] = cmulc(v[n1], tw[n1]);
        }
        fftk::Dft<FX1, true>::run(v);
    }
    __device__ __forceinline__ void stride_inverse(cf *v, const StrideItem &s) {
        cf tw[FX1];
        stride_twiddles(tw, s);
        stride_inverse(v, s, tw);
    }

    // ---- radix-16 passes of the row transforms, fused with the Hermitian separation ------
    // Block k1 of a pair holds the frequencies k1 + FX1 k2 (k2 = 0 .. 15); the mirror
    // frequency FX - k sits in block FX1 - k1 at 15 - k2 (block 0 mirrors onto itself at
    // (16 - k2) % 16, block FX1 / 2 of an even FX1 at 15 - k2).  Work item (row pair, slot);
    // 32 pairs x 2 slots make a wavefront, and the slots are ordered 1, FX1 - 1, 2, FX1 - 2,
    // ..., 0, FX1 / 2, so that the two slots of a wavefront are a block and its mirror
    // block: the lanes l and l ^ 32 exchange what the other one needs.
    // A work item separates the eight frequencies <= FX / 2 of its block (k2 < 8; block 0 the
    // Nyquist frequency on top) -- sixteen values for the sixteen slots it read.  So the
    // COLUMNS OF T ARE NOT IN FREQUENCY ORDER: frequency k1 + FX1 k2 lives in column
    // 8 k1 + k2, the Nyquist frequency in column FX / 2 (column_of; K^ is stored in the same
    // order, and the column stage does not care which frequency a column holds).  A work
    // item then writes exactly the slots it has read, in both directions: the pass is in
    // place without a barrier inside, and no work item touches another one's slots.
    static constexpr int kDouble = (FX1 - 1) / 2;  // blocks 1 .. kDouble have a distinct mirror block
    static constexpr int kSlots = FX1 + (FX1 & 1);  // slots per group of 32 pairs (even)
    static constexpr int kGroupsPerTrip = (kThreads / 32) / kSlots;
    static_assert(kGroupsPerTrip >= 1, "slots of a pair group in one trip");
    static __device__ __forceinline__ int slot_block(int slot) {
        if (slot < 2 * kDouble) return (slot & 1) ? FX1 - (slot / 2 + 1) : slot / 2 + 1;
        return slot == 2 * kDouble ? 0 : FX1 / 2;
    }
    // Lanes without a work item of their own (pairs beyond the last one in the last group
    // of 32, the pad slot of an odd FX1) repeat the work item of a neighbour -- the last
    // pair, block 0 -- and store the same values to the same slots: no lane is masked, and
    // the wavefronts that have no work item at all (`wave_on`, uniform) skip the pass.
    // (what the four radix-16 passes of a band need of it, in two registers; the item of the
    // first trip is worked out once per band and kept -- `first_block_item` --, the passes used
    // to spend a sixth of their instructions on recomputing it)
    struct BlockItem {
        int z;     // slot of the pair's rows in the first column of the block: 8 k1 SY + sk(2 j)
        int kind;  // 0: block with a mirror block in the partner half-wave, 1: block 0, 2: block FX1 / 2
        bool wave_on;
    };
    BlockItem first;
    __device__ __forceinline__ BlockItem block_item(int group0) const {
        BlockItem b;
        const int grp = tid >> 5;
        const int slot = grp % kSlots;
        int j = (tid & 31) + 32 * (group0 + grp / kSlots);
        const int wgrp = __builtin_amdgcn_readfirstlane(tid >> 6) * 2;  // first group of the wavefront
        b.wave_on = wgrp < kGroupsPerTrip * kSlots && 32 * (group0 + wgrp / kSlots) < n_pairs;
        if (j >= n_pairs) j = n_pairs - 1;
        const int k1 = slot < FX1 ? slot_block(slot) : 0;
        b.z = 8 * k1 * C::SY + sk(2 * j);
        b.kind = slot < 2 * kDouble ? 0 : k1 == 0 ? 1 : 2;
        return b;
    }
    __device__ __forceinline__ void first_block_item() {
        first = block_item(0);
        // (opaque: values the compiler cannot recompute stay in their registers)
        asm volatile("" : "+v"(first.z), "+v"(first.kind));
    }
    __device__ __forceinline__ BlockItem trip_item(int trip) const {
        return trip == 0 ? first : block_item(trip * kGroupsPerTrip);
    }
    __device__ __forceinline__ int block_trips() const {
        // (one trip whenever every pair group of the tallest frame fits the workgroup)
        if (kGroupsPerTrip >= (C::FY / 2 + 31) / 32) return 1;
        return ((n_pairs + 31) / 32 + kGroupsPerTrip - 1) / kGroupsPerTrip;
    }
    static __device__ __forceinline__ cf from_partner(cf v) {
        return cf{__shfl_xor(v.x, 32, 64), __shfl_xor(v.y, 32, 64)};
    }
    // Xa = za + conj(zb), Xb = -i (za - conj(zb))   (the 1/2 lives in K^); one packed addition
    // each, the swaps of halves and the signs in the operand modifiers (the compiler builds a
    // swapped operand with two register copies more often than not)
    static __device__ __forceinline__ void sep(cf za, cf zb, cf &xa, cf &xb) {
        // (za.x + zb.x, za.y - zb.y)
        asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(xa) : "v"(za), "v"(zb));
        // (za.y + zb.y, zb.x - za.x)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(xb) : "v"(za), "v"(zb));
    }
    static __device__ __forceinline__ cf plus(cf xa, cf xb) {  // Xa + i Xb = (xa.x - xb.y, xa.y + xb.x)
        cf r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(xa), "v"(xb));
        return r;
    }
    static __device__ __forceinline__ cf minus(cf xa, cf xb) {  // conj(Xa) + i conj(Xb) = (xa.x + xb.y, xb.x - xa.y)
        cf r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(xa), "v"(xb));
        return r;
    }

    // forward: radix-16 pass of every pair and separation into the pair's rows
    __device__ __forceinline__ void blocks_forward() {
        const int trips = block_trips();
        for (int trip = 0; trip < trips; ++trip) {
            const BlockItem b = trip_item(trip);
            if (!b.wave_on) continue;
            float2 *z = T + b.z;
            cf va[kF2], xa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                va[2 * i] = ld(z[i * C::SY]);
                va[2 * i + 1] = ld(z[i * C::SY + 1]);
            }
            fftk::Dft<kF2, false>::run(va);
            // (the stores sit in every branch: a common tail would have the three branches
            // hand over xa / xb through sixteen register copies each)
            auto store = [&] {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    z[k2 * C::SY] = st(xa[k2]);
                    z[k2 * C::SY + 1] = st(xb[k2]);
                }
            };
            if (b.kind == 0) {  // uniform over the wavefront
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) sep(va[k2], from_partner(va[15 - k2]), xa[k2], xb[k2]);
                store();
            } else if (b.kind == 1) {
                // (two branches: a select between va[a] and va[b] would be compiled into a
                // select of the index, i.e. a dynamically indexed register array)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) sep(va[k2], va[(16 - k2) & 15], xa[k2], xb[k2]);
                // the Nyquist frequency: real like the DC term (both imaginary parts are exact
                // zeros), it takes the imaginary part of the DC term's slot (Conv::columns)
                cf na, nb;
                sep(va[8], va[8], na, nb);
                xa[0].y = na.x;
                xb[0].y = nb.x;
                store();
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) sep(va[k2], va[15 - k2], xa[k2], xb[k2]);
                store();
            }
        }
        lds_barrier();
    }

    // inverse: the pair's elements <- radix-16 pass of (Xa + i Xb) rebuilt from its rows;
    // `mid` runs between the loads and the stores of a trip (global loads whose latency the
    // rest of the pass hides)
    template <typename F>
    __device__ __forceinline__ void blocks_inverse(F &&mid) {
        const int trips = block_trips();
        for (int trip = 0; trip < trips; ++trip) {
            const BlockItem b = trip_item(trip);
            if (!b.wave_on) {
                if (trip == 0) mid();
                continue;
            }
            float2 *z = T + b.z;
            cf va[kF2], xa[8], xb[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                xa[k2] = ld(z[k2 * C::SY]);
                xb[k2] = ld(z[k2 * C::SY + 1]);
            }
            // (the global loads of `mid` go out behind the LDS loads, in front of the branches;
            // transform and stores in every branch: see blocks_forward)
            __builtin_amdgcn_sched_barrier(0);
            if (trip == 0) mid();
            __builtin_amdgcn_sched_barrier(0);
            auto tail = [&] {
                fftk::Dft<kF2, true>::run(va);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    z[i * C::SY] = st(va[2 * i]);
                    z[i * C::SY + 1] = st(va[2 * i + 1]);
                }
            };
            if (b.kind == 0) {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    va[k2] = plus(xa[k2], xb[k2]);
                    va[15 - k2] = from_partner(minus(xa[k2], xb[k2]));
                }
                tail();
            } else if (b.kind == 1) {
                // (slot 0 of the rows: DC term in the real, Nyquist term in the imaginary part)
                va[0] = cf{xa[0].x, xb[0].x};
                va[8] = cf{xa[0].y, xb[0].y};
#pragma unroll
                for (int k2 = 1; k2 < 8; ++k2) {
                    va[k2] = plus(xa[k2], xb[k2]);
                    va[16 - k2] = minus(xa[k2], xb[k2]);
                }
                tail();
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    va[k2] = plus(xa[k2], xb[k2]);
                    va[15 - k2] = minus(xa[k2], xb[k2]);
                }
                tail();
            }
        }
        lds_barrier();
    }
};


// ---- model rows rendered inside the kernel (Blend.get_model, blend.py:200-244) --------------
// Without a model cube (`model == nullptr`) a stride-pass work item builds its 2 x FX1 model
// pixels itself: rows y, y + 1 at the columns 16 n1 + n2, every pixel the sum of sed_k[c] *
// morph_k over the components whose box holds it, in ascending component order with one fma
// per term -- the arithmetic of render_kernel (kernels.hip), so the rows are bit for bit what
// the cube would have held; the cube's round trip through HBM (write, read) and a launch per
// iteration go away.
//
// Component metadata sits one component per lane; the wavefront (four row pairs = eight
// frame rows) walks through the components whose box meets its rows.  A box of width w holds
// at most M = (w + 14) / 16 + 1 columns of one residue class, n1 = n1lo .. n1lo + M - 1 with
// n1lo = ox / 16: every component costs 2 M loads whatever its position -- a STATIC number, so
// that the loads of the next component are in flight while this one is accumulated (with
// loads behind branches the compiler's wait-counter pass falls back to vmcnt(0)); lanes whose
// column or row lies outside the box (or the frame) get an out-of-range offset, for which the
// buffer load returns 0 without touching memory, and fma(sed, 0, acc) = acc.  Only the
// accumulation branches (wave-uniformly) on n1lo, to keep the register indices static.
// (FX1 blocks per row, of which the first NX1 can hold frame columns)
template <int FX1, int NX1, int M>
struct ModelGather {
    static constexpr uint32_t kNone = 0x40000000u;  // rowbase + column stays out of range
    const BatchView &v;
    int band_c, H, W, y, n2, ylo;
    // metadata of 64 components, one per lane (all that `open` needs comes from here by
    // v_readlane: no scalar address arithmetic, no load per component)
    struct Lanes {
        int oy, ox, h, w, mo;
        float sd;
    };
    // pops the first component of `todo` and issues its loads into mv; returns n1lo, or -1
    // with nothing left -- the loads are still issued then (all out of range: no memory
    // access), so that the number of loads in flight never depends on a branch
    __device__ __forceinline__ int open(unsigned long long &todo, const Lanes &l, float &sd,
                                        cf (&mv)[M]) const {
        const bool none = todo == 0;
        const int kl = none ? 0 : __builtin_ctzll(todo);
        todo &= todo - 1;  // (0 stays 0)
        const int oy = __builtin_amdgcn_readlane(l.oy, kl), hh = __builtin_amdgcn_readlane(l.h, kl);
        const int ox = __builtin_amdgcn_readlane(l.ox, kl), w = __builtin_amdgcn_readlane(l.w, kl);
        const float *mbase = v.morph + __builtin_amdgcn_readlane(l.mo, kl);
        const plane_t r = band_plane(mbase, hh * w);
        sd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l.sd), kl));
        const int n1lo = (ox > 0 ? ox : 0) >> 4;
        const uint32_t hlim = none ? 0u : (uint32_t)min(hh, H - oy);
        const uint32_t wlim4 = 4u * (uint32_t)min(w, W - ox);
        const uint32_t ry = (uint32_t)(y - oy), w4 = 4u * (uint32_t)w;
        uint32_t row0, row1;  // (24-bit multiplies, full rate; the compiler picks v_mul_lo_u32)
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(row0) : "v"(ry), "s"(w4));
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(row1) : "v"(ry + 1u), "s"(w4));
        row0 = ry < hlim ? row0 : kNone;
        row1 = ry + 1u < hlim ? row1 : kNone;
        const uint32_t x4 = (uint32_t)(16 * n1lo + n2 - ox) * 4u;  // byte offset of slot 0 in its row
#pragma unroll
        for (int m = 0; m < M; ++m) {
            uint32_t t = x4 + 64u * m;
            t = t < wlim4 ? t : kNone;
            mv[m].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, row0 + t, 0, 0));
            mv[m].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, row1 + t, 0, 0));
        }
        return none ? -1 : n1lo;
    }
    __device__ __forceinline__ void close(int n1lo_, float sd_, const cf (&mv)[M], cf (&acc)[FX1]) const {
        const cf sd = cf{sd_, sd_};
        fftk::static_for<0, NX1>([&](auto lo) {
            constexpr int n1lo = decltype(lo)::value;
            if (n1lo_ == n1lo) {
#pragma unroll
                for (int m = 0; m < M; ++m)
                    if (n1lo + m < NX1) acc[n1lo + m] = fftk::fma2(mv[m], sd, acc[n1lo + m]);
                // (keeps the cases apart: merged, they index `acc` dynamically and the
                // array moves to scratch memory)
                asm volatile("; n1lo = %0" ::"n"(n1lo));
            }
        });
    }
    __device__ __forceinline__ void run(int b, int lane, cf (&acc)[FX1]) const {
#pragma unroll
        for (int n1 = 0; n1 < NX1; ++n1) acc[n1] = cf{0.f, 0.f};
        const int cs = v.comp_start[b], ce = v.comp_start[b + 1];
        for (int kb = cs; kb < ce; kb += 64) {
            const int kk = kb + lane;
            const bool have = kk < ce;
            Lanes l;
            l.oy = have ? v.c_oy[kk] : 0;
            l.ox = have ? v.c_ox[kk] : 0;
            l.h = have ? v.c_h[kk] : 0;
            l.w = have ? v.c_w[kk] : 0;
            l.mo = have ? (int)v.c_moff[kk] : 0;  // packed offsets fit 31 bits
            l.sd = have ? v.sed[kk * v.C + band_c] : 0.f;
            const bool hit = have && l.oy < ylo + 8 && l.oy + l.h > ylo && l.oy < H && l.ox < W &&
                             l.ox + l.w > 0;
            unsigned long long todo = __ballot(hit);
            if (!todo) continue;
            cf mva[M], mvb[M];
            float sda, sdb;
            int na = open(todo, l, sda, mva);
            for (;;) {
                const int nb = open(todo, l, sdb, mvb);
                close(na, sda, mva, acc);
                if (nb < 0) break;
                na = open(todo, l, sda, mva);
                close(nb, sdb, mvb, acc);
                if (na < 0) break;
            }
        }
    }
};

extern __shared__ __attribute__((aligned(16))) float2 lds_conv[];

// model: the model cube [nb][C][H][W] (render_kernel)
// mode 0: full (writes the gradient image G[nb][C][H][W] and the loss partial)
// mode 1: forward only (writes the rendered cube instead and the loss partial)
template <int FY1, int FX1, int ZB>
__global__ __launch_bounds__(kThreads) void fused_conv_kernel(BatchView v, const float *model,
                                                              const float2 *Kt, int k_bands,
                                                              int k_per_blend, float *out,
                                                              int mode, long long *dbg) {
    using C = Cfg<FY1, FX1, ZB>;
    constexpr int NX1 = C::NX1;
    // XCD-aware placement: consecutive logical ids (the bands of one blend) share an
    // XCD and therefore its L2 (morphologies, data of neighbouring bands)
    const int total = gridDim.x;
    int lid = blockIdx.x;
    if (total % 8 == 0) lid = (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8;
    const int b = lid / v.C + v.blend0, c = lid % v.C;
    if (v.state[b] >= 2) return;
    const int tid = threadIdx.x;
    const int H = v.H, W = v.W;

    Conv<FY1, FX1, ZB> cv;
    cv.T = lds_conv;
    cv.twy = cv.T + C::NT * C::SY;
    cv.twx = cv.twy + C::FY;
    cv.tid = tid;
    cv.n_pairs = (H + 1) / 2;
    cv.first_block_item();
    cv.qtab = cv.twx + C::FX;
    double *loss_part = reinterpret_cast<double *>(cv.qtab + C::FY);
    const int64_t band = ((int64_t)b * v.C + c) * H * W;
    const int n_items = cv.stride_items();
    using Item = typename Conv<FY1, FX1, ZB>::StrideItem;

    // the first model rows are requested before the twiddle tables are made
    // (model == nullptr: the rows are rendered here, ModelGather)
    const bool have_cube = model != nullptr;
    const plane_t r_model = band_plane(have_cube ? model + band : v.data + band, H * W);
    cf mrow[FX1];
    auto fetch_model = [&](const Item &s) {
        const PairRows a = pair_rows(2 * s.j, s.n2, W);
        fftk::static_for<0, NX1>([&](auto n1c) {
            constexpr int n1 = decltype(n1c)::value;
            mrow[n1] = pair_load<n1>(r_model, a);
        });
    };
    auto render_rows = [&](const Item &s) {
        // the four row pairs of this wavefront start at pair (item / 64) * 4
        const int j0 = __builtin_amdgcn_readfirstlane(s.j - ((tid & 63) >> 4));
        if (v.render_slots <= 4) {
            const ModelGather<FX1, NX1, 4> g{v, c, H, W, 2 * s.j, s.n2, 2 * j0};
            g.run(b, tid & 63, mrow);
        } else {
            const ModelGather<FX1, NX1, (FX1 < 6 ? FX1 : 6)> g{v, c, H, W, 2 * s.j, s.n2, 2 * j0};
            g.run(b, tid & 63, mrow);
        }
    };
    if (have_cube) fetch_model(cv.stride_item(tid));
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
    {   // zero padding rows inside the blocks that the column passes read (Conv::zero_tail)
        const int y0 = 2 * cv.n_pairs, tail = kF2 * C::NY1 - y0;
        for (int i = tid; i < tail * C::NT; i += kThreads)
            cv.T[(i / tail) * C::SY + sk(y0 + i % tail)] = make_float2(0.f, 0.f);
    }
    const float2 *K = Kt + ((int64_t)(k_per_blend ? b : 0) * k_bands + (k_bands == 1 ? 0 : c)) *
                               C::FY * C::NKX;
    for (int j = tid; j < C::FY; j += kThreads) cv.qtab[j] = K[(int64_t)(C::FX / 2) * C::FY + j];
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
    // row transforms
    // (rendering walks through the components with wave-wide votes: every lane of a wavefront
    // that has an item stays in the loop)
    for (int it = tid; (have_cube ? it : it - (tid & 63)) < n_items; it += kThreads) {
        const Item s = cv.stride_item(it);
        if (have_cube) {
            if (it != tid) fetch_model(s);
#pragma unroll
            for (int n1 = 0; n1 < NX1; ++n1)  // the zero padding beyond column W - 1
                if (kF2 * n1 + s.n2 >= W) mrow[n1] = cf{0.f, 0.f};
        } else {
            render_rows(s);
        }
        if (s.on) cv.stride_forward(mrow, s);
    }
    lds_barrier();
    SMI_STAMP(6);
    cv.blocks_forward();
    SMI_STAMP(1);
    // ---- B: columns, x K^ --------------------------------------------------------
    cv.columns(K, false);
    SMI_STAMP(2);
    // ---- C: rendered rows -> residual, loss, forward rows of the residual ------------
    // (observation.py:147-170)  One pass between the two radix-16 passes: inverse
    // butterfly, w (m - d) and the loss on the thread's 2 x FX1 pixels, forward butterfly.
    double loss = 0.0;
    {
        cf loss2 = cf{0.f, 0.f};
        // data and weights by row pairs: one 16-byte load per pixel pair (BatchView::dw)
        const int64_t pband = ((int64_t)b * v.C + c) * cv.n_pairs * W;
        const char *p_dw = reinterpret_cast<const char *>(v.dw + pband);
        const plane_t r_rendered = band_plane(out + band, H * W);
        // data / weights of the first SMI_CONV_PRE columns of the butterfly are requested
        // before the inverse radix-16 pass (their HBM latency hides behind it, the registers
        // are live across it), the others between its loads and its stores.  With the 16-byte
        // loads of BatchView::dw all eight fit (0, 2, 4, 8 measured: conv 0.549 / 0.554 / 0.554 /
        // 0.543 ms over 100 iterations, 0.598 / 0.599 / 0.600 / 0.589 in the driver window)
#ifndef SMI_CONV_PRE
#define SMI_CONV_PRE 8
#endif
        constexpr int kPre = SMI_CONV_PRE < NX1 ? SMI_CONV_PRE : NX1;
        cf dv[NX1], wv[NX1];
        auto fetch = [&](const Item &s, auto from, auto to) {
            // (a thread without an item of its own reads the last pair; columns beyond W read
            // what follows the row -- the allocation has sixteen elements of slack -- and are
            // discarded below.  One offset register, the column step an immediate.)
            uint32_t o = (uint32_t)(min(s.j, cv.n_pairs - 1) * W + s.n2) * 16u;
            asm volatile("" : "+v"(o));
            fftk::static_for<decltype(from)::value, decltype(to)::value>([&](auto n1c) {
                constexpr int n1 = decltype(n1c)::value;
                dw_load<n1>(p_dw, o, dv[n1], wv[n1]);
            });
        };
        using std::integral_constant;
        const Item s0 = cv.stride_item(tid);
        fetch(s0, integral_constant<int, 0>{}, integral_constant<int, kPre>{});
        // (a butterfly of ten columns has no registers for all of them next to its twiddles:
        // the last kLate columns are requested behind the butterfly)
        constexpr int kLate = NX1 > 8 ? NX1 - 8 : 0;
        cv.blocks_inverse([&] { fetch(s0, integral_constant<int, kPre>{}, integral_constant<int, NX1 - kLate>{}); });
        SMI_STAMP(10);
        for (int it = tid; it < n_items; it += kThreads) {
            const Item s = cv.stride_item(it);
            const int y = 2 * s.j;
            if (it != tid) fetch(s, integral_constant<int, 0>{}, integral_constant<int, NX1 - kLate>{});
            cf m[FX1];
            cv.stride_inverse(m, s);
            fetch(s, integral_constant<int, NX1 - kLate>{}, integral_constant<int, NX1>{});
#pragma unroll
            for (int n1 = 0; n1 < NX1; ++n1) {
                const int x = kF2 * n1 + s.n2;
                if (mode == 1) {
                    plane_store(r_rendered, y, x, W, m[n1].x);
                    plane_store(r_rendered, y + 1, x, W, m[n1].y);
                }
                // row H of an odd frame carries weight 0 (BatchView::dw), columns beyond W
                // whatever follows the row: a lane select.  Packed: d = m - data, r = w d,
                // loss += r d (per thread in float32 -- 2 FX1 terms --, across threads in double)
                const cf w = x < W ? wv[n1] : cf{0.f, 0.f};
                const cf d = m[n1] - dv[n1];
                m[n1] = w * d;
                loss2 = fftk::fma2(m[n1], d, loss2);
            }
            cv.stride_forward(m, s);
            loss += (double)loss2.x + (double)loss2.y;
            loss2 = cf{0.f, 0.f};
        }
        wave_partial(loss, loss_part);
        lds_barrier();
        SMI_STAMP(11);
        cv.blocks_forward();
    }
    // (behind the barrier of the pass; the other wavefronts go on)
    if (tid == 0) v.loss_partial[(int64_t)b * v.n_partial + c] = sum_partials(loss_part);
    SMI_STAMP(3);
    if (mode == 1) return;
    // ---- B': columns, x conj(K^) ---------------------------------------------------
    cv.columns(K, true);
    SMI_STAMP(4);
    // ---- D: gradient image rows -----------------------------------------------------
    {
        // (stores beyond row H - 1 or column W - 1 are dropped by the descriptor's range check)
        const plane_t r_out = band_plane(out + band, H * W);
        cv.blocks_inverse([] {});
        SMI_STAMP(12);
        for (int it = tid; it < n_items; it += kThreads) {
            const Item s = cv.stride_item(it);
            cf g[FX1];
            cv.stride_inverse(g, s);
#pragma unroll
            for (int k1 = 0; k1 < NX1; ++k1) {
                plane_store(r_out, 2 * s.j, kF2 * k1 + s.n2, W, g[k1].x);
                plane_store(r_out, 2 * s.j + 1, kF2 * k1 + s.n2, W, g[k1].y);
            }
        }
    }
    SMI_STAMP(5);
#undef SMI_STAMP
}

#ifndef SMI_CONV_SHORT_ROWS
// natural-order spectrum (rocFFT, [img][ky][kx]) -> [img][pos_y(ky)][kx], scaled
template <int FY1>
__global__ void permute_kernel_spectrum(const float2 *Khat, float2 *Kt, int NKX, float scale) {
    const int Fx = 2 * (NKX - 1);
    constexpr int FY = FY1 * kF2;
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FY * NKX) return;
    const int ky = i / NKX, kx = i - ky * NKX;
    const float2 k = Khat[(int64_t)img * FY * NKX + i];
    Kt[((int64_t)img * NKX + column_of(kx, Fx)) * FY + pos<FY1>(ky)] =
        make_float2(k.x * scale, k.y * scale);
}

// Columns 0 and FX / 2 of the kernel spectrum (K0 and KN: the DC and the Nyquist frequency of
// the rows) -> P = (K0 + KN) / 2 and Q = (K0 - KN) / 2, what Conv::columns multiplies the packed
// column 0 of T and its mirrored conjugate by.
__global__ void pack_dc_nyquist(float2 *Kt, int Fy, int NKX) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Fy) return;
    float2 *k0 = Kt + (int64_t)blockIdx.y * NKX * Fy + i, *kn = k0 + (int64_t)(NKX - 1) * Fy;
    const float2 a = *k0, b = *kn;
    *k0 = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
    *kn = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
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
    Kt[((int64_t)img * NKX + column_of(kx, 2 * (NKX - 1))) * FY + pos<FY1>(ky)] =
        make_float2((float)(re * scale), (float)(im * scale));
}

#endif  // SMI_CONV_SHORT_ROWS

template <int FY1, int FX1, int ZB>
int launch_zb(const BatchView &v, const float *model, const float2 *Kt, int k_bands,
              int k_per_blend, float *out, int mode, long long *dbg, hipStream_t s) {
    using C = Cfg<FY1, FX1, ZB>;
    auto kern = fused_conv_kernel<FY1, FX1, ZB>;
    static size_t configured[kMaxDevices] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), C::lds_bytes, configured))
        return rc;
    hipLaunchKernelGGL(kern, dim3(v.nb * v.C), dim3(kThreads), C::lds_bytes, s, v, model, Kt,
                       k_bands, k_per_blend, out, mode, dbg);
    return SMI_OK;
}

// the kernel variant for the frame: as many all-padding blocks (of both axes) as it has, up to 2
template <int FY1, int FX1>
int launch_impl(const BatchView &v, const float *model, const float2 *Kt, int k_bands,
                int k_per_blend, float *out, int mode, long long *dbg, hipStream_t s) {
    const int zy = FY1 - (v.H + kF2 - 1) / kF2, zx = FX1 - (v.W + kF2 - 1) / kF2;
    const int zb = zy < zx ? zy : zx;
    if (zb >= 2) return launch_zb<FY1, FX1, 2>(v, model, Kt, k_bands, k_per_blend, out, mode, dbg, s);
    if (zb == 1) return launch_zb<FY1, FX1, 1>(v, model, Kt, k_bands, k_per_blend, out, mode, dbg, s);
    return launch_zb<FY1, FX1, 0>(v, model, Kt, k_bands, k_per_blend, out, mode, dbg, s);
}

}  // namespace

#ifndef SMI_CONV_SHORT_ROWS
// supported (FY, FX): multiples of 16 with first radix in {4,5,6,8,10}; LDS must fit
bool fused_conv_supported(int Fy, int Fx) {
    auto ok = [](int f) { return f == 64 || f == 80 || f == 96 || f == 128 || f == 160; };
    if (!ok(Fy) || !ok(Fx)) return false;
    return conv_lds_bytes(Fy, Fx) <= 160 * 1024;
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
    // 512 or 1024 threads per workgroup.  With the row transforms in place a band needs 44 ..
    // 114 KB of LDS, so two or three 512-thread workgroups share a CU for most shapes -- and
    // two workgroups that are out of step overlap each other's LDS and arithmetic phases,
    // which the wavefronts of ONE workgroup, all in the same pass between the same barriers,
    // do not.  tools/conv_sizes.py, ms per 512 blends x 5 bands, 512 / 1024 threads: 64^2 0.070 /
    // 0.103, 80^2 0.087 / 0.121, 96^2 0.121 / 0.152, 128^2 0.178 / 0.212, 64 x 128 0.109 / 0.145,
    // 128 x 64 0.101 / 0.131, 160 x 64 0.112 / 0.143, 160 x 80 0.144 / 0.174, 160 x 96 0.175 / 0.203,
    // 96 x 128 0.154 / 0.189, 128 x 96 0.140 / 0.175, 64 x 160 0.135 / 0.173; but 80 x 160 0.206 /
    // 0.194, 96 x 160 0.262 / 0.233, 128 x 160 0.303 / 0.268, 160 x 128 0.272 / 0.250, 160^2 0.357 /
    // 0.310 (one workgroup per CU either way, or rows of 160 elements).
    static const char *force = getenv("SMI_CONV_WORKGROUP");  // development aid: "512" / "1024"
    const bool large = (Fx == 160 && Fy >= 80) || (Fy == 160 && Fx >= 128);
    if (force ? force[0] == '5' : !large)
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
    hipLaunchKernelGGL(pack_dc_nyquist, dim3((Fy + 255) / 256, n_img), dim3(256), 0, s, Kt, Fy, NKX);
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
    hipLaunchKernelGGL(pack_dc_nyquist, dim3((Fy + 255) / 256, n_img), dim3(256), 0, s, Kt, Fy, NKX);
    return SMI_OK;
}

#endif  // SMI_CONV_SHORT_ROWS

}  // namespace smi
