// In-register small DFTs for the LDS-resident FFT convolution (fused_conv.hip).
//
// dft<R, INV>(v): v[0..R) -> its DFT (sign -1 forward, +1 inverse, unnormalised),
// natural order in and out, everything statically indexed so the array lives in
// VGPRs.  Base butterflies 2, 3, 4, 5; composite radices are split R = RA * RB
// (Cooley-Tukey in registers) with compile-time twiddles.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace fftk {

constexpr double kPi = 3.14159265358979323846264338327950288;

// constexpr sin/cos of 2 pi k / R by Taylor series on an argument reduced to (-pi, pi]
// (clang does not constant-evaluate the libm builtins)
constexpr double series_sin(double x) {
    double term = x, sum = x;
    for (int n = 1; n < 24; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double series_cos(double x) {
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 24; ++n) {
        term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
constexpr double reduced_angle(int k, int R) {
    k %= R;
    if (k < 0) k += R;
    if (2 * k > R) k -= R;
    return 2.0 * kPi * (double)k / (double)R;
}
// exp(-2 pi i k / R) = (cos, -sin)
constexpr float tw_re(int k, int R) {
    k = ((k % R) + R) % R;
    if (4 * k == R || 4 * k == 3 * R) return 0.f;
    return (float)series_cos(reduced_angle(k, R));
}
constexpr float tw_im(int k, int R) {
    k = ((k % R) + R) % R;
    if (k == 0 || 2 * k == R) return 0.f;
    return (float)-series_sin(reduced_angle(k, R));
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Complex numbers are two-element vectors so that the compiler emits the packed f32
// instructions of CDNA3/4 (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32: one instruction per
// complex add, two per multiplication by a constant): the FFT passes are VALU-issue
// bound.  Swapping the halves of an operand is free (op_sel), so a rotation by +-i is
// folded into the addition that consumes it (`add_rot` / `sub_rot`) instead of being
// materialised.
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf ld(const float2 &a) { return cf{a.x, a.y}; }
__device__ __forceinline__ float2 st(cf a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ cf swp(cf a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cf fma2(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// a * b = a.xx * b + a.yy * (-b.y, b.x): two packed operations.  The second one reads b with
// its halves swapped (op_sel) and the low product negated (neg_lo); the compiler cannot
// express a negation of one half of a packed operand and spends a third operation on it,
// hence the inline assembly (same products, same single rounding of the fused step).
__device__ __forceinline__ cf cmul(cf a, cf b) {
    const cf t = a.xx * b;
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * conj(b) = a.xx * (b.x, -b.y) + a.yy * (b.y, b.x)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
// b + w d and b - w d with w = -i (forward) or +i (inverse), scaled by s:
// -i d = (d.y, -d.x), +i d = (-d.y, d.x)
template <bool INV>
__device__ __forceinline__ cf add_rot(cf b, cf d, float s = 1.f) {
    return fma2(swp(d), INV ? cf{-s, s} : cf{s, -s}, b);
}
template <bool INV>
__device__ __forceinline__ cf sub_rot(cf b, cf d, float s = 1.f) {
    return fma2(swp(d), INV ? cf{s, -s} : cf{-s, s}, b);
}
// a * exp(-/+ 2 pi i k / R) with the twiddle as immediate constants
template <int K, int R, bool INV>
__device__ __forceinline__ cf twiddle(cf a) {
    constexpr float c = tw_re(K, R);
    constexpr float s = INV ? -tw_im(K, R) : tw_im(K, R);
    if constexpr (((K % R) + R) % R == 0) {
        return a;
    } else if constexpr (c == 0.f) {
        return swp(a) * cf{-s, s};  // (-a.y s, a.x s)
    } else if constexpr (s == 0.f) {
        return a * c;
    } else {
        return fma2(a.yy, cf{-s, c}, a.xx * cf{c, s});
    }
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cf *v) {
        const cf a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <bool INV>
struct Dft<3, INV> {
    static __device__ __forceinline__ void run(cf *v) {
        constexpr float s = 0.86602540378443864676f;  // sin(pi/3)
        const cf t = v[1] + v[2];
        const cf d = v[1] - v[2];
        const cf m = fma2(t, cf{-0.5f, -0.5f}, v[0]);
        v[0] = v[0] + t;
        v[1] = add_rot<INV>(m, d, s);  // m -/+ i s (v1 - v2)
        v[2] = sub_rot<INV>(m, d, s);
    }
};

template <bool INV>
struct Dft<4, INV> {
    static __device__ __forceinline__ void run(cf *v) {
        const cf a = v[0] + v[2], b = v[0] - v[2];
        const cf c = v[1] + v[3], d = v[1] - v[3];
        v[0] = a + c;
        v[1] = add_rot<INV>(b, d);
        v[2] = a - c;
        v[3] = sub_rot<INV>(b, d);
    }
};

template <bool INV>
struct Dft<5, INV> {
    static __device__ __forceinline__ void run(cf *v) {
        constexpr float c1 = 0.30901699437494742410f;   // cos(2pi/5)
        constexpr float c2 = -0.80901699437494742410f;  // cos(4pi/5)
        constexpr float s1 = 0.95105651629515357212f;   // sin(2pi/5)
        constexpr float s2 = 0.58778525229247312917f;   // sin(4pi/5)
        const cf a1 = v[1] + v[4], b1 = v[1] - v[4];
        const cf a2 = v[2] + v[3], b2 = v[2] - v[3];
        const cf m1 = fma2(a2, cf{c2, c2}, fma2(a1, cf{c1, c1}, v[0]));
        const cf m2 = fma2(a2, cf{c1, c1}, fma2(a1, cf{c2, c2}, v[0]));
        // forward: X1 = m1 - i (s1 b1 + s2 b2), X2 = m2 - i (s2 b1 - s1 b2)
        const cf q1 = fma2(b2, cf{s2, s2}, b1 * s1);
        const cf q2 = fma2(b2, cf{-s1, -s1}, b1 * s2);
        v[0] = v[0] + (a1 + a2);
        v[1] = add_rot<INV>(m1, q1);
        v[4] = sub_rot<INV>(m1, q1);
        v[2] = add_rot<INV>(m2, q2);
        v[3] = sub_rot<INV>(m2, q2);
    }
};

// R = RA * RB in registers: n = RB n1 + n2, k = k1 + RA k2
template <int RA, int RB, bool INV>
struct DftComposite {
    static __device__ __forceinline__ void run(cf *v) {
        constexpr int R = RA * RB;
        cf t[RA * RB];  // t[k1 * RB + n2]
        static_for<0, RB>([&](auto n2c) {
            constexpr int n2 = decltype(n2c)::value;
            cf c[RA];
            static_for<0, RA>([&](auto n1c) {
                constexpr int n1 = decltype(n1c)::value;
                c[n1] = v[RB * n1 + n2];
            });
            Dft<RA, INV>::run(c);
            static_for<0, RA>([&](auto k1c) {
                constexpr int k1 = decltype(k1c)::value;
                t[k1 * RB + n2] = twiddle<k1 * n2, R, INV>(c[k1]);
            });
        });
        static_for<0, RA>([&](auto k1c) {
            constexpr int k1 = decltype(k1c)::value;
            cf c[RB];
            static_for<0, RB>([&](auto n2c) {
                constexpr int n2 = decltype(n2c)::value;
                c[n2] = t[k1 * RB + n2];
            });
            Dft<RB, INV>::run(c);
            static_for<0, RB>([&](auto k2c) {
                constexpr int k2 = decltype(k2c)::value;
                v[k1 + RA * k2] = c[k2];
            });
        });
    }
};

template <bool INV>
struct Dft<6, INV> : DftComposite<2, 3, INV> {};
template <bool INV>
struct Dft<8, INV> : DftComposite<2, 4, INV> {};
template <bool INV>
struct Dft<10, INV> : DftComposite<2, 5, INV> {};
template <bool INV>
struct Dft<12, INV> : DftComposite<3, 4, INV> {};
template <bool INV>
struct Dft<16, INV> : DftComposite<4, 4, INV> {};

}  // namespace fftk
