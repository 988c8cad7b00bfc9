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

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// a * exp(-/+ 2 pi i k / R) with the twiddle as immediate constants
template <int K, int R, bool INV>
__device__ __forceinline__ float2 twiddle(float2 a) {
    constexpr float c = tw_re(K, R);
    constexpr float s = INV ? -tw_im(K, R) : tw_im(K, R);
    if constexpr (((K % R) + R) % R == 0) {
        return a;
    } else {
        return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
    }
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
    static __device__ __forceinline__ void run(float2 *v) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <bool INV>
struct Dft<3, INV> {
    static __device__ __forceinline__ void run(float2 *v) {
        constexpr float s = 0.86602540378443864676f;  // sin(pi/3)
        const float2 t = cadd(v[1], v[2]);
        const float2 d = rot90<INV>(csub(v[1], v[2]));  // -/+ i (v1 - v2)
        const float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
        v[0] = cadd(v[0], t);
        v[1] = make_float2(m.x + s * d.x, m.y + s * d.y);
        v[2] = make_float2(m.x - s * d.x, m.y - s * d.y);
    }
};

template <bool INV>
struct Dft<4, INV> {
    static __device__ __forceinline__ void run(float2 *v) {
        const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
        const float2 c = cadd(v[1], v[3]), d = rot90<INV>(csub(v[1], v[3]));
        v[0] = cadd(a, c);
        v[1] = cadd(b, d);
        v[2] = csub(a, c);
        v[3] = csub(b, d);
    }
};

template <bool INV>
struct Dft<5, INV> {
    static __device__ __forceinline__ void run(float2 *v) {
        constexpr float c1 = 0.30901699437494742410f;   // cos(2pi/5)
        constexpr float c2 = -0.80901699437494742410f;  // cos(4pi/5)
        constexpr float s1 = 0.95105651629515357212f;   // sin(2pi/5)
        constexpr float s2 = 0.58778525229247312917f;   // sin(4pi/5)
        const float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
        const float2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
        const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
        const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
        // forward: X1 = m1 - i (s1 b1 + s2 b2), X2 = m2 - i (s2 b1 - s1 b2)
        const float2 n1 = rot90<INV>(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
        const float2 n2 = rot90<INV>(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = cadd(m1, n1);
        v[4] = csub(m1, n1);
        v[2] = cadd(m2, n2);
        v[3] = csub(m2, n2);
    }
};

// R = RA * RB in registers: n = RB n1 + n2, k = k1 + RA k2
template <int RA, int RB, bool INV>
struct DftComposite {
    static __device__ __forceinline__ void run(float2 *v) {
        constexpr int R = RA * RB;
        float2 t[RA * RB];  // t[k1 * RB + n2]
        static_for<0, RB>([&](auto n2c) {
            constexpr int n2 = decltype(n2c)::value;
            float2 c[RA];
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
            float2 c[RB];
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
