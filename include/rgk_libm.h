/*
 * rgk_libm.h -- the five transcendental functions on the path-tracing hot path, DEFINED here.
 *
 * The reference calls the C library (std::sin / cos on the disc and sphere samples, src/random_utils.hpp:12-60;
 * glm::angle = acos, src/LTC/ltc.cpp:62,122; asin / atan2 of the envmap lookup, src/scene.cpp:752-753), so its results
 * depend on the libm of the machine that runs it -- glibc's and ROCm ocml's sinf / cosf / acosf differ by 1-2 ulp, and a
 * 1-ulp direction now and then flips a discrete decision of a path (which triangle, which side of an edge): measured,
 * ~5e-6 of all paths, the whole per-pixel difference between the CPU oracle and the HIP path in round 1.
 *
 * This build pins them: every function below is a fixed sequence of IEEE-754 double operations (+, -, *, /, sqrt, compare;
 * no fused multiply-add, no table, no libm call), so the oracle (g++, -ffp-contract=off) and the device code (hipcc
 * gfx950, -ffp-contract=off) produce the SAME BITS for the same input, on any machine.  Each result is the double value
 * rounded once to float: error <= 0.5 ulp + 1e-9 relative (tests/test_oracle_cpu.py measures them against libm).
 *
 * Shared by oracle/ and rgk_amd/csrc (like rgk.h): a definition both sides include, not code of either.
 */
#ifndef RGK_LIBM_H
#define RGK_LIBM_H

#if defined(__HIPCC__)
#define RGK_HD __host__ __device__ __forceinline__
#else
#define RGK_HD inline
#endif

#define RGK_M_PI 3.14159265358979323846
#define RGK_M_PI_2 1.57079632679489661923

/* sqrt of a double in [0, 1] to ~1e-15 relative: the correctly rounded FLOAT square root (IEEE on both sides) refined by one
 * Newton step in double -- additions, multiplications and one division only, so no library or hardware sqrt-f64 sequence
 * is involved.  (Not only for bit parity: with __builtin_sqrt(double) inlined into the shading kernel, hipcc 7.2's gfx950 code
 * gave results that changed from run to run on LTC materials -- tools/gpu_zoo_debug.py, DESIGN.md "Numerics".) */
#if defined(__HIPCC__)
RGK_HD float rgk_sqrtf_ieee(float x) { return __builtin_sqrtf(x); }
#else
#include <cmath>
RGK_HD float rgk_sqrtf_ieee(float x) { return std::sqrt(x); }
#endif
RGK_HD double rgk_sqrt(double x) {
#if defined(RGK_LIBM_BUILTIN_SQRT) /* diagnosis only (tools/gpu_zoo_debug.py): the form that gave run-to-run differences in round 2 */
    return x > 0.0 ? __builtin_sqrt(x) : 0.0;
#endif
    if (!(x > 0.0)) return 0.0;
    const double s0 = (double)rgk_sqrtf_ieee((float)x);
    if (!(s0 > 0.0)) return 0.0; /* x below the float range: the path never gets there (x = (1 - |cos|) / 2 of float cosines) */
    return 0.5 * (s0 + x / s0);
}

/* sin and cos of a float argument.  n = nearest integer to x * 2/pi (round-half-even through the 1.5 * 2^52 constant),
 * r = x - n * pi/2 in two steps (pi/2 = hi + lo), Taylor polynomials of degree 11 / 12 on |r| <= pi/4 (truncation
 * < 2e-11), quadrant from n.  Exact for the arguments the path makes (|x| <= 2 pi); defined, if less accurate, up to
 * |x| ~ 1e9 (beyond: NaN in, NaN out; huge finite values lose the reduction's guard bits but stay deterministic). */
typedef struct rgk_sincos { float s, c; } rgk_sincos;
/* (by value: through two pointers the device compiler merges the bodies of two call sites into one block that stores through a
 * SELECTED pointer, which keeps the four results in scratch memory -- 24 bytes and 17 scratch instructions in every shading kernel) */
RGK_HD rgk_sincos rgk_sincosf_v(float x) {
    const double xd = (double)x;
    const double q = xd * 0.63661977236758134308; /* 2/pi */
    const double big = 6755399441055744.0;        /* 1.5 * 2^52 */
    const double nd = (q + big) - big;
    const double r1 = xd - nd * 1.57079632673412561417e+00;  /* pi/2, leading 33 bits */
    const double r = r1 - nd * 6.07710050650619224932e-11;   /* pi/2 - the above */
    const double z = r * r;
    const double ps = r + r * z * (-1.66666666666666666667e-01 + z * (8.33333333333333333333e-03 + z * (-1.98412698412698412698e-04 +
                      z * (2.75573192239858906526e-06 + z * (-2.50521083854417187751e-08)))));
    const double pc = 1.0 + z * (-0.5 + z * (4.16666666666666666667e-02 + z * (-1.38888888888888888889e-03 + z * (2.48015873015873015873e-05 +
                      z * (-2.75573192239858906526e-07 + z * (2.08767569878680989792e-09))))));
    const long long n = (long long)nd;
    const int k = (int)(n & 3);
    const double sv = (k == 0) ? ps : (k == 1) ? pc : (k == 2) ? -ps : -pc;
    const double cv = (k == 0) ? pc : (k == 1) ? -ps : (k == 2) ? -pc : ps;
    rgk_sincos r_;
    r_.s = (float)sv;
    r_.c = (float)cv;
    return r_;
}
RGK_HD void rgk_sincosf(float x, float* s, float* c) { const rgk_sincos r_ = rgk_sincosf_v(x); *s = r_.s; *c = r_.c; }
RGK_HD float rgk_sinf(float x) { return rgk_sincosf_v(x).s; }
RGK_HD float rgk_cosf(float x) { return rgk_sincosf_v(x).c; }

/* asin(t) for 0 <= t <= 0.5 in double: t + t^3 * P(t^2), P = the Taylor series of (asin(t) - t) / t^3 to 14 terms
 * (the 15th is < 3e-11 at t = 0.5). */
RGK_HD double rgk_asin_core(double t) {
    const double z = t * t;
    const double p = 1.66666666666666666667e-01 + z * (7.50000000000000000000e-02 + z * (4.46428571428571428571e-02 + z * (3.03819444444444444444e-02 +
                     z * (2.23721590909090909091e-02 + z * (1.73527644230769230769e-02 + z * (1.39648437500000000000e-02 + z * (1.15518008961397058824e-02 +
                     z * (9.76160952919407894737e-03 + z * (8.39033580961681547619e-03 + z * (7.31252587359884510870e-03 + z * (6.44721031188964843750e-03 +
                     z * (5.74003767084192346644e-03 + z * (5.15330968231990419585e-03)))))))))))));
    return t + t * z * p;
}
/* acos of a float in [-1, 1] (NaN outside, as libm).  |x| <= 0.5: pi/2 - asin(x); x > 0.5: 2 asin(sqrt((1 - x) / 2));
 * x < -0.5: pi - 2 asin(sqrt((1 + x) / 2)). */
RGK_HD float rgk_acosf(float x) {
    const double xd = (double)x;
    if (!(xd >= -1.0 && xd <= 1.0)) return (float)((xd - xd) / (xd - xd)); /* NaN */
    if (xd > 0.5) return (float)(2.0 * rgk_asin_core(rgk_sqrt((1.0 - xd) * 0.5)));
    if (xd < -0.5) return (float)(RGK_M_PI - 2.0 * rgk_asin_core(rgk_sqrt((1.0 + xd) * 0.5)));
    return (float)(RGK_M_PI_2 - (xd >= 0.0 ? rgk_asin_core(xd) : -rgk_asin_core(-xd)));
}
RGK_HD float rgk_asinf(float x) {
    const double xd = (double)x;
    if (!(xd >= -1.0 && xd <= 1.0)) return (float)((xd - xd) / (xd - xd)); /* NaN */
    const double a = xd >= 0.0 ? xd : -xd;
    const double v = a <= 0.5 ? rgk_asin_core(a) : RGK_M_PI_2 - 2.0 * rgk_asin_core(rgk_sqrt((1.0 - a) * 0.5));
    return (float)(xd >= 0.0 ? v : -v);
}
/* atan(u) for |u| <= tan(pi/8) in double: alternating Taylor series to u^27 (next term < 2e-12). */
RGK_HD double rgk_atan_core(double u) {
    const double z = u * u;
    const double p = 1.0 + z * (-3.33333333333333333333e-01 + z * (2.00000000000000000000e-01 + z * (-1.42857142857142857143e-01 + z * (1.11111111111111111111e-01 +
                     z * (-9.09090909090909090909e-02 + z * (7.69230769230769230769e-02 + z * (-6.66666666666666666667e-02 + z * (5.88235294117647058824e-02 +
                     z * (-5.26315789473684210526e-02 + z * (4.76190476190476190476e-02 + z * (-4.34782608695652173913e-02 + z * (4.00000000000000000000e-02 +
                     z * (-3.70370370370370370370e-02)))))))))))));
    return u * p;
}
/* atan2(y, x) of floats.  t = min(|y|, |x|) / max(|y|, |x|) in [0, 1]; atan(t) = atan_core(t) below tan(pi/8), else
 * pi/4 + atan_core((t - 1) / (t + 1)); then the octant and the signs.  atan2(+-0, x >= 0) = +-0, atan2(+-0, x < 0) = +-pi,
 * atan2(0, 0) = 0 (the path only calls it on unit directions). */
RGK_HD float rgk_atan2f(float y, float x) {
    const double yd = (double)y, xd = (double)x;
    if (yd != yd || xd != xd) return (float)(yd + xd);
    const double ay = yd >= 0.0 ? yd : -yd, ax = xd >= 0.0 ? xd : -xd;
    double a;
    if (ay == 0.0 && ax == 0.0) a = 0.0;
    else {
        const double hi = ay > ax ? ay : ax, lo = ay > ax ? ax : ay;
        const double t = lo / hi;
        a = t <= 0.41421356237309504880 ? rgk_atan_core(t) : 0.78539816339744830962 + rgk_atan_core((t - 1.0) / (t + 1.0));
        if (ay > ax) a = RGK_M_PI_2 - a;
    }
    if (xd < 0.0) a = RGK_M_PI - a;
    return (float)(yd < 0.0 || (yd == 0.0 && 1.0 / yd < 0.0) ? -a : a);
}

#endif /* RGK_LIBM_H */
