/* hrt_libm.h -- the four float libm functions the reference's shading calls (sinf, cosf, expf,
 * acosf; src/compute_paths.c:310,325,372-374,379,395-396,694), restated so that the DEVICE
 * returns the same bits as the host libm the reference links against.
 *
 * Third-party dependency: GNU libc 2.35 (Ubuntu 22.04, x86-64), not vendored by the
 * reference.  Restated from its published algorithms:
 *   sinf/cosf  sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c
 *              (Arm Optimized Routines: double-precision polynomials, reduce_fast for
 *              |x| < 120)
 *   expf       sysdeps/ieee754/flt-32/e_expf.c, math/e_exp2f_data.c (32-entry 2^(i/32) table)
 *   acosf      sysdeps/ieee754/flt-32/e_acosf.c (fdlibm rational approximation, float)
 * On FMA-capable x86-64 CPUs glibc's ifunc selects the builds of sinf/cosf/expf compiled with
 * -mfma (sysdeps/x86_64/fpu/multiarch), in which gcc contracted every a*b+c of those
 * sources; acosf has no such variant.  The fma() calls below reproduce exactly that
 * contraction pattern, and everything else is contraction-free.
 *
 * Pinned, not assumed: oracle/libm_probe.c compiles this very header for the host and
 * compares each function with the host libm over EVERY float in the domain the tracer can
 * produce -- sinf/cosf |x| < 120 (2.2e9 values each), expf |x| < 88, acosf
 * |x| <= 1 -- 0 mismatches on glibc 2.35 (record in DESIGN.md).  Outside those domains (never
 * reached by the tracer: angles are in [0, pi], the exponent in [-4 pi, 0]) the functions
 * fall back to the double-precision routine rounded to float.
 *
 * Usable from HIP device code and from host C/C++ (the probe).
 */
#ifndef HRT_LIBM_H
#define HRT_LIBM_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define HRT_HD __host__ __device__ __forceinline__
#else
#define HRT_HD static inline
#endif

HRT_HD uint32_t hrt_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
HRT_HD float hrt_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
HRT_HD uint64_t hrt_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
HRT_HD double hrt_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
HRT_HD uint32_t hrt_abstop12(float x) { return (hrt_f2u(x) >> 20) & 0x7ffu; }

/* polynomial of sin (quadrant even) or cos (odd) on [-pi/4, pi/4]; `neg` selects the
 * negated cosine coefficients used in quadrants 2 and 3 */
HRT_HD float hrt_sincos_poly(double x, double x2, int n, int neg)
{
    if ((n & 1) == 0) {
        const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7,
                     s3c = -0x1.994eb3774cf24p-13;
        double x3 = x * x2;
        double s1 = fma(x2, s3c, s2c);
        double x7 = x3 * x2;
        double s = fma(x3, s1c, x);
        return (float)fma(x7, s1, s);
    } else {
        double c0 = 0x1p0, c1c = -0x1.ffffffd0c621cp-2, c2c = 0x1.55553e1068f19p-5,
               c3c = -0x1.6c087e89a359dp-10, c4c = 0x1.99343027bf8c3p-16;
        if (neg) { c0 = -c0; c1c = -c1c; c2c = -c2c; c3c = -c3c; c4c = -c4c; }
        double x4 = x2 * x2;
        double c2 = fma(x2, c4c, c3c);
        double c1 = fma(x2, c1c, c0);
        double x6 = x4 * x2;
        double c = fma(x4, c2c, c1);
        return (float)fma(x6, c2, c);
    }
}

/* x mod pi/2 into [-pi/4, pi/4] and the quadrant, for |x| < 120 (2/pi prescaled by 2^24) */
HRT_HD double hrt_reduce_fast(double x, int *np)
{
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, 0x1.921FB54442D18p0, x);
}

HRT_HD float hrt_sinf(float y)
{
    double x = y;
    if (hrt_abstop12(y) < hrt_abstop12(0x1.921FB6p-1f)) {
        if (hrt_abstop12(y) < hrt_abstop12(0x1p-12f)) return y;
        return hrt_sincos_poly(x, x * x, 0, 0);
    }
    if (hrt_abstop12(y) < hrt_abstop12(120.0f)) {
        int n;
        x = hrt_reduce_fast(x, &n);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return hrt_sincos_poly(x * sgn, x * x, n, n & 2);
    }
    return (float)sin(x);
}

HRT_HD float hrt_cosf(float y)
{
    double x = y;
    if (hrt_abstop12(y) < hrt_abstop12(0x1.921FB6p-1f)) {
        if (hrt_abstop12(y) < hrt_abstop12(0x1p-12f)) return 1.0f;
        return hrt_sincos_poly(x, x * x, 1, 0);
    }
    if (hrt_abstop12(y) < hrt_abstop12(120.0f)) {
        int n;
        x = hrt_reduce_fast(x, &n);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return hrt_sincos_poly(x * sgn, x * x, n ^ 1, n & 2);
    }
    return (float)cos(x);
}

/* sinf AND cosf of one argument, and cosf alone, through ONE branch-free path for |y| < 120.
 * glibc's small-argument branches are special cases of its reduction branch: for |y| < 0.75
 * reduce_fast gives n = 0 and x unchanged (fma(-0, pi/2, x) == x), so the same polynomial sees the
 * same operands; the sine polynomial is odd and the negated-coefficient cosine polynomial is the
 * negation, bit for bit, so each polynomial is evaluated ONCE on (x, x^2) and the quadrant only
 * picks which one and its sign.  On a wavefront that removes the divergence between lanes in
 * different branches/quadrants (every lane used to pay for all of them) and shares the
 * reduction: 18 double operations for the pair instead of up to 46.  |y| < 2^-12 returns (y, 1)
 * like glibc (only -0 differs otherwise).  Pinned by oracle/libm_probe.c like the others. */
HRT_HD void hrt_sincos_core(float y, float *ps, float *pc, int *pn)
{
    int n;
    const double x = hrt_reduce_fast((double)y, &n);
    const double x2 = x * x;
    *ps = hrt_sincos_poly(x, x2, 0, 0);
    *pc = hrt_sincos_poly(x, x2, 1, 0);
    *pn = n;
}
HRT_HD float hrt_negate_if(float v, int cond) { return hrt_u2f(hrt_f2u(v) ^ (cond ? 0x80000000u : 0u)); }

HRT_HD void hrt_sincosf(float y, float *sp, float *cp)
{
    if (!(hrt_abstop12(y) < hrt_abstop12(120.0f))) {
        *sp = hrt_sinf(y);
        *cp = hrt_cosf(y);
        return;
    }
    float ps, pc;
    int n;
    hrt_sincos_core(y, &ps, &pc, &n);
    float s = (n & 1) ? hrt_negate_if(pc, n & 2) : hrt_negate_if(ps, n & 2);
    float c = (n & 1) ? hrt_negate_if(ps, (n & 3) == 1) : hrt_negate_if(pc, n & 2);
    if (hrt_abstop12(y) < hrt_abstop12(0x1p-12f)) { s = y; c = 1.0f; }
    *sp = s;
    *cp = c;
}

HRT_HD float hrt_cosf_nb(float y)
{
    if (!(hrt_abstop12(y) < hrt_abstop12(120.0f))) return hrt_cosf(y);
    float ps, pc;
    int n;
    hrt_sincos_core(y, &ps, &pc, &n);
    const float c = (n & 1) ? hrt_negate_if(ps, (n & 3) == 1) : hrt_negate_if(pc, n & 2);
    return (hrt_abstop12(y) < hrt_abstop12(0x1p-12f)) ? 1.0f : c;
}

/* bits of 2^(i/32) minus (i << 47): so that adding (k << 47) with k = 32*e + i yields the
 * bits of 2^(k/32) */
HRT_HD uint64_t hrt_exp2_tab(uint32_t i)
{
    const uint64_t t[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    return t[i];
}

HRT_HD float hrt_expf(float xf)
{
    const uint32_t at = hrt_abstop12(xf);
    /* |x| >= 88, inf, NaN: glibc's special-case branch; not reachable from the tracer */
    if (at >= hrt_abstop12(88.0f)) return (float)exp((double)xf);
    const double N = 32.0;
    const double inv_ln2_n = 0x1.71547652b82fep+0 * N, shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / N / N / N, c1 = 0x1.ebfce50fac4f3p-3 / N / N,
                 c2 = 0x1.62e42ff0c52d6p-1 / N;
    const double xd = xf;
    double kd = fma(inv_ln2_n, xd, shift);
    const uint64_t ki = hrt_d2u(kd);
    kd -= shift;
    const double r = fma(inv_ln2_n, xd, -kd);
    const uint64_t t = hrt_exp2_tab((uint32_t)(ki & 31u)) + (ki << 47);
    const double s = hrt_u2d(t);
    const double z = fma(c0, r, c1);
    const double r2 = r * r;
    double y = fma(c2, r, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (float)y;
}

HRT_HD float hrt_acosf(float x)
{
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f,
                pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f,
                pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
                qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = (int32_t)hrt_f2u(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;
    const float s = sqrtf(z);
    const float df = hrt_u2f(hrt_f2u(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

#endif /* HRT_LIBM_H */
