// prt_devmath.h -- transcendental functions shared by the HIP kernels and the host-side
// checks.  The reference calls libm's cosf/sinf (path_tracer.cpp:153,184) and powf
// (material.cpp:27); a GPU libm would round differently, so the kernels carry their own
// versions that reproduce glibc 2.35's results bit for bit on the argument range the
// path uses.  glibc's sinf/cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h:
// the ARM "optimized routines" single-step reduction + double-precision polynomial) are
// restated here from their published algorithm; tests/test_devmath.py checks every one of
// the 2^23 arguments theta = 2*pi*r1 the path can produce against the libm of the box.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PRT_HD __host__ __device__ __forceinline__
#else
#define PRT_HD static inline
#endif

// PRT_SINCOS_FMA selects fused multiply-adds in the double-precision polynomial, as
// glibc's x86-64 ifunc variant built with -mfma does.
#ifndef PRT_SINCOS_FMA
#define PRT_SINCOS_FMA 1
#endif

PRT_HD uint32_t prt_f2u(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}

PRT_HD float prt_u2f(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

PRT_HD double prt_mad(double a, double b, double c)
{
#if PRT_SINCOS_FMA
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}

// 4/pi in 24 overlapping 32-bit windows, 8 bits apart (glibc sincosf_data.c: __inv_pio4: window i holds bits [8 i - 24, 8 i + 8)
// of a2f9836e 4e441529 fc2757d1 f534ddc0 db629599 3c439041), for arguments of 120 and beyond.  A table in constant memory:
// computed in registers, or reached through a call, the rare path cost the shade pass 2-12 % of a frame in spills.
PRT_HD uint32_t prt_inv_pio4(uint32_t i)
{
    static const uint32_t tab[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                     0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
                                     0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    return tab[i];
}

// Reduction of |y| >= 120 by exact integer arithmetic on the bits of 4/pi (glibc sincosf.h: reduce_large): returns x in
// [-pi/4, pi/4] and the quadrant.
PRT_HD double prt_reduce_large(uint32_t xi, int* np)
{
    const uint32_t base = (xi >> 26) & 15u;
    const int shift = (int)((xi >> 23) & 7u);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * prt_inv_pio4(base)); // the low 32 bits of the product, as the 32-bit multiply gives them
    const uint64_t res1 = (uint64_t)xi * prt_inv_pio4(base + 4);
    const uint64_t res2 = (uint64_t)xi * prt_inv_pio4(base + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
}

// The two polynomials on the reduced argument x in [-pi/4, pi/4] (glibc sincosf.h: sincosf_poly): n picks which of them is the
// sine, q the sign of the sine's argument and whether the cosine's coefficients are negated.
PRT_HD void prt_sincosf_poly(double x, int n, int q, float* sinp, float* cosp)
{
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                 c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    // sign table {1,-1,-1,1}[q&3] for the sine argument; polynomial negated when q&2
    double sgn = ((q + 1) & 2) ? -1.0 : 1.0;
    double neg = (q & 2) ? -1.0 : 1.0;
    double x2 = x * x;
    double xs = x * sgn;
    // sine polynomial on xs (sine coefficients are not negated in the second table)
    double x3 = xs * x2;
    double sp1 = prt_mad(x2, s3, s2);
    double x7 = x3 * x2;
    double sv = prt_mad(x3, s1, xs);
    double sres = prt_mad(x7, sp1, sv);
    // cosine polynomial (coefficients negated when q & 2)
    double x4 = x2 * x2;
    double cp2 = prt_mad(x2, neg * c4, neg * c3);
    double cp1 = prt_mad(x2, neg * c1, neg * c0);
    double x6 = x4 * x2;
    double cv = prt_mad(x4, neg * c2, cp1);
    double cres = prt_mad(x6, cp2, cv);
    if (n & 1) {
        *sinp = (float)cres;
        *cosp = (float)sres;
    } else {
        *sinp = (float)sres;
        *cosp = (float)cres;
    }
}

// |y| >= 120, infinities and NaNs (glibc: reduce_large; the sign of y moves the sign and the table choice, not the sine /
// cosine choice).  The path's own angles never come here, an environment map's can (light.cpp:121-125).
PRT_HD void prt_sincosf_large(float y, float* sinp, float* cosp)
{
    const uint32_t bits = prt_f2u(y);
    if ((bits & 0x7f800000u) == 0x7f800000u) {
        *sinp = *cosp = y - y; // NaN (and the invalid exception in libm)
        return;
    }
    int n;
    const double x = prt_reduce_large(bits, &n);
    prt_sincosf_poly(x, n, n + (int)(bits >> 31), sinp, cosp);
}

// One shared reduction: |theta| in [pi/4, 120) -> (x in [-pi/4,pi/4], quadrant n) in one double-precision step; below pi/4
// n = 0 with no reduction.
PRT_HD void prt_sincosf(float y, float* sinp, float* cosp)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23; // 2/pi * 2^24
    const double hpi = 0x1.921FB54442D18p0;       // pi/2
    double x = (double)y;
    int n = 0;
    uint32_t top = (prt_f2u(y) >> 20) & 0x7ff;
    const uint32_t top_pio4 = (0x3f490fdbu >> 20) & 0x7ff;
    const uint32_t top_tiny = (0x39800000u >> 20) & 0x7ff; // 0x1p-12f
    if (top < top_pio4) {
        if (top < top_tiny) { // |y| < 2^-12: sin = y, cos = 1
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
    } else if (top < ((0x42f00000u >> 20) & 0x7ff)) { // |y| < 120
        double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = prt_mad(-(double)n, hpi, x);
    } else {
        prt_sincosf_large(y, sinp, cosp);
        return;
    }
    prt_sincosf_poly(x, n, n, sinp, cosp);
}

// powf(x, 2.2f) for x >= 0 as glibc 2.35 computes it (sysdeps/ieee754/flt-32/e_powf.c: log2 by a
// 16-entry table + degree-5 polynomial in double, exp2 by a 32-entry table + cubic), restated
// from the published algorithm.  material.cpp:24-28 (degamma) is the only caller on the path;
// its arguments are bilinear texel mixes in [0, 1].  Checked against the box's libm over every
// float in [2^-24, 1] by tests/test_devmath.py.
PRT_HD double prt_u2d(uint64_t u)
{
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}
PRT_HD uint64_t prt_d2u(double d)
{
    uint64_t u;
    __builtin_memcpy(&u, &d, 8);
    return u;
}

PRT_HD void prt_powf_log2_tab(int i, double* invc, double* logc)
{
    // c near the centre of the i-th sixteenth of [0x1.66p-1, 0x1.66p0): invc ~ 1/c, logc ~ log2(c)
    switch (i) {
    case 0: *invc = 0x1.661ec79f8f3bep+0; *logc = -0x1.efec65b963019p-2; break;
    case 1: *invc = 0x1.571ed4aaf883dp+0; *logc = -0x1.b0b6832d4fca4p-2; break;
    case 2: *invc = 0x1.49539f0f010bp+0; *logc = -0x1.7418b0a1fb77bp-2; break;
    case 3: *invc = 0x1.3c995b0b80385p+0; *logc = -0x1.39de91a6dcf7bp-2; break;
    case 4: *invc = 0x1.30d190c8864a5p+0; *logc = -0x1.01d9bf3f2b631p-2; break;
    case 5: *invc = 0x1.25e227b0b8eap+0; *logc = -0x1.97c1d1b3b7afp-3; break;
    case 6: *invc = 0x1.1bb4a4a1a343fp+0; *logc = -0x1.2f9e393af3c9fp-3; break;
    case 7: *invc = 0x1.12358f08ae5bap+0; *logc = -0x1.960cbbf788d5cp-4; break;
    case 8: *invc = 0x1.0953f419900a7p+0; *logc = -0x1.a6f9db6475fcep-5; break;
    case 9: *invc = 0x1p+0; *logc = 0x0p+0; break;
    case 10: *invc = 0x1.e608cfd9a47acp-1; *logc = 0x1.338ca9f24f53dp-4; break;
    case 11: *invc = 0x1.ca4b31f026aap-1; *logc = 0x1.476a9543891bap-3; break;
    case 12: *invc = 0x1.b2036576afce6p-1; *logc = 0x1.e840b4ac4e4d2p-3; break;
    case 13: *invc = 0x1.9c2d163a1aa2dp-1; *logc = 0x1.40645f0c6651cp-2; break;
    case 14: *invc = 0x1.886e6037841edp-1; *logc = 0x1.88e9c2c1b9ff8p-2; break;
    default: *invc = 0x1.767dcf5534862p-1; *logc = 0x1.ce0a44eb17bccp-2; break;
    }
}

PRT_HD float prt_powf_2p2(float x)
{
    uint32_t ix = prt_f2u(x);
    if (ix == 0x3f800000u) return 1.0f; // x == 1
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        // zero, subnormal, negative, inf or nan
        if ((ix << 1) == 0) return 0.0f;                 // pow(+-0, 2.2) = +0
        if (ix == 0x7f800000u) return x;                 // +inf
        if (ix > 0x7f800000u) return prt_u2f(0x7fc00000u); // negative or nan -> nan (never on the path)
        // subnormal: normalise as glibc does (x * 2^23, exponent - 23)
        ix = prt_f2u(x * 0x1p23f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    // log2_inline
    const uint32_t OFF = 0x3f330000u;
    uint32_t tmp = ix - OFF;
    int i = (int)((tmp >> (23 - 4)) % 16u);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc, logc;
    prt_powf_log2_tab(i, &invc, &logc);
    double z = (double)prt_u2f(iz);
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    double r = prt_mad(z, invc, -1.0);
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = prt_mad(A0, r, A1);
    double p = prt_mad(A2, r, A3);
    double r4 = r2 * r2;
    double q = prt_mad(A4, r, y0);
    q = prt_mad(p, r2, q);
    y = prt_mad(y, r4, q);
    double ylogx = (double)2.2f * y;
    // (overflow/underflow of the result cannot happen for x in [0,1], y = 2.2: 2.2*log2(x) > -330 only matters
    //  below 2^-57; such inputs underflow to the correctly signed 0/denormal through the scaling below)
    if (ylogx <= -150.0) return 0.0f;
    // exp2_inline, N = 32, no sign bias
    const double SHIFT = 0x1.8p+52 / 32.0;
    double kd = ylogx + SHIFT;
    uint64_t ki = prt_d2u(kd);
    kd -= SHIFT;
    double rr = ylogx - kd;
    // T[j] = bits(2^(j/32)) - (j << 47)
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
    };
    uint64_t t = T[ki % 32u];
    t += ki << (52 - 5);
    double s = prt_u2d(t);
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double zz = prt_mad(C0, rr, C1);
    double rr2 = rr * rr;
    double yy = prt_mad(C2, rr, 1.0);
    yy = prt_mad(zz, rr2, yy);
    yy = yy * s;
    return (float)yy;
}
