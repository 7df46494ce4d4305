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

// One shared reduction: theta in [pi/4, 120) -> (x in [-pi/4,pi/4], quadrant n).
// theta below pi/4 uses n = 0 with no reduction.
PRT_HD void prt_sincosf(float y, float* sinp, float* cosp)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23; // 2/pi * 2^24
    const double hpi = 0x1.921FB54442D18p0;       // pi/2
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                 c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;

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
    } else {
        double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = prt_mad(-(double)n, hpi, x);
    }
    // sign table {1,-1,-1,1}[n&3] for the sine argument; polynomial negated when n&2
    double sgn = ((n + 1) & 2) ? -1.0 : 1.0;
    double neg = (n & 2) ? -1.0 : 1.0;
    double x2 = x * x;
    double xs = x * sgn;
    // sine polynomial on xs (sine coefficients are not negated in the second table)
    double x3 = xs * x2;
    double sp1 = prt_mad(x2, s3, s2);
    double x7 = x3 * x2;
    double sv = prt_mad(x3, s1, xs);
    double sres = prt_mad(x7, sp1, sv);
    // cosine polynomial (coefficients negated when n & 2)
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
