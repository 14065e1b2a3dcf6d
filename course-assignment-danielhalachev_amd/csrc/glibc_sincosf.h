// sinf / cosf for the GI mode's sample directions (reference: SourceCode/src/RayTracer.cpp:342-345, `std::cosf(angle1)`,
// `std::sinf(angle1)`, `std::cosf(angle2)`, `std::sinf(angle2)` with angle1 in [0, pi), angle2 in [0, 2 pi)).
//
// The reference calls libm, and glibc's sinf / cosf are not correctly rounded (documented error 0.56 ULP), so no closed
// form reproduces them: this header restates the routine.
//
// Third-party dependency restated here: GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11),
// sysdeps/ieee754/flt-32/s_sinf.c + s_cosf.c + sincosf.h + s_sincosf_data.c (the Arm Optimized Routines sinf / cosf: the
// argument goes to binary64, one multiply by 2^24 * 2/pi and an integer shift give the quadrant, one multiply-subtract by
// pi/2 the reduced argument, and a degree-7 (sine) or degree-8 (cosine) polynomial in binary64 the result, rounded to
// binary32 once).  Restriction: |y| < 120 (the "fast" reduction; the GI angles lie in [0, 2 pi]).  On x86-64 glibc dispatches
// these functions through an IFUNC to FMA builds on CPUs with FMA; over ALL 1,086,918,637 floats in [0, 2 pi] the sequence
// below gives the same bits with every a*b+c fused or not, and the same bits as this image's libm sinf, cosf and sincosf
// (the exhaustive run is recorded in DESIGN.md section 8; tests/test_gi.py repeats a sample of it on every run).
#pragma once

#include <stdint.h>
#include <string.h>

#ifndef CRT_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD static inline
#endif
#endif

// quadrant polynomials: sine (n even) in x, x^3 .. x^7; cosine (n odd) in x^2 .. x^8, negated in quadrants 2 and 3
CRT_HD float crt_sincos_poly(double x, double x2, bool negated, int n) {
    const double sgn = negated ? -1.0 : 1.0;
    if ((n & 1) == 0) {
        const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
        const double x3 = x * x2;
        const double s1 = s2c + x2 * s3c;
        const double x7 = x3 * x2;
        const double s = x + x3 * s1c;
        return (float)(s + x7 * s1);
    }
    const double c0 = sgn * 0x1p0, c1c = sgn * -0x1.ffffffd0c621cp-2, c2c = sgn * 0x1.55553e1068f19p-5, c3c = sgn * -0x1.6c087e89a359dp-10,
                 c4c = sgn * 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double c2 = c3c + x2 * c4c;
    const double c1 = c0 + x2 * c1c;
    const double x6 = x4 * x2;
    const double c = c1 + x4 * c2c;
    return (float)(c + x6 * c2);
}

CRT_HD uint32_t crt_abstop12(float x) {
    uint32_t u;
    memcpy(&u, &x, sizeof(u));
    return (u >> 20) & 0x7ffu;
}

// want_cos = false: sinf(y); true: cosf(y)
CRT_HD float crt_sincosf(float y, bool want_cos) {
    double x = (double)y;
    if (crt_abstop12(y) < crt_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
        if (crt_abstop12(y) < crt_abstop12(0x1p-12f)) return want_cos ? 1.0f : y;
        return crt_sincos_poly(x, x * x, false, want_cos ? 1 : 0);
    }
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;  // quadrant: round(x * 2/pi)
    x = x - (double)n * hpi;
    const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // sign[4] = {1, -1, -1, 1}
    return crt_sincos_poly(x * sg, x * x, (n & 2) != 0, want_cos ? (n ^ 1) : n);
}
CRT_HD float crt_sinf(float y) { return crt_sincosf(y, false); }
CRT_HD float crt_cosf(float y) { return crt_sincosf(y, true); }
