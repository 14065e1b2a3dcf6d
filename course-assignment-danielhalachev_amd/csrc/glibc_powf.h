// pow(x, 5) for the Fresnel term of RayTracer::calculateRefraction
// (reference: SourceCode/src/RayTracer.cpp:407, `std::powf(1.0f - cosineAlpha, 5)`).
//
// The reference calls libm.  `std::powf(q, 2)` on the line above it (:406) is folded to q*q by both
// gcc and clang at -O2 (checked in the reference object code: exactly one call to powf remains), so only
// the fifth power reaches the C library -- glibc 2.35 in this image -- whose powf is NOT correctly
// rounded (documented error 0.82 ULP), so no closed form reproduces it: this header restates the routine.
//
// Third-party dependency restated here: GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11),
// sysdeps/ieee754/flt-32/e_powf.c + e_powf_log2_data.c + e_exp2f_data.c (the Arm Optimized Routines powf:
// log2 via a 16-entry table and a degree-5 polynomial, exp2 via a 32-entry table and a cubic, all in
// binary64).  On x86-64 glibc dispatches powf through an IFUNC to `__powf_fma` on every CPU with FMA and
// AVX2 (sysdeps/x86_64/fpu/multiarch/e_powf.c) -- which is every host this project runs on -- and in that
// variant the compiler contracted each `a*b + c` of the source into one fused multiply-add.  The operation
// sequence below (which products are fused, operand order of the additions) and the table values were read
// from that build's machine code / .rodata, and are checked on the CPU against libm's powf itself over
// millions of arguments by tests/test_powf.py before the same header is compiled for the GPU.
// Restriction: exponent fixed to 5 (an odd positive integer), round-to-nearest.
#pragma once

#include <stdint.h>

#ifndef CRT_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD static inline
#endif
#endif

namespace crt_powf_detail {

struct LogEntry { double invc, logc; };

CRT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
CRT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
CRT_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
CRT_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// log2(x) for a normalised positive binary32 bit pattern (glibc e_powf.c log2_inline, FMA build)
CRT_HD double log2_inline(uint32_t ix) {
    // __powf_log2_data.tab: {1/c, log2(c)} for 16 subintervals of [0x1.66p-1, 0x1.66p0)
    const LogEntry T[16] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
        {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
        {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
        {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
        {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
    };
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;  // __powf_log2_data.poly
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)top >> 23;
    const double z = (double)u2f(iz);
    const double r = __builtin_fma(z, T[i].invc, -1.0);
    const double y0 = (double)k + T[i].logc;
    const double r2 = r * r;
    double y = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    return y;
}

// 2^xd as binary32 (glibc e_powf.c exp2_inline, FMA build, TOINT_INTRINSICS == 0)
CRT_HD float exp2_inline(double xd, uint64_t sign_bias) {
    const uint64_t T[32] = {  // __exp2f_data.tab: bits of 2^(i/32) with i << 47 subtracted
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
    };
    const double SHIFT = 0x1.8p+47;  // __exp2f_data.shift_scaled = 0x1.8p52 / 32
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;  // __exp2f_data.poly
    double kd = xd + SHIFT;
    const uint64_t ki = d2u(kd);
    kd -= SHIFT;
    const double r = xd - kd;
    uint64_t t = T[ki & 31u];
    t += (ki + sign_bias) << 47;
    const double s = u2d(t);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

}  // namespace crt_powf_detail

// glibc 2.35 powf(x, 5.0f), x86-64 FMA variant, round-to-nearest.
CRT_HD float crt_pow5(float x) {
    using namespace crt_powf_detail;
    uint32_t ix = f2u(x);
    uint64_t sign_bias = 0;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        // x < 0x1p-126, or inf, or nan (y = 5 is none of 0 / inf / nan)
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {  // zeroinfnan(ix)
            float x2 = x * x;
            if (ix & 0x80000000u) x2 = -x2;  // checkint(5) == 1: odd
            return x2;                        // y > 0
        }
        if (ix & 0x80000000u) {  // finite x < 0, odd integer exponent: the result is negative
            sign_bias = 1ull << (5 + 11);  // SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11)
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {  // subnormal: normalise so that the exponent becomes negative
            ix = f2u(u2f(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double logx = log2_inline(ix);
    const double ylogx = 5.0 * logx;  // cannot overflow
    if (((d2u(ylogx) >> 47) & 0xffffu) >= (d2u(126.0) >> 47)) {
        // |y * log2(x)| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();  // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                                      // __math_uflowf
        if (ylogx < -149.0) return u2f(sign_bias ? 0x80000001u : 0x00000001u);  // __math_may_uflowf: +-0x1.9p-150 -> +-0x1p-149
    }
    return exp2_inline(ylogx, sign_bias);
}
