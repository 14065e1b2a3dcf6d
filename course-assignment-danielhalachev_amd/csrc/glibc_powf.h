// pow(x, 5) for the Fresnel term of RayTracer::calculateRefraction (reference:
// SourceCode/src/RayTracer.cpp:407, `std::powf(1.0f - cosineAlpha, 5)`).
//
// The reference calls glibc's powf (glibc 2.35 in this image).  `std::powf(x, 2)` on the line
// above it (:406) is folded to x*x by both gcc and clang at -O2, so only the fifth power reaches
// libm.  This header restates that libm routine for the device; see DESIGN.md "powf".
#pragma once

#ifndef CRT_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD static inline
#endif
#endif

// First cut: fifth power evaluated in binary64 and rounded once to binary32.  x^5 of a binary32
// value carries 120 significant bits, so the three binary64 products below have a relative error
// below 3*2^-53 before the final rounding -- this is the correctly rounded result except when
// x^5 lies within ~2^-29 ulp of a rounding boundary.
CRT_HD float crt_pow5(float x) {
    double d = (double)x;
    double d2 = d * d;
    double d4 = d2 * d2;
    return (float)(d4 * d);
}
