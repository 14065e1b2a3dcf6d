// The GI / multi-sample mode's random numbers (reference: SourceCode/src/RayTracer.cpp:28-30, 62-63, 340-341).
//
// The reference draws from a thread_local std::default_random_engine seeded with clock() ^ hash(thread id): no two of its runs
// agree and the draw order depends on which thread renders which bucket, so only the DISTRIBUTION of its images is defined.
// This path replaces that stream by a counter-based one that depends on neither evaluation order nor thread count -- every
// shootRay invocation has a 32-bit key that says where it sits in its pixel's ray tree:
//     primary ray of sample s of pixel p (p = row * width + col):   key = mix(mix(seed, p), s)         (seed: crt_options::gi_seed)
//     child c of the invocation with key K:                          key = mix(K, 0x40000000 + c)
//                                                                    c = 0 reflection ray, 1 transmission ray, 2 + i GI sample i
// and the d-th uniform number an invocation draws is u(K, d) = (mix(K, 0x80000000 + d) >> 8) * 2^-24, uniform on [0, 1) with
// 24 bits like std::uniform_real_distribution<float>(0, 1): d = 0, 1 the pixel jitter of a primary ray (offsetX, offsetY),
// d = 2 + 2i, 3 + 2i the two angles of GI sample i.  mix = two rounds of the MurmurHash3 finaliser.
// oracle/cpu_ref.c states the same rules independently; the GPU frame equals the oracle's bit for bit for a given seed, and
// both match the reference's image distribution (tests/test_gi.py, tests/golden/gi_stats.npz).
#pragma once

#include <stdint.h>

#ifndef CRT_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD static inline
#endif
#endif

CRT_HD uint32_t crt_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
CRT_HD uint32_t crt_gi_mix(uint32_t a, uint32_t b) { return crt_fmix32(crt_fmix32(a ^ 0x9E3779B9u) + (b ^ 0x7F4A7C15u) * 0x9E3779B1u); }
CRT_HD float crt_gi_uniform(uint32_t key, uint32_t d) { return (float)(crt_gi_mix(key, 0x80000000u + d) >> 8) * 0x1p-24f; }
CRT_HD uint32_t crt_gi_child_key(uint32_t key, uint32_t c) { return crt_gi_mix(key, 0x40000000u + c); }
