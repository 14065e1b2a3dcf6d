// crt_testhooks.hip -- unit-test hooks: in libcrt_hip_test.so only (the tests load that flavour; crt_main, crt_animation and bench.py the plain one).
#include "crt_internal.h"
#include "crt_multi.h"
#include "glibc_powf.h"
#include "glibc_sincosf.h"
#include "gi_random.h"

namespace {
// unit-test kernel: the device build of csrc/glibc_powf.h
__global__ void pow5_kernel(const float *x, float *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = crt_pow5(x[i]);
}

}  // namespace

// Unit-test hook: evaluates the device build of the restated glibc powf(x, 5) (csrc/glibc_powf.h) on host arrays.
extern "C" int crt_test_pow5(int device, const float *x, float *out, uint64_t n) {
    if (!x || !out) return CRT_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CRT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CRT_ERR_HIP;
    float *dx = nullptr, *dout = nullptr;
    if (hipMalloc((void **)&dx, n * sizeof(float)) != hipSuccess) return CRT_ERR_NOMEM;
    if (hipMalloc((void **)&dout, n * sizeof(float)) != hipSuccess) { (void)hipFree(dx); return CRT_ERR_NOMEM; }
    int rc = CRT_OK;
    if (hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = CRT_ERR_HIP;
    if (rc == CRT_OK && n) {
        hipLaunchKernelGGL(pow5_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, dx, dout, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_ERR_HIP;
    }
    (void)hipFree(dx);
    (void)hipFree(dout);
    return rc;
}

__host__ __device__ static inline uint32_t gi_test_value(uint32_t what, uint32_t a, uint32_t b) {
    float f;
    memcpy(&f, &a, sizeof(f));
    float r = 0;
    switch (what) {
        case 0: r = crt_sinf(f); break;
        case 1: r = crt_cosf(f); break;
        case 2: r = crt_gi_uniform(a, b); break;
        default: return crt_gi_mix(a, b);
    }
    uint32_t u;
    memcpy(&u, &r, sizeof(u));
    return u;
}
__global__ void gi_test_kernel(uint32_t what, const uint32_t *a, const uint32_t *b, uint32_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gi_test_value(what, a[i], b ? b[i] : 0u);
}
// Unit-test hook: the GI mode's sinf / cosf / generator (csrc/glibc_sincosf.h, csrc/gi_random.h) on host arrays.
extern "C" int crt_test_gi(int device, uint32_t what, const uint32_t *a, const uint32_t *b, uint32_t *out, uint64_t n) {
    if (!a || !out || what > 3u || (what >= 2u && !b)) return CRT_ERR_INVALID;
    if (device < 0) {
        for (uint64_t i = 0; i < n; i++) out[i] = gi_test_value(what, a[i], b ? b[i] : 0u);
        return CRT_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return CRT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CRT_ERR_HIP;
    uint32_t *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = CRT_OK;
    if (hipMalloc((void **)&da, (n ? n : 1) * 4) != hipSuccess || hipMalloc((void **)&db, (n ? n : 1) * 4) != hipSuccess ||
        hipMalloc((void **)&dout, (n ? n : 1) * 4) != hipSuccess) rc = CRT_ERR_NOMEM;
    if (rc == CRT_OK && n) {
        if (hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice) != hipSuccess || (b && hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice) != hipSuccess)) rc = CRT_ERR_HIP;
        if (rc == CRT_OK) {
            hipLaunchKernelGGL(gi_test_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, what, da, b ? db : nullptr, dout, n);
            if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_ERR_HIP;
        }
    }
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return rc;
}

extern "C" int crt_debug_multi_force_staged(crt_multi *M, int on) {
    if (!M) return CRT_ERR_INVALID;
    M->force_staged = on != 0;
    M->cached_rects.clear();  // the next render allocates the pinned buffers
    return CRT_OK;
}

extern "C" int crt_debug_multi_fail_next_alloc(crt_multi *M) {
    if (!M) return CRT_ERR_INVALID;
    M->fail_next_alloc = true;
    return CRT_OK;
}
