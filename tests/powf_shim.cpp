// Test shim: exposes the product's restatement of glibc powf(x, 5) (csrc/glibc_powf.h) and libm's own powf
// to tests/test_powf.py, both evaluated over whole arrays.
#include <cmath>
#include <cstddef>
#include "../course-assignment-danielhalachev_amd/csrc/glibc_powf.h"

extern "C" void shim_pow5(const float *x, float *out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = crt_pow5(x[i]);
}
extern "C" void libm_pow5(const float *x, float *out, size_t n) {
  volatile float five = 5.0f;  // keep the call to libm
  for (size_t i = 0; i < n; i++) out[i] = powf(x[i], five);
}
