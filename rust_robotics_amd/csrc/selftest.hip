// selftest.hip -- evaluates the arithmetic contract (include/rr_detmath.h) on the device so
// tests can assert that gfx950 and the host produce identical bits (rr_selftest_math).
#include <hip/hip_runtime.h>

#include <string>

#include "rr_common.hpp"
#include "rr_pf.h"
#include "rr_pf_spec.h"

namespace {
__global__ void k_selftest(int fn, size_t n, const double* __restrict__ a, const double* __restrict__ b,
                           double* __restrict__ o0, double* __restrict__ o1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r0 = 0.0, r1 = 0.0;
  switch (fn) {
    case 0: r0 = rr_exp(a[i]); break;
    case 1: r0 = rr_log(a[i]); break;
    case 2: rr_sincos(a[i], &r0, &r1); break;
    case 3: rr_sincos2pi(a[i], &r0, &r1); break;
    case 4: r0 = rr_atan2(a[i], b[i]); break;
    case 5: r0 = rr_sqrt(a[i]); break;
    case 6: r0 = a[i] / b[i]; break;
    case 7: rr_normal2(rr_d2u(a[0]), RR_STREAM_MOTION, (uint32_t)rr_d2u(b[0]), i, &r0, &r1); break;
    case 8: r0 = rr_fma(a[i], b[i], a[i]); break;
    case 9:  // the bare square-root core of the fused likelihood / Box-Muller radius (device-only function)
#if defined(__HIP_DEVICE_COMPILE__)
      r0 = rr_sqrt_core(a[i]);
#endif
      break;
    default: break;
  }
  o0[i] = r0;
  if (o1) o1[i] = r1;
}
}  // namespace

extern "C" rr_status rr_selftest_math(int32_t device, int32_t fn, size_t n, const double* a, const double* b,
                                      double* out0, double* out1) {
  if (!a || !out0 || n == 0) return rr::fail(RR_INVALID_PARAMETER, "selftest: null input/output");
  if (fn < 0 || fn > 9) return rr::fail(RR_INVALID_PARAMETER, "selftest: unknown function id");
  RR_HIP_TRY(hipSetDevice(device));
  double *da = nullptr, *db = nullptr, *d0 = nullptr, *d1 = nullptr;
  const size_t nb = n * sizeof(double);
  RR_HIP_TRY(rr::dev_malloc(&da, nb));
  RR_HIP_TRY(rr::dev_malloc(&db, nb));
  RR_HIP_TRY(rr::dev_malloc(&d0, nb));
  RR_HIP_TRY(rr::dev_malloc(&d1, nb));
  RR_HIP_TRY(hipMemcpy(da, a, nb, hipMemcpyHostToDevice));
  if (b) RR_HIP_TRY(hipMemcpy(db, b, nb, hipMemcpyHostToDevice));
  else RR_HIP_TRY(hipMemset(db, 0, nb));
  hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, n, da, db, d0, d1);
  RR_HIP_TRY(hipGetLastError());
  RR_HIP_TRY(hipDeviceSynchronize());
  RR_HIP_TRY(hipMemcpy(out0, d0, nb, hipMemcpyDeviceToHost));
  if (out1) RR_HIP_TRY(hipMemcpy(out1, d1, nb, hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(d0);
  (void)hipFree(d1);
  return RR_OK;
}
