// rr_common.hpp -- host-side plumbing shared by the engine translation units:
// error reporting behind the C ABI, HIP call checking, wave64 primitives.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "rr_pf.h"

namespace rr {

std::string& last_error_slot();

inline rr_status fail(rr_status code, const std::string& msg) {
  last_error_slot() = msg;
  return code;
}

#define RR_HIP_TRY(expr)                                                                  \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      return ::rr::fail(RR_RUNTIME_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    }                                                                                     \
  } while (0)

// ---- wave64 primitives (gfx950: a wavefront is 64 lanes) ------------------------------
constexpr int kWave = 64;

__device__ inline double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double t = __shfl_xor(v, o, kWave);
    v = t > v ? t : v;
  }
  return v;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// HIP's 64-bit shuffles are declared on (unsigned) long long; uint64_t is unsigned long here
using ull = unsigned long long;
__device__ inline uint64_t shfl_xor_u64(uint64_t v, int o) { return (uint64_t)__shfl_xor((ull)v, o, kWave); }
__device__ inline uint64_t shfl_up_u64(uint64_t v, int o) { return (uint64_t)__shfl_up((ull)v, o, kWave); }
__device__ inline uint64_t shfl_u64(uint64_t v, int src) { return (uint64_t)__shfl((ull)v, src, kWave); }

__device__ inline uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += shfl_xor_u64(v, o);
  return v;
}

// inclusive prefix sum across the 64 lanes
__device__ inline uint64_t wave_scan_u64(uint64_t v, int lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    uint64_t t = shfl_up_u64(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

struct u128 {
  uint64_t hi, lo;
};

__device__ inline void atomic_max_u64(uint64_t* p, uint64_t v) { atomicMax((ull*)p, (ull)v); }

__host__ __device__ inline u128 add128(u128 a, u128 b) {
  u128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
  return r;
}

__device__ inline u128 wave_sum_u128(u128 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    u128 t;
    t.hi = shfl_xor_u64(v.hi, o);
    t.lo = shfl_xor_u64(v.lo, o);
    v = add128(v, t);
  }
  return v;
}

}  // namespace rr
