// valu_rates.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the
// MCL propagate+weight kernel is made of, measured on the machine at hand.  Development tool
// (docs/DESIGN_NOTES.md section 4 "instruction budget"); not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                 \
  do {                                                                           \
    hipError_t _e = (e);                                                         \
    if (_e != hipSuccess) {                                                      \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));               \
      std::exit(1);                                                              \
    }                                                                            \
  } while (0)

constexpr int kIters = 256;   // loop trips
constexpr int kUnroll = 8;    // independent chains per trip

// Every kernel: each thread keeps kUnroll independent values, applies OP to each per trip.
#define RATE_KERNEL(NAME, TYPE, INIT, BODY)                                           \
  __global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE a, TYPE b) {            \
    TYPE v[kUnroll];                                                                  \
    for (int k = 0; k < kUnroll; ++k) v[k] = INIT;                                    \
    for (int i = 0; i < kIters; ++i) {                                                \
      _Pragma("unroll") for (int k = 0; k < kUnroll; ++k) { BODY; }                   \
    }                                                                                 \
    TYPE s = v[0];                                                                    \
    for (int k = 1; k < kUnroll; ++k) s += v[k];                                      \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
  }

RATE_KERNEL(k_fma_f64, double, (a + k + threadIdx.x), asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b)))
RATE_KERNEL(k_mul_f64, double, (a + k + threadIdx.x), asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_add_f64, double, (a + k + threadIdx.x), asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_min_f64, double, (a + k + threadIdx.x), asm volatile("v_min_f64 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_rsq_f64, double, (a + k + threadIdx.x), asm volatile("v_rsq_f64 %0, %0" : "+v"(v[k])))
RATE_KERNEL(k_rcp_f64, double, (a + k + threadIdx.x), asm volatile("v_rcp_f64 %0, %0" : "+v"(v[k])))
RATE_KERNEL(k_sqrt_f64, double, (a + k + threadIdx.x), asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[k])))
RATE_KERNEL(k_ldexp_f64, double, (a + k + threadIdx.x), asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v[k])))
RATE_KERNEL(k_rndne_f64, double, (a + k + threadIdx.x), asm volatile("v_rndne_f64 %0, %0" : "+v"(v[k])))
RATE_KERNEL(k_fract_f64, double, (a + k + threadIdx.x), asm volatile("v_fract_f64 %0, %0" : "+v"(v[k])))
RATE_KERNEL(k_cvt_f64_f32_rt, double, (a + k + threadIdx.x),
            { float t; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(v[k])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[k]) : "v"(t)); })
RATE_KERNEL(k_fma_f32, float, (a + k + threadIdx.x), asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b)))
RATE_KERNEL(k_rsq_f32, float, (a + k + threadIdx.x), asm volatile("v_rsq_f32 %0, %0" : "+v"(v[k])))
RATE_KERNEL(k_pk_fma_f32, double, (a + k + threadIdx.x), asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b)))
RATE_KERNEL(k_add_u32, unsigned, (a + k + threadIdx.x), asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_xor_b32, unsigned, (a + k + threadIdx.x), asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_mul_lo_u32, unsigned, (a + k + threadIdx.x), asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_mul_hi_u32, unsigned, (a + k + threadIdx.x), asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_mad_u64_u32, unsigned long long, (a + k + threadIdx.x),
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v[k]) : "v"((unsigned)a), "v"((unsigned)b) : "vcc"))
RATE_KERNEL(k_alignbit, unsigned, (a + k + threadIdx.x), asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(v[k])))
RATE_KERNEL(k_cndmask, unsigned, (a + k + threadIdx.x), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a) : "vcc"))
RATE_KERNEL(k_lshl_add_u64, unsigned long long, (a + k + threadIdx.x), asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(v[k]) : "v"(a)))
RATE_KERNEL(k_cvt_f64_u32, double, (a + k + threadIdx.x),
            { unsigned t = (unsigned)k; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[k]) : "v"(t)); })
RATE_KERNEL(k_cmp_f64, double, (a + k + threadIdx.x), asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(v[k]), "v"(a) : "vcc"))
RATE_KERNEL(k_div_scale_f64, double, (a + k + threadIdx.x), asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(v[k]) : "v"(a) : "vcc"))
RATE_KERNEL(k_div_fmas_f64, double, (a + k + threadIdx.x), asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b) : "vcc"))
RATE_KERNEL(k_div_fixup_f64, double, (a + k + threadIdx.x), asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b)))

// dependent chain latency: one value, kIters * kUnroll back-to-back dependent ops
#define CHAIN_KERNEL(NAME, TYPE, BODY)                                            \
  __global__ __launch_bounds__(64) void NAME(TYPE* out, TYPE a, TYPE b) {         \
    TYPE v = a + threadIdx.x;                                                     \
    for (int i = 0; i < kIters; ++i) {                                            \
      _Pragma("unroll") for (int k = 0; k < kUnroll; ++k) { BODY; }               \
    }                                                                             \
    out[blockIdx.x * 64 + threadIdx.x] = v;                                       \
  }
CHAIN_KERNEL(c_fma_f64, double, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b)))
CHAIN_KERNEL(c_rsq_f64, double, asm volatile("v_rsq_f64 %0, %0" : "+v"(v)))
CHAIN_KERNEL(c_fma_f32, float, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b)))
CHAIN_KERNEL(c_mad_u64_u32, unsigned long long, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"((unsigned)a), "v"((unsigned)b) : "vcc"))

template <typename T, typename K>
static void run(const char* name, K kern, int waves_per_simd, void* out, int block = 256) {
  // grid fills every SIMD with waves_per_simd waves: 256 CUs * 4 SIMDs
  const int waves = 256 * 4 * waves_per_simd;
  const int blocks = waves * 64 / block;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), 0, 0, (T*)out, (T)1.0000001, (T)0.5);
  CHECK(hipEventRecord(a, 0));
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), 0, 0, (T*)out, (T)1.0000001, (T)0.5);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double sec = ms * 1e-3 / reps;
  const double insts_per_simd = (double)waves_per_simd * kIters * kUnroll;
  std::printf("%-18s waves/SIMD=%d  %8.2f us  ns per wave-instr per SIMD = %.3f  (= %.2f cycles @2.4GHz)\n", name, waves_per_simd,
              sec * 1e6, sec * 1e9 / insts_per_simd, sec * 2.4e9 / insts_per_simd);
}

int main() {
  void* out;
  CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * 8 * 2));
#define R(K, T) run<T>(#K, K, 8, out)
  R(k_fma_f64, double);
  R(k_mul_f64, double);
  R(k_add_f64, double);
  R(k_min_f64, double);
  R(k_rsq_f64, double);
  R(k_rcp_f64, double);
  R(k_sqrt_f64, double);
  R(k_ldexp_f64, double);
  R(k_rndne_f64, double);
  R(k_fract_f64, double);
  R(k_cvt_f64_f32_rt, double);
  R(k_cvt_f64_u32, double);
  R(k_cmp_f64, double);
  R(k_div_scale_f64, double);
  R(k_div_fmas_f64, double);
  R(k_div_fixup_f64, double);
  R(k_fma_f32, float);
  R(k_rsq_f32, float);
  R(k_pk_fma_f32, double);
  R(k_add_u32, unsigned);
  R(k_xor_b32, unsigned);
  R(k_mul_lo_u32, unsigned);
  R(k_mul_hi_u32, unsigned);
  R(k_mad_u64_u32, unsigned long long);
  R(k_alignbit, unsigned);
  R(k_cndmask, unsigned);
  R(k_lshl_add_u64, unsigned long long);
  run<double>("k_fma_f64", k_fma_f64, 4, out);
  run<double>("k_fma_f64", k_fma_f64, 2, out);
  run<double>("k_fma_f64", k_fma_f64, 1, out);
  std::printf("-- dependent chains, one wave per SIMD: cycles per instruction = latency\n");
  run<double>("c_fma_f64", c_fma_f64, 1, out, 64);
  run<double>("c_rsq_f64", c_rsq_f64, 1, out, 64);
  run<float>("c_fma_f32", c_fma_f32, 1, out, 64);
  run<unsigned long long>("c_mad_u64_u32", c_mad_u64_u32, 1, out, 64);
  return 0;
}
