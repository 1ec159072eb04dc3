// Accuracy of the hardware reciprocal-square-root seed (v_rsq_f64) and how many refinement steps the
// correctly rounded root needs:   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off rsq_accuracy.hip -o rsq_accuracy
// Prints the largest |y*sqrt(x) - 1| over 2^30 pseudo-random x in [1, 4) (one binade pair covers every mantissa/parity
// case) and, for several refinement variants, how many results differ from the correctly rounded sqrt().
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ void k(unsigned long long* out, double* maxerr, uint64_t base, int rounds) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad[4] = {0, 0, 0, 0};
  double worst = 0.0;
  for (int r = 0; r < rounds; ++r) {
    const uint64_t u = mix(base + tid * (uint64_t)rounds + r);
    // x in [1, 4): exponent 0x3ff or 0x400, 52 random mantissa bits
    const uint64_t bits = ((0x3ffull + (u >> 63)) << 52) | (u & 0x000fffffffffffffull);
    const double x = __longlong_as_double((long long)bits);
    const double ref = sqrt(x);  // correctly rounded (device library)
    const double y = __builtin_amdgcn_rsq(x);
    // seed error: e = y*sqrt(x) - 1, via g = x*y ~ sqrt(x): e ~ (g*g - x)/(2x) evaluated with an exact residual
    const double g0 = x * y;
    const double res = __builtin_fma(g0, g0, -x);  // g0^2 - x (g0's own rounding error is ~2^-53, far below the seed's)
    const double e = fabs(res / (2.0 * x));
    if (e > worst) worst = e;
    const double h0 = 0.5 * y;
    // variant 0: the D-spec's core (Goldschmidt step + two corrections)
    {
      double g = g0, h = h0;
      const double rr = __builtin_fma(-h, g, 0.5);
      g = __builtin_fma(g, rr, g);
      h = __builtin_fma(h, rr, h);
      double d = __builtin_fma(-g, g, x);
      g = __builtin_fma(d, h, g);
      d = __builtin_fma(-g, g, x);
      g = __builtin_fma(d, h, g);
      bad[0] += g != ref;
    }
    // variant 1: Goldschmidt step + ONE correction
    {
      double g = g0, h = h0;
      const double rr = __builtin_fma(-h, g, 0.5);
      g = __builtin_fma(g, rr, g);
      h = __builtin_fma(h, rr, h);
      const double d = __builtin_fma(-g, g, x);
      g = __builtin_fma(d, h, g);
      bad[1] += g != ref;
    }
    // variant 2: no Goldschmidt step, two corrections
    {
      double g = g0;
      double d = __builtin_fma(-g, g, x);
      g = __builtin_fma(d, h0, g);
      d = __builtin_fma(-g, g, x);
      g = __builtin_fma(d, h0, g);
      bad[2] += g != ref;
    }
    // variant 3: no Goldschmidt step, one correction
    {
      double g = g0;
      const double d = __builtin_fma(-g, g, x);
      g = __builtin_fma(d, h0, g);
      bad[3] += g != ref;
    }
  }
  for (int v = 0; v < 4; ++v) atomicAdd(&out[v], bad[v]);
  // maximum of non-negative doubles through their bit patterns
  atomicMax((unsigned long long*)maxerr, (unsigned long long)__double_as_longlong(worst));
}

int main() {
  unsigned long long* d_out;
  double* d_max;
  (void)hipMalloc(&d_out, 4 * sizeof(unsigned long long));
  (void)hipMalloc(&d_max, sizeof(double));
  (void)hipMemset(d_out, 0, 4 * sizeof(unsigned long long));
  (void)hipMemset(d_max, 0, sizeof(double));
  const int blocks = 8192, threads = 256, rounds = 32768;  // 2^36 samples
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, d_max, 12345ull, rounds);
  unsigned long long out[4];
  double worst;
  (void)hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
  (void)hipMemcpy(&worst, d_max, sizeof(double), hipMemcpyDeviceToHost);
  std::printf("samples %llu  max seed error %.3e = 2^%.2f\n", (unsigned long long)blocks * threads * rounds, worst, std::log2(worst));
  const char* names[4] = {"goldschmidt + 2 corrections (D-spec core)", "goldschmidt + 1 correction", "2 corrections", "1 correction"};
  for (int v = 0; v < 4; ++v) std::printf("%-44s mismatches vs correctly rounded sqrt: %llu\n", names[v], out[v]);
  return 0;
}
