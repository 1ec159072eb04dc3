// mfma_pair.hip -- A/B of the MCL pair loop (particle_filter.rs:316-329: sum over landmarks of (d - |p - l|)^2) with the squared
// distance formed on the FP64 VALU (what k_step_lazy does: rr_pf_weight_fused_rows, include/rr_pf_spec.h) against the squared
// distance formed on the FP64 MATRIX pipe (v_mfma_f64_16x16x4_f64: |p - l|^2 = [-2lx, -2ly, 1, lx^2+ly^2] . [x, y, x^2+y^2, 1], a
// K = 4 contraction of 16 landmarks x 16 particles per instruction), the VALU keeping the square root, the residual and the
// accumulation.  VERDICT r4 item 4.  Development tool (profiles/r05_mfma_f64_ab.md); not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -disable-machine-licm -I../../include -o mfma_pair mfma_pair.hip && ./mfma_pair
// Prints, for n particles x L landmarks: microseconds per launch of both kernels (HIP events over `reps` launches), pairs/s,
// the largest relative difference of the per-particle sums, and what the matrix instruction's arithmetic is: whether
// D = C + sum_k A_k B_k equals, bit for bit, a chain of IEEE fused multiply-adds in ascending or descending k on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "rr_detmath.h"
#include "rr_pf_spec.h"

#define CHECK(e)                                                   \
  do {                                                             \
    hipError_t _e = (e);                                           \
    if (_e != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); \
      std::exit(1);                                                \
    }                                                              \
  } while (0)

#if !defined(__HIP_DEVICE_COMPILE__)
__device__ static inline double rr_sqrt_core(double x) { return x; }  // (host pass of the compiler: never called)
#endif
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256, kRows = 2, kSlots = kBlock * kRows;  // k_step_lazy's tile: 512 particles per workgroup, 128 per wave

// ---- A: the product's loop (same statement order as rr_pf_weight_fused_rows, the sum returned instead of exp(...))
__global__ __launch_bounds__(kBlock, 4) void k_pair_valu(const double* __restrict__ px, const double* __restrict__ py,
                                                         const double* __restrict__ obs, int n_obs, double* __restrict__ out, uint64_t n) {
  extern __shared__ double s_obs[];
  for (int i = threadIdx.x; i < 3 * n_obs; i += kBlock) s_obs[i] = obs[i];
  __syncthreads();
  double x[kRows], y[kRows], ss[kRows];
  const uint64_t base = (uint64_t)blockIdx.x * kSlots;
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const uint64_t k = base + (uint64_t)r * kBlock + threadIdx.x;
    x[r] = k < n ? px[k] : 0.0;
    y[r] = k < n ? py[k] : 0.0;
    ss[r] = 0.0;
  }
#pragma unroll 4
  for (int l = 0; l < n_obs; ++l) {
    const double d = s_obs[3 * l], lx = s_obs[3 * l + 1], ly = s_obs[3 * l + 2];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const double dx = x[r] - lx, dy = y[r] - ly;
      const double q = rr_fma(dy, dy, rr_fma(dx, dx, RR_PF_Q_FLOOR));
      const double diff = d - rr_sqrt_core(q);
      ss[r] = rr_fma(diff, diff, ss[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const uint64_t k = base + (uint64_t)r * kBlock + threadIdx.x;
    if (k < n) out[k] = ss[r];
  }
}

// ---- B: squared distances from the matrix pipe.  Per wave: 128 particles = 8 column groups of 16; per launch: n_obs / 16 row
// blocks of 16 landmarks (n_obs a multiple of 16 here).  Operand layout of v_mfma_f64_16x16x4_f64 (lane l):
//   A (16 x 4, rows = landmarks):  A[l % 16][l / 16]         B (4 x 16, columns = particles):  B[l / 16][l % 16]
//   D (16 x 16), register v = 0..3:  D[4 * v + l / 16][l % 16]     (found by the probe below; NOT 4 * (l / 16) + v)
// so a lane accumulates for ONE particle (column l % 16) over four landmarks per instruction, and the four lanes
// {c, c + 16, c + 32, c + 48} that share a particle are added up once at the end (through LDS, in lane order).
// MODE 0: the real thing.  MODE 1: the matrix instructions alone (their results folded with 32-bit integer XORs: nothing on the
// FP64 vector pipe).  MODE 2: the vector work alone (the square root, residual and accumulation of the same number of pairs, the
// squared distances made up from registers with one integer add).  T(0) ~ max(T(1), T(2)): the pipes overlap; ~ T(1) + T(2): they do not.
template <int MODE>
__global__ __launch_bounds__(kBlock, 4) void k_pair_mfma(const double* __restrict__ px, const double* __restrict__ py,
                                                         const double* __restrict__ obs, int n_obs, double* __restrict__ out, uint64_t n) {
  extern __shared__ double lds[];
  const int n_blk = n_obs / 16;
  double* const s_a = lds;                           // [n_blk][64]: the landmark operand of every row block, lane order
  double* const s_d = s_a + (size_t)n_blk * 64;      // [n_obs]: observed ranges
  double* const s_p = s_d + n_obs;                   // [4 waves][3][128]: x, y, x^2 + y^2 of the wave's particles
  double* const s_acc = s_p + 4 * 3 * 128;           // [4 waves][8 groups][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < n_blk * 64; i += kBlock) {
    const int blk = i >> 6, l = i & 63, m = 16 * blk + (l & 15), k = l >> 4;
    const double lx = obs[3 * m + 1], ly = obs[3 * m + 2];
    s_a[i] = k == 0 ? -2.0 * lx : (k == 1 ? -2.0 * ly : (k == 2 ? 1.0 : rr_fma(ly, ly, lx * lx)));
  }
  for (int i = tid; i < n_obs; i += kBlock) s_d[i] = obs[3 * i];
  const uint64_t base = (uint64_t)blockIdx.x * kSlots + (uint64_t)wv * 128;  // (a wave takes 128 CONSECUTIVE particles here)
  double* const wp = s_p + wv * 3 * 128;
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const uint64_t k = base + (uint64_t)r * 64 + lane;
    const double x = k < n ? px[k] : 0.0, y = k < n ? py[k] : 0.0;
    wp[r * 64 + lane] = x;
    wp[128 + r * 64 + lane] = y;
    wp[256 + r * 64 + lane] = rr_fma(y, y, x * x);
  }
  __syncthreads();
  // the particle operands of the wave's 8 column groups: field l / 16 of particle 16 g + l % 16 ([x, y, x^2+y^2, 1])
  double b[8];
  const int fld = lane >> 4, col = lane & 15;
#pragma unroll
  for (int g = 0; g < 8; ++g) b[g] = fld == 3 ? 1.0 : wp[fld * 128 + 16 * g + col];
  double acc[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) acc[g] = 0.0;
  const d4 c0 = {RR_PF_Q_FLOOR, RR_PF_Q_FLOOR, RR_PF_Q_FLOOR, RR_PF_Q_FLOOR};
  for (int blk = 0; blk < n_blk; ++blk) {
    const double a = s_a[blk * 64 + lane];
    const double* dd = s_d + 16 * blk + fld;  // the observed ranges of this lane's four rows: l / 16 + 4 v
    const double d0 = dd[0], d1 = dd[4], d2 = dd[8], d3 = dd[12];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      d4 q;
      if (MODE != 2) {
        q = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[g], c0, 0, 0, 0);
      } else {  // made-up squared distances in [1, 2): one 32-bit integer add each, no FP64 instruction
        const uint64_t u = 0x3ff0000000000000ull + ((uint64_t)(unsigned)(blk * 8 + g + lane) << 32);
        q[0] = rr_u2d(u), q[1] = rr_u2d(u + (1ull << 40)), q[2] = rr_u2d(u + (2ull << 40)), q[3] = rr_u2d(u + (3ull << 40));
      }
      if (MODE == 1) {
        acc[g] = rr_u2d(rr_d2u(acc[g]) ^ rr_d2u(q[0]) ^ rr_d2u(q[1]) ^ rr_d2u(q[2]) ^ rr_d2u(q[3]));
        continue;
      }
      const double e0 = d0 - rr_sqrt_core(q[0]);
      const double e1 = d1 - rr_sqrt_core(q[1]);
      const double e2 = d2 - rr_sqrt_core(q[2]);
      const double e3 = d3 - rr_sqrt_core(q[3]);
      acc[g] = rr_fma(e3, e3, rr_fma(e2, e2, rr_fma(e1, e1, rr_fma(e0, e0, acc[g]))));
    }
  }
  double* const wa = s_acc + wv * 8 * 64;
#pragma unroll
  for (int g = 0; g < 8; ++g) wa[g * 64 + lane] = acc[g];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const int i = r * 64 + lane, g = i >> 4, c = i & 15;  // particle i of the wave: column c of group g
    const double* q = wa + g * 64 + c;
    const uint64_t k = base + (uint64_t)i;
    if (k < n) out[k] = ((q[0] + q[16]) + q[32]) + q[48];
  }
}

// ---- what the matrix instruction computes: one 16 x 16 x 4 product of random operands, every element back to the host
__global__ void k_mfma_probe(const double* __restrict__ A /* [16][4] */, const double* __restrict__ B /* [4][16] */,
                             const double* __restrict__ C /* [16][16] */, double* __restrict__ D /* [16][16] */) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
  d4 c;
  for (int v = 0; v < 4; ++v) c[v] = C[(4 * v + (l >> 4)) * 16 + (l & 15)];
  const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * v + (l >> 4)) * 16 + (l & 15)] = d[v];
}

static uint64_t bits(double x) {
  uint64_t u;
  std::memcpy(&u, &x, 8);
  return u;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000;
  const int L = argc > 2 ? std::atoi(argv[2]) : 32;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 200;
  if (L % 16) {
    std::fprintf(stderr, "L must be a multiple of 16\n");
    return 1;
  }
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> pos(-25.0, 25.0), unit(0.5, 2.0);
  // ---- the instruction's arithmetic
  {
    int asc = 0, desc = 0, layout_ok = 0, trials = 200;
    double *dA, *dB, *dC, *dD;
    CHECK(hipMalloc(&dA, 64 * 8));
    CHECK(hipMalloc(&dB, 64 * 8));
    CHECK(hipMalloc(&dC, 256 * 8));
    CHECK(hipMalloc(&dD, 256 * 8));
    for (int t = 0; t < trials; ++t) {
      double A[64], B[64], C[256], D[256];
      for (double& v : A) v = pos(rng) * unit(rng);
      for (double& v : B) v = pos(rng) * unit(rng);
      for (double& v : C) v = t % 2 ? pos(rng) : 0.0;
      CHECK(hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      CHECK(hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost));
      bool a_ok = true, d_ok = true, near = true;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double up = C[i * 16 + j], down = C[i * 16 + j];
          for (int k = 0; k < 4; ++k) up = std::fma(A[i * 4 + k], B[k * 16 + j], up);
          for (int k = 3; k >= 0; --k) down = std::fma(A[i * 4 + k], B[k * 16 + j], down);
          a_ok &= bits(up) == bits(D[i * 16 + j]);
          d_ok &= bits(down) == bits(D[i * 16 + j]);
          near &= std::fabs(up - D[i * 16 + j]) <= 1e-9 * (1.0 + std::fabs(up));
        }
      if (t == 0 && !near) {  // the assumed register layout is wrong: say where every element of the product really landed
        std::printf("{\"layout_probe\": \"D[i][j] as stored by the kernel under the ASSUMED layout holds the product's element (i', j'):\", \"map\": [");
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            int fi = -1, fj = -1;
            for (int a = 0; a < 16 && fi < 0; ++a)
              for (int b2 = 0; b2 < 16; ++b2) {
                double up = C[a * 16 + b2];  // (t == 0: C is random, so C's own placement is part of what is probed)
                for (int k = 0; k < 4; ++k) up = std::fma(A[a * 4 + k], B[k * 16 + b2], up);
                if (std::fabs(up - D[i * 16 + j]) <= 1e-9 * (1.0 + std::fabs(up))) { fi = a; fj = b2; break; }
              }
            if (j < 2 || j == 15) std::printf("[%d,%d,%d,%d],", i, j, fi, fj);
          }
        std::printf("[]]}\n");
      }
      asc += a_ok;
      desc += d_ok;
      layout_ok += near;
    }
    std::printf("{\"mfma_f64_16x16x4\": {\"trials\": %d, \"operand_layout_as_assumed\": %d, \"equals_fma_chain_ascending_k\": %d, "
                "\"equals_fma_chain_descending_k\": %d}}\n", trials, layout_ok, asc, desc);
  }
  // ---- the A/B
  std::vector<double> x(n), y(n), obs(3 * (size_t)L);
  for (uint64_t i = 0; i < n; ++i) x[i] = pos(rng), y[i] = pos(rng);
  for (int l = 0; l < L; ++l) {
    obs[3 * l + 1] = pos(rng);
    obs[3 * l + 2] = pos(rng);
    obs[3 * l] = std::hypot(obs[3 * l + 1] - 1.0, obs[3 * l + 2] + 2.0) + 0.1 * unit(rng);
  }
  double *dx, *dy, *dobs, *o1, *o2, *o3;
  CHECK(hipMalloc(&dx, n * 8));
  CHECK(hipMalloc(&dy, n * 8));
  CHECK(hipMalloc(&dobs, obs.size() * 8));
  CHECK(hipMalloc(&o1, n * 8));
  CHECK(hipMalloc(&o2, n * 8));
  CHECK(hipMalloc(&o3, n * 8));
  CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dobs, obs.data(), obs.size() * 8, hipMemcpyHostToDevice));
  const unsigned grid = (unsigned)((n + kSlots - 1) / kSlots);
  const size_t lds_v = 3 * (size_t)L * 8;
  const size_t lds_m = ((size_t)(L / 16) * 64 + L + 4 * 3 * 128 + 4 * 8 * 64) * 8;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms_v = 0, ms_m = 0, ms_m1 = 0, ms_m2 = 0;
  for (int pass = 0; pass < 2; ++pass) {  // (first pass warms the device up)
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_pair_valu, dim3(grid), dim3(kBlock), lds_v, 0, dx, dy, dobs, L, o1, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_v, e0, e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_pair_mfma<0>, dim3(grid), dim3(kBlock), lds_m, 0, dx, dy, dobs, L, o2, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_m, e0, e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_pair_mfma<1>, dim3(grid), dim3(kBlock), lds_m, 0, dx, dy, dobs, L, o3, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_m1, e0, e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_pair_mfma<2>, dim3(grid), dim3(kBlock), lds_m, 0, dx, dy, dobs, L, o3, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_m2, e0, e1));
  }
  CHECK(hipGetLastError());
  std::vector<double> h1(n), h2(n);
  CHECK(hipMemcpy(h1.data(), o1, n * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(h2.data(), o2, n * 8, hipMemcpyDeviceToHost));
  double worst = 0.0, worst_host = 0.0;
  for (uint64_t i = 0; i < n; ++i) {
    worst = std::fmax(worst, std::fabs(h1[i] - h2[i]) / std::fmax(std::fabs(h1[i]), 1e-300));
    if (i < 20000) {  // the literal form on the host (libm sqrt, no FMA), a sample
      double s = 0.0;
      for (int l = 0; l < L; ++l) {
        const double ddx = x[i] - obs[3 * l + 1], ddy = y[i] - obs[3 * l + 2], df = obs[3 * l] - std::sqrt(ddx * ddx + ddy * ddy);
        s += df * df;
      }
      worst_host = std::fmax(worst_host, std::fabs(h2[i] - s) / std::fmax(std::fabs(s), 1e-300));
    }
  }
  const double pairs = (double)n * L;
  std::printf("{\"n\": %llu, \"L\": %d, \"reps\": %d, \"valu_us\": %.3f, \"mfma_us\": %.3f, \"valu_pairs_per_s\": %.4g, \"mfma_pairs_per_s\": %.4g, "
              "\"matrix_pipe_alone_us\": %.3f, \"vector_rest_alone_us\": %.3f, \"mfma_over_valu_time\": %.4f, \"max_rel_diff_of_sums_mfma_vs_valu\": %.3g, \"max_rel_diff_of_sums_mfma_vs_literal_host\": %.3g}\n",
              (unsigned long long)n, L, reps, 1e3 * ms_v / reps, 1e3 * ms_m / reps, pairs / (1e-3 * ms_v / reps), pairs / (1e-3 * ms_m / reps),
              1e3 * ms_m1 / reps, 1e3 * ms_m2 / reps, ms_m / ms_v, worst, worst_host);
  return 0;
}
