// Does the landmark-major PLANE layout of the FastSLAM map (field f of landmark l of particle p at
// ((3 + 6 l + f) * n + p)) cost bandwidth once ~0.7 us of arithmetic sits between a wave's loads and its stores?
// Same traffic, same grid and software pipeline as k_fs1_observe (1e5 particles x 200 landmarks, 25 chunks), synthetic
// arithmetic of K dependent FMAs per field, two layouts:
//   A  planes:  field f of 64 consecutive particles = 512 contiguous bytes, six 800 KB-strided pieces per update
//   B  blocks:  [landmark][particle / 64][field][particle % 64] -- the six fields of a wave's update are 3 KB contiguous
// plus the same planes with two particles per thread and a plain streaming copy of the same bytes as the reference.
// Measured (MI355X, gpurun): planes 5.2-5.35 TB/s, blocks 5.3-5.45, 2 particles/thread 5.1-5.2, plain copy 4.9-5.2.
//   hipcc --offload-arch=gfx950 -O3 plane_layout.hip -o plane_layout && ./plane_layout
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int LAYOUT, int K>
__global__ __launch_bounds__(256) void k(const double* src, double* dst, uint64_t n, int L, int chunk_len) {
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int l0 = blockIdx.y * chunk_len, l1 = min(l0 + chunk_len, L);
  auto addr = [&](int l, int f) -> uint64_t {
    if (LAYOUT == 0) return ((uint64_t)(3 + 6 * l + f)) * n + p;
    const uint64_t nb = (n + 63) / 64;  // blocks per landmark
    return 3 * n + (((uint64_t)l * nb + p / 64) * 6 + f) * 64 + (p % 64);
  };
  double nxt[6];
#pragma unroll
  for (int f = 0; f < 6; ++f) nxt[f] = src[addr(l0, f)];
  double acc = 1.0;
  for (int l = l0; l < l1; ++l) {
    double e[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) e[f] = nxt[f];
    if (l + 1 < l1) {
#pragma unroll
      for (int f = 0; f < 6; ++f) nxt[f] = src[addr(l + 1, f)];
    }
    // K dependent FMAs per field, ILP 6 (about what the EKF offers)
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      double v = e[f];
#pragma unroll
      for (int i = 0; i < K; ++i) v = __builtin_fma(v, 0.999999, 1e-9);
      e[f] = v;
    }
    acc *= e[0] + e[5];
#pragma unroll
    for (int f = 0; f < 6; ++f) dst[addr(l, f)] = e[f];
  }
  dst[p] = acc;
}


// planes, two consecutive particles per thread (16-byte accesses, 1 KB per wave per plane)
template <int K>
__global__ __launch_bounds__(256) void k2(const double* src, double* dst, uint64_t n, int L, int chunk_len) {
  const uint64_t p = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (p >= n) return;
  const int l0 = blockIdx.y * chunk_len, l1 = min(l0 + chunk_len, L);
  auto addr = [&](int l, int f) -> uint64_t { return ((uint64_t)(3 + 6 * l + f)) * n + p; };
  double2 nxt[6];
#pragma unroll
  for (int f = 0; f < 6; ++f) nxt[f] = *reinterpret_cast<const double2*>(src + addr(l0, f));
  double acc = 1.0;
  for (int l = l0; l < l1; ++l) {
    double2 e[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) e[f] = nxt[f];
    if (l + 1 < l1) {
#pragma unroll
      for (int f = 0; f < 6; ++f) nxt[f] = *reinterpret_cast<const double2*>(src + addr(l + 1, f));
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      double v = e[f].x, u = e[f].y;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        v = __builtin_fma(v, 0.999999, 1e-9);
        u = __builtin_fma(u, 0.999999, 1e-9);
      }
      e[f].x = v;
      e[f].y = u;
    }
    acc *= e[0].x + e[5].y;
#pragma unroll
    for (int f = 0; f < 6; ++f) *reinterpret_cast<double2*>(dst + addr(l, f)) = e[f];
  }
  dst[p] = acc;
}

__global__ __launch_bounds__(256) void k_stream(const double2* __restrict__ src, double2* __restrict__ dst, uint64_t n2) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

template <int LAYOUT, int K>
float run(const double* src, double* dst, uint64_t n, int L, int chunks) {
  const int len = (L + chunks - 1) / chunks;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)chunks);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LAYOUT, K>), grid, dim3(256), 0, 0, src, dst, n, L, len);
  (void)hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LAYOUT, K>), grid, dim3(256), 0, 0, src, dst, n, L, len);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const uint64_t n = 100000;
  const int L = 200, chunks = 25;
  const size_t doubles = (3 + 6 * (size_t)L) * (((n + 63) / 64) * 64) + 64;
  double *src, *dst;
  (void)hipMalloc(&src, doubles * sizeof(double));
  (void)hipMalloc(&dst, doubles * sizeof(double));
  (void)hipMemset(src, 0, doubles * sizeof(double));
  (void)hipMemset(dst, 0, doubles * sizeof(double));
  const double gb = 96.0 * n * L / 1e9;
  float t;
#define RUN(LAY, KK) t = run<LAY, KK>(src, dst, n, L, chunks); std::printf("layout %s  K=%3d (%4d FMAs per update)  %.1f us  %.2f TB/s\n", LAY ? "blocks" : "planes", KK, 6 * KK, t * 1e3, gb / t)
  RUN(0, 0); RUN(1, 0);
  std::printf("-- in place (dst == src)\n");
  { double* keep = dst; dst = src; RUN(0, 0); RUN(0, 50); RUN(1, 50); dst = keep; }
  std::printf("-- out of place\n");
  {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const uint64_t n2 = (3 + 6 * (uint64_t)L) * n / 2;
    for (int g : {2048, 8192, 32768}) {
      hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, (const double2*)src, (double2*)dst, n2);
      (void)hipEventRecord(a);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, (const double2*)src, (double2*)dst, n2);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      std::printf("plain streaming copy of the same bytes, grid %5d: %.1f us  %.2f TB/s\n", g, ms / 10 * 1e3, 16.0 * n2 * 2 / 1e9 / (ms / 10));
    }
    for (int kk : {0, 50}) {
      const int len = (L + chunks - 1) / chunks;
      dim3 grid((unsigned)((n / 2 + 255) / 256), (unsigned)chunks);
      if (kk == 0) hipLaunchKernelGGL((k2<0>), grid, dim3(256), 0, 0, src, dst, n, L, len); else hipLaunchKernelGGL((k2<50>), grid, dim3(256), 0, 0, src, dst, n, L, len);
      (void)hipEventRecord(a);
      for (int i = 0; i < 10; ++i) { if (kk == 0) hipLaunchKernelGGL((k2<0>), grid, dim3(256), 0, 0, src, dst, n, L, len); else hipLaunchKernelGGL((k2<50>), grid, dim3(256), 0, 0, src, dst, n, L, len); }
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      std::printf("planes, 2 particles per thread, K=%d: %.1f us  %.2f TB/s\n", kk, ms / 10 * 1e3, gb / (ms / 10));
    }
  }
  RUN(0, 16); RUN(1, 16);
  RUN(0, 32); RUN(1, 32);
  RUN(0, 50); RUN(1, 50);
  RUN(0, 80); RUN(1, 80);
  // shape sweep of the plane pattern (K = 50): observation chunks x workgroups per CU (capped by dynamic LDS)
  std::printf("-- planes, K=50: chunks x resident workgroups per CU\n");
  for (int ch : {8, 13, 25, 50, 100, 200}) {
    for (int per_cu : {2, 4, 6, 8}) {
      const size_t lds = (size_t)(150 * 1024) / per_cu;  // bytes of dynamic LDS per workgroup: at most per_cu fit in 160 KB
      const int len = (L + ch - 1) / ch;
      dim3 grid((unsigned)((n + 255) / 256), (unsigned)ch);
      hipEvent_t a, b;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipFuncSetAttribute((const void*)k<0, 50>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((k<0, 50>), grid, dim3(256), lds, 0, src, dst, n, L, len);
      (void)hipEventRecord(a);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<0, 50>), grid, dim3(256), lds, 0, src, dst, n, L, len);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      std::printf("chunks %3d  <= %d WG/CU: %.1f us  %.2f TB/s\n", ch, per_cu, ms / 5 * 1e3, gb / (ms / 5));
    }
  }
  return 0;
}
