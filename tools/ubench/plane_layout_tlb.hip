// k_fs1_observe runs at 0.66 of the HBM peak at 1e5 particles x 200 landmarks and at 0.61 at 1e6 x 200: is it
// the plane layout's page spread?  With the landmark-major planes (field f of landmark l of particle p at ((3 + 6 l + f) n + p)) the
// six 512-byte pieces a wave touches per update lie n * 8 bytes apart: 0.8 MB at 1e5 particles (the whole 963 MB map is ~480 pages
// of 2 MB), 8 MB at 1e6 (9.6 GB: ~4 800 pages per buffer set, and every workgroup in flight sits in 6 of them that nobody near it
// in the dispatch order shares once the chunks spread out).
// This runs the kernel's access pattern (same grid shape and software pipeline, K dependent FMAs per field standing in for the
// EKF) at any n, in place like the kernel, in three layouts:
//   planes           the engine's
//   blocked B        [p / B][3 + 6 L][B]: a block of B particles keeps all its planes in one contiguous B * 9.6 KB region; the wave
//                    access stays 512 contiguous bytes and a monotone gather stays monotone inside a block
// and reports TB/s by algorithmic bytes (96 per update).  Pair it with the counters tools/fs1_tlb_probe.sh collects.
//   hipcc --offload-arch=gfx950 -O3 plane_layout_tlb.hip -o plane_layout_tlb && ./plane_layout_tlb 100000 25 && ./plane_layout_tlb 1000000 3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int K>
__global__ __launch_bounds__(256) void k(double* buf, uint64_t n, int L, int chunk_len, uint64_t B /* 0: planes */) {
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int l0 = blockIdx.y * chunk_len, l1 = min(l0 + chunk_len, L);
  const uint64_t np = 3 + 6 * (uint64_t)L;
  const uint64_t base = B ? (p / B) * np * B + (p % B) : p;
  const uint64_t stride = B ? B : n;
  auto addr = [&](int l, int f) -> uint64_t { return base + (uint64_t)(3 + 6 * l + f) * stride; };
  double nxt[6];
#pragma unroll
  for (int f = 0; f < 6; ++f) nxt[f] = buf[addr(l0, f)];
  double acc = 1.0;
  for (int l = l0; l < l1; ++l) {
    double e[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) e[f] = nxt[f];
    if (l + 1 < l1) {
#pragma unroll
      for (int f = 0; f < 6; ++f) nxt[f] = buf[addr(l + 1, f)];
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      double v = e[f];
#pragma unroll
      for (int i = 0; i < K; ++i) v = __builtin_fma(v, 0.999999, 1e-9);
      e[f] = v;
    }
    acc *= e[0] + e[5];
#pragma unroll
    for (int f = 0; f < 6; ++f) buf[addr(l, f)] = e[f];
  }
  if (acc == 12345.0) buf[base] = acc;
}

template <int K>
static float run(double* buf, uint64_t n, int L, int chunks, uint64_t B) {
  const int len = (L + chunks - 1) / chunks;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)chunks);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<K>), grid, dim3(256), 0, 0, buf, n, L, len, B);
  (void)hipEventRecord(a);
  const int reps = 6;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<K>), grid, dim3(256), 0, 0, buf, n, L, len, B);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000;
  const int L = 200;
  const uint64_t np = 3 + 6 * (uint64_t)L;
  const uint64_t n_pad = ((n + 32767) / 32768) * 32768;  // room for the largest block size
  double* buf;
  if (hipMalloc(&buf, np * n_pad * sizeof(double)) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, np * n_pad * sizeof(double));
  const double gb = 96.0 * (double)n * L / 1e9;
  std::printf("{\"particles\": %llu, \"landmarks\": %d, \"GB_per_launch\": %.3f, \"rows\": [\n", (unsigned long long)n, L, gb);
  bool first = true;
  for (int a = 2; a < (argc > 2 ? argc : 3); ++a) {
    const int chunks = argc > 2 ? std::atoi(argv[a]) : 25;
    for (uint64_t B : {(uint64_t)0, (uint64_t)2048, (uint64_t)8192, (uint64_t)32768}) {
      const float t0 = run<0>(buf, n, L, chunks, B), t50 = run<50>(buf, n, L, chunks, B);
      std::printf("%s {\"layout\": \"%s\", \"block\": %llu, \"chunks\": %d, \"K0_ms\": %.4f, \"K0_TBps\": %.3f, \"K50_ms\": %.4f, \"K50_TBps\": %.3f}",
                  first ? "" : ",\n", B ? "blocked" : "planes", (unsigned long long)B, chunks, t0, gb / t0, t50, gb / t50);
      first = false;
    }
  }
  std::printf("\n]}\n");
  (void)hipFree(buf);
  return 0;
}
