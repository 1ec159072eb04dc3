// What does this GPU copy memory at?  A few shapes of dst[i] = src[i] over 962 MB (the size of one FastSLAM buffer set at
// 1e5 x 200), as the reference for k_fs1_observe's 5.2 TB/s:   hipcc --offload-arch=gfx950 -O3 copy_rates.hip -o copy_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ src_, double2* __restrict__ dst_, uint64_t n2) {
  const v2d* __restrict__ src = reinterpret_cast<const v2d*>(src_);
  v2d* __restrict__ dst = reinterpret_cast<v2d*>(dst_);
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
    v2d v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n2; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ src, double* __restrict__ out, uint64_t n2) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
    const double2 v = src[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_write(double2* __restrict__ dst, uint64_t n2) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) dst[i] = make_double2(1.0, 2.0);
}

template <typename F>
float timeit(F f) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  f();
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / 10;
}

int main() {
  const uint64_t n2 = (3 + 6 * 200ull) * 100000 / 2;  // double2 elements
  const double gb = 16.0 * n2 / 1e9;
  double2 *src, *dst;
  (void)hipMalloc(&src, n2 * 16);
  (void)hipMalloc(&dst, n2 * 16);
  (void)hipMemset(src, 0, n2 * 16);
  (void)hipMemset(dst, 0, n2 * 16);
  for (int g : {1024, 2048, 4096, 8192, 32768, 131072}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(g), dim3(256), 0, 0, src, dst, n2); });
    std::printf("copy grid %6d unroll 1      : %.1f us  %.2f TB/s (read + write)\n", g, t * 1e3, 2 * gb / t);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, src, dst, n2); });
    std::printf("copy grid %6d unroll 4      : %.1f us  %.2f TB/s\n", g, t * 1e3, 2 * gb / t);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, src, dst, n2); });
    std::printf("copy grid %6d unroll 4 nt   : %.1f us  %.2f TB/s\n", g, t * 1e3, 2 * gb / t);
  }
  float t = timeit([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, src, (double*)dst, n2); });
  std::printf("read only  grid 8192: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, dst, n2); });
  std::printf("write only grid 8192: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { (void)hipMemcpyAsync(dst, src, n2 * 16, hipMemcpyDeviceToDevice, 0); });
  std::printf("hipMemcpyAsync D2D  : %.1f us  %.2f TB/s (read + write)\n", t * 1e3, 2 * gb / t);
  return 0;
}
