// Is hipMemset(device memory) finished when it returns?  (CUDA's rule: no -- it is asynchronous with respect to the host.)
// The engine's streams are non-blocking ones, which do not order themselves against the null stream a hipMemset runs on: a kernel
// launched on such a stream right after the call sees the fill only if the call itself waited.  Probe: fill 1 GiB (poisoned first),
// launch at once on a non-blocking stream a kernel that samples the buffer's far end, count the samples that still hold the poison.
//   hipcc --offload-arch=gfx950 -O2 -o memset_sync_probe memset_sync_probe.hip && ./memset_sync_probe [rounds] [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void sample(const unsigned char* p, size_t n, unsigned* stale) {
  const size_t i = n - 1 - (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4096;
  if (__builtin_nontemporal_load(p + i) != 0) atomicAdd(stale, 1u);
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 20;
  const size_t n = (size_t)(argc > 2 ? std::atoi(argv[2]) : 1024) << 20;
  unsigned char* p = nullptr;
  unsigned *stale = nullptr, h = 0;
  hipStream_t s;
  CK(hipMalloc(&p, n));
  CK(hipMalloc(&stale, 4));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int hit_plain = 0, hit_async_sync = 0, hit_h2d = 0;
  unsigned worst = 0;
  unsigned char* host = (unsigned char*)std::calloc(n, 1);  // pageable zeros: mode 2 copies them over the poison with a plain hipMemcpy
  for (int mode = 0; mode < 3; ++mode)
    for (int r = 0; r < rounds; ++r) {
      CK(hipMemsetAsync(p, 0xA5, n, s));
      CK(hipMemsetAsync(stale, 0, 4, s));
      CK(hipDeviceSynchronize());
      if (mode == 0) {
        CK(hipMemset(p, 0, n));  // the call under test
      } else if (mode == 2) {
        CK(hipMemcpy(p, host, n, hipMemcpyHostToDevice));  // (pageable source: may the DMA still be under way when the call returns?)
      } else {
        CK(hipMemsetAsync(p, 0, n, s));  // the engine's replacement: on the consumer's stream, and waited for
        CK(hipStreamSynchronize(s));
      }
      hipLaunchKernelGGL(sample, dim3(64), dim3(256), 0, s, p, n, stale);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost));
      if (h) (mode == 0 ? hit_plain : mode == 1 ? hit_async_sync : hit_h2d) += 1;
      if (h > worst) worst = h;
    }
  std::printf("{\"MiB\": %zu, \"rounds\": %d, \"hipMemset_then_kernel_on_nonblocking_stream_saw_stale\": %d, \"memsetAsync_sync_saw_stale\": %d, \"pageable_hipMemcpy_H2D_then_kernel_saw_stale\": %d, \"worst_stale_samples_of_16384\": %u}\n",
              n >> 20, rounds, hit_plain, hit_async_sync, hit_h2d, worst);
  return 0;
}
