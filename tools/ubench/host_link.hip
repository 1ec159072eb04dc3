// host_link.hip -- where should the command block of the resident step kernel (csrc/resident_core.hpp) live?
// Ping-pong between the host and ONE resident wave: the host writes command i (a 16-byte {bits, seq} pair), the wave polls
// for it and answers with a 16-byte pair in pinned host memory, the host polls for that.  Command block in
//   (a) pinned host memory (hipHostMalloc): the wave's poll is a read across the host link;
//   (b) fine-grained device memory (hipExtMallocWithFlags) written by the host through the PCIe BAR: the wave polls its own
//       memory, the host's write is posted.
// A forked child probes whether the host may touch (b) at all (a fault there must not kill the measurement).
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/host_link tools/ubench/host_link.hip
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

struct alignas(16) Pair { uint64_t bits, seq; };
using u4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ inline void ld16(const Pair* p, uint64_t& bits, uint64_t& seq) {
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  bits = (uint64_t)v.x | ((uint64_t)v.y << 32);
  seq = (uint64_t)v.z | ((uint64_t)v.w << 32);
}
__device__ inline void st16(Pair* p, uint64_t bits, uint64_t seq) {
  u4 v;
  v.x = (unsigned)bits; v.y = (unsigned)(bits >> 32); v.z = (unsigned)seq; v.w = (unsigned)(seq >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// width: pairs polled per look (lanes 0 .. width-1), all must carry the command's seq
__global__ void k_pong(const Pair* cmd, Pair* rsp, uint64_t n, int width, uint64_t* polls_out) {
  const int lane = threadIdx.x;
  uint64_t polls = 0;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  for (uint64_t i = 1; i <= n; ++i) {
    const uint64_t t0 = wall_clock64();
    for (;;) {
      uint64_t b = 0, s = 0;
      if (lane < width) ld16(&cmd[lane], b, s);
      ++polls;
      const uint64_t m = __ballot(s == i);
      const uint64_t want = width == 64 ? ~0ull : ((1ull << width) - 1);
      if ((m & want) == want) break;
      if (wall_clock64() - t0 > 200000000ull) return;  // 2 s: the host died
    }
    if (lane < 4) st16(&rsp[lane], i * 3 + lane, i);
  }
  if (lane == 0) {
    polls_out[0] = polls;
    polls_out[1] = clock64() - c0;   // shader clock cycles ...
    polls_out[2] = wall_clock64() - w0;  // ... per 100 MHz ticks: the clock a lone resident wave runs at
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at %s\"}\n", hipGetErrorString(e_), #x); return 1; } } while (0)

static double g_mhz = 0.0;
static double pingpong(Pair* cmd_host_view, const Pair* cmd_dev_view, Pair* rsp, int width, int n, double* p50, uint64_t* polls) {
  uint64_t* d_polls;
  hipMalloc(&d_polls, 24);
  hipMemset(d_polls, 0, 24);
  std::memset(rsp, 0, 8 * sizeof(Pair));
  for (int k = 0; k < 64; ++k) { cmd_host_view[k].bits = 0; cmd_host_view[k].seq = 0; }
  hipDeviceSynchronize();
  hipStream_t st;
  hipStreamCreate(&st);
  hipLaunchKernelGGL(k_pong, dim3(1), dim3(64), 0, st, cmd_dev_view, rsp, (uint64_t)n, width, d_polls);
  std::vector<double> ts(n);
  for (int i = 1; i <= n; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = width - 1; k >= 0; --k) {
      cmd_host_view[k].bits = (uint64_t)i * 7 + k;
      __atomic_store_n(&cmd_host_view[k].seq, (uint64_t)i, __ATOMIC_RELEASE);
    }
    long spins = 0;
    while (__atomic_load_n(&rsp[3].seq, __ATOMIC_ACQUIRE) != (uint64_t)i || __atomic_load_n(&rsp[0].seq, __ATOMIC_ACQUIRE) != (uint64_t)i) {
      if (++spins > 400000000L) { printf("{\"error\": \"no answer to command %d\"}\n", i); return -1.0; }
    }
    ts[i - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  hipStreamSynchronize(st);
  uint64_t pc[3];
  hipMemcpy(pc, d_polls, 24, hipMemcpyDeviceToHost);
  *polls = pc[0];
  g_mhz = pc[2] ? 100.0 * (double)pc[1] / (double)pc[2] : 0.0;
  hipStreamDestroy(st);
  hipFree(d_polls);
  double sum = 0;
  for (int i = n / 10; i < n; ++i) sum += ts[i];
  std::sort(ts.begin(), ts.end());
  *p50 = ts[n / 2];
  return sum / (n - n / 10);
}

int main() {
  CK(hipSetDevice(0));
  Pair *cmd_host, *rsp;
  CK(hipHostMalloc(&cmd_host, 64 * sizeof(Pair), hipHostMallocDefault));
  CK(hipHostMalloc(&rsp, 8 * sizeof(Pair), hipHostMallocDefault));
  const int n = 20000;
  printf("{");
  for (int width : {1, 15, 64}) {
    double p50; uint64_t polls;
    const double m = pingpong(cmd_host, cmd_host, rsp, width, n, &p50, &polls);
    printf("\"host_pinned_w%d\": {\"mean_us\": %.3f, \"p50_us\": %.3f, \"polls_per_cmd\": %.2f, \"shader_mhz\": %.0f}, ", width, m, p50, (double)polls / n, g_mhz);
  }
  // (b) device memory the host writes through the BAR
  struct { const char* name; unsigned flags; } kinds[] = {{"finegrained", hipDeviceMallocFinegrained}, {"uncached", hipDeviceMallocUncached}, {"default", hipDeviceMallocDefault}};
  for (auto& kd : kinds) {
    Pair* cmd_dev = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&cmd_dev, 64 * sizeof(Pair), kd.flags);
    if (e != hipSuccess) { printf("\"device_%s\": {\"alloc\": \"%s\"}, ", kd.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    hipMemset(cmd_dev, 0, 64 * sizeof(Pair));
    hipDeviceSynchronize();
    fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) {  // may the host touch it?
      volatile uint64_t* q = reinterpret_cast<volatile uint64_t*>(cmd_dev);
      q[0] = 0x1234;
      const uint64_t v = q[0];
      _exit(v == 0x1234 ? 0 : 3);
    }
    int status = 0;
    waitpid(pid, &status, 0);
    const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
    if (!ok) { printf("\"device_%s\": {\"host_access\": false, \"status\": %d}, ", kd.name, status); hipFree(cmd_dev); continue; }
    for (int width : {1, 15, 64}) {
      double p50; uint64_t polls;
      const double m = pingpong(cmd_dev, cmd_dev, rsp, width, n, &p50, &polls);
      printf("\"device_%s_w%d\": {\"mean_us\": %.3f, \"p50_us\": %.3f, \"polls_per_cmd\": %.2f}, ", kd.name, width, m, p50, (double)polls / n);
    }
    hipFree(cmd_dev);
  }
  printf("\"n\": %d}\n", n);
  return 0;
}
