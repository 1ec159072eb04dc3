// rr_common.hpp -- host-side plumbing shared by the engine translation units:
// error reporting behind the C ABI, HIP call checking, wave64 primitives.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

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
      (void)hipGetLastError(); /* HIP's last error is sticky: reported once, here */        \
      return ::rr::fail(RR_RUNTIME_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    }                                                                                     \
  } while (0)

// ---- HIP-event timing of kernel launches on one stream (measurement hooks of the C ABI)
struct Profiler {
  struct Ev {
    int id;
    hipEvent_t a, b;
  };
  bool on = false;
  // dominant-only mode: only the engine's dominant kernel is timed, through the start/stop events
  // of its own dispatch packet (hipExtLaunchKernelGGL) -- no extra packets in the stream, so the
  // kernel runs exactly as in an un-instrumented loop.  ScopedTimer is inert in this mode.
  bool dispatch_only = false;
  std::vector<Ev> events;
  std::vector<hipEvent_t> pool;
  std::vector<uint64_t> launches;
  std::vector<double> ms;
  explicit Profiler(int n_ids) : launches(n_ids, 0), ms(n_ids, 0.0) {}
  hipEvent_t take() {
    hipEvent_t e;
    if (!pool.empty()) {
      e = pool.back();
      pool.pop_back();
    } else {
      (void)hipEventCreate(&e);
    }
    return e;
  }
  void drain() {
    for (auto& e : events) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) {
        ms[e.id] += t;
        launches[e.id] += 1;
      }
      pool.push_back(e.a);
      pool.push_back(e.b);
    }
    events.clear();
  }
  void reset() {
    drain();
    for (auto& v : launches) v = 0;
    for (auto& v : ms) v = 0.0;
  }
  void destroy() {
    for (auto& e : events) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
    events.clear();
    pool.clear();
  }
};

// brackets the launches issued during its lifetime with two events on `stream`
struct ScopedTimer {
  Profiler& p;
  hipStream_t stream;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(Profiler& p_, hipStream_t s, int id_) : p(p_), stream(s), id(id_) {
    if (!p.on || p.dispatch_only) return;
    a = p.take();
    b = p.take();
    (void)hipEventRecord(a, stream);
  }
  ~ScopedTimer() {
    if (!p.on || p.dispatch_only) return;
    (void)hipEventRecord(b, stream);
    p.events.push_back({id, a, b});
  }
};

// The one-launch resample plans (k_quantize_plan_mark, k_shard_plan_mark) spin inside the kernel until every workgroup
// of their grid has arrived.  Two such kernels running at the same time on one device could each hold the slots the
// other's missing workgroups need, so per device (and process) only ONE handle at a time may have spinning kernels in
// flight: a handle asks before it launches one (spin_permit) and keeps the permit until its stream has been
// synchronised (spin_release); a handle that is refused takes the multi-launch plan -- identical results, and kernels
// that do not spin always finish, so the permit holder's grid gets its slots.  (Found by running two 1e6-particle
// filters side by side: without this each waited ~1.5 s for the other and gave up.)
struct SpinGate {
  std::mutex m;
  const void* holder = nullptr;
};
// `device` is a gate number: the device ordinal, or -- for a handle whose stream is confined to its own part of the CUs
// (RR_P2P_CU_PARTITION, pf_engine.hip) -- device + 64 * (part + 1): spinning kernels on disjoint CUs cannot starve each other
constexpr int kSpinGates = 64 * 17;
inline SpinGate& spin_gate(int device) {
  static SpinGate g[kSpinGates];
  return g[(unsigned)device % (unsigned)kSpinGates];
}
inline bool spin_permit(int device, const void* handle) {
  SpinGate& g = spin_gate(device);
  std::lock_guard<std::mutex> lock(g.m);
  if (g.holder == nullptr) g.holder = handle;
  return g.holder == handle;
}
inline void spin_release(int device, const void* handle) {  // the handle's stream is idle (or the handle is going away)
  SpinGate& g = spin_gate(device);
  std::lock_guard<std::mutex> lock(g.m);
  if (g.holder == handle) g.holder = nullptr;
}

// how long a workgroup of a one-launch plan waits for the others before the launch degrades to the serial plan
// (resample_core.hpp, k_quantize_plan_mark), in ticks of the 100 MHz wall clock: RR_PF_PLAN_TIMEOUT_US, default 2000 us
// (0: give up at once -- the test hook that exercises the serial plan on an idle device)
inline uint64_t plan_giveup_ticks() {
  static const uint64_t ticks = [] {
    const char* e = std::getenv("RR_PF_PLAN_TIMEOUT_US");
    const double us = e ? std::atof(e) : 2000.0;
    return (uint64_t)((us >= 0.0 ? us : 2000.0) * 100.0);
  }();
  return ticks;
}

// ---- RR_DEBUG_POISON_ALLOC=1: every device allocation of the engine is filled with 0xA5 before anybody uses it -------------------
// hipMalloc does not promise zeroed memory, and in practice returns zeros on a fresh box and whatever an earlier PROCESS left on a
// used one -- a read of something the engine never wrote passes every test on the former and fails somewhere, sometimes, on the
// latter (round 6: the unsharded 2e6-particle reference filter of ONE of eight processes sharing a device came out different, and
// only when other tests had run before).  With the switch on, the whole GPU suite runs against poisoned allocations
// (tests/test_gpu_poison.py); a buffer the engine reads before it writes shows as a parity failure, deterministically.
inline hipError_t dev_malloc(void** p, size_t bytes) {
  // 1: the byte 0xA5 (a huge integer, a denormal-sized negative double); any other value: that byte (0x3f: doubles of ~5e-4 and
  // integers of ~1e9; 0xff: NaNs and all-ones markers; 0x01: small counters) -- what a stale tenant left looks like SOMETHING
  static const int poison = [] { const char* e = std::getenv("RR_DEBUG_POISON_ALLOC"); return e ? (int)std::strtol(e, nullptr, 0) : 0; }();
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && poison && bytes) {
    // on a stream of its own that waits for nobody: the null stream would wait for every blocking stream of the process -- among
    // them a linked shard's step that waits, on the device, for the very shard this allocation belongs to (in-process worlds)
    hipStream_t s = nullptr;
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemsetAsync(*p, poison == 1 ? 0xA5 : (poison & 0xff), bytes, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (s) (void)hipStreamDestroy(s);
  }
  return e;
}
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc(reinterpret_cast<void**>(p), bytes); }

// ---- fills that are DONE when the call returns ------------------------------------------------------------------------------------
// hipMemset(device memory) is queued on the NULL stream and may return before it has run (CUDA's rule, and this runtime's: seen in
// round 6 as a first resample plan that marked into an earlier tenant's bytes).  Every stream of the engine is a non-blocking one,
// which the null stream does not order itself against -- so a plain hipMemset at create time is a race with the filter's first
// kernels.  memset_on fills on the stream whose kernels will use the memory and waits for it: ordered for that stream by
// construction, and finished for everybody else (a peer that maps the memory through hipIpc, another shard of the process).
inline hipError_t memset_on(hipStream_t s, void* p, int value, size_t bytes) {
  hipError_t e = hipMemsetAsync(p, value, bytes, s);
  return e == hipSuccess ? hipStreamSynchronize(s) : e;
}

// ---- rr_pf_warm / rr_fs1_warm: bring the device (and the runtime) to the state the thousandth step finds --------------
// An MI355X that has been idle runs its first ~50 ms of work at reduced clocks (measured, MCL 1e6 x 32: 52.6 us/step right after
// create, 47.9 after 1000 steps; k_step_lazy 35.8 -> 31.2 us), and the HIP runtime pays one-off costs along its first few thousand
// launches.  A caller's first steps are then 10 % slower than its later ones.  device_warm enqueues `ms` milliseconds of FP64 work
// on the handle's stream as launches of the size, argument block and cadence of a step kernel (25 us each, every CU busy, the
// stream drained every 100 launches as a stepping caller drains it), so that the first real step runs at the steady rate.
struct WarmArg {
  double v[288];  // a step kernel's argument block carries the observations: ~2.3 KB
};
static __global__ __launch_bounds__(256) void k_warm(WarmArg a, uint64_t ticks /* of the 100 MHz wall clock */, double* sink) {
  const uint64_t t0 = wall_clock64();
  double x = a.v[threadIdx.x] + (double)threadIdx.x, y = 1.0000001;
  do {
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_fma(x, y, 1e-9);
  } while (wall_clock64() - t0 < ticks);
  if (x == 12345.678 && sink) *sink = x;  // (never: keeps the chain alive)
}
inline hipError_t device_warm(hipStream_t stream, int device, double ms) {
  int cus = 0;
  hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (e != hipSuccess) return e;
  static WarmArg arg{};  // zeros
  const int launches = (int)(ms * 1000.0 / 25.0 + 0.5);
  for (int i = 0; i < launches; ++i) {
    hipLaunchKernelGGL(k_warm, dim3((unsigned)cus * 8u), dim3(256), 0, stream, arg, (uint64_t)2500, (double*)nullptr);
    if ((i + 1) % 100 == 0 && (e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

// ---- wave64 primitives (gfx950: a wavefront is 64 lanes) ------------------------------
constexpr int kWave = 64;

// Exact, order-independent reductions (integer sums, maxima) run on DPP -- lane data moved inside the VALU, ~10 cycles a
// step -- instead of __shfl (ds_bpermute_b32 through the LDS crossbar, ~100+ cycles a step, and the steps depend on each
// other): row_shr 1 / 2 / 4 / 8 give an inclusive scan inside each row of 16 lanes, row_bcast:15 (rows 1, 3) and
// row_bcast:31 (rows 2, 3) carry it across the rows.  Lane 63 ends up with the reduction of the whole wave.
// (Floating-point sums take the same lane movements in one FIXED order -- wave_sum_dpp, wave_sum4 below -- since round 4; their
// value depends on the order, which is therefore part of the deterministic specification's GPU side.)
template <int CTRL, int ROW_MASK>
__device__ inline unsigned int dpp_u32(unsigned int v) {  // lanes without a source lane (or outside ROW_MASK) read 0
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ inline uint64_t dpp_u64(uint64_t v) {
  return ((uint64_t)dpp_u32<CTRL, ROW_MASK>((unsigned int)(v >> 32)) << 32) | dpp_u32<CTRL, ROW_MASK>((unsigned int)v);
}
__device__ inline unsigned int umax(unsigned int a, unsigned int b) { return a > b ? a : b; }
__device__ inline uint64_t umax(uint64_t a, uint64_t b) { return a > b ? a : b; }
#define RR_DPP_SCAN(V, STEP)                  \
  do {                                        \
    V = STEP(V, 0x111, 0xf); /* row_shr:1 */  \
    V = STEP(V, 0x112, 0xf); /* row_shr:2 */  \
    V = STEP(V, 0x114, 0xf); /* row_shr:4 */  \
    V = STEP(V, 0x118, 0xf); /* row_shr:8 */  \
    V = STEP(V, 0x142, 0xa); /* row_bcast:15 */ \
    V = STEP(V, 0x143, 0xc); /* row_bcast:31 */ \
  } while (0)

__device__ inline uint64_t last_lane_u64(uint64_t v) {
  return ((uint64_t)(unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), 63) << 32) |
         (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, 63);
}

// inclusive running maximum across the 64 lanes (0 is the identity: unsigned values)
__device__ inline unsigned int wave_scan_max_u32(unsigned int v) {
#define RR_STEP_MAX32(V, C, M) umax(V, dpp_u32<C, M>(V))
  RR_DPP_SCAN(v, RR_STEP_MAX32);
#undef RR_STEP_MAX32
  return v;
}

// inclusive running maximum of unsigned 64-bit values across the 64 lanes
__device__ inline uint64_t wave_scan_max_u64(uint64_t v) {
#define RR_STEP_MAX64S(V, C, M) umax(V, dpp_u64<C, M>(V))
  RR_DPP_SCAN(v, RR_STEP_MAX64S);
#undef RR_STEP_MAX64S
  return v;
}

// maximum of non-negative doubles (NaN / negative lanes must have been replaced by 0): the bit patterns order like the values
__device__ inline double wave_max(double v) {
  uint64_t u = (uint64_t)__double_as_longlong(v);
#define RR_STEP_MAX64(V, C, M) umax(V, dpp_u64<C, M>(V))
  RR_DPP_SCAN(u, RR_STEP_MAX64);
#undef RR_STEP_MAX64
  return __longlong_as_double((long long)last_lane_u64(u));
}

// floating-point sum of the 64 lanes in a FIXED order (row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then row_bcast:15 and
// row_bcast:31 across the rows): reproducible for a given input, lane 63 holds it
__device__ inline double wave_sum_dpp(double v) {
#define RR_STEP_ADDF(V, C, M) (V + __longlong_as_double((long long)dpp_u64<C, M>((uint64_t)__double_as_longlong(V))))
  RR_DPP_SCAN(v, RR_STEP_ADDF);
#undef RR_STEP_ADDF
  return v;
}

// ... handed to every lane.  (Until round 4 this was a __shfl_xor butterfly: twelve dependent ds_bpermute_b32 per sum, ~1.5 us
// for the 4 - 5 sums of an estimate -- most of what the in-step estimate added to the plan kernel's tail and of the adaptive
// step's 5 us estimate phase.  The order of the additions, hence the last bits of means and covariances, changed with it.)
__device__ inline double wave_sum(double v) {
  v = wave_sum_dpp(v);
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  return __longlong_as_double((long long)last_lane_u64(u));
}

// FOUR floating-point sums of the 64 lanes in a fixed order for little more than the price of one: the lanes of a quad first
// split the work -- after two exchanges (quad_perm, lane ^ 1 then lane ^ 2) each lane holds its quad's total of ONE of the four
// values --, then one value per lane goes down the rows (row_shr 4 / 8) and across them (lane ^ 16, lane ^ 32: commutative, so
// every row ends with the same bits).  7 additions and 6 + 4 lane exchanges instead of 24 and 48.
// Returns, in lanes 12 .. 15 of every row, the wave's sums of a[0], a[2], a[1], a[3] (wave_sum4_field(lane) names the value).
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  return __longlong_as_double((long long)dpp_u64<CTRL, 0xf>((uint64_t)__double_as_longlong(v)));
}
__device__ inline int wave_sum4_field(int lane) { return ((lane & 1) << 1) | ((lane >> 1) & 1); }
__device__ inline double wave_sum4(const double (&a)[4]) {
  const int lane = (int)(threadIdx.x & 63u);
  const bool odd = (lane & 1) != 0, up = (lane & 2) != 0;
  // lane ^ 1: even lanes go on with a[0], a[1], odd lanes with a[2], a[3]
  const double keep0 = odd ? a[2] : a[0], keep1 = odd ? a[3] : a[1];
  const double send0 = odd ? a[0] : a[2], send1 = odd ? a[1] : a[3];
  const double b0 = keep0 + dpp_f64<0xB1>(send0);  // quad_perm:[1,0,3,2]
  const double b1 = keep1 + dpp_f64<0xB1>(send1);
  // lane ^ 2: lanes 0, 1 of the quad go on with the first of their two values, lanes 2, 3 with the second
  const double keep = up ? b1 : b0, send = up ? b0 : b1;
  double c = keep + dpp_f64<0x4E>(send);  // quad_perm:[2,3,0,1]
  c = c + dpp_f64<0x114>(c);              // row_shr:4 (the first quad of a row adds +0)
  c = c + dpp_f64<0x118>(c);              // row_shr:8: lanes 12 .. 15 hold the row's totals
  c = c + __shfl_xor(c, 16, kWave);
  c = c + __shfl_xor(c, 32, kWave);
  return c;
}

// HIP's 64-bit shuffles are declared on (unsigned) long long; uint64_t is unsigned long here
using ull = unsigned long long;
__device__ inline uint64_t shfl_xor_u64(uint64_t v, int o) { return (uint64_t)__shfl_xor((ull)v, o, kWave); }
__device__ inline uint64_t shfl_up_u64(uint64_t v, int o) { return (uint64_t)__shfl_up((ull)v, o, kWave); }
__device__ inline uint64_t shfl_u64(uint64_t v, int src) { return (uint64_t)__shfl((ull)v, src, kWave); }

// inclusive prefix sum across the 64 lanes
__device__ inline uint64_t wave_scan_u64(uint64_t v, int /*lane*/) {
#define RR_STEP_ADD64(V, C, M) (V + dpp_u64<C, M>(V))
  RR_DPP_SCAN(v, RR_STEP_ADD64);
#undef RR_STEP_ADD64
  return v;
}

__device__ inline uint64_t wave_sum_u64(uint64_t v) { return last_lane_u64(wave_scan_u64(v, 0)); }

struct u128 {
  uint64_t hi, lo;
};

// Running maximum of a value that only grows within a kernel.  Device-scope atomics on ONE address are carried out one
// after the other at the memory side -- measured 7.5 ns each: the 1954 workgroups of k_step_lazy at 1e6 particles spent
// 14.6 us queueing there -- so look first: a coherent load (never newer than the truth, and the truth only grows) lets
// every caller that cannot raise the maximum skip its atomic.
__device__ inline void atomic_max_u64(uint64_t* p, uint64_t v) {
  const uint64_t seen = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v > seen) atomicMax((ull*)p, (ull)v);
}

__host__ __device__ inline u128 add128(u128 a, u128 b) {
  u128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
  return r;
}

__device__ inline u128 wave_sum_u128(u128 v) {
#define RR_STEP_ADD128(V, C, M) add128(V, u128{dpp_u64<C, M>(V.hi), dpp_u64<C, M>(V.lo)})
  RR_DPP_SCAN(v, RR_STEP_ADD128);
#undef RR_STEP_ADD128
  return u128{last_lane_u64(v.hi), last_lane_u64(v.lo)};
}

}  // namespace rr

