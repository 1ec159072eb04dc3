// resident_core.hpp -- a one-workgroup kernel that STAYS on the device and is fed steps through host-visible memory.
//
// Why: every caller in the reference runs its filter at 100 - 1 200 particles through the synchronous try_step
// (headless_localizers.rs:39-56, render_gif_particle_filter.rs:77-79, ros2_nodes/ekf_localizer_node/src/main.rs:273).  At
// that size a step is 3 - 6 us of arithmetic in one workgroup; a kernel launch plus the completion round trip is 13 - 20 us
// on top.  A resident kernel pays neither: the host writes the step's inputs into a pinned command block, the kernel (which
// keeps the particles in registers between steps) finds them by polling, runs the step and stores the estimate into a
// pinned response block the host polls.
//
// Protocol (one producer, one consumer, one command in flight):
//   * Both blocks are arrays of 16-byte PAIRS {bits, seq}.  The writer stores `bits` then `seq` (host: two ordered stores
//     into one cache line; device: ONE 16-byte store = one write request) and the reader takes a pair with ONE 16-byte load
//     (device) or seq-then-bits (host, acquire): a pair whose seq is the awaited one carries that command's bits -- every
//     word vouches for itself, so no "payload, fence, flag" round trip in either direction and no ordering assumption
//     between different words on the PCIe / fabric path.
//   * cmd[0] = {op | count << 8}, cmd[1 .. count] = payload doubles (PF: u0, u1, then the observation rows).
//   * rsp[0 .. 3] = the estimate, rsp[4] = flags, rsp[5] = EXIT marker {last command consumed, launch id}.
//   * The kernel never waits unboundedly: it leaves after `idle_ticks` without a command and after `life_ticks` in any case
//     (so work queued behind it on a shared hardware queue, a hipDeviceSynchronize of another thread, or a process that
//     died without saying goodbye are delayed, never deadlocked), storing the particle set back to HBM and posting the EXIT
//     marker.  A command that crosses an exit is not lost: the host sees the marker (last consumed < its command), launches
//     the kernel again on the same stream and the new incarnation finds the command waiting.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rr {

struct alignas(16) MailPair {
  uint64_t bits;
  uint64_t seq;
};

constexpr int kResCmdPairs = 512;  // header + up to 511 payload doubles
constexpr int kResRspPairs = 16;  // [8 .. 15]: in-kernel stamps of the instrumented build (RR_PLAN_TIMELINE)
constexpr int kResRspFlags = 4;
constexpr int kResRspExit = 5;
enum ResidentOp : int { kResOpNone = 0, kResOpStep = 1, kResOpQuit = 2, kResOpIdle = 3 /* device-side: nothing came */ };

struct alignas(64) ResidentRing {
  MailPair cmd[kResCmdPairs];
  MailPair rsp[kResRspPairs];
};

struct ResidentArgs {
  int on;               // != 0: serve commands from the ring instead of running K steps
  int payload_cap;      // payload doubles the LDS staging area holds
  uint64_t first_seq;   // the first command this incarnation waits for
  uint64_t idle_ticks;  // 100 MHz wall-clock ticks without a command after which the kernel leaves
  uint64_t life_ticks;  // ... and after which it leaves no matter what
  uint64_t launch_id;   // stamps the EXIT marker
};

using res_u4 = __attribute__((ext_vector_type(4))) unsigned int;

// one 16-byte system-scope load / store (around every cache: the other side is the host CPU)
__device__ inline void load_pair_sys(const MailPair* p, uint64_t& bits, uint64_t& seq) {
  res_u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  bits = (uint64_t)v.x | ((uint64_t)v.y << 32);
  seq = (uint64_t)v.z | ((uint64_t)v.w << 32);
}
__device__ inline void store_pair_sys(MailPair* p, uint64_t bits, uint64_t seq) {
  res_u4 v;
  v.x = (unsigned int)bits;
  v.y = (unsigned int)(bits >> 32);
  v.z = (unsigned int)seq;
  v.w = (unsigned int)(seq >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// Wait for command `want`.  Every thread of the workgroup calls it; returns the op (uniform) with the payload in
// s_pay[0 .. count) and count in s_hdr[1].  Wave 0 polls the first `guess` pairs in one go (the header and, when the
// command has as many words as the one before -- the common case --, the whole payload: detection and transfer in the
// same round trip); longer commands are completed by the other threads.  kResOpIdle: nothing came within idle_ticks / the
// incarnation's life is over / a long command did not arrive completely (it is then left for the next incarnation).
template <int BLOCK>
__device__ inline int resident_fetch(const ResidentRing* __restrict__ ring, uint64_t want, uint64_t idle_ticks, uint64_t deadline,
                                     int payload_cap, int& guess, double* s_pay, int* s_hdr) {
  const int tid = threadIdx.x, lane = tid & 63;
  __syncthreads();  // nobody reads the previous command's s_pay / s_hdr any more
  if (tid < 64) {
    const uint64_t t0 = wall_clock64();
    int op = kResOpNone, count = 0;
    unsigned polls = 0;
    for (;;) {
      uint64_t bits = 0, seq = 0;
      if (lane < guess) load_pair_sys(&ring->cmd[lane], bits, seq);
      const uint64_t vm = __ballot(seq == want);
      if (vm & 1ull) {
        const uint64_t hdr = __shfl(bits, 0, 64);
        op = (int)(hdr & 0xffull);
        count = (int)(hdr >> 8);
        if (count > payload_cap || count >= kResCmdPairs) {  // (the host never sends such a command)
          op = kResOpQuit;
          count = 0;
        }
        const int need = 1 + count, need0 = need < 64 ? need : 64;
        const uint64_t mask = need0 == 64 ? ~0ull : ((1ull << need0) - 1ull);
        guess = need0;
        if ((vm & mask) == mask) {
          if (lane >= 1 && lane < need0) s_pay[lane - 1] = __longlong_as_double((long long)bits);
          break;
        }
        // the header is here, part of the payload was not read (or is not visible yet): look again, wide enough
      }
      const uint64_t now = wall_clock64();
      if (now - t0 > idle_ticks || now > deadline) {
        op = kResOpIdle;
        count = 0;
        break;
      }
      if (vm & 1ull) continue;
      if (++polls > 256u) __builtin_amdgcn_s_sleep(32);  // ~1 us naps once the caller has been quiet for a while
    }
    if (lane == 0) {
      s_hdr[0] = op;
      s_hdr[1] = count;
    }
  }
  __syncthreads();
  int op = s_hdr[0];
  const int count = s_hdr[1];
  if (op == kResOpStep && 1 + count > 64) {  // the rest of a long command: one pair per thread
    int ok = 1;
    for (int i = 64 + tid; i < 1 + count; i += BLOCK) {
      uint64_t bits = 0, seq = 0;
      const uint64_t t0 = wall_clock64();
      for (;;) {
        load_pair_sys(&ring->cmd[i], bits, seq);
        if (seq == want) break;
        if (wall_clock64() - t0 > 100000ull) {  // 1 ms
          ok = 0;
          break;
        }
      }
      s_pay[i - 1] = __longlong_as_double((long long)bits);
    }
    if (!__syncthreads_and(ok)) op = kResOpIdle;
  }
  return op;
}

// ---- host side of the pairs
inline void ring_put(MailPair* p, uint64_t bits, uint64_t seq) {
  p->bits = bits;
  __atomic_store_n(&p->seq, seq, __ATOMIC_RELEASE);
}
inline bool ring_take(const MailPair* p, uint64_t seq, uint64_t* bits) {
  if (__atomic_load_n(&p->seq, __ATOMIC_ACQUIRE) != seq) return false;
  *bits = __atomic_load_n(&p->bits, __ATOMIC_RELAXED);
  return true;
}

}  // namespace rr
