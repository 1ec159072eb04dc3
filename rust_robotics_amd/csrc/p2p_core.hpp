// p2p_core.hpp -- device-initiated exchange between the shards of a filter over xGMI, shared by
// the PF/MCL engine and the FastSLAM engine (include/rr_pf.h "peer-to-peer transport").
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>

#include <cstdlib>
#include <cstring>
#include <string>

#include "resample_core.hpp"
#include "resident_core.hpp"
#include "rr_common.hpp"

namespace rr {

// ------------------------------------------------------------------------------------------
// Device-initiated exchange over xGMI: no host code and no collective library inside a step.
// Every rank owns a fine-grained mailbox that the peers write their records into (WMAX / SUMS: self-vouching
// 16-byte pairs, P2PPairSlot; DONE: a release-stamped slot); k_p2p_exchange (one workgroup, one thread per peer)
// publishes this rank's record to every peer and waits, bounded, for every peer's record.  Particles that cross
// ranks at resample time are delivered into the owning rank's INBOX: a fine-grained (uncached, device-coherent) mirror of one buffer set, the
// kind of memory RCCL uses for its own peer-written buffers -- a store from a remote kernel is
// visible to a later local kernel without any assumption about what either GPU's L2 still holds.
// The consumer reads an "in place" slot from its inbox instead of its slab.
constexpr int kMaxP2P = 16;
// what gave up (the value latched in P2PState::err; the error message names it)
constexpr int kGaveUpWmax = 1, kGaveUpSums = 2, kGaveUpRecord = 3, kGaveUpInboxSlot = 4, kGaveUpPlanFlag = 5;
constexpr int kP2PHandleBytes = 256;
constexpr int kBusIdBytes = 32;  // the PCI bus id of the exporting device, after the three IPC handles
struct P2PSlot {
  uint64_t v[3];
  uint64_t seq;
};
// WMAX / SUMS records vouch for themselves (round 5): each payload word travels as ONE 16-byte store {word, seq ^ mix(word)}
// (MailPair, resident_core.hpp: the same store the resident service answers the host with).  The sender stores and is done -- no
// wait for the words' acknowledgement before a separate stamp; the reader polls the pairs themselves and has the payload the moment
// the last one fits -- no second round trip for it.  mix is one-to-one, so a pair torn in flight (it is a single aligned 16-byte
// transaction; nothing promises that on every fabric) fits only if the word is the one the tag was made for.  Only n_ranks threads of
// ONE workgroup poll here, so looking at the pairs directly costs nothing (489 workgroups doing the same inside the plan kernels is
// another matter: resample_core.hpp).  Measured at world size 1, two builds alternating on one box: 56.8 -> 56.1 us per sharded step
// (profiles/r05g_ab_shard_exchange_pairs.log).
struct alignas(64) P2PPairSlot {
  MailPair p[3];
};
__device__ inline uint64_t p2p_tag(uint64_t seq, uint64_t word) { return seq ^ (word * 0x9E3779B97F4A7C15ull); }
struct P2PMailbox {
  P2PPairSlot wmax[kMaxP2P];
  P2PPairSlot sums[kMaxP2P];
  P2PSlot done[kMaxP2P];
};
struct P2PPeers {
  P2PMailbox* mbox[kMaxP2P];
  double* slab[kMaxP2P];   // the peers' state slabs (the plain, eager protocol stores straight into them)
  double* inbox[kMaxP2P];  // the peers' fine-grained inboxes: where the lazy protocol delivers cross-rank particles
  int n_ranks;
  int rank;
  uint64_t timeout_ticks;  // bound of one peer wait in 100 MHz wall-clock ticks (RR_P2P_TIMEOUT_MS, default 2 s)
};
enum { kP2PWmax = 0, kP2PSums = 1, kP2PDone = 2 };

// A record's pair and the transport's give-up flag in ONE round trip: the flag's load (device scope) is issued first and waited for
// together with the pair's (system scope).
__device__ inline void load_pair_and_flag(const MailPair* p, const int* flag, uint64_t& bits, uint64_t& tag, int& flag_value) {
  res_u4 v;
  int f;
  asm volatile("global_load_dword %1, %3, off sc1\n\tglobal_load_dwordx4 %0, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v), "=&v"(f)
               : "v"(p), "v"(flag)
               : "memory");
  bits = (uint64_t)v.x | ((uint64_t)v.y << 32);
  tag = (uint64_t)v.z | ((uint64_t)v.w << 32);
  flag_value = f;
}

__device__ inline P2PPairSlot* p2p_pairs(P2PMailbox* m, int kind, int idx) { return kind == kP2PWmax ? &m->wmax[idx] : &m->sums[idx]; }

// One exchange round, executed block-uniformly by a workgroup of >= 64 threads (WMAX / DONE: thread g < n_ranks
// talks to peer g; SUMS: thread 3 g + k trades word k with peer g).  payload (valid in every thread of the first
// wave): kind WMAX -> {bits of the local max weight}, SUMS -> {T, q2_hi, q2_lo}, DONE -> nothing.  Every rank's payload is collected in LDS
// (gathered[g*3..]).  Post-processing by thread 0: WMAX -> *wmax_out = global max; SUMS -> finalize_plan with
// the global totals (what k_shard_plan does in the RCCL path).
// (Until round 5 the payloads were collected in a scratch array in DEVICE memory, with ordinary stores and loads.  Inside
// k_shard_plan_mark two DIFFERENT workgroups run an exchange in one launch -- workgroup 0 the maxima, the last arrival the sums --
// normally on two XCDs, each with an L2 of its own: two dirty copies of the same words.  When a kernel boundary of ANOTHER stream
// wrote both L2s back and invalidated them between the last arrival's store and its own load, the load came back with workgroup
// 0's maximum instead of the sum just stored -- T = the bit pattern of w_max, a wrong resample, ~1 in 150 runs of a 5 000-particle
// shard stepping beside another filter: tools/soak_shard_estimate.py, profiles/r05_shard_race.md.  Nothing that only one
// workgroup needs has any business in device memory.)
// `totals` (SUMS, optional, LDS): thread 0 leaves the global total, this rank's base, its local total and the squares' sum there
// INSTEAD of running finalize_plan -- the caller has something more urgent to do with them first (k_shard_plan_mark: tell the
// waiting workgroups) and finalizes afterwards; ok = 0 when the exchange gave up.
struct P2PSumsTotals {
  uint64_t total, base, local;
  u128 q2;
  int ok;
};
__device__ inline void p2p_exchange(const P2PPeers& peers, int kind, uint64_t seq, uint64_t v0, uint64_t v1, uint64_t v2,
                                    Ctl* __restrict__ ctl, double* __restrict__ wmax_out, const PlanArgs& pa, int* __restrict__ err,
                                    P2PSumsTotals* totals = nullptr) {
  __shared__ uint64_t gathered[3 * kMaxP2P];
  const int g = threadIdx.x;
  const int n_words = kind == kP2PWmax ? 1 : 3;
  // a peer is gone: do not hang the device.  The first exchanges of a filter get ten times the
  // budget -- process start-up and code-object loading skew the ranks by far more than a step does
  const uint64_t patience = seq <= 3 ? 10 * peers.timeout_ticks : peers.timeout_ticks;
  // Once a wait has given up, every later exchange of this filter gives up at once (the host reads the flag with
  // rr_pf_p2p_status); only the first one costs the time-out.  The flag travels with the first look at the peer's record -- one
  // round trip for both; up front, as until round 5, it was a device-scope round trip and a barrier on the critical path of every
  // exchange.
  bool bad = false;
  if (kind != kP2PDone && g < n_words * peers.n_ranks) {
    // WMAX (one word) / SUMS (three): self-vouching pairs, see P2PPairSlot.  One LANE per pair -- thread 3 r + k sends word k to
    // rank r and waits for word k of rank r -- so that a look at a peer's record is one round trip to the mailbox, not three one
    // after the other (a pair is read by ONE instruction that waits for its answer; until round 5 one thread per peer read the three
    // pairs of a SUMS record in turn: tools/shard_plan_timeline.py)
    const int r = g / n_words, k = g - r * n_words;
    const uint64_t mine = k == 0 ? v0 : (k == 1 ? v1 : v2);
    store_pair_sys(&p2p_pairs(peers.mbox[r], kind, peers.rank)->p[k], mine, p2p_tag(seq, mine));
    const MailPair* in = &p2p_pairs(peers.mbox[peers.rank], kind, r)->p[k];
    const uint64_t t0 = wall_clock64();  // 100 MHz
    uint64_t got = 0, tag = 0;
    int dead = 0;
    load_pair_and_flag(in, err, got, tag, dead);
    if (dead) bad = true;  // (the transport gave up earlier: whatever the record says)
    while (!bad && tag != p2p_tag(seq, got)) {
      if (wall_clock64() - t0 > patience) {
        bad = true;
        if (atomicCAS(err, 0, kind == kP2PWmax ? kGaveUpWmax : kGaveUpSums) == 0) {  // the FIRST give-up names the wait and says what it saw
          uint64_t* d = reinterpret_cast<uint64_t*>(err) + 1;  // (the flag's allocation has room for this: P2PState::local_setup)
          d[0] = (uint64_t)r;
          d[1] = seq;
          d[2] = tag ^ (got * 0x9E3779B97F4A7C15ull);  // the sequence number the pair that IS there vouches for
          d[3] = seq;                                   // (this rank's own record went out before the wait)
        }
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      load_pair_sys(in, got, tag);
    }
    if (!bad) gathered[3 * r + k] = got;
  } else if (kind == kP2PDone && g < peers.n_ranks) {
    // DONE of the eager / FastSLAM protocols vouches for bulk data written with ORDINARY stores by other workgroups and
    // earlier kernels: a full system-scope release (L2 write-back) has to come first, the stamp is a release store and the
    // wait an acquire
    P2PSlot* out = &peers.mbox[g]->done[peers.rank];
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    P2PSlot* in = &peers.mbox[peers.rank]->done[g];
    const uint64_t t0 = wall_clock64();
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) bad = true;  // (not a hot path: asked up front)
    while (!bad && __hip_atomic_load(&in->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > patience) {
        bad = true;
        if (atomicCAS(err, 0, kGaveUpRecord) == 0) {
          uint64_t* d = reinterpret_cast<uint64_t*>(err) + 1;
          d[0] = (uint64_t)g;
          d[1] = seq;
          d[2] = __hip_atomic_load(&in->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          d[3] = __hip_atomic_load(&out->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);  // (also: gathered[] is complete for thread 0)
  if (g != 0) return;
  if (any_bad) {
    ctl->fired = 0;  // nothing downstream may act on incomplete data
    if (kind == kP2PWmax) *wmax_out = 0.0;
    if (totals) totals->ok = 0;
    return;
  }
  if (kind == kP2PWmax) {
    double m = 0.0;
    for (int k = 0; k < peers.n_ranks; ++k) {
      const double w = rr_u2d(gathered[3 * k]);
      if (w > m) m = w;
    }
    *wmax_out = m;
  } else if (kind == kP2PSums) {
    uint64_t total = 0, base = 0;
    u128 qq = {0, 0};
    for (int k = 0; k < peers.n_ranks; ++k) {
      if (k == peers.rank) base = total;
      total += gathered[3 * k];
      qq = add128(qq, u128{gathered[3 * k + 1], gathered[3 * k + 2]});
    }
    if (totals) {
      totals->total = total;
      totals->base = base;
      totals->local = gathered[3 * peers.rank];
      totals->q2 = qq;
      totals->ok = 1;
    } else {
      finalize_plan(ctl, total, base, gathered[3 * peers.rank], qq, pa);
    }
  }
}

// stand-alone exchange (one workgroup of 64): payload from device memory written by the previous kernel
static __global__ void k_p2p_exchange(P2PPeers peers, int kind, uint64_t seq, const uint64_t* __restrict__ payload,
                                      Ctl* __restrict__ ctl, double* __restrict__ wmax_out, PlanArgs pa, int* __restrict__ err) {
  uint64_t v0 = 0, v1 = 0, v2 = 0;
  if (kind == kP2PWmax) v0 = payload[0];
  if (kind == kP2PSums) { v0 = payload[0]; v1 = payload[1]; v2 = payload[2]; }
  p2p_exchange(peers, kind, seq, v0, v1, v2, ctl, wmax_out, pa, err);
}

// k_scan_tiles + the SUMS exchange in one single-workgroup launch: exclusive scan of the tile
// totals (in place), this shard's sums to every peer, everybody's sums back, gate + plan.
static __global__ __launch_bounds__(kScanThreads) void k_scan_exchange(P2PPeers peers, uint64_t seq,
                                                                      uint64_t* __restrict__ tile_total,
                                                                      const uint64_t* __restrict__ tile_q2,
                                                                      uint64_t n_tiles, Ctl* __restrict__ ctl, PlanArgs pa,
                                                                      int* __restrict__ err) {
  __shared__ uint64_t s_pay[3];
  uint64_t total = 0;
  u128 qq = {0, 0};
  scan_tiles_block<kScanThreads>(tile_total, tile_q2, n_tiles, &total, &qq);
  if (threadIdx.x == 0) {
    ctl->total_local = total;
    s_pay[0] = total;
    s_pay[1] = qq.hi;
    s_pay[2] = qq.lo;
  }
  __syncthreads();
  p2p_exchange(peers, kP2PSums, seq, s_pay[0], s_pay[1], s_pay[2], ctl, &ctl->wmax, pa, err);
}

// Inbox traffic goes around every cache: system-scope (sc0 sc1) stores on the delivering side, system-scope loads on the
// reading side.  The reader consumes a delivery INSIDE a kernel (as soon as the slot's seal fits, below), not behind a kernel
// boundary, so nothing may depend on which lines an L2 -- eight of them per device, one per XCD -- still holds from the
// step before or on how a peer's mapping of the buffer is cached.  Only the slots that cross a shard boundary take this
// path (10^3 - 10^4 of 10^6 in steady state).
__device__ inline void st_sys(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void st_sys_u64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline double ld_sys(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// A delivered slot vouches for itself: the inbox has a fifth plane of SEALS, and a slot's seal is a 64-bit mix of the step's
// sequence number and the slot's four fields.  The deliverer stores the five words in one go, in any order, without waiting for
// anything; the reader takes all five and accepts them when the seal fits the fields it has just read -- a slot of which only
// a part has arrived yet (or none: the words of an earlier step, or the zeros of a fresh inbox) does not fit, and the reader
// looks again.  No DONE message, no election of a last workgroup to send it, nobody waits who has nothing to receive, and no
// "fields, wait for their acknowledgement, then the tag" on the deliverer's side (4 - 5 us of a system-scope round trip that
// the receiving shard's next step sat through).  Multiplying by odd constants is one-to-one, so a slot with exactly ONE
// stale word never fits; two or more stale words fit with probability 2^-64 per look.
__device__ inline uint64_t inbox_seal(uint64_t seq, double x, double y, double yaw, double v) {
  const uint64_t z = seq * 0x9E3779B97F4A7C15ull + rr_d2u(x) * 0xC2B2AE3D27D4EB4Full + rr_d2u(y) * 0x165667B19E3779F9ull +
                     rr_d2u(yaw) * 0xD6E8FEB86659FD93ull + rr_d2u(v) * 0xFF51AFD7ED558CCDull;
  return z ^ (z >> 31);  // one-to-one, 0 only for z == 0: a zeroed slot (seal 0, fields 0) fits no seq >= 1
}
__device__ inline void inbox_put(double* inbox, uint64_t n_local, uint64_t li, uint64_t seq, double x, double y, double yaw, double v) {
  st_sys(inbox + li, x);
  st_sys(inbox + n_local + li, y);
  st_sys(inbox + 2 * n_local + li, yaw);
  st_sys(inbox + 3 * n_local + li, v);
  __hip_atomic_store(reinterpret_cast<uint64_t*>(inbox + 4 * n_local + li), inbox_seal(seq, x, y, yaw, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
// the slot's fields (x, y, yaw, v) of step `seq`; seq == 0: whatever is there (an earlier kernel of the stream filled the slot).
// false (and *err latched) if the delivery does not show up in time -- f then holds what was there.
__device__ inline bool inbox_take(const double* inbox, uint64_t n_local, uint64_t li, uint64_t seq, uint64_t timeout_ticks,
                                  int* __restrict__ err, double f[4]) {
  const uint64_t* seal = reinterpret_cast<const uint64_t*>(inbox + 4 * n_local + li);
  auto look = [&]() {  // all five words; true if the seal vouches for the fields just read
    f[0] = ld_sys(inbox + li);
    f[1] = ld_sys(inbox + n_local + li);
    f[2] = ld_sys(inbox + 2 * n_local + li);
    f[3] = ld_sys(inbox + 3 * n_local + li);
    return __hip_atomic_load(seal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (seq == 0) {  // an earlier kernel of the stream filled the slot (RCCL transport: its inbox has no seal plane)
    f[0] = ld_sys(inbox + li);
    f[1] = ld_sys(inbox + n_local + li);
    f[2] = ld_sys(inbox + 2 * n_local + li);
    f[3] = ld_sys(inbox + 3 * n_local + li);
    return true;
  }
  uint64_t seen = look();
  if (seen == inbox_seal(seq, f[0], f[1], f[2], f[3])) return true;
  // Not there yet.  From here on only the SEAL is polled (one load instead of five), with a back-off that grows to ~3.5 us, and
  // the fields are read again only when the seal has changed: after a resample that moves most of a shard (a filter that lost
  // track: ~10^6 slots cross a boundary at once) every lane of the consuming kernel waits here, and what it polls with arrives
  // through the same memory system as the deliveries it waits for.
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;  // the filter is broken already: do not wait again
  const uint64_t t0 = wall_clock64();
  const uint64_t limit = seq <= 3 ? 10 * timeout_ticks : timeout_ticks;
  int nap = 0;
  for (;;) {
    if (nap < 4) {
      __builtin_amdgcn_s_sleep(8 << 0);
    } else if (nap < 8) {
      __builtin_amdgcn_s_sleep(32);
    } else {
      __builtin_amdgcn_s_sleep(127);
    }
    ++nap;
    const uint64_t now = __hip_atomic_load(seal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (now != seen || (nap & 3) == 0) {  // (the seal may have been the FIRST of the five words to land: look at the fields now and then anyway)
      seen = look();
      if (seen == inbox_seal(seq, f[0], f[1], f[2], f[3])) return true;
      // (a delivery in flight: its words land in any order -- keep looking, briefly at full pace)
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_s_sleep(4);
        seen = look();
        if (seen == inbox_seal(seq, f[0], f[1], f[2], f[3])) return true;
      }
    }
    if (wall_clock64() - t0 > limit) {
      if (atomicCAS(err, 0, kGaveUpInboxSlot) == 0) {  // (first give-up of the filter: leave what was seen)
        uint64_t* d = reinterpret_cast<uint64_t*>(err) + 1;
        (void)look();
        d[0] = li;
        d[1] = seq;
        d[2] = __hip_atomic_load(seal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        d[3] = inbox_seal(seq, f[0], f[1], f[2], f[3]);
        d[4] = 0;  // the sequence number (within +-16 of the awaited one) whose seal fits what IS there: a complete delivery of another step?
        for (uint64_t q = seq > 16 ? seq - 16 : 1; q <= seq + 16; ++q)
          if (inbox_seal(q, f[0], f[1], f[2], f[3]) == d[2]) d[4] = q;
      }
      return false;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The plan of a sharded systematic step in ONE launch: WMAX exchange, integer image, SUMS exchange and the marking of
// this shard's sources -- k_p2p_exchange + k_quantize_reduce + k_scan_exchange + k_mark, four launches whose ~6 us
// boundaries (resample_core.hpp, k_quantize_plan_mark) cost more than their work.  Same in-kernel hand-over as
// k_quantize_plan_mark, with the two exchanges between the GPUs inside it:
//   workgroup 0 trades the weight maximum with the peers and raises flag 0 (the global maximum);
//   every workgroup quantises its tile under it, stores its tile sums in its record, takes an arrival ticket;
//   the last arrival scans the records (prefix per tile, this shard's sums), settles Ctl as k_quantize_reduce would,
//   trades the sums with the peers (p2p_exchange -> finalize_plan: gate, base, plan) and raises flag 1;
//   every workgroup reads its prefix, the base and the global totals and marks its sources exactly as k_mark does.
// rec layout: [n_tiles][kRecWords] input records; everything handed BACK (the global maximum; every tile's prefix, the base, the
// global total, the gate decision) travels as self-vouching pairs (resample_core.hpp, TagPair / plan_pairs).
// All workgroups must be resident at once (host: n_tiles <= grid capacity).  A local wait is bounded by twelve peer
// time-outs (the first exchanges of a filter are allowed ten); giving up sets *err like a peer time-out does.
__device__ inline bool wait_flag(const uint64_t* flag, uint64_t epoch, uint64_t limit_ticks) {
  const uint64_t t0 = wall_clock64();
  while (ld_dev(flag) != epoch) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > limit_ticks) return false;
  }
  return true;
}
template <int N>
__device__ inline bool wait_pairs(const TagPair* const (&p)[N], uint64_t epoch, uint64_t limit_ticks, uint64_t (&word)[N]) {
  const uint64_t t0 = wall_clock64();
  while (!take_pairs<N>(p, epoch, word)) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > limit_ticks) return false;
  }
  return true;
}

static __global__ __launch_bounds__(kTileBlock) void k_shard_plan_mark(
    P2PPeers peers, uint64_t seq, const double* __restrict__ w, Ctl* __restrict__ ctl, ImageArgs a, uint64_t* __restrict__ rec,
    unsigned int* __restrict__ ticket, uint64_t epoch, int settle, uint64_t n_tiles, PlanArgs pa,
    unsigned int* __restrict__ markers, unsigned int* __restrict__ carry, int* __restrict__ err, uint64_t slot_pad, int wmax_posted
#if defined(RR_DEBUG_TRACE)
    , uint64_t* trace
#endif
    ) {
  constexpr int W = kTileBlock / kWave;
  __shared__ uint64_t s_posted[kMaxP2P];
  __shared__ P2PSumsTotals s_tot;
  __shared__ uint64_t s4[4 * W];
  __shared__ uint64_t s_w[W];
  __shared__ uint64_t s_pay[3];
  __shared__ double s_wmax;
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  TagPair* const pp = plan_pairs(rec);  // [tile]: its exclusive prefix; [kTileBlock + 0 .. 2]: base, global total, gate decision; [+ 5]: the global maximum
  uint64_t* const flag0 = rec + n_tiles * kRecWords;  // the two state words of the launch (= epoch when raised)
  uint64_t* const flag1 = flag0 + 1;
  const uint64_t limit = 12 * peers.timeout_ticks;
#if defined(RR_PLAN_TIMELINE)
  // instrumented build (tools/shard_plan_timeline.py): 0 start, 1 the global maximum known, 2 record stored, 3 ticket taken,
  // 4 prefix + totals seen, 5 markers written; the last arrival alone: 6 records scanned (before the SUMS exchange), 7 sums traded
  uint64_t* const tl = rec + (kTileBlock + 1) * kRecWords + 16;
#endif
  RR_TL(0);
  // reads of Ctl that the last arrival's settle / finalize could race with come first (they precede this workgroup's ticket)
  const bool forced_uniform = a.honour_uniform_flag && ctl->weights_uniform;
  // ---- 0: the global maximum
  if (wmax_posted) {
    // RR_P2P_WMAX_EARLY=1 (off by default, pf_engine.hip: rr_pf_shard_step_p2p): the ranks' maxima are already on their way, or
    // here -- the last workgroup of every rank's STEP kernel sent its shard's maximum to every mailbox as it finished
    // (k_step_lazy<kSrcWindow>, WindowArgs.post_peers).  Every workgroup takes them from this rank's own mailbox itself -- n_ranks
    // self-vouching pairs, one round trip to local memory -- instead of workgroup 0 opening the launch with an exchange (store,
    // fabric, poll) and a flag for the others to wait on.  This kernel gets 1.65 us shorter by it and the step kernel as much
    // longer (profiles/r05o_wmax_early.md).  (The records cannot be overwritten while somebody still reads them: a rank sends the
    // next maximum after its plan kernel has traded the SUMS of this sequence number, which every rank posts only after ALL its
    // workgroups have passed this point -- they need the maximum to form the records the last arrival adds up.)
    if (tid < peers.n_ranks) {
      const P2PPairSlot* in = &peers.mbox[peers.rank]->wmax[tid];
      uint64_t got = 0, tag = 0;
      bool ok = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;  // (a transport that gave up stays down)
      const uint64_t t0 = wall_clock64();
      while (ok) {
        load_pair_sys(&in->p[0], got, tag);
        if (tag == p2p_tag(seq, got)) break;
        if (wall_clock64() - t0 > (seq <= 3 ? 10 * peers.timeout_ticks : peers.timeout_ticks)) {
          ok = false;
          if (atomicCAS(err, 0, kGaveUpWmax) == 0) {  // the first give-up names the wait and says what it saw
            uint64_t* d = reinterpret_cast<uint64_t*>(err) + 1;
            d[0] = (uint64_t)tid;
            d[1] = seq;
            d[2] = tag ^ (got * 0x9E3779B97F4A7C15ull);
            d[3] = seq;
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      s_posted[tid] = ok ? got : ~0ull;
    }
    __syncthreads();
    if (tid == 0) {
      double m = 0.0;
      bool ok = true;
      for (int k = 0; k < peers.n_ranks; ++k) {
        ok &= s_posted[k] != ~0ull;
        const double wk = rr_u2d(s_posted[k]);
        if (wk > m) m = wk;
      }
      s_wmax = ok ? m : 0.0;  // (gave up: the SUMS exchange below sees the flag, shuts the gate, and nothing is marked)
    }
  } else {
    if (blockIdx.x == 0) {
      const uint64_t local_bits = ctl->wmax_bits;
      p2p_exchange(peers, kP2PWmax, seq, local_bits, 0, 0, ctl, &s_wmax, pa, err);  // thread 0 leaves the maximum in s_wmax
      if (tid == 0) {
        put_pair(&pp[kTileBlock + 5], rr_d2u(s_wmax), epoch);  // vouches for itself (resample_core.hpp, TagPair): the flag follows at once
        st_dev(flag0, epoch);
#if defined(RR_DEBUG_TRACE)
        if (trace) {
          trace[10] = local_bits;
          trace[11] = rr_d2u(s_wmax);
          trace[12] = seq;
          trace[13] = epoch;
        }
#endif
      }
    }
    if (tid == 0) {
      const TagPair* const want[1] = {&pp[kTileBlock + 5]};
      uint64_t got[1] = {0};
      if (!wait_flag(flag0, epoch, limit) || !wait_pairs<1>(want, epoch, limit, got)) (void)atomicCAS(err, 0, kGaveUpPlanFlag);
      s_wmax = rr_u2d(got[0]);
    }
  }
  __syncthreads();
  RR_TL(1);
  const double wmax = s_wmax;
  // ---- A: the integer image of this tile (k_quantize_plan_mark, phase A)
  const bool usable = !forced_uniform && wmax > 0.0 && wmax < INFINITY;
  const int mode = usable ? (int)kImageWeights : (forced_uniform ? (int)kImageUniform : a.degenerate);
  const int shift = usable ? rr_fix_shift(wmax, a.n_global) : 0;
  TileScan t;
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)tid * kItems;
  uint64_t run = 0;
  u128 q2 = {0, 0};
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    t.q[j] = quantize_at(w, i0 + j, a.n, mode, shift, a.gid0, a.n_global);
    run += t.q[j];
    t.c[j] = run;
    u128 sq;
    rr_mul64wide(t.q[j], t.q[j], &sq.hi, &sq.lo);
    q2 = add128(q2, sq);
  }
  const uint64_t incl = wave_scan_u64(run, lane);
  q2 = wave_sum_u128(q2);
  if (lane == 63) s_w[wv] = incl;
  if (lane == 0) {
    s4[wv] = q2.hi;
    s4[W + wv] = q2.lo;
  }
  __syncthreads();
  uint64_t off = incl - run;
  for (int k = 0; k < wv; ++k) off += s_w[k];
  t.thread_off = off;
  if (tid == 0) {
    uint64_t tt = 0;
    u128 qq = {0, 0};
    for (int k = 0; k < W; ++k) {
      tt += s_w[k];
      qq = add128(qq, u128{s4[k], s4[W + k]});
    }
    uint64_t* r = rec + (uint64_t)blockIdx.x * kRecWords;
    st_dev(&r[0], tt);
    st_dev(&r[1], qq.hi);
    st_dev(&r[2], qq.lo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RR_TL(2);
    s_last = last_arrival(ticket, blockIdx.x, (unsigned int)n_tiles, /*fence=*/false) ? 1 : 0;
    RR_TL(3);
  }
  __syncthreads();
  // ---- the last arrival: prefixes, this shard's sums, Ctl, the exchange with the peers, flag 1
  if (s_last) {
    uint64_t tk = 0;
    u128 qk = {0, 0};
    if ((uint64_t)tid < n_tiles) {
      const uint64_t* r = rec + (uint64_t)tid * kRecWords;
      tk = ld_dev(&r[0]);
      qk.hi = ld_dev(&r[1]);
      qk.lo = ld_dev(&r[2]);
    }
    const uint64_t inc = wave_scan_u64(tk, lane);
    const u128 qw = wave_sum_u128(qk);
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    if (lane == 0) {
      s4[wv] = qw.hi;
      s4[W + wv] = qw.lo;
    }
    __syncthreads();
    uint64_t wave_off = 0;
    for (int k = 0; k < wv; ++k) wave_off += s_w[k];
    if ((uint64_t)tid < n_tiles) put_pair(&pp[tid], wave_off + inc - tk, epoch);
    if (tid == 0) {
      uint64_t tt = 0;
      u128 qq = {0, 0};
      for (int k = 0; k < W; ++k) {
        tt += s_w[k];
        qq = add128(qq, u128{s4[k], s4[W + k]});
      }
      s_pay[0] = tt;
      s_pay[1] = qq.hi;
      s_pay[2] = qq.lo;
      // what k_quantize_reduce's first thread leaves in Ctl (finalize_plan, inside the exchange, reads it)
      if (settle && ctl->pending) {
        ctl->cur ^= 1;
        ctl->pending = 0;
      }
      ctl->usable = usable ? 1 : 0;
      ctl->image_mode = mode;
      ctl->shift = shift;
      ctl->wmax = wmax;
      ctl->total_local = tt;
    }
    __syncthreads();
    RR_TL(6);
    p2p_exchange(peers, kP2PSums, seq, s_pay[0], s_pay[1], s_pay[2], ctl, &ctl->wmax, pa, err, &s_tot);
    RR_TL(7);
    if (tid == 0) {
      // What everybody is waiting for goes out FIRST: this rank's base, the global total, the gate's decision.  What finalize_plan
      // leaves in Ctl -- the sums, the plan, the served range: a 64-bit division, two 128-bit slot counts -- is for LATER kernels
      // and follows below, while the other workgroups mark (until round 5 it came first: 2.1 us of one thread on everybody's
      // critical path, tools/shard_plan_timeline.py).
      TileSums ts;
      ts.pre = 0;
      ts.tot = s_tot.total;
      ts.q2 = s_tot.q2;
      const int fire = s_tot.ok ? gate_decision(mode, ts, pa) : 0;  // (gave up: Ctl.fired is 0, nothing is marked)
      put_pair(&pp[kTileBlock + 0], s_tot.base, epoch);
      put_pair(&pp[kTileBlock + 1], s_tot.total, epoch);
      put_pair(&pp[kTileBlock + 2], (uint64_t)fire, epoch);
      st_dev(flag1, epoch);  // (right behind the pairs, not after their acknowledgement: they vouch for themselves)
    }
  }
  // the plan's one uniform: drawn while this workgroup has nothing to do but wait; the last arrival draws it now that everybody
  // has been told, and leaves in Ctl what later kernels need
  double rho = s_last ? plan_rho(pa) : plan_rho_early(pa);
  if (s_last && tid == 0 && s_tot.ok) finalize_plan(ctl, s_tot.total, s_tot.base, s_tot.local, s_tot.q2, pa, rho, mode, shift);
#if defined(RR_DEBUG_TRACE)
  if (s_last && tid == 0 && trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace[14] = s_pay[0];
    trace[15] = ctl->total;
    trace[16] = (uint64_t)ctl->fired;
    trace[17] = (uint64_t)ctl->pending;
    trace[18] = (uint64_t)ctl->cur;
    trace[19] = ctl->base;
    trace[20] = ctl->q2_lo;
    trace[21] = ctl->served_first;
    trace[22] = ctl->served_count;
    trace[23] = (uint64_t)(unsigned int)ctl->shift;
    trace[24] = (uint64_t)blockIdx.x;
    trace[25] = rr_d2u(wmax);
    trace[26] = rr_d2u(ctl->rho);
    trace[27] = (uint64_t)ctl->image_mode;
    trace[28] = wall_clock64();
  }
#endif
  // ---- everybody: this tile's prefix, the base, the global total and the gate decision, as soon as they vouch for this launch
  if (tid == 0) {
    const TagPair* const want[4] = {&pp[blockIdx.x], &pp[kTileBlock + 0], &pp[kTileBlock + 1], &pp[kTileBlock + 2]};
    uint64_t got[4] = {0, 0, 0, 0};
    const bool ok = wait_flag(flag1, epoch, limit) && wait_pairs<4>(want, epoch, limit, got);
    if (!ok) (void)atomicCAS(err, 0, kGaveUpPlanFlag);
    s4[0] = got[0];
    s4[1] = got[1];
    s4[2] = got[2];
    s4[3] = ok ? got[3] : 0;  // gave up: as if the gate were shut -- nothing is marked with unknown sums
  }
  __syncthreads();
  RR_TL(4);
  const uint64_t pre = s4[0], base = s4[1], total = s4[2];
  const bool fired = s4[3] != 0;
#if defined(RR_DEBUG_TRACE)
  if (trace && tid == 0 && blockIdx.x < 3) trace[32 + 4 * blockIdx.x] = pre, trace[33 + 4 * blockIdx.x] = base, trace[34 + 4 * blockIdx.x] = total,
                                           trace[35 + 4 * blockIdx.x] = (uint64_t)fired | ((uint64_t)(unsigned int)shift << 8) | ((uint64_t)mode << 40);
#endif
  if (fired) {
    // ---- B: k_mark
    const rr_sys_plan plan = rr_sys_plan_make(rho, total, pa.n_global);
    // marker position of global slot s: s + slot_pad (resolve_tile_window); the overhang over the own block is
    // k_push_window's (see docs/DESIGN_NOTES.md section 5 for the variant that delivered it from here, and why it lost)
    mark_sources(t, base + pre + t.thread_off, i0, a.n, plan, total, (uint64_t)0 - slot_pad, markers, carry);
  }
  RR_TL(5);
}

// ---- host side: what a handle owns for the transport
struct P2PState {
  P2PMailbox* mbox = nullptr;  // fine-grained: peers write their records here
  double* inbox = nullptr;     // fine-grained: peers deliver cross-rank particles here (engine-defined layout)
  size_t inbox_doubles = 0;
  P2PPeers peers{};            // device pointers of every rank's mailbox and slab
  bool ready = false;
  int n_sharing = 1;               // ranks of this filter whose shard lives on THIS device (this one included)
  int share_ordinal = 0;           // ... and how many of them have a lower rank than this one
  void* opened[3 * kMaxP2P] = {};  // IPC mappings to close
  int n_opened = 0;
  uint64_t seq = 0;
  uint64_t* scratch = nullptr;  // [4] local payload of the stand-alone exchanges (behind 3 * kMaxP2P words no longer used)
  int* err = nullptr;           // device: set when a wait timed out
  int* err_host = nullptr;
  P2PPeers* peers_dev = nullptr;       // `peers` in device memory, for the kernel that posts a record without being handed the struct
                                       // (k_step_lazy<kSrcWindow>: WindowArgs.post_peers)
  unsigned int* post_ticket = nullptr; // kTicketWords: "the last workgroup of the step kernel" (zero between launches)

  uint64_t* local3() const { return scratch + 3 * kMaxP2P; }

  void teardown() {
    for (int k = 0; k < n_opened; ++k) (void)hipIpcCloseMemHandle(opened[k]);
    n_opened = 0;
    (void)hipFree(mbox);
    (void)hipFree(inbox);
    inbox = nullptr;
    (void)hipFree(scratch);
    (void)hipFree(err);
    (void)hipFree(peers_dev);
    (void)hipFree(post_ticket);
    peers_dev = nullptr;
    post_ticket = nullptr;
    if (err_host) (void)hipHostFree(err_host);
    mbox = nullptr;
    scratch = nullptr;
    err = nullptr;
    err_host = nullptr;
    ready = false;
  }

  static uint64_t timeout_ticks_from_env() {
    const char* e = std::getenv("RR_P2P_TIMEOUT_MS");
    double ms = e ? std::atof(e) : 2000.0;
    if (!(ms > 0.0)) ms = 2000.0;
    return (uint64_t)(ms * 1e5);  // wall_clock64 ticks at 100 MHz
  }

  // (re)connecting: forget every record of an earlier connection.  Called before the handles are
  // exchanged (export) or before any rank is linked (in-process), i.e. before a peer can write.
  rr_status reset_records() {
    RR_HIP_TRY(hipMemset(mbox, 0, sizeof(P2PMailbox)));
    RR_HIP_TRY(hipMemset(err, 0, 64));
    RR_HIP_TRY(hipMemset(inbox, 0, inbox_doubles * sizeof(double)));  // (the seal plane: sequence numbers start over)
    RR_HIP_TRY(hipDeviceSynchronize());
    seq = 0;
    return RR_OK;
  }

  rr_status local_setup(size_t inbox_doubles_) {
    if (mbox) return RR_OK;
    inbox_doubles = inbox_doubles_;
    RR_HIP_TRY(hipExtMallocWithFlags((void**)&inbox, inbox_doubles * sizeof(double), hipDeviceMallocFinegrained));
    RR_HIP_TRY(hipMemset(inbox, 0, inbox_doubles * sizeof(double)));  // the engines' seal planes must not hold an earlier filter's seals
    RR_HIP_TRY(hipExtMallocWithFlags((void**)&mbox, sizeof(P2PMailbox), hipDeviceMallocFinegrained));
    RR_HIP_TRY(hipMemset(mbox, 0, sizeof(P2PMailbox)));
    RR_HIP_TRY(rr::dev_malloc(&scratch, (3 * kMaxP2P + 4) * sizeof(uint64_t)));
    RR_HIP_TRY(rr::dev_malloc(&err, 64));  // the flag + 7 words of detail a give-up leaves behind (kP2PErrWords)
    RR_HIP_TRY(hipMemset(err, 0, 64));
    RR_HIP_TRY(hipHostMalloc(&err_host, 64));
    RR_HIP_TRY(rr::dev_malloc(&peers_dev, sizeof(P2PPeers)));
    RR_HIP_TRY(rr::dev_malloc(&post_ticket, kTicketWords * sizeof(unsigned int)));
    RR_HIP_TRY(hipMemset(post_ticket, 0, kTicketWords * sizeof(unsigned int)));
    RR_HIP_TRY(hipDeviceSynchronize());
    return RR_OK;
  }

  // the connection is made: `peers` goes to device memory as well (synchronous: nothing of this handle runs yet)
  rr_status publish_peers() {
    RR_HIP_TRY(hipMemcpy(peers_dev, &peers, sizeof(P2PPeers), hipMemcpyHostToDevice));
    RR_HIP_TRY(hipMemset(post_ticket, 0, kTicketWords * sizeof(unsigned int)));
    RR_HIP_TRY(hipDeviceSynchronize());  // (a hipMemset may return before it has run, and the shard's stream does not wait for the null stream)
    return RR_OK;
  }

  rr_status export_handles(double* slab, size_t inbox_doubles_, uint8_t out[kP2PHandleBytes], size_t slab_bytes = 0) {
    const bool again = mbox != nullptr;
    rr_status s = local_setup(inbox_doubles_);
    if (s != RR_OK) return s;
    if (again && (s = reset_records()) != RR_OK) return s;
    static_assert(3 * sizeof(hipIpcMemHandle_t) <= kP2PHandleBytes, "handle blob too small");
    // An allocation of 2 GiB or more must not go through hipIpc on this driver (ROCm 7.2, dmabuf IPC): hipIpcOpenMemHandle of the
    // importing rank does not return (round 6, tools/ipc_connect_probe.py: 2 processes x a 2.4 GB slab hung for minutes; 1.2 GB
    // opens in 0.1 s).  FastSLAM shards therefore export no slab at all (slab == nullptr: their peers only ever write the inbox),
    // and an inbox or slab that large is refused here with a message instead of a hang there.
    constexpr size_t kIpcMaxBytes = (size_t)1 << 31;
    if (inbox_doubles_ * sizeof(double) >= kIpcMaxBytes || slab_bytes >= kIpcMaxBytes)
      return fail(RR_INVALID_PARAMETER, "peer-to-peer transport: this shard's inbox / state slab is 2 GiB or larger, which hipIpc cannot map on this "
                                        "driver (the importing rank would hang): use more ranks (smaller shards) or the RCCL transport");
    hipIpcMemHandle_t hs[3];
    std::memset(hs, 0, sizeof hs);
    if (slab) RR_HIP_TRY(hipIpcGetMemHandle(&hs[0], slab));
    RR_HIP_TRY(hipIpcGetMemHandle(&hs[1], mbox));
    RR_HIP_TRY(hipIpcGetMemHandle(&hs[2], inbox));
    std::memset(out, 0, kP2PHandleBytes);
    std::memcpy(out, hs, sizeof hs);
    // which physical device the shard lives on (ranks that share one must leave each other room, see n_sharing)
    static_assert(sizeof hs + kBusIdBytes <= kP2PHandleBytes, "handle blob too small");
    int dev = 0;
    RR_HIP_TRY(hipGetDevice(&dev));
    RR_HIP_TRY(hipDeviceGetPCIBusId(reinterpret_cast<char*>(out) + sizeof hs, kBusIdBytes, dev));
    return RR_OK;
  }

  rr_status connect_ipc(double* slab, size_t inbox_doubles_, const uint8_t* all_handles, int n_ranks, int rank) {
    rr_status s = local_setup(inbox_doubles_);
    if (s != RR_OK) return s;
    P2PPeers p{};
    p.n_ranks = n_ranks;
    p.rank = rank;
    char own_bus[kBusIdBytes] = {};
    int dev = 0;
    RR_HIP_TRY(hipGetDevice(&dev));
    RR_HIP_TRY(hipDeviceGetPCIBusId(own_bus, kBusIdBytes, dev));
    n_sharing = 1;
    share_ordinal = 0;
    for (int g = 0; g < n_ranks; ++g) {
      if (g == rank) {
        p.slab[g] = slab;
        p.mbox[g] = mbox;
        p.inbox[g] = inbox;
        continue;
      }
      hipIpcMemHandle_t hs[3];
      std::memcpy(hs, all_handles + (size_t)g * kP2PHandleBytes, sizeof hs);
      const char* bus = reinterpret_cast<const char*>(all_handles) + (size_t)g * kP2PHandleBytes + sizeof hs;
      if (bus[0] && std::strncmp(bus, own_bus, kBusIdBytes) == 0) {
        n_sharing += 1;
        if (g < rank) share_ordinal += 1;
      }
      void *ps = nullptr, *pm = nullptr, *pi = nullptr;
      static const hipIpcMemHandle_t kNoHandle{};
      if (std::memcmp(&hs[0], &kNoHandle, sizeof kNoHandle) != 0) {  // (FastSLAM shards export no slab: nobody writes a peer's planes)
        RR_HIP_TRY(hipIpcOpenMemHandle(&ps, hs[0], hipIpcMemLazyEnablePeerAccess));
        opened[n_opened++] = ps;
      }
      RR_HIP_TRY(hipIpcOpenMemHandle(&pm, hs[1], hipIpcMemLazyEnablePeerAccess));
      opened[n_opened++] = pm;
      RR_HIP_TRY(hipIpcOpenMemHandle(&pi, hs[2], hipIpcMemLazyEnablePeerAccess));
      opened[n_opened++] = pi;
      p.slab[g] = (double*)ps;
      p.mbox[g] = (P2PMailbox*)pm;
      p.inbox[g] = (double*)pi;
    }
    p.timeout_ticks = timeout_ticks_from_env();
    peers = p;
    if ((s = publish_peers()) != RR_OK) return s;
    ready = true;
    seq = 0;
    return RR_OK;
  }

  // RR_RUNTIME_ERROR once a peer wait has given up: from then on no exchange runs and no resample is
  // applied, so results read back after this point would silently be those of a non-resampled filter
  rr_status check(hipStream_t stream) {
    if (!ready || !err) return RR_OK;
    RR_HIP_TRY(hipMemcpyAsync(err_host, err, 64, hipMemcpyDeviceToHost, stream));
    RR_HIP_TRY(hipStreamSynchronize(stream));
    if (*err_host) {
      static const char* const what[] = {"", "the peers' weight maxima", "the peers' integer sums", "a peer's record",
                                         "a particle a peer serves (inbox slot)", "the plan kernel's own hand-over"};
      const int k = *err_host > 0 && *err_host <= 5 ? *err_host : 3;
      // what the waiter saw when it gave up (exchange waits only): peer, the sequence number it waited for, the one that was there
      const uint64_t* d = reinterpret_cast<const uint64_t*>(err_host) + 1;
      char detail[320] = "";
      if (k <= 3) std::snprintf(detail, sizeof detail, " [peer %llu: waited for sequence %llu, its slot held %llu; this rank had posted %llu]",
                                (unsigned long long)d[0], (unsigned long long)d[1], (unsigned long long)d[2], (unsigned long long)d[3]);
      if (k == 4) std::snprintf(detail, sizeof detail, " [local slot %llu of step sequence %llu: seal there %016llx, seal of the fields read %016llx; the slot holds a complete delivery of sequence %llu (0: of none nearby)]",
                                (unsigned long long)d[0], (unsigned long long)d[1], (unsigned long long)d[2], (unsigned long long)d[3], (unsigned long long)d[4]);
      return fail(RR_RUNTIME_ERROR, std::string("peer-to-peer exchange: a wait for a peer's record timed out (RR_P2P_TIMEOUT_MS; waiting for ") +
                                        what[k] + ")" + detail + "; the sharded filter is no longer consistent -- reconnect or recreate it");
    }
    return RR_OK;
  }

  rr_status status(hipStream_t stream, int32_t* timed_out) {
    *timed_out = 0;
    if (!err) return RR_OK;
    RR_HIP_TRY(hipMemcpyAsync(err_host, err, sizeof(int), hipMemcpyDeviceToHost, stream));
    RR_HIP_TRY(hipStreamSynchronize(stream));
    *timed_out = *err_host != 0 ? 1 : 0;
    return RR_OK;
  }
};

// in-process wiring: slabs[g], states[g], devices[g] of rank g
inline rr_status p2p_link_local(P2PState* const* states, double* const* slabs, const size_t* inbox_doubles, const int* devices,
                                int n_ranks) {
  for (int g = 0; g < n_ranks; ++g) {
    RR_HIP_TRY(hipSetDevice(devices[g]));
    const bool again = states[g]->mbox != nullptr;
    rr_status s = states[g]->local_setup(inbox_doubles[g]);
    if (s != RR_OK) return s;
    if (again && (s = states[g]->reset_records()) != RR_OK) return s;
  }
  for (int g = 0; g < n_ranks; ++g) {
    P2PPeers p{};
    p.n_ranks = n_ranks;
    p.rank = g;
    RR_HIP_TRY(hipSetDevice(devices[g]));
    for (int k = 0; k < n_ranks; ++k) {
      if (devices[k] != devices[g]) {
        int can = 0;
        RR_HIP_TRY(hipDeviceCanAccessPeer(&can, devices[g], devices[k]));
        if (!can) return fail(RR_RUNTIME_ERROR, "devices cannot access each other's memory");
        hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return fail(RR_RUNTIME_ERROR, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
      p.slab[k] = slabs[k];
      p.mbox[k] = states[k]->mbox;
      p.inbox[k] = states[k]->inbox;
    }
    p.timeout_ticks = P2PState::timeout_ticks_from_env();
    states[g]->peers = p;
    if (rr_status s = states[g]->publish_peers(); s != RR_OK) return s;
    states[g]->ready = true;
    states[g]->seq = 0;
    states[g]->n_sharing = 0;
    states[g]->share_ordinal = 0;
    for (int k = 0; k < n_ranks; ++k) {
      states[g]->n_sharing += devices[k] == devices[g] ? 1 : 0;
      if (k < g && devices[k] == devices[g]) states[g]->share_ordinal += 1;
    }
  }
  return RR_OK;
}

}  // namespace rr
