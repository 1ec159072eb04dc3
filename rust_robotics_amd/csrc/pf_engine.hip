// pf_engine.hip -- MI355X (gfx950) particle-filter / fixed-N MCL engine behind include/rr_pf.h.
//
// Replaces the CPU hot path of
//   rust_robotics_localization/src/particle_filter.rs:255-301 (propagate), :310-334 (weight),
//   :337-345,416-473 (N_eff gate + resample), :382-413 (mean / covariance)
//   rust_robotics_localization/src/monte_carlo_localization.rs:209-288,322-365 (fixed-N mode)
// (paths under /root/reference/crates).  Not a translation: the particle set lives in HBM as
// structure-of-arrays, the observation block is staged in LDS, exact reductions (integer sums, maxima) run on DPP,
// and the resampling CDF is an integer reduce-then-scan whose value is independent of summation order
// (include/rr_pf_spec.h).
//
// Kernels (one HIP stream per filter, no host synchronisation inside a step):
//  the fused step (rr_pf_step_async, systematic resampling): 2 launches
//   k_step_lazy            resolve the previous resample's markers, read x,y,yaw through them, propagate, weigh,
//                          write x,y,yaw,v,w, running maximum of w                                 72 B / particle
//   k_quantize_plan_mark   integer image + tile sums handed over inside the launch + gate + slot-run markers
//                          (+ the mean try_step returns); beyond 2^20 particles k_quantize_reduce + k_plan_mark
//                          (resample_core.hpp)                                                      8-12 B / particle
//  multinomial: k_step_lazy<kSrcDraw, PACKED> (draws and searches the previous resample's sources for its own slots through
//               the guide table over the target space), k_quantize_reduce, k_plan_cdf (CDF + guide markers), k_guide_resolve;
//               k_resample_guide_mn is the same search as a launch of its own (accessors, RR_MN_DEFER=0; then
//               k_step_lazy<kSrcLidx, PACKED> reads through lidx); sharded / adaptive / beyond 8.4e6 particles:
//               k_resample_gather_mn, coarse table of the CDF in LDS
//  the separate entry points (predict / update / resample, the RCCL sharded step, the adaptive filter):
//   k_propagate_weight     x,y,yaw -> x,y,yaw,v,w + maximum of w                                   64 B / particle
//   k_quantize_reduce      w -> per-tile integer totals, sum q^2                                    8 B / particle
//   k_scan_tiles           tile totals -> exclusive offsets, gate decision (single block)
//   k_cdf / k_mark         inclusive integer CDF / slot-run markers of this shard                  16 B / particle
//   k_resolve_gather, k_resample_gather_mn   markers or CDF search -> SoA gather into the other buffer set
//   k_moments(+final)      weighted first/second moments about particle 0                          40 B / particle
//  sharded over the peer-to-peer transport: k_step_lazy<sharded>, k_shard_plan_mark (p2p_core.hpp), k_resolve_push
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "p2p_core.hpp"
#include "rccl_core.hpp"
#include "resample_core.hpp"
#include "resident_core.hpp"
#include "rr_common.hpp"
#include "rr_pf.h"
#include "rr_pf_spec.h"

namespace rr {
std::string& last_error_slot() {
  static thread_local std::string s;
  return s;
}
}  // namespace rr

using rr::fail;
using rr::u128;
using rr::Ctl;
using rr::ImageArgs;
using rr::PlanArgs;
using rr::kBlock;
using rr::kTile;
using rr::kScanThreads;
using rr::kMaxObsKernarg;
using rr::kMomentBlocks;
using rr::kNumMoments;
using rr::P2PPeers;
using rr::kMaxP2P;
using rr::rccl;
using rr::rccl_load;
using rr::Rccl;
using rr::kNcclUint64;
using rr::kNcclFloat64;
using rr::kNcclMax;
using rr::ncclUniqueIdPod;

namespace {

struct Bufs {
  double* x[2];
  double* y[2];
  double* yaw[2];
  double* v[2];
};

struct ObsArg {
  double v[3 * kMaxObsKernarg];
};

struct StepParams {
  uint64_t n;          // particles in this shard
  uint64_t n_global;
  uint64_t first_gid;
  uint64_t seed;
  unsigned int step;
  unsigned int rstep;
  int n_obs;
  int lik_mode;
  double u0, u1, dt;
  double sigma_v, sigma_w;
  rr_pf_lik lik;
  int dyn_n;  // KLD-adaptive filter: the particle count is Ctl.n_active (k_propagate_weight only)
};

// ------------------------------------------------------------------------------------------
// K1: propagate + weight.  One thread per particle; the observation block (d, lx, ly) x n_obs
// is staged once per workgroup in LDS and read back with wave-uniform addresses (broadcast).
// Reads x,y,yaw (24 B), writes x,y,yaw,v,w (40 B).  The per-particle maximum weight is reduced
// across the wave with shuffles, across waves through LDS, and folded into Ctl with one
// atomicMax per workgroup (bit pattern of a non-negative double is order preserving).
template <bool PREDICT, bool WEIGHT, bool EXPLICIT_NOISE, bool OBS_KERNARG>
__global__ __launch_bounds__(kBlock) void k_propagate_weight(Bufs b, double* __restrict__ w,
                                                            Ctl* __restrict__ ctl, StepParams p,
                                                            ObsArg obs_arg,
                                                            const double* __restrict__ obs_dev,
                                                            const double* __restrict__ nv,
                                                            const double* __restrict__ nw) {
  extern __shared__ double s_obs[];
  __shared__ double s_wmax[kBlock / rr::kWave];
  const int tid = threadIdx.x;
  if (WEIGHT) {
    for (int i = tid; i < 3 * p.n_obs; i += kBlock) s_obs[i] = OBS_KERNARG ? obs_arg.v[i] : obs_dev[i];
    __syncthreads();
  }
  const int cur = ctl->cur;
  if (p.dyn_n) p.n = ctl->n_active;
  double* __restrict__ bx = b.x[cur];
  double* __restrict__ by = b.y[cur];
  double* __restrict__ byaw = b.yaw[cur];
  double* __restrict__ bv = b.v[cur];
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  uint64_t i = (uint64_t)blockIdx.x * kBlock + tid;
  double wmax_local = 0.0;
  // grid-stride over the particles with the next particle's state prefetched while the current
  // one is being computed (the kernel is FP64-VALU bound; this keeps its HBM reads off the
  // critical path)
  double x = 0.0, y = 0.0, yaw = 0.0;
  if (i < p.n) {
    x = bx[i];
    y = by[i];
    if (PREDICT) yaw = byaw[i];
  }
  while (i < p.n) {
    const uint64_t inext = i + stride;
    double nx = 0.0, ny = 0.0, nyaw = 0.0;
    if (inext < p.n) {
      nx = bx[inext];
      ny = by[inext];
      if (PREDICT) nyaw = byaw[inext];
    }
    if (PREDICT) {
      double v, a, c;
      if (EXPLICIT_NOISE) {
        a = nv[i];
        c = nw[i];
      } else {
        rr_pf_motion_noise(p.seed, p.step, p.first_gid + i, p.sigma_v, p.sigma_w, &a, &c);
      }
      rr_pf_propagate_one(&x, &y, &yaw, &v, p.u0, p.u1, p.dt, a, c);
      bx[i] = x;
      by[i] = y;
      byaw[i] = yaw;
      bv[i] = v;
    }
    if (WEIGHT) {
      const double wgt = p.lik_mode == RR_LIK_PRODUCT ? rr_pf_weight_product(x, y, s_obs, p.n_obs, p.lik)
                                                      : rr_pf_weight_fused(x, y, s_obs, p.n_obs, p.lik);
      w[i] = wgt;
      if (wgt > wmax_local) wmax_local = wgt;  // NaN and negatives drop out
    }
    i = inext;
    x = nx;
    y = ny;
    yaw = nyaw;
  }
  if (WEIGHT) {
    double m = rr::wave_max(wmax_local);
    if ((tid & 63) == 0) s_wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      double bm = s_wmax[0];
      for (int k = 1; k < kBlock / rr::kWave; ++k) bm = s_wmax[k] > bm ? s_wmax[k] : bm;
      if (bm > 0.0) rr::atomic_max_u64(&ctl->wmax_bits, rr_d2u(bm));
      if (blockIdx.x == 0) ctl->weights_uniform = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1 of the fused step (rr_pf_step_async, systematic resampling): propagate + weight with the
// previous step's resample gather folded into its loads.  If Ctl.pending is set the last plan
// kernel left markers instead of moved particles: each workgroup resolves the markers of its
// 512-slot tiles to source indices (running maximum, rr::resolve_tile) and reads x,y,yaw of slot k
// from particle idx[k] of the live buffer set, writing the propagated particle to slot k of the
// OTHER set; k_quantize_reduce (next in the stream) then flips Ctl.cur.  Without a pending
// resample it runs in place.  This removes a whole 72 B/particle pass over HBM per step.
//
// Sharded (rr_pf_shard_step_p2p, StepSrc::kSrcWindow): the markers live in the GLOBAL slot index (rr::resolve_tile_window);
// the own slots inside the window this shard serves are resolved and read exactly as on one GPU, the few outside it were
// delivered into this rank's inbox by the peer that serves them (k_push_window) -- such a slot waits, bounded, until the seal
// of the previous step fits its fields (rr::inbox_take).  Round 2 resolved every served slot in a separate launch
// (k_resolve_push -> lidx, 7.9 us at 1e6 particles) and waited for a DONE message in a fourth one.
constexpr unsigned int kInPlace = 0xffffffffu;
constexpr unsigned kPushGrid = 64;  // workgroups of k_push_window (grid-stride over the foreign tiles: usually a handful)

//
// Shape of the kernel, each point measured at 1e6 x 32 (A/B on one box, tools/ab_bench.sh):
//  * ONE 512-slot tile per workgroup, no grid-stride loop.  With the loop every polynomial coefficient of log / sincos /
//    exp was loop-invariant, the compiler kept them all in registers for the whole kernel (128 VGPRs, 106 SGPRs with
//    spills into VGPR lanes, 4 waves per SIMD); without it they are materialised where they are used: 52 VGPRs,
//    8 waves per SIMD, 44.7 -> 37.7 us.
//  * the likelihood form is a template argument (only one form's code and constants in the kernel);
//  * the weights of the thread's two rows come from ONE pass over the observation block (rr_pf_weight_fused_rows):
//    half the LDS reads and two independent chains per wave, -3 us;
//  * the motion noise (a function of seed, step and slot only) is evaluated before the marker / particle loads are
//    needed, so the kernel's first microseconds are not spent waiting, -1.3 us;
//  * the weight maximum goes through rr::atomic_max_u64's look-first form: one same-address atomic per workgroup cost
//    14.6 us of queueing at L = 1 and 2 us at L = 32.
//  * PACKED (the lazy MULTINOMIAL resample of one GPU: sources are iid draws, so the reads through `lidx` are random):
//    the kernel also keeps an array-of-structures mirror {x, y, yaw, v} of the set it writes, and reads a source
//    particle from the mirror of the live set -- one random 32-byte record instead of three random 8-byte words
//    from three arrays.  The mirror is only ever read while a resample is pending, and a resample only becomes
//    pending right after a launch of this kernel has written the mirror of the then-live set.
// where the sources of a pending resample come from
enum StepSrc {
  kSrcMarkers = 0,  // one GPU, systematic: slot-run markers, resolved here (rr::resolve_tile)
  kSrcLidx = 1,     // one GPU, multinomial: the search kernel left one source index per slot
  kSrcWindow = 2,   // a shard of the peer-to-peer transport: markers over the global slot index (rr::resolve_tile_window);
                    // own slots outside the window this shard serves were delivered into the inbox by a peer
  kSrcDraw = 3      // one GPU, multinomial, the search not run yet: this kernel draws and searches for its own slots
                    // (mn_guide_search) -- the latency-bound search hides under this kernel's FP64 work instead of being a
                    // launch of its own (k_resample_guide_mn: 17.7 us at 1e6 particles)
};
constexpr size_t kEstSlotWords = 4 * (kBlock / rr::kWave);  // doubles per slot tile of the deferred estimate: one quadruple per wave
struct WindowArgs {
  const double* inbox;         // this rank's inbox [4 fields + tag][n]
  int* err;
  uint64_t pad;                // marker position of global slot s = s + pad
  uint64_t wait_seq;           // the tag a peer-served slot must carry: the sequence number of the step being consumed
  uint64_t timeout_ticks;
  int n_ranks;                 // 0: nothing to wait for (RCCL transport: an earlier kernel of the stream filled the inbox)
  // kSrcDraw: the pending multinomial resample's CDF, guide table and draw stream
  const uint64_t* cdf;
  const unsigned int* guide;
  uint64_t n_src;
  unsigned int rstep;
  int guide_log2;
  // every source kind but kSrcWindow: the previous step asked for its estimate in the deferred form (rr::kEstDeferred) -- when its
  // resample fired (Ctl.pending), this launch adds up the fields of the sources it gathers, per slot tile -> est_partials[tile][4]
  double* est_partials;
};

// The multinomial draw of output slot `slot` through the guide table (resample_core.hpp: buckets of the target space, built
// by k_plan_cdf's markers and k_guide_resolve): two adjacent table entries bracket the answer, the CDF is only read inside
// the bracket -- not at all when a heavy particle spans the whole bucket.  Same index as the lower bound over the whole CDF
// (tests: identical to the coarse-table kernel and to the literal walk).
struct __attribute__((packed, aligned(4))) GuidePair {
  unsigned int lo, hi;
};
__device__ inline uint64_t mn_guide_search(const Ctl* __restrict__ ctl, const uint64_t* __restrict__ cdf,
                                           const unsigned int* __restrict__ guide, int guide_log2, uint64_t target, uint64_t n_src) {
  const uint64_t total = ctl->total;
  const int s = rr::guide_shift(total, guide_log2);
  const uint64_t bucket = target >> s;
  const GuidePair g = *reinterpret_cast<const GuidePair*>(guide + bucket);
  const uint64_t hi = bucket < (total >> s) ? (uint64_t)g.hi : n_src - 1;  // the last bucket ends with the last source
  uint64_t j = g.lo, end = hi;  // the answer lies in [j, end]: no CDF entry is read when the bracket is one source
  while (j < end) {
    const uint64_t mid = j + ((end - j) >> 1);
    if (cdf[mid] >= target) end = mid;
    else j = mid + 1;
  }
  return j;
}

// The multinomial draws of one slot tile (slot = tile_base + r * kBlock + tid), the rows' searches in LOCKSTEP: every dependent
// read of the chain (guide pair -> CDF probes) is issued for all rows before any of them is waited for.  (mn_guide_search per
// row: the bracket loop's trip count depends on the data, so the second row's chain only started when the first row's had
// ended: +2.6 us at 1e6 slots.)  Same indices as mn_guide_search.
__device__ inline void mn_draw_rows(const Ctl* __restrict__ ctl, const uint64_t* __restrict__ cdf, const unsigned int* __restrict__ guide,
                                    int guide_log2, uint64_t n_src, uint64_t seed, unsigned int rstep, uint64_t first_gid,
                                    uint64_t tile_base, uint64_t n, unsigned int (&idx)[rr::kResolveRows]) {
  const int tid = threadIdx.x;
  const uint64_t total = ctl->total;
  const int gs = rr::guide_shift(total, guide_log2);
  uint64_t target[rr::kResolveRows], lo[rr::kResolveRows], hi[rr::kResolveRows];
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
    target[r] = k < n ? rr::resample_target(ctl, RR_RESAMPLE_MULTINOMIAL, first_gid + k, seed, rstep, nullptr, k) : 0ull;
  }
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
    lo[r] = hi[r] = 0;
    if (k < n) {
      const uint64_t bucket = target[r] >> gs;
      const GuidePair g = *reinterpret_cast<const GuidePair*>(guide + bucket);
      lo[r] = g.lo;
      hi[r] = bucket < (total >> gs) ? (uint64_t)g.hi : n_src - 1;  // the last bucket ends with the last source
    }
  }
  for (;;) {
    bool open = false;
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) open |= lo[r] < hi[r];
    if (!__any(open)) break;
    uint64_t c[rr::kResolveRows], mid[rr::kResolveRows];
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      mid[r] = lo[r] + ((hi[r] - lo[r]) >> 1);
      c[r] = lo[r] < hi[r] ? cdf[mid[r]] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      if (lo[r] < hi[r]) {
        if (c[r] >= target[r]) hi[r] = mid[r];
        else lo[r] = mid[r] + 1;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) idx[r] = (unsigned int)lo[r];
}

// EST: a build that can add up the deferred estimate (WindowArgs.est_partials).  A build of its own because the code's mere presence
// costs the headline kernel 1.5 us at 1e6 x 32 (six more VGPRs live through the observation loop), executed or not.
template <bool OBS_KERNARG, int SRC, int LIK, bool PACKED = false, bool EST = false>
__global__ __launch_bounds__(kBlock, 4) void k_step_lazy(Bufs b, double* __restrict__ w, Ctl* __restrict__ ctl,
                                                     StepParams p, ObsArg obs_arg,
                                                     const double* __restrict__ obs_dev,
                                                     unsigned int* __restrict__ markers,
                                                     const unsigned int* __restrict__ carry,
                                                     unsigned int* __restrict__ idx_out,
                                                     WindowArgs wa, double* pk0, double* pk1) {
  extern __shared__ double s_obs[];
  __shared__ double s_wmax[kBlock / rr::kWave];
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * p.n_obs; i += kBlock) s_obs[i] = OBS_KERNARG ? obs_arg.v[i] : obs_dev[i];
  __syncthreads();
  const int pending = ctl->pending;
  const int src = ctl->cur, dst = pending ? src ^ 1 : src;
  const double* __restrict__ sx = b.x[src];
  const double* __restrict__ sy = b.y[src];
  const double* __restrict__ syaw = b.yaw[src];
  const uint64_t n_tiles = (p.n + rr::kResolveSlots - 1) / rr::kResolveSlots;
  double wmax_local = 0.0;
  bool est_now = false;
  double est_acc[4] = {0.0, 0.0, 0.0, 0.0};
  const uint64_t tile = blockIdx.x;  // one tile per workgroup: nothing is loop-invariant, so no constant outlives its use
  if (tile < n_tiles) {
    unsigned int idx[rr::kResolveRows];
    const uint64_t tile_base = tile * rr::kResolveSlots;
    double na[rr::kResolveRows], nc[rr::kResolveRows];
    // The noise of a slot depends on (seed, step, slot) only: evaluate it first, so that this FP64 work runs while
    // the first dependent loads of the tile (control word, markers) are in flight and the workgroups of a CU, which
    // all start together, do not all sit in their load prologue at the same time.
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r)
      rr_pf_motion_noise(p.seed, p.step, p.first_gid + tile_base + (uint64_t)r * kBlock + tid, p.sigma_v, p.sigma_w, &na[r], &nc[r]);
    // kSrcWindow: positions of the window this shard serves, of this tile's slots
    uint64_t win_lo = 0, win_hi = 0, pos0 = 0;
    if (pending && SRC == kSrcLidx) {
#pragma unroll
      for (int r = 0; r < rr::kResolveRows; ++r) {
        const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
        idx[r] = k < p.n ? markers[k] : 0u;  // `markers` is the lidx array here
      }
    } else if (pending && SRC == kSrcDraw) {
      mn_draw_rows(ctl, wa.cdf, wa.guide, wa.guide_log2, wa.n_src, p.seed, wa.rstep, p.first_gid, tile_base, p.n, idx);
#pragma unroll
      for (int r = 0; r < rr::kResolveRows; ++r) {
        const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
        if (k < p.n && idx_out) idx_out[k] = idx[r];
      }
    } else if (pending && SRC == kSrcWindow) {
      const uint64_t own0 = p.first_gid + wa.pad;  // position of own slot 0: a multiple of kResolveSlots
      win_lo = ctl->served_first + wa.pad;
      win_hi = win_lo + ctl->served_count;
      pos0 = own0 + tile_base;
      rr::resolve_tile_window(markers, carry, own0 / rr::kResolveSlots + tile, win_lo, win_hi, own0, own0 + p.n, idx);
      const uint64_t pos1 = pos0 + rr::kResolveSlots < own0 + p.n ? pos0 + rr::kResolveSlots : own0 + p.n;
      (void)pos1;  // (a slot a peer serves waits for its own delivery below: the seal plane of the inbox)
    } else if (pending) {
      rr::resolve_tile(markers, carry, p.n, tile, idx);
    } else {
#pragma unroll
      for (int r = 0; r < rr::kResolveRows; ++r) idx[r] = (unsigned int)(tile_base + (uint64_t)r * kBlock + tid);
    }
    // issue every row's loads before the (long) arithmetic of the first row
    double x[rr::kResolveRows], y[rr::kResolveRows], yaw[rr::kResolveRows];
    est_now = EST && SRC != kSrcWindow && wa.est_partials != nullptr && pending;
    double v_src[rr::kResolveRows];  // the sources' v: only the estimate reads it (propagate overwrites v)
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
      x[r] = y[r] = yaw[r] = v_src[r] = 0.0;
      if (k < p.n) {
        const uint64_t j = idx[r];
        if (SRC == kSrcWindow && pending && (pos0 + (uint64_t)r * kBlock + tid < win_lo || pos0 + (uint64_t)r * kBlock + tid >= win_hi)) {
          // delivered by a peer into this rank's inbox [4 fields + tag][n]: wait (bounded) for THIS slot's tag of the step
          // whose resample is being consumed, then read its fields.  (n_ranks == 0: the RCCL transport -- an earlier kernel
          // of this stream filled the inbox.)
          double f[4];
          (void)rr::inbox_take(wa.inbox, p.n, k, wa.n_ranks > 0 ? wa.wait_seq : (uint64_t)0, wa.timeout_ticks, wa.err, f);
          x[r] = f[0];
          y[r] = f[1];
          yaw[r] = f[2];
        } else if (PACKED && pending) {
          const double4 rec = *reinterpret_cast<const double4*>((src ? pk1 : pk0) + 4 * j);
          x[r] = rec.x;
          y[r] = rec.y;
          yaw[r] = rec.z;
          v_src[r] = rec.w;
        } else {
          x[r] = sx[j];
          y[r] = sy[j];
          yaw[r] = syaw[j];
          if (EST && est_now) v_src[r] = b.v[src][j];
        }
      }
    }
    if (EST && est_now) {  // the mean of the resampled set, before it is propagated: the thread's rows now, the wave's sums at the end
      double f[4][rr::kResolveRows];
#pragma unroll
      for (int r = 0; r < rr::kResolveRows; ++r) {
        f[0][r] = x[r];
        f[1][r] = y[r];
        f[2][r] = yaw[r];
        f[3][r] = v_src[r];
      }
      rr::est_rows_sum<rr::kResolveRows>(f, est_acc);
    }
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
      if (k < p.n) {
        double v;
        rr_pf_propagate_one(&x[r], &y[r], &yaw[r], &v, p.u0, p.u1, p.dt, na[r], nc[r]);
        b.x[dst][k] = x[r];
        b.y[dst][k] = y[r];
        b.yaw[dst][k] = yaw[r];
        b.v[dst][k] = v;
        if (PACKED) *reinterpret_cast<double4*>((dst ? pk1 : pk0) + 4 * k) = make_double4(x[r], y[r], yaw[r], v);
        if (SRC == kSrcLidx) {
          if (pending) markers[k] = kInPlace;
        } else if (SRC == kSrcMarkers && pending && idx_out) {
          idx_out[k] = idx[r];
        }
      }
    }
    double wgt[rr::kResolveRows];
    if (LIK == RR_LIK_PRODUCT) {
#pragma unroll
      for (int r = 0; r < rr::kResolveRows; ++r) wgt[r] = rr_pf_weight_product(x[r], y[r], s_obs, p.n_obs, p.lik);
    } else {
      // one pass over the observation block for the thread's rows, independent chains per row (a row past the end
      // of the set weighs a dummy particle at the origin and stores nothing)
      rr_pf_weight_fused_rows<rr::kResolveRows>(x, y, s_obs, p.n_obs, p.lik, wgt);
    }
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
      if (k < p.n) {
        w[k] = wgt[r];
        if (wgt[r] > wmax_local) wmax_local = wgt[r];
      }
    }
  }
  if (EST && est_now) rr::est_wave_store<kBlock>(est_acc, wa.est_partials, tile);
  double m = rr::wave_max(wmax_local);
  if ((tid & 63) == 0) s_wmax[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    double bm = s_wmax[0];
    for (int k = 1; k < kBlock / rr::kWave; ++k) bm = s_wmax[k] > bm ? s_wmax[k] : bm;
    if (bm > 0.0) rr::atomic_max_u64(&ctl->wmax_bits, rr_d2u(bm));
    if (blockIdx.x == 0) ctl->weights_uniform = 0;
  }
}

// ------------------------------------------------------------------------------------------
// K5: resample gather.  The plan kernel has already flipped Ctl.cur, so the particles are read
// from buffer set cur^1 and written to set cur (or to `staging` for the sharded exchange).
struct GatherArgs {
  uint64_t n_src;       // CDF entries (particles of this shard)
  uint64_t first_slot;  // global index of output slot 0 handled here
  uint64_t n_slots;
  uint64_t seed;
  unsigned int rstep;
  int scheme;
  int to_staging;  // 1 => write n_slots x (x, y, yaw, v) records to `staging`
};

__device__ inline void copy_particle(const Bufs& b, int src, int dst, uint64_t j, uint64_t k, bool to_staging,
                                     double* __restrict__ staging) {
  const double x = b.x[src][j], y = b.y[src][j], yaw = b.yaw[src][j], v = b.v[src][j];
  if (to_staging) {  // one contiguous 32-byte record per slot
    staging[4 * k] = x;
    staging[4 * k + 1] = y;
    staging[4 * k + 2] = yaw;
    staging[4 * k + 3] = v;
  } else {
    b.x[dst][k] = x;
    b.y[dst][k] = y;
    b.yaw[dst][k] = yaw;
    b.v[dst][k] = v;
  }
}

// Systematic (fastslam1.rs:219-231): the plan kernel left one marker per source with offspring
// (resample_core.hpp "WITHOUT any search"); resolve them to source indices with a running maximum
// per 1024-slot tile and move the particles in the same pass.  Reads 4 B marker + 32 B particle,
// writes 32 B particle (+ 4 B marker clear) per slot; no CDF array, no binary search.
__global__ __launch_bounds__(kBlock) void k_resolve_gather(Bufs b, const Ctl* __restrict__ ctl,
                                                          unsigned int* __restrict__ markers,
                                                          const unsigned int* __restrict__ carry,
                                                          unsigned int* __restrict__ idx_out,
                                                          double* __restrict__ staging, uint64_t n_slots,
                                                          int to_staging, int lazy) {
  // eager: the plan kernel flipped Ctl.cur already (read cur^1, write cur), runs iff fired;
  // lazy (materialise a pending resample for an accessor): read cur, write cur^1, k_settle flips
  if (lazy ? !ctl->pending : !ctl->fired) return;
  unsigned int idx[rr::kResolveRows];
  rr::resolve_tile(markers, carry, n_slots, blockIdx.x, idx);
  const int dst = lazy ? ctl->cur ^ 1 : ctl->cur, src = dst ^ 1;
  const uint64_t tile_base = (uint64_t)blockIdx.x * rr::kResolveSlots;
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + threadIdx.x;
    if (k < n_slots) {
      copy_particle(b, src, dst, idx[r], k, to_staging != 0, staging);
      if (idx_out) idx_out[k] = idx[r];
    }
  }
}

__global__ void k_settle(Ctl* ctl) {
  if (ctl->pending) {
    ctl->cur ^= 1;
    ctl->pending = 0;
  }
}

// Multinomial (particle_filter.rs:455-470): independent draws, one thread per output slot.  A
// lower bound over the whole CDF is ~20 DEPENDENT probes of HBM/L2 per draw (137 us for 1e6
// draws); instead every workgroup stages the coarse table (every 2^coarse_log2-th CDF entry,
// <= 144 KB: 64-entry windows up to 1.18e6 particles) in LDS, searches that, and finishes inside one
// 2^coarse_log2-entry window of the full CDF (6 probes over 4 cache lines at 64 entries).
__global__ __launch_bounds__(1024) void k_resample_gather_mn(Bufs b, const Ctl* __restrict__ ctl,
                                                              const uint64_t* __restrict__ cdf,
                                                              const uint64_t* __restrict__ coarse, int coarse_log2,
                                                              uint64_t n_coarse,
                                                              const double* __restrict__ r_explicit,
                                                              unsigned int* __restrict__ idx_out,
                                                              unsigned int* __restrict__ lidx_out, GatherArgs a) {
  if (!ctl->fired) return;
  extern __shared__ uint64_t s_coarse[];
  for (uint64_t i = threadIdx.x; i < n_coarse; i += blockDim.x) s_coarse[i] = coarse[i];
  __syncthreads();
  const int dst = ctl->cur, src = dst ^ 1;
  // grid-stride: a workgroup stages the coarse table once and serves several blocks of slots
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.n_slots; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t target = rr::resample_target(ctl, RR_RESAMPLE_MULTINOMIAL, a.first_slot + k, a.seed, a.rstep, r_explicit, k);
    const uint64_t blk = rr_lower_bound_u64(s_coarse, n_coarse, target);  // first window whose last entry >= target
    const uint64_t lo = blk << coarse_log2;
    const uint64_t len = lo + (1ull << coarse_log2) <= a.n_src ? (1ull << coarse_log2) : a.n_src - lo;
    const uint64_t j = lo + rr_lower_bound_u64(cdf + lo, len, target);
    if (lidx_out) lidx_out[k] = (unsigned int)j;  // lazy: the next propagate kernel reads through it
    else copy_particle(b, src, dst, j, k, false, nullptr);
    if (idx_out) idx_out[k] = (unsigned int)j;
  }
}

// The same draws through the guide table (mn_guide_search): no LDS table to stage, one draw per thread.
__global__ __launch_bounds__(kBlock) void k_resample_guide_mn(Bufs b, const Ctl* __restrict__ ctl,
                                                             const uint64_t* __restrict__ cdf,
                                                             const unsigned int* __restrict__ guide, int guide_log2,
                                                             const double* __restrict__ r_explicit,
                                                             unsigned int* __restrict__ idx_out,
                                                             unsigned int* __restrict__ lidx_out, GatherArgs a) {
  if (!ctl->fired) return;
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= a.n_slots) return;
  const uint64_t target = rr::resample_target(ctl, RR_RESAMPLE_MULTINOMIAL, a.first_slot + k, a.seed, a.rstep, r_explicit, k);
  const uint64_t j = mn_guide_search(ctl, cdf, guide, guide_log2, target, a.n_src);
  const int dst = ctl->cur, src = dst ^ 1;
  if (lidx_out) lidx_out[k] = (unsigned int)j;  // lazy: the next propagate kernel reads through it
  else copy_particle(b, src, dst, j, k, false, nullptr);
  if (idx_out) idx_out[k] = (unsigned int)j;
}

// The synchronous try_step of a large multinomial filter (rr_pf_step) needs the resampled set's mean NOW, not when the next step
// moves the particles: the draws are searched here -- per slot tile, rows in lockstep, exactly k_step_lazy<kSrcDraw>'s search --,
// the source indices go to `lidx` (the next step reads through them instead of searching again) and the sources' fields, read out
// of the packed mirror, into the deferred estimate's per-wave sums (rr::est_rows_sum / est_wave_store: the bits
// rr_pf_step_async_estimate + rr_pf_last_step_estimate produce).
__global__ __launch_bounds__(kBlock) void k_mn_search_est(const Ctl* __restrict__ ctl, const uint64_t* __restrict__ cdf,
                                                         const unsigned int* __restrict__ guide, int guide_log2, GatherArgs a,
                                                         unsigned int* __restrict__ idx_out, unsigned int* __restrict__ lidx_out,
                                                         const double* __restrict__ pk0, const double* __restrict__ pk1,
                                                         double* __restrict__ est_partials) {
  if (!ctl->fired) return;
  const int tid = threadIdx.x;
  const uint64_t tile_base = (uint64_t)blockIdx.x * rr::kResolveSlots;
  unsigned int idx[rr::kResolveRows];
  mn_draw_rows(ctl, cdf, guide, guide_log2, a.n_src, a.seed, a.rstep, a.first_slot, tile_base, a.n_slots, idx);
  const double* __restrict__ pk = ctl->cur ? pk1 : pk0;  // (the lazy resample has not flipped Ctl.cur: the sources are the live set)
  double f[4][rr::kResolveRows];
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
    f[0][r] = f[1][r] = f[2][r] = f[3][r] = 0.0;
    if (k < a.n_slots) {
      const double4 rec = *reinterpret_cast<const double4*>(pk + 4 * (uint64_t)idx[r]);
      f[0][r] = rec.x;
      f[1][r] = rec.y;
      f[2][r] = rec.z;
      f[3][r] = rec.w;
      lidx_out[k] = idx[r];
      if (idx_out) idx_out[k] = idx[r];
    }
  }
  double acc[4];
  rr::est_rows_sum<rr::kResolveRows>(f, acc);
  rr::est_wave_store<kBlock>(acc, est_partials, blockIdx.x);
}

// sharded adopt: unpack the received n x (x, y, yaw, v) records into the live buffer set (the
// plan kernel already made it the other one)
// slots [self_lo, self_hi) are the ones this rank serves to ITSELF: their records are taken straight from the send
// buffer (`in_self`, record 0 = slot self_lo) instead of travelling through a send / receive to the same rank
__global__ __launch_bounds__(kBlock) void k_adopt(Bufs b, const Ctl* __restrict__ ctl,
                                                 const double* __restrict__ in, uint64_t n,
                                                 const double* __restrict__ in_self, uint64_t self_lo, uint64_t self_hi) {
  if (!ctl->fired) return;
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const int dst = ctl->cur;
  const double* __restrict__ r = (k >= self_lo && k < self_hi) ? in_self + 4 * (k - self_lo) : in + 4 * k;
  b.x[dst][k] = r[0];
  b.y[dst][k] = r[1];
  b.yaw[dst][k] = r[2];
  b.v[dst][k] = r[3];
}

// ------------------------------------------------------------------------------------------
// K7: weighted moments about the shift point p0 = particle 0 of the live set (always inside
// the cloud, so the one-pass covariance does not cancel): sum w, sum w d, sum w d d^T with
// d = p - p0.  Grid-stride, 15 accumulators per thread, wave shuffle + LDS, per-block partials
// combined by k_moments_final in block order.  force_uniform => w_i = 1.
__global__ __launch_bounds__(kBlock) void k_moments(Bufs b, const double* __restrict__ w,
                                                   const Ctl* __restrict__ ctl, uint64_t n,
                                                   int force_uniform, double* __restrict__ partials) {
  __shared__ double s_acc[kBlock / rr::kWave][kNumMoments];
  const int cur = ctl->cur;
  const bool uniform = force_uniform || ctl->weights_uniform;
  const double p0x = b.x[cur][0], p0y = b.y[cur][0], p0a = b.yaw[cur][0], p0v = b.v[cur][0];
  double acc[kNumMoments];
#pragma unroll
  for (int k = 0; k < kNumMoments; ++k) acc[k] = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * kBlock) {
    const double wi = uniform ? 1.0 : w[i];
    const double d0 = b.x[cur][i] - p0x, d1 = b.y[cur][i] - p0y, d2 = b.yaw[cur][i] - p0a, d3 = b.v[cur][i] - p0v;
    const double w0 = wi * d0, w1 = wi * d1, w2 = wi * d2, w3 = wi * d3;
    acc[0] += wi;
    acc[1] += w0; acc[2] += w1; acc[3] += w2; acc[4] += w3;
    acc[5] += w0 * d0; acc[6] += w0 * d1; acc[7] += w0 * d2; acc[8] += w0 * d3;
    acc[9] += w1 * d1; acc[10] += w1 * d2; acc[11] += w1 * d3;
    acc[12] += w2 * d2; acc[13] += w2 * d3;
    acc[14] += w3 * d3;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kNumMoments; ++k) {
    double s = rr::wave_sum(acc[k]);
    if (lane == 0) s_acc[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < kNumMoments) {
    double s = 0.0;
    for (int k = 0; k < kBlock / rr::kWave; ++k) s += s_acc[k][threadIdx.x];
    partials[(uint64_t)blockIdx.x * kNumMoments + threadIdx.x] = s;
  }
}

// Host-visible mailbox of a filter (pinned, host-coherent memory): the synchronous try_step of a small filter reads the
// estimate the kernel wrote there instead of paying two device-to-host copies and a stream synchronisation (~20 us) for
// four doubles; the host polls `seq`.
struct HostMail {
  double est[4];
  uint64_t flags;  // k_est_mail: != 0 => Ctl holds something the host has to look at (a degraded plan) -- take the long way
  uint64_t seq;
};
// one wave per moment: lanes stride over the per-block partials (fixed order per lane), then a
// shuffle tree -- deterministic for a given grid.  mail != null: the estimate (rr_pf_estimate: shift point + first moments / W,
// formed as compute_moments forms it on the host) also goes to the host mailbox, stamped `seq`; flags != 0 when the weights do
// not sum to a positive finite number (the host then takes the uniform-weights retry of particle_filter.rs:433-438)
__global__ __launch_bounds__(kNumMoments * 64) void k_moments_final(Bufs b, Ctl* __restrict__ ctl,
                                                                   const double* __restrict__ partials,
                                                                   int n_blocks, HostMail* __restrict__ mail, uint64_t seq) {
  __shared__ double s_m[kNumMoments];
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int j = lane; j < n_blocks; j += 64) s += partials[j * kNumMoments + k];
  s = rr::wave_sum(s);
  if (lane == 0) {
    ctl->moments[k] = s;
    s_m[k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int cur = ctl->cur;
    const double p0[4] = {b.x[cur][0], b.y[cur][0], b.yaw[cur][0], b.v[cur][0]};
    for (int q = 0; q < 4; ++q) ctl->shift_point[q] = p0[q];
    if (mail) {
      const double W = s_m[0];
      const bool ok = W > 0.0 && W < INFINITY;
      for (int q = 0; q < 4; ++q)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(&mail->est[q]), (uint64_t)__double_as_longlong(p0[q] + s_m[1 + q] / W), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&mail->flags, (uint64_t)(ok ? 0 : 1) | ((uint64_t)(ctl->grid_timeout != 0) << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------
// SMALL particle sets: the whole step -- and K of them -- in ONE launch of ONE workgroup.
// Every caller in the reference runs 100 - 1200 particles (headless_localizers.rs:56, render_gif_particle_filter.rs:77-79,
// playground/src/localization.rs:58-66, tests/unified_filter_comparison.rs:286-295).  At that size the fused step of the
// large filters (2 launches, 5 for the multinomial resampler) is nothing but launch latency: 16 us a step for 2 us of work.
// Here the particles live in REGISTERS from the first step of the launch to the last (thread t owns the R consecutive slots
// [t R, t R + R), R = 1 / 2 / 4 for N <= 512 / 1024 / 2048), the maximum and the integer sums are workgroup reductions, the
// resample runs through LDS (systematic: slot-run markers + running maximum, exactly the large kernels' scheme; multinomial:
// the integer CDF and one binary search per slot), the gather is an LDS round trip, and the mean try_step returns
// (particle_filter.rs:488-497) is a workgroup sum.  Same per-element arithmetic (include/rr_pf_spec.h), same Philox
// counters, same integer image: bit-identical to the large path and to the D-spec (tests/test_gpu_small_n.py).
// rr_pf_step_many hands K controls and observation blocks over in one buffer; the single-step entry points use K = 1 with the
// observations in the launch packet.
constexpr uint64_t kSmallMaxParticles = 2048;

struct SmallArgs {
  uint64_t n;
  uint64_t seed;
  unsigned int step0, rstep0;
  int n_obs, K;
  int gate, scheme;       // rr_resample_gate, rr_resample_scheme
  double neff_threshold;  // N * resample_threshold
  double dt, sigma_v, sigma_w;
  double u0, u1;          // K == 1: the control (else in steps_in)
  rr_pf_lik lik;
  int want_est;
  int inputs_in_kernarg;  // K == 1 and the observations fit the launch packet
  uint64_t mail_seq;      // != 0: the last step's estimate also goes to the host mailbox, stamped with this number
  rr::ResidentArgs res;   // res.on: the kernel stays and serves one step per command of the ring (resident_core.hpp); K is ignored
};

constexpr int kEstRing = 32;  // per-step estimates of rr_pf_step_many gather in LDS and leave in blocks of this many steps

// The synchronous try_step of a LARGE filter (rr_pf_step, fused systematic step): the step's estimate is the sum of the plan
// kernel's per-tile partial sums in tile order, divided by Ctl.est_denom -- formed here exactly as rr_pf_last_step_estimate
// forms it on the host (same order, same operations) and left in the mailbox, so the host polls a stamp instead of copying
// 16 KB of partial sums and Ctl back behind a stream synchronisation (~20 us of a 70 us synchronous step at 1e6 particles).
__global__ void k_est_mail(const Ctl* __restrict__ ctl, const double* __restrict__ partials, uint64_t n_tiles,
                           HostMail* __restrict__ mail, uint64_t seq) {
  // the sums are serial (tile order), the loads must not be: stage 512 tiles at a time in LDS with all threads
  constexpr int kStage = 512;
  __shared__ double s_part[4 * kStage];
  const int k = threadIdx.x;
  double acc = 0.0;
  for (uint64_t t0 = 0; t0 < n_tiles; t0 += kStage) {
    const uint64_t m = n_tiles - t0 < (uint64_t)kStage ? n_tiles - t0 : (uint64_t)kStage;
    __syncthreads();
    for (uint64_t i = threadIdx.x; i < 4 * m; i += blockDim.x) s_part[i] = partials[4 * t0 + i];
    __syncthreads();
    if (k < 4)
      for (uint64_t t = 0; t < m; ++t) acc += s_part[4 * t + k];
  }
  if (k < 4) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(&mail->est[k]), (uint64_t)__double_as_longlong(acc / ctl->est_denom), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (k == 0) __hip_atomic_store(&mail->flags, (uint64_t)(ctl->grid_timeout != 0) | ((uint64_t)(ctl->est_step == 0) << 1), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (k == 0) __hip_atomic_store(&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same for a filter whose estimate may be in either form (rr::EstArgs; the multinomial scheme): the plan tiles' partial sums in
// tile order when the gate stayed shut, else the slot tiles' per-wave sums in est_slots_total's order -- kEstChunks interleaved
// chunks (chunk c = entries c, c + kEstChunks, c + 2 kEstChunks, ...: the threads of one pass read consecutive entries), each added
// up from its first entry on, then the chunks in order (the host adds them the same way).
constexpr int kEstChunks = 256;
__global__ __launch_bounds__(kEstChunks) void k_est_mail_any(const Ctl* __restrict__ ctl, const double* __restrict__ plan_partials,
                                                             uint64_t n_tiles, const double* __restrict__ slot_partials, uint64_t n_slot_part,
                                                             HostMail* mail, uint64_t seq) {
  __shared__ double s_cs[4][kEstChunks];
  const int c = threadIdx.x;
  const bool slots = ctl->est_kind == rr::kEstSlotTiles;
  const double* __restrict__ part = slots ? slot_partials : plan_partials;
  const uint64_t n_part = slots ? n_slot_part : n_tiles;
  // (plan tiles: ONE chunk holds everything -- the sequential order of k_est_mail and rr_pf_last_step_estimate)
  const uint64_t first = slots ? (uint64_t)c : 0, stride = slots ? (uint64_t)kEstChunks : 1;
  double cs[4] = {0.0, 0.0, 0.0, 0.0};
  if (slots || c == 0) {
#pragma unroll 8
    for (uint64_t i = first; i < n_part; i += stride) {
      const double4 v = *reinterpret_cast<const double4*>(part + 4 * i);
      cs[0] += v.x;
      cs[1] += v.y;
      cs[2] += v.z;
      cs[3] += v.w;
    }
  }
  for (int k = 0; k < 4; ++k) s_cs[k][c] = cs[k];
  __syncthreads();
  if (c < 4) {
    double acc = 0.0;
    if (slots)
      for (int q = 0; q < kEstChunks; ++q) acc += s_cs[c][q];
    else
      acc = s_cs[c][0];
    __hip_atomic_store(reinterpret_cast<uint64_t*>(&mail->est[c]), (uint64_t)__double_as_longlong(acc / ctl->est_denom), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (c == 0) __hip_atomic_store(&mail->flags, (uint64_t)(ctl->grid_timeout != 0) | ((uint64_t)(ctl->est_step == 0) << 1), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (c == 0) __hip_atomic_store(&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// workgroup-wide helpers of the small kernel (kSmallBlock threads); every thread gets the result
template <int BLOCK>
__device__ inline double small_block_max(double v, double* s_red) {
  const int tid = threadIdx.x;
  const double m = rr::wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
  __syncthreads();
  double r = s_red[0];
#pragma unroll
  for (int k = 1; k < BLOCK / rr::kWave; ++k) r = s_red[k] > r ? s_red[k] : r;
  return r;
}
// four sums at once (the estimate): DPP inside the waves, one LDS exchange
template <int BLOCK>
__device__ inline void small_block_sum4(double (&v)[4], double* s_red4 /* [4][BLOCK / 64] */) {
  constexpr int W = BLOCK / rr::kWave;
  const int tid = threadIdx.x;
  double m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = rr::wave_sum_dpp(v[k]);
  __syncthreads();
  if ((tid & 63) == 63) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s_red4[k * W + (tid >> 6)] = m[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double r = 0.0;
#pragma unroll
    for (int q = 0; q < W; ++q) r += s_red4[k * W + q];
    v[k] = r;
  }
}

// instrumented build (make timeline): 100 MHz stamps of a resident step's phases, returned in rsp[8 .. 15] (tools/resident_timeline.py)
#if defined(RR_PLAN_TIMELINE)
#define RR_RES_TL(K_) do { if (resident && threadIdx.x == 0) res_tl[(K_)] = wall_clock64(); } while (0)
#else
#define RR_RES_TL(K_) do { } while (0)
#endif
template <int BLOCK, int R, int LIK>
__global__ __launch_bounds__(BLOCK) void k_step_small(Bufs b, double* __restrict__ w, Ctl* __restrict__ ctl, SmallArgs a,
                                                            ObsArg obs_arg, const double* __restrict__ steps_in,
                                                            unsigned int* __restrict__ idx_out, double* __restrict__ est_out,
                                                            double* __restrict__ est_partials, HostMail* mail,
                                                            rr::ResidentRing* __restrict__ ring) {
  // [3 n_obs] observations | [4][n] gather fields | [n + 1] markers (u32) or [n] CDF (u64);
  // resident: [payload_cap] command payload (u0, u1, observations) in front instead of the observations
  extern __shared__ double s_dyn[];
  constexpr int W = BLOCK / rr::kWave;
  __shared__ uint64_t s_u[4 * W];
  __shared__ double s_red[4 * W];
  __shared__ unsigned int s_mx[W];
  __shared__ double s_ring[kEstRing * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t n = a.n;
  const bool resident = a.res.on != 0;
  __shared__ int s_hdr[2];
  double* const s_obs = resident ? s_dyn + 2 : s_dyn;
  double* const s_f = s_dyn + (resident ? (size_t)a.res.payload_cap : 3 * (size_t)a.n_obs);
  uint64_t* const s_cdf = reinterpret_cast<uint64_t*>(s_f + 4 * n);
  unsigned int* const s_mark = reinterpret_cast<unsigned int*>(s_f + 4 * n);
  const int cur = ctl->cur;
  const uint64_t k0 = (uint64_t)tid * R;
  double x[R], y[R], yaw[R], v[R], wgt[R];
  unsigned int last_idx[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint64_t k = k0 + j;
    x[j] = k < n ? b.x[cur][k] : 0.0;
    y[j] = k < n ? b.y[cur][k] : 0.0;
    yaw[j] = k < n ? b.yaw[cur][k] : 0.0;
    v[j] = k < n ? b.v[cur][k] : 0.0;
    wgt[j] = 0.0;
    last_idx[j] = (unsigned int)k;
  }
  // what the last step leaves in Ctl (thread 0 writes it once, after the loop)
  int c_usable = 0, c_mode = rr::kImageUniform, c_shift = 0, c_fired = 0, any_fired = 0;
  uint64_t c_total = 0;
  u128 c_q2 = {0, 0};
  double c_wmax = 0.0, c_rho = 0.0, c_est[4] = {0.0, 0.0, 0.0, 0.0}, c_den = 1.0;
  rr_sys_plan c_plan = {};
#if defined(RR_PLAN_TIMELINE)
  uint64_t res_tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  int n_obs = a.n_obs, steps_done = 0, res_guess = 3 + 3 * a.n_obs < 64 ? 3 + 3 * a.n_obs : 64, res_last_op = rr::kResOpNone;
  const uint64_t res_deadline = resident ? wall_clock64() + a.res.life_ticks : 0;
  for (int s = 0; resident || s < a.K; ++s) {
    // ---- inputs of this step
    double u0 = a.u0, u1 = a.u1;
    // the step's random numbers are functions of (seed, step counters, slot) alone -- not of the state, not of the inputs: a
    // resident incarnation draws them BEFORE it waits for the command, in time the host spends turning the last answer around
    // (the Philox rounds, the logarithm and the sine/cosine of the Box-Muller pair are ~40 % of a small step's dependent chain)
    double pre_na[R], pre_nc[R], pre_r[R];
    if (resident) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        double dummy;
        rr_pf_motion_noise(a.seed, a.step0 + (unsigned int)s, k0 + j, a.sigma_v, a.sigma_w, &pre_na[j], &pre_nc[j]);
        // multinomial: this slot's draw; systematic: the one offset (index 0 of the stream), the same in every slot
        rr_uniform2(a.seed, RR_STREAM_RESAMPLE, a.rstep0 + (unsigned int)s, a.scheme == RR_RESAMPLE_SYSTEMATIC ? 0ull : k0 + j, &pre_r[j], &dummy);
      }
    }
    RR_RES_TL(0);  // random numbers drawn, about to wait
    if (resident) {  // wait for the host's next command; anything but a step ends this incarnation
      res_last_op = rr::resident_fetch<BLOCK>(ring, a.res.first_seq + (uint64_t)s, a.res.idle_ticks, res_deadline, a.res.payload_cap,
                                              res_guess, s_dyn, s_hdr);
      if (res_last_op != rr::kResOpStep) break;
      n_obs = (s_hdr[1] - 2) / 3;
      u0 = s_dyn[0];
      u1 = s_dyn[1];
    } else {
      __syncthreads();
      if (a.inputs_in_kernarg) {
        for (int i = tid; i < 3 * n_obs; i += BLOCK) s_obs[i] = obs_arg.v[i];
      } else {
        const double* in = steps_in + (size_t)s * (2 + 3 * (size_t)n_obs);
        u0 = in[0];
        u1 = in[1];
        for (int i = tid; i < 3 * n_obs; i += BLOCK) s_obs[i] = in[2 + i];
      }
      __syncthreads();
    }
    steps_done = s + 1;
    RR_RES_TL(1);  // command here
    // ---- propagate + weight (particle_filter.rs:279-296, :310-329)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      double na, nc;
      if (resident) {
        na = pre_na[j];
        nc = pre_nc[j];
      } else {
        rr_pf_motion_noise(a.seed, a.step0 + (unsigned int)s, k0 + j, a.sigma_v, a.sigma_w, &na, &nc);
      }
      rr_pf_propagate_one(&x[j], &y[j], &yaw[j], &v[j], u0, u1, a.dt, na, nc);
    }
    if (LIK == RR_LIK_PRODUCT) {
#pragma unroll
      for (int j = 0; j < R; ++j) wgt[j] = rr_pf_weight_product(x[j], y[j], s_obs, n_obs, a.lik);
    } else {
      rr_pf_weight_fused_rows<R>(x, y, s_obs, n_obs, a.lik, wgt);
    }
    double wl = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j)
      if (k0 + j < n && wgt[j] > wl) wl = wgt[j];  // NaN and negatives drop out
    RR_RES_TL(2);  // propagated, weighted
    const double wmax = small_block_max<BLOCK>(wl, s_red);
    RR_RES_TL(3);  // maximum
    // ---- integer image, sums (resample_core.hpp: quantize_reduce_tile / tile_scan)
    const bool usable = wmax > 0.0 && wmax < INFINITY;
    const int mode = usable ? (int)rr::kImageWeights : (int)rr::kImageUniform;  // PF / MCL: sum w <= 0 => uniform (:433-438)
    const int shift = usable ? rr_fix_shift(wmax, n) : 0;
    uint64_t q[R], c[R], run = 0;
    u128 q2 = {0, 0};
#pragma unroll
    for (int j = 0; j < R; ++j) {
      q[j] = k0 + j >= n ? 0ull : (mode == rr::kImageWeights ? rr_fix_quantize(wgt[j], shift) : 1ull);
      run += q[j];
      c[j] = run;
      u128 sq;
      rr_mul64wide(q[j], q[j], &sq.hi, &sq.lo);
      q2 = rr::add128(q2, sq);
    }
    const uint64_t incl = rr::wave_scan_u64(run, lane);
    q2 = rr::wave_sum_u128(q2);
    __syncthreads();
    if (lane == 63) s_u[wv] = incl;
    if (lane == 0) {
      s_u[W + wv] = q2.hi;
      s_u[2 * W + wv] = q2.lo;
    }
    __syncthreads();
    uint64_t off = incl - run, total = 0;
    u128 qq = {0, 0};
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (k < wv) off += s_u[k];
      total += s_u[k];
      qq = rr::add128(qq, u128{s_u[W + k], s_u[2 * W + k]});
    }
    rr::TileSums ts;
    ts.pre = 0;
    ts.tot = total;
    ts.q2 = qq;
    PlanArgs pa{};
    pa.n_global = n;
    pa.neff_threshold = a.neff_threshold;
    pa.gate = a.gate;
    pa.mode = 0;
    const int fire = rr::gate_decision(mode, ts, pa);
    const unsigned int rstep = a.rstep0 + (unsigned int)s;
    c_usable = usable ? 1 : 0;
    c_mode = mode;
    c_shift = shift;
    c_fired = fire;
    c_total = total;
    c_q2 = qq;
    c_wmax = wmax;
    double est_acc[4] = {0.0, 0.0, 0.0, 0.0};
    RR_RES_TL(4);  // integer image, sums, gate
    if (!fire) {
      if (a.want_est) {  // sum_j q_j p_j / T  (the cache refreshed at particle_filter.rs:332)
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const double cq = (double)q[j];
          est_acc[0] = rr_fma(cq, x[j], est_acc[0]);
          est_acc[1] = rr_fma(cq, y[j], est_acc[1]);
          est_acc[2] = rr_fma(cq, yaw[j], est_acc[2]);
          est_acc[3] = rr_fma(cq, v[j], est_acc[3]);
        }
        c_den = (double)total;
      }
    } else {
      any_fired = 1;
      unsigned int idx[R];
      if (a.scheme == RR_RESAMPLE_SYSTEMATIC) {
        double rho, dummy;
        if (resident) rho = pre_r[0];
        else rr_uniform2(a.seed, RR_STREAM_RESAMPLE, rstep, 0, &rho, &dummy);
        const rr_sys_plan plan = rr_sys_plan_make(rho, total, n);
        c_rho = rho;
        c_plan = plan;
        const rr_sys_inv inv = rr_sys_inv_make(plan, total);
        __syncthreads();
        for (uint64_t k = tid; k <= n; k += BLOCK) s_mark[k] = 0;
        __syncthreads();
        uint64_t h_run = rr_sys_slots_upto(plan, inv, total, off);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (q[j] == 0 || k0 + j >= n) continue;
          const uint64_t h = rr_sys_slots_upto(plan, inv, total, off + c[j]);
          if (h > h_run) {
            s_mark[h_run] = (unsigned int)(k0 + j + 1);
            h_run = h;
          }
        }
        __syncthreads();
        // running maximum over the slots in slot order: serial inside the thread, DPP across the wave, LDS across waves
        unsigned int m[R], mrun = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const unsigned int mk = k0 + j < n ? s_mark[k0 + j] : 0u;
          mrun = mk > mrun ? mk : mrun;
          m[j] = mrun;
        }
        const unsigned int mincl = rr::wave_scan_max_u32(mrun);
        if (lane == 63) s_mx[wv] = mincl;
        __syncthreads();
        unsigned int pre = 0;
#pragma unroll
        for (int k = 0; k < W; ++k)
          if (k < wv) pre = s_mx[k] > pre ? s_mx[k] : pre;
        // exclusive carry into this thread: the maximum of the lanes before it in the wave and of the waves before that
        unsigned int before = __shfl_up(mincl, 1, rr::kWave);
        if (lane == 0) before = 0;
        before = before > pre ? before : pre;
#pragma unroll
        for (int j = 0; j < R; ++j) idx[j] = (m[j] > before ? m[j] : before) - 1u;
      } else {  // multinomial (particle_filter.rs:455-470; monte_carlo_localization.rs:343-355,387-392)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; ++j)
          if (k0 + j < n) s_cdf[k0 + j] = off + c[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; ++j) {
          double r, dummy;
          if (resident) r = pre_r[j];
          else rr_uniform2(a.seed, RR_STREAM_RESAMPLE, rstep, k0 + j, &r, &dummy);
          idx[j] = k0 + j < n ? (unsigned int)rr_lower_bound_u64(s_cdf, n, rr_fix_target_multinomial(r, total)) : 0u;
        }
      }
      // ---- gather through LDS (particle_filter.rs:467-469)
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (k0 + j < n) {
          s_f[k0 + j] = x[j];
          s_f[n + k0 + j] = y[j];
          s_f[2 * n + k0 + j] = yaw[j];
          s_f[3 * n + k0 + j] = v[j];
        }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (k0 + j < n) {
          const unsigned int i = idx[j];
          x[j] = s_f[i];
          y[j] = s_f[n + i];
          yaw[j] = s_f[2 * n + i];
          v[j] = s_f[3 * n + i];
          last_idx[j] = i;
          if (a.want_est) {  // the mean of the resampled set, uniform weights (:343)
            est_acc[0] += x[j];
            est_acc[1] += y[j];
            est_acc[2] += yaw[j];
            est_acc[3] += v[j];
          }
        }
      c_den = (double)n;
    }
    RR_RES_TL(5);  // resampled / gathered
    if (a.want_est) {
      small_block_sum4<BLOCK>(est_acc, s_red);
#pragma unroll
      for (int k = 0; k < 4; ++k) c_est[k] = est_acc[k];
      if (est_out) {  // K x 4 estimates: collected in LDS, flushed (coalesced) every kEstRing steps -- no global store per step
        if (tid == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) s_ring[(s % kEstRing) * 4 + k] = c_est[k] / c_den;
        }
        if ((s % kEstRing) == kEstRing - 1 || s == a.K - 1) {
          __syncthreads();
          const int s0 = s - (s % kEstRing), cnt = (s - s0 + 1) * 4;
          if (tid < cnt) est_out[4 * (size_t)s0 + tid] = s_ring[tid];
        }
      }
      if (resident && tid < 4) {  // the step's answer: four self-vouching pairs, no fence, no separate stamp
        const double e = tid == 0 ? c_est[0] : tid == 1 ? c_est[1] : tid == 2 ? c_est[2] : c_est[3];
        rr::store_pair_sys(&ring->rsp[tid], (uint64_t)__double_as_longlong(e / c_den), a.res.first_seq + (uint64_t)s);
      }
#if defined(RR_PLAN_TIMELINE)
      if (resident && tid == 0) {  // [6]: the estimate summed and its four pairs issued; [7]: did the resample fire
        res_tl[6] = wall_clock64();
        res_tl[7] = (uint64_t)fire;
        for (int k = 0; k < 8; ++k) rr::store_pair_sys(&ring->rsp[8 + k], res_tl[k], a.res.first_seq + (uint64_t)s);
      }
#endif
    }
  }
  if (resident) __syncthreads();
  // a resident incarnation that served no step leaves the particle set, the weights and Ctl as it found them
  const bool write_back = !resident || steps_done > 0;
  // ---- the state after the last step
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint64_t k = k0 + j;
    if (k < n && write_back) {
      b.x[cur][k] = x[j];
      b.y[cur][k] = y[j];
      b.yaw[cur][k] = yaw[j];
      b.v[cur][k] = v[j];
      w[k] = wgt[j];
      if (idx_out && any_fired) idx_out[k] = last_idx[j];
    }
  }
  if (tid == 0 && write_back) {
    ctl->weights_uniform = c_fired ? 1 : 0;
    ctl->usable = c_usable;
    ctl->image_mode = c_mode;
    ctl->shift = c_shift;
    ctl->wmax = c_wmax;
    ctl->pending = 0;
    PlanArgs pa{};
    pa.n_global = n;
    pa.mode = 2;  // sums only: the decision, the plan and the flags are set below
    rr::finalize_plan(ctl, c_total, 0, c_total, c_q2, pa);
    ctl->fired = c_fired;
    ctl->wmax_bits = 0;
    if (c_fired && a.scheme == RR_RESAMPLE_SYSTEMATIC) {
      ctl->rho = c_rho;
      ctl->plan = c_plan;
      ctl->served_first = 0;
      ctl->served_count = n;
    }
    if (a.want_est) {
      for (int k = 0; k < 4; ++k) est_partials[k] = c_est[k];
      ctl->est_denom = c_den;
      ctl->est_step = (uint64_t)(a.rstep0 + (unsigned int)(resident ? steps_done : a.K) - 1) + 1;
      ctl->est_kind = rr::kEstPlanTiles;
      if (a.mail_seq) {
        for (int k = 0; k < 4; ++k)
          __hip_atomic_store(reinterpret_cast<uint64_t*>(&mail->est[k]), (uint64_t)__double_as_longlong(c_est[k] / c_den), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mail->seq, a.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (resident && tid == 0) {  // EXIT marker: the last command this incarnation consumed (a quit counts), stamped with the launch id
    const uint64_t consumed = a.res.first_seq + (uint64_t)steps_done - 1 + (res_last_op == rr::kResOpQuit ? 1 : 0);
    rr::store_pair_sys(&ring->rsp[rr::kResRspExit], consumed, a.res.launch_id);
  }
}

// initial clouds
__global__ __launch_bounds__(kBlock) void k_init(Bufs b, double* __restrict__ w, uint64_t n,
                                                uint64_t n_global, uint64_t first_gid,
                                                uint64_t seed, int jitter, double s0, double s1,
                                                double s2, double s3) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double x = 0.0, y = 0.0, yaw = 0.0, v = 0.0;
  if (jitter) {
    const double st[4] = {s0, s1, s2, s3};
    rr_pf_init_one(seed, first_gid + i, st, &x, &y, &yaw, &v);
  }
  b.x[0][i] = x;
  b.y[0][i] = y;
  b.yaw[0][i] = yaw;
  b.v[0][i] = v;
  w[i] = 1.0 / (double)n_global;
}

// AoS (x,y,yaw,v,w) <-> SoA
__global__ __launch_bounds__(kBlock) void k_pack_aos(Bufs b, const double* __restrict__ w,
                                                    const Ctl* __restrict__ ctl, uint64_t n,
                                                    uint64_t n_global, double* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int cur = ctl->cur;
  double wi;
  if (ctl->weights_uniform || ctl->image_mode != rr::kImageWeights) wi = 1.0 / (double)n_global;
  else wi = w[i] / ctl->sum;
  out[5 * i] = b.x[cur][i];
  out[5 * i + 1] = b.y[cur][i];
  out[5 * i + 2] = b.yaw[cur][i];
  out[5 * i + 3] = b.v[cur][i];
  out[5 * i + 4] = wi;
}

__global__ __launch_bounds__(kBlock) void k_unpack_aos(Bufs b, double* __restrict__ w, Ctl* __restrict__ ctl,
                                                      uint64_t n, const double* __restrict__ in) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const int cur = ctl->cur;
  double wi = 0.0;
  if (i < n) {
    b.x[cur][i] = in[5 * i];
    b.y[cur][i] = in[5 * i + 1];
    b.yaw[cur][i] = in[5 * i + 2];
    b.v[cur][i] = in[5 * i + 3];
    wi = in[5 * i + 4];
    w[i] = wi;
  }
  double m = wi > 0.0 ? wi : 0.0;
  m = rr::wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.0) rr::atomic_max_u64(&ctl->wmax_bits, rr_d2u(m));
  if (i == 0) ctl->weights_uniform = 0;
}

}  // namespace

// resolve + gather of the slots this rank serves, stored straight into the owners' slabs:
// global slot s -> rank s / n_local, local index s % n_local, buffer set `Ctl.cur` (every rank
// flips in lockstep: the gate decision is a function of the global integer sums)
__global__ __launch_bounds__(kBlock) void k_resolve_gather_p2p(Bufs b, const Ctl* __restrict__ ctl,
                                                              unsigned int* __restrict__ markers,
                                                              const unsigned int* __restrict__ carry,
                                                              P2PPeers peers, uint64_t n_local) {
  if (!ctl->fired) return;
  const uint64_t first = ctl->served_first, n_slots = ctl->served_count;
  const uint64_t tile_base = (uint64_t)blockIdx.x * rr::kResolveSlots;
  if (tile_base >= n_slots) return;  // uniform per workgroup
  unsigned int idx[rr::kResolveRows];
  rr::resolve_tile(markers, carry, n_slots, blockIdx.x, idx);
  const int dst = ctl->cur, src = dst ^ 1;
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + threadIdx.x;
    if (k < n_slots) {
      const uint64_t s = first + k;
      const uint64_t d = s / n_local, li = s - d * n_local;
      const uint64_t j = idx[r];
      double* __restrict__ out = peers.slab[d] + (size_t)(4 * dst) * n_local;
      out[li] = b.x[src][j];
      out[n_local + li] = b.y[src][j];
      out[2 * n_local + li] = b.yaw[src][j];
      out[3 * n_local + li] = b.v[src][j];
    }
  }
}

// Sharded lazy resample, phase D: deliver what this shard serves to OTHER ranks.  The window of positions this shard's
// sources feed, [served_first, +served_count) + pad, sticks out of its own block [own0, own0 + n) on either side by the
// drift of the cumulative weight across the block boundaries (10^3 - 10^4 slots of 10^6 in steady state, everything in
// the worst case): those positions are resolved here, tile by tile (grid-stride over the foreign tiles only), and each
// particle is stored into the owning rank's fine-grained inbox, its four fields and the seal that vouches for them in one go
// (rr::inbox_put).  Own positions are left to the next step's k_step_lazy.
__global__ __launch_bounds__(kBlock) void k_push_window(Bufs b, const Ctl* __restrict__ ctl,
                                                       unsigned int* __restrict__ markers,
                                                       const unsigned int* __restrict__ carry, P2PPeers peers,
                                                       uint64_t n_local, uint64_t pad, uint64_t seq) {
  if (!ctl->fired) return;
  const uint64_t own0 = (uint64_t)peers.rank * n_local + pad, own1 = own0 + n_local;
  const uint64_t win_lo = ctl->served_first + pad, win_hi = win_lo + ctl->served_count;
  const int src = ctl->cur;  // lazy: Ctl.cur flips when the next step settles
  // foreign positions: left of the own block [l_lo, l_hi), right of it [r_lo, r_hi)
  const uint64_t S = rr::kResolveSlots;
  const uint64_t l_lo = win_lo, l_hi = own0 < win_hi ? own0 : win_hi;
  const uint64_t r_lo = own1 > win_lo ? own1 : win_lo, r_hi = win_hi;
  const uint64_t lt0 = l_lo / S, n_left = l_lo < l_hi ? (l_hi + S - 1) / S - lt0 : 0;
  const uint64_t rt0 = r_lo / S, n_right = r_lo < r_hi ? (r_hi + S - 1) / S - rt0 : 0;
  const uint64_t n_tiles = n_left + n_right;
  for (uint64_t q = blockIdx.x; q < n_tiles; q += gridDim.x) {
    const uint64_t tile = q < n_left ? lt0 + q : rt0 + (q - n_left);
    const uint64_t lo = q < n_left ? l_lo : r_lo, hi = q < n_left ? l_hi : r_hi;  // positions this kernel consumes
    unsigned int idx[rr::kResolveRows];
    rr::resolve_tile_window(markers, carry, tile, win_lo, win_hi, lo, hi, idx);
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      const uint64_t pos = tile * S + (uint64_t)r * kBlock + threadIdx.x;
      if (pos >= lo && pos < hi) {
        const uint64_t s = pos - pad;  // global slot
        const uint64_t d = s / n_local, li = s - d * n_local;
        const uint64_t j = idx[r];
        // fine-grained, [4 fields + seal][n_local]: the five words in one go, the seal vouches for them (rr::inbox_put)
        rr::inbox_put(peers.inbox[d], n_local, li, seq, b.x[src][j], b.y[src][j], b.yaw[src][j], b.v[src][j]);
      }
    }
  }
}

// RCCL transport, the same overhang into a SEND BUFFER instead of the peers' inboxes: record q (x, y, yaw, v) is the q-th
// foreign position of the window in ascending order -- left overhang, then right overhang -- which is also ascending
// destination rank, so the buffer is cut into one contiguous segment per destination.  If the overhang does not fit the
// buffer (`cap` records) the kernel leaves everything as it is and the host, which learns the size a moment later, grows
// the buffer and launches it again.
__global__ __launch_bounds__(kBlock) void k_pack_window(Bufs b, const Ctl* __restrict__ ctl, unsigned int* __restrict__ markers,
                                                       const unsigned int* __restrict__ carry, int rank, uint64_t n_local,
                                                       uint64_t pad, double* __restrict__ send, uint64_t cap) {
  if (!ctl->fired) return;
  const uint64_t own0 = (uint64_t)rank * n_local + pad, own1 = own0 + n_local;
  const uint64_t win_lo = ctl->served_first + pad, win_hi = win_lo + ctl->served_count;
  const int src = ctl->cur;
  const uint64_t S = rr::kResolveSlots;
  const uint64_t l_lo = win_lo, l_hi = own0 < win_hi ? own0 : win_hi;
  const uint64_t r_lo = own1 > win_lo ? own1 : win_lo, r_hi = win_hi;
  const uint64_t n_lpos = l_lo < l_hi ? l_hi - l_lo : 0, n_rpos = r_lo < r_hi ? r_hi - r_lo : 0;
  if (n_lpos + n_rpos > cap) return;
  const uint64_t lt0 = l_lo / S, n_left = n_lpos ? (l_hi + S - 1) / S - lt0 : 0;
  const uint64_t rt0 = r_lo / S, n_right = n_rpos ? (r_hi + S - 1) / S - rt0 : 0;
  for (uint64_t q = blockIdx.x; q < n_left + n_right; q += gridDim.x) {
    const bool left = q < n_left;
    const uint64_t tile = left ? lt0 + q : rt0 + (q - n_left);
    const uint64_t lo = left ? l_lo : r_lo, hi = left ? l_hi : r_hi;
    unsigned int idx[rr::kResolveRows];
    rr::resolve_tile_window(markers, carry, tile, win_lo, win_hi, lo, hi, idx);
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      const uint64_t pos = tile * S + (uint64_t)r * kBlock + threadIdx.x;
      if (pos >= lo && pos < hi) {
        const uint64_t rec = left ? pos - l_lo : n_lpos + (pos - r_lo);
        const uint64_t j = idx[r];
        double* __restrict__ o = send + 4 * rec;
        o[0] = b.x[src][j];
        o[1] = b.y[src][j];
        o[2] = b.yaw[src][j];
        o[3] = b.v[src][j];
      }
    }
  }
}

// received records -> this rank's inbox: record q is the q-th own slot OUTSIDE the own window, in ascending order
// (`below` of them lie below the window, the rest above the `self` slots this rank serves to itself)
__global__ __launch_bounds__(kBlock) void k_unpack_inbox(const double* __restrict__ recv, uint64_t n_recv, uint64_t below, uint64_t self,
                                                        uint64_t n, double* __restrict__ inbox) {
  const uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n_recv) return;
  const uint64_t k = q < below ? q : q + self;
  if (k >= n) return;
  const double* __restrict__ r = recv + 4 * q;
  inbox[k] = r[0];
  inbox[n + k] = r[1];
  inbox[2 * n + k] = r[2];
  inbox[3 * n + k] = r[3];
}

// accessors: make a pending window resample real -- own slots inside the window through the markers, the others out of
// the inbox (peer-to-peer transport: each as soon as its seal fits); k_settle flips the live set afterwards
__global__ __launch_bounds__(kBlock) void k_resolve_gather_window(Bufs b, const Ctl* __restrict__ ctl,
                                                                 unsigned int* __restrict__ markers,
                                                                 const unsigned int* __restrict__ carry, uint64_t n,
                                                                 uint64_t first_gid, uint64_t pad,
                                                                 const double* __restrict__ inbox,
                                                                 unsigned int* __restrict__ idx_out, uint64_t wait_seq,
                                                                 uint64_t timeout_ticks, int* __restrict__ err) {
  if (!ctl->pending) return;
  const uint64_t own0 = first_gid + pad;
  const uint64_t win_lo = ctl->served_first + pad, win_hi = win_lo + ctl->served_count;
  unsigned int idx[rr::kResolveRows];
  rr::resolve_tile_window(markers, carry, own0 / rr::kResolveSlots + blockIdx.x, win_lo, win_hi, own0, own0 + n, idx);
  const int src = ctl->cur, dst = src ^ 1;
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = (uint64_t)blockIdx.x * rr::kResolveSlots + (uint64_t)r * kBlock + threadIdx.x;
    if (k >= n) continue;
    const uint64_t pos = own0 + k;
    if (pos < win_lo || pos >= win_hi) {
      double f[4];
      (void)rr::inbox_take(inbox, n, k, wait_seq, timeout_ticks, err, f);
      b.x[dst][k] = f[0];
      b.y[dst][k] = f[1];
      b.yaw[dst][k] = f[2];
      b.v[dst][k] = f[3];
      if (idx_out) idx_out[k] = kInPlace;
    } else {
      const uint64_t j = idx[r];
      b.x[dst][k] = b.x[src][j];
      b.y[dst][k] = b.y[src][j];
      b.yaw[dst][k] = b.yaw[src][j];
      b.v[dst][k] = b.v[src][j];
      if (idx_out) idx_out[k] = (unsigned int)j;
    }
  }
}

// make a pending sharded-lazy resample real (accessors): copy the locally-sourced slots, leave
// the ones a peer stored where they are; k_settle flips the live set afterwards
__global__ __launch_bounds__(kBlock) void k_gather_lidx(Bufs b, const Ctl* __restrict__ ctl,
                                                       unsigned int* __restrict__ lidx, uint64_t n,
                                                       const double* __restrict__ inbox) {
  if (!ctl->pending) return;
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const unsigned int j = lidx[k];
  const int src = ctl->cur, dst = src ^ 1;
  if (j == kInPlace) {  // delivered by a peer: take it out of the inbox (multinomial, unsharded: never happens)
    if (!inbox) return;
    b.x[dst][k] = inbox[k];
    b.y[dst][k] = inbox[n + k];
    b.yaw[dst][k] = inbox[2 * n + k];
    b.v[dst][k] = inbox[3 * n + k];
    return;
  }
  b.x[dst][k] = b.x[src][j];
  b.y[dst][k] = b.y[src][j];
  b.yaw[dst][k] = b.yaw[src][j];
  b.v[dst][k] = b.v[src][j];
  lidx[k] = kInPlace;
}

// The deferred in-step estimate when an accessor moved the particles before the next step did (materialise): the same sums over
// the live set -- slot k now HOLDS its source's fields --, the same slot tiles, the same order as k_step_lazy (est_slots_partial).
__global__ __launch_bounds__(kBlock) void k_est_slots(Bufs b, const Ctl* __restrict__ ctl, uint64_t n, double* __restrict__ partials) {
  if (ctl->est_kind != rr::kEstSlotTiles) return;  // the gate stayed shut: the plan kernel has formed the weighted mean
  const int cur = ctl->cur, tid = threadIdx.x;
  const uint64_t tile_base = (uint64_t)blockIdx.x * rr::kResolveSlots;
  double f[4][rr::kResolveRows];
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
    const bool in = k < n;
    f[0][r] = in ? b.x[cur][k] : 0.0;
    f[1][r] = in ? b.y[cur][k] : 0.0;
    f[2][r] = in ? b.yaw[cur][k] : 0.0;
    f[3][r] = in ? b.v[cur][k] : 0.0;
  }
  double acc[4];
  rr::est_rows_sum<rr::kResolveRows>(f, acc);
  rr::est_wave_store<kBlock>(acc, partials, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// KLD-adaptive resampling (monte_carlo_localization.rs:322-385).  The reference draws one
// particle at a time and stops at the first count that satisfies the KLD bound for the number of
// bins occupied so far.  Here all max_particles candidate draws are evaluated at once:
//   k_kld_draw    draw m -> source index (multinomial, as k_resample_gather_mn) and its bin,
//   k_kld_insert  exact "first draw with this bin" through an open-addressing table keyed by the
//                 full (x, y, yaw) bin triple: a slot is claimed once (CAS) by some draw, later
//                 draws compare their bin with the claimant's and keep the minimum draw index,
//   k_kld_count   one workgroup: occupied-bin count after every draw (scan of the first-occurrence
//                 flags), running maximum of rr_kld_required, first draw satisfying rr_kld_stop.
// The new count goes to the host (the only synchronisation of an adaptive step); a plain gather
// of that many particles follows.
constexpr unsigned int kKldEmpty = 0xffffffffu;

// ------------------------------------------------------------------------------------------
// Sharded MULTINOMIAL resample (the resampler MonteCarloLocalizer uses, monte_carlo_localization.rs:322-365,387-392,
// and particle_filter.rs:441-473).  Draw k belongs to output slot k of the GLOBAL particle index and is a pure
// function of (seed, resample step, k), so every shard can evaluate all n_global targets; the shard whose CDF
// interval (base, base + T_local] contains target_k serves slot k.  Unlike the systematic plan the served slots
// are scattered, so they are COMPACTED in slot order: tiles of 2048 slots that never straddle a destination rank
// (destination d owns slots [d n_local, (d+1) n_local)), count -> scan -> write.  The send buffer is therefore
// ordered by global slot, i.e. grouped by destination, and its per-destination counts are row `rank` of the
// exchange matrix.  Records are 5 doubles: x, y, yaw, v and the destination's LOCAL slot index.
struct MnSelectArgs {
  uint64_t n_local, n_global, tiles_per_dest;
  uint64_t seed;
  unsigned int rstep;
  int n_shards;
};

__device__ inline bool mn_slot_target(const Ctl* __restrict__ ctl, const MnSelectArgs& a, uint64_t tile, int item,
                                      uint64_t* slot_out, uint64_t* target_out) {
  const uint64_t d = tile / a.tiles_per_dest, t = tile - d * a.tiles_per_dest;
  const uint64_t li = t * kTile + (uint64_t)threadIdx.x * rr::kItems + item;
  if (li >= a.n_local) return false;
  const uint64_t slot = d * a.n_local + li;
  double r, dummy;
  rr_uniform2(a.seed, RR_STREAM_RESAMPLE, a.rstep, slot, &r, &dummy);
  const uint64_t target = rr_fix_target_multinomial(r, ctl->total);
  *slot_out = slot;
  *target_out = target;
  return target > ctl->base && target <= ctl->base + ctl->total_local;
}

__global__ __launch_bounds__(rr::kTileBlock) void k_mn_select_count(const Ctl* __restrict__ ctl, MnSelectArgs a,
                                                           unsigned int* __restrict__ tile_cnt) {
  __shared__ unsigned int s_c[rr::kTileBlock / rr::kWave];
  unsigned int c = 0;
  if (ctl->fired) {
#pragma unroll
    for (int j = 0; j < rr::kItems; ++j) {
      uint64_t slot, target;
      c += mn_slot_target(ctl, a, blockIdx.x, j, &slot, &target) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, rr::kWave);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int k = 0; k < rr::kTileBlock / rr::kWave; ++k) t += s_c[k];
    tile_cnt[blockIdx.x] = t;
  }
}

// one workgroup: exclusive scan of the tile counts (in place), counts per destination
__global__ __launch_bounds__(kScanThreads) void k_mn_select_scan(unsigned int* __restrict__ tile_cnt, uint64_t n_tiles,
                                                                uint64_t tiles_per_dest, int n_shards,
                                                                uint64_t* __restrict__ counts_out) {
  __shared__ uint64_t s_w[kScanThreads / rr::kWave];
  __shared__ uint64_t s_dest[kMaxP2P + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
  const uint64_t lo = (uint64_t)tid * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
  uint64_t local = 0;
  for (uint64_t k = lo; k < hi; ++k) local += tile_cnt[k];
  const uint64_t incl = rr::wave_scan_u64(local, lane);
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  uint64_t run = incl - local;
  for (int k = 0; k < wv; ++k) run += s_w[k];
  for (uint64_t k = lo; k < hi; ++k) {
    if (k % tiles_per_dest == 0) s_dest[k / tiles_per_dest] = run;  // first tile of a destination: its block starts here
    const unsigned int t = tile_cnt[k];
    tile_cnt[k] = (unsigned int)run;
    run += t;
  }
  if (tid == kScanThreads - 1 || hi == n_tiles) {
    if (hi == n_tiles && lo < hi) s_dest[n_shards] = run;  // grand total (the thread that owns the last tile)
  }
  __syncthreads();
  if (tid < n_shards) counts_out[tid] = s_dest[tid + 1] - s_dest[tid];
}

__global__ __launch_bounds__(rr::kTileBlock) void k_mn_select_pack(Bufs b, const Ctl* __restrict__ ctl, MnSelectArgs a,
                                                          const unsigned int* __restrict__ tile_off,
                                                          const uint64_t* __restrict__ cdf, uint64_t n_src,
                                                          double* __restrict__ out) {
  if (!ctl->fired) return;
  __shared__ unsigned int s_c[rr::kTileBlock / rr::kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint64_t slot[rr::kItems], target[rr::kItems];
  bool mine[rr::kItems];
  unsigned int c = 0;
#pragma unroll
  for (int j = 0; j < rr::kItems; ++j) {
    mine[j] = mn_slot_target(ctl, a, blockIdx.x, j, &slot[j], &target[j]);
    c += mine[j] ? 1u : 0u;
  }
  unsigned int incl = c;
#pragma unroll
  for (int o = 1; o < rr::kWave; o <<= 1) {
    const unsigned int t = __shfl_up(incl, o, rr::kWave);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_c[wv] = incl;
  __syncthreads();
  unsigned int pos = tile_off[blockIdx.x] + incl - c;
  for (int k = 0; k < wv; ++k) pos += s_c[k];
  const int src = ctl->cur ^ 1;  // the plan kernel has flipped Ctl.cur: the weighted set is the other one
#pragma unroll
  for (int j = 0; j < rr::kItems; ++j) {
    if (!mine[j]) continue;
    const uint64_t i = rr_lower_bound_u64(cdf, n_src, target[j]);
    double* __restrict__ o = out + 5 * (uint64_t)pos;
    o[0] = b.x[src][i];
    o[1] = b.y[src][i];
    o[2] = b.yaw[src][i];
    o[3] = b.v[src][i];
    o[4] = (double)(slot[j] % a.n_local);
    ++pos;
  }
}

// received records -> the live buffer set, each to the local slot it names
__global__ __launch_bounds__(kBlock) void k_adopt_records(Bufs b, const Ctl* __restrict__ ctl,
                                                         const double* __restrict__ in, uint64_t n) {
  if (!ctl->fired) return;
  const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= n) return;
  const int dst = ctl->cur;
  const uint64_t k = (uint64_t)in[5 * r + 4];
  if (k >= n) return;
  b.x[dst][k] = in[5 * r];
  b.y[dst][k] = in[5 * r + 1];
  b.yaw[dst][k] = in[5 * r + 2];
  b.v[dst][k] = in[5 * r + 3];
}

__global__ __launch_bounds__(kBlock) void k_kld_draw(Bufs b, const Ctl* __restrict__ ctl,
                                                    const uint64_t* __restrict__ cdf,
                                                    const uint64_t* __restrict__ coarse, int coarse_log2,
                                                    uint64_t n_coarse, const double* __restrict__ r_explicit,
                                                    unsigned int* __restrict__ idx, int32_t* __restrict__ keys,
                                                    uint64_t n_src, uint64_t n_draws, uint64_t seed, unsigned int rstep, int dyn_n) {
  extern __shared__ uint64_t s_coarse[];
  if (dyn_n) {  // the current particle count lives on the device (Ctl.n_active); the launch was sized for the capacity
    n_src = ctl->n_active;
    n_coarse = (n_src + (1ull << coarse_log2) - 1) >> coarse_log2;
  }
  for (uint64_t i = threadIdx.x; i < n_coarse; i += kBlock) s_coarse[i] = coarse[i];
  __syncthreads();
  const uint64_t m = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= n_draws) return;
  const int src = ctl->cur ^ 1;  // the plan kernel flipped Ctl.cur already
  const uint64_t target = rr::resample_target(ctl, RR_RESAMPLE_MULTINOMIAL, m, seed, rstep, r_explicit, m);
  const uint64_t blk = rr_lower_bound_u64(s_coarse, n_coarse, target);
  const uint64_t lo = blk << coarse_log2;
  const uint64_t len = lo + (1ull << coarse_log2) <= n_src ? (1ull << coarse_log2) : n_src - lo;
  uint64_t j = lo + rr_lower_bound_u64(cdf + lo, len, target);
  if (j >= n_src) j = n_src - 1;
  idx[m] = (unsigned int)j;
  int32_t xb, yb, ab;
  rr_kld_bin(b.x[src][j], b.y[src][j], b.yaw[src][j], &xb, &yb, &ab);
  keys[3 * m] = xb;
  keys[3 * m + 1] = yb;
  keys[3 * m + 2] = ab;
}

__device__ inline uint64_t kld_hash(int32_t a, int32_t b, int32_t c) {
  uint64_t h = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
  h ^= (uint64_t)(uint32_t)c * 0x9e3779b97f4a7c15ull;
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return h;
}

// One lane per DISTINCT value of `slot` among the wave's valid lanes: the lowest such lane (its draw index is the smallest of
// the group, draw indices ascend with the lane).  A tracking filter's draws fall into a few dozen bins, and same-address
// atomics are carried out one after the other at the memory side (7.5 ns each): 5000 draws claiming and lowering ~100
// slots took 36 us; with one atomic per bin and wave, and none when a look shows that nothing would change, 5 us.
__device__ inline bool wave_first_of_slot(uint32_t slot, bool valid) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  bool first = false;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t s_l = (uint32_t)__builtin_amdgcn_readlane((int)slot, leader);
    const unsigned long long same = __ballot(valid && slot == s_l);
    if (lane == leader) first = true;
    todo &= ~same;
  }
  return first;
}

// draw m's bin into the table (all lanes of a wave call this together; valid == false: the lane only takes part in the
// wave-wide steps)
__device__ inline void kld_insert_one(uint64_t m, bool valid, const int32_t* __restrict__ keys, unsigned int* __restrict__ table,
                                      unsigned int* __restrict__ minslot, unsigned int* __restrict__ myslot, uint64_t hash_size) {
  int32_t a = 0, bb = 0, c = 0;
  if (valid) {
    a = keys[3 * m];
    bb = keys[3 * m + 1];
    c = keys[3 * m + 2];
  }
  uint64_t s = kld_hash(a, bb, c) & (hash_size - 1);
  // the home slot of a bin is claimed once per wave, not once per draw
  if (wave_first_of_slot((uint32_t)s, valid) && __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kKldEmpty)
    (void)atomicCAS(&table[s], kKldEmpty, (unsigned int)m);
  bool placed = !valid;
  while (!placed) {
    unsigned int o = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o == kKldEmpty) {
      o = atomicCAS(&table[s], kKldEmpty, (unsigned int)m);
      if (o == kKldEmpty) o = (unsigned int)m;
    }
    if (o == m || (keys[3 * (uint64_t)o] == a && keys[3 * (uint64_t)o + 1] == bb && keys[3 * (uint64_t)o + 2] == c)) placed = true;
    else s = (s + 1) & (hash_size - 1);  // another bin lives here: linear probing (the table is at most half full)
  }
  if (valid) myslot[m] = (unsigned int)s;
  // the bin's smallest draw index: the wave's smallest draw of the bin speaks for the wave, and only if a look says it matters
  if (wave_first_of_slot((uint32_t)s, valid) && __hip_atomic_load(&minslot[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned int)m)
    atomicMin(&minslot[s], (unsigned int)m);
}

__global__ __launch_bounds__(kBlock) void k_kld_insert(const int32_t* __restrict__ keys, unsigned int* __restrict__ table,
                                                      unsigned int* __restrict__ minslot,
                                                      unsigned int* __restrict__ myslot, uint64_t n_draws,
                                                      uint64_t hash_size) {
  const uint64_t m = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  kld_insert_one(m, m < n_draws, keys, table, minslot, myslot, hash_size);
}

constexpr int kKldThreads = 1024;
constexpr uint64_t kKldWipeInKernel = 16384;  // candidate draws up to which k_kld_count wipes the bin table itself
// One workgroup of kKldThreads: the number of draws the reference's loop makes before it stops (:340-352) -- occupied-bin
// count after every draw (scan of the first-occurrence flags), running maximum of the KLD bound, first draw that satisfies
// the stop rule.  Every thread gets the result.  Then the bin table and the first-occurrence slots are wiped for the NEXT
// resample (both arrays are cleared once when the filter is created).
__device__ inline uint64_t kld_count_body(unsigned int* __restrict__ minslot, const unsigned int* __restrict__ myslot, uint64_t n_draws,
                                          const rr_mcl_adaptive& kld, unsigned int* __restrict__ table, uint64_t hash_size) {
  __shared__ uint64_t s_cnt[kKldThreads / rr::kWave];
  __shared__ uint64_t s_req[kKldThreads / rr::kWave];
  __shared__ uint64_t s_stop;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint64_t k_carry = 0, req_carry = kld.min_particles;
  if (tid == 0) s_stop = ~0ull;
  __syncthreads();
  for (uint64_t base = 0; base < n_draws; base += kKldThreads) {
    const uint64_t m = base + tid;
    // (device-scope load: in the one-launch adaptive step the minima were formed by atomics of this very launch)
    const uint64_t flag = (m < n_draws && __hip_atomic_load(&minslot[myslot[m]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)m) ? 1ull : 0ull;
    // occupied bins after draw m
    uint64_t incl = rr::wave_scan_u64(flag, lane);
    if (lane == 63) s_cnt[wv] = incl;
    __syncthreads();
    uint64_t off = k_carry;
    for (int q = 0; q < wv; ++q) off += s_cnt[q];
    const uint64_t k = off + incl;
    uint64_t chunk_total = 0;
    for (int q = 0; q < kKldThreads / rr::kWave; ++q) chunk_total += s_cnt[q];
    // running maximum of the bound (:350: required = required.max(...))
    uint64_t req = m < n_draws ? rr_kld_required(k, kld.min_particles, kld.max_particles, kld.kld_epsilon, kld.kld_z) : 0;
#pragma unroll
    for (int o = 1; o < rr::kWave; o <<= 1) {
      const uint64_t t = rr::shfl_up_u64(req, o);
      if (lane >= o && t > req) req = t;
    }
    if (lane == 63) s_req[wv] = req;
    __syncthreads();
    uint64_t pre = req_carry;
    for (int q = 0; q < wv; ++q) pre = s_req[q] > pre ? s_req[q] : pre;
    if (pre > req) req = pre;
    uint64_t chunk_req = req_carry;
    for (int q = 0; q < kKldThreads / rr::kWave; ++q) chunk_req = s_req[q] > chunk_req ? s_req[q] : chunk_req;
    if (m < n_draws && rr_kld_stop(m, req, kld.min_particles)) atomicMin((unsigned long long*)&s_stop, (unsigned long long)m);
    __syncthreads();
    if (s_stop != ~0ull) break;  // uniform: read after the barrier
    k_carry += chunk_total;
    req_carry = chunk_req;
    __syncthreads();
  }
  const uint64_t n_new = s_stop == ~0ull ? n_draws : s_stop + 1;  // :342: at most max_particles
  // every read of minslot[] above happened before a barrier all threads have passed (the loop ends with one, or breaks
  // right after one): the table can go
  // (one workgroup wipes the slots the draws used; beyond kKldWipeInKernel draws the host wipes both arrays with two wide
  // memsets behind the launch instead -- 2 x 4 x hash_size bytes through one workgroup would cost more than the step)
  __syncthreads();
  if (n_draws <= kKldWipeInKernel) {
    for (uint64_t m = tid; m < n_draws; m += kKldThreads) {
      const unsigned int slot = myslot[m];
      table[slot] = kKldEmpty;
      minslot[slot] = kKldEmpty;
    }
  }
  (void)hash_size;
  return n_new;
}

__global__ __launch_bounds__(kKldThreads) void k_kld_count(unsigned int* __restrict__ minslot,
                                                          const unsigned int* __restrict__ myslot, uint64_t n_draws,
                                                          rr_mcl_adaptive kld, uint64_t* __restrict__ out,
                                                          unsigned int* __restrict__ table, uint64_t hash_size, Bufs b,
                                                          Ctl* __restrict__ ctl, const unsigned int* __restrict__ idx, int gather) {
  const uint64_t n_new = kld_count_body(minslot, myslot, n_draws, kld, table, hash_size);
  const int tid = threadIdx.x;
  if (tid == 0) out[0] = n_new;
  if (gather) {  // a filter of the reference's sizes (<= 16 384 candidate draws): k_kld_gather_dyn's work on the way, one launch less
    const int dst = ctl->cur, src = dst ^ 1;
    for (uint64_t k = tid; k < n_new; k += kKldThreads) copy_particle(b, src, dst, idx[k], k, false, nullptr);
    if (tid == 0) ctl->n_active = n_new;  // (nothing in this launch reads it)
  }
}

// ------------------------------------------------------------------------------------------
// The adaptive step of a filter of the reference's sizes (MonteCarloLocalizationConfig::default(): 100 - 5 000 particles) in
// ONE launch of ONE workgroup: propagate + weight (k_propagate_weight), integer image and CDF (k_quantize_reduce, k_plan_cdf,
// finalize_plan), the max_particles candidate draws and their bins (k_kld_draw), the bin table (k_kld_insert), the stop rule
// (k_kld_count) and the gather -- six launches of a few microseconds of work each otherwise (36 us a step).  Same
// per-element arithmetic, same integer sums, same draws: bit-identical to the six kernels (tests/test_gpu_kld_adaptive.py runs
// both routes).  The particle count comes from Ctl.n_active and goes back there.
struct AdaptSmallArgs {
  ImageArgs img;
  PlanArgs plan;
  rr_mcl_adaptive kld;
  uint64_t max_draws;
  uint64_t hash_size;
  rr::ResidentArgs res;  // res.on: the kernel stays and serves one step per command of the ring (resident_core.hpp)
};
// Resident mode (rr_pf_set_resident on an adaptive filter): the body below runs once per command; the particle set, the CDF and
// the bin table live in HBM anyway (one workgroup: its writes are its own reads after a barrier), so an incarnation can leave
// at any command boundary without a write-back.  While it waits it draws the NEXT step's random numbers -- the motion noise of
// the live particles and the uniforms of all max_particles candidate draws are functions of (seed, counters, index) alone --
// into `pre` ([cap] noise v | [cap] noise w | [cap] uniforms).
__global__ __launch_bounds__(kKldThreads) void k_mcl_adaptive_small(Bufs b, double* __restrict__ w, Ctl* __restrict__ ctl, StepParams p,
                                                                   ObsArg obs_arg, AdaptSmallArgs a, uint64_t* __restrict__ cdf,
                                                                   unsigned int* __restrict__ idx, int32_t* __restrict__ keys,
                                                                   unsigned int* __restrict__ table, unsigned int* __restrict__ minslot,
                                                                   unsigned int* __restrict__ myslot, uint64_t* __restrict__ out,
                                                                   uint64_t* __restrict__ coarse, int coarse_log2, int front_only,
                                                                   HostMail* __restrict__ mail, uint64_t mail_seq,
                                                                   rr::ResidentRing* __restrict__ ring, double* __restrict__ pre,
                                                                   uint64_t cap) {
  extern __shared__ double s_dyn_a[];  // [3 n_obs] observations; resident: the command's payload (u0, u1, observations)
  constexpr int W = kKldThreads / rr::kWave;
  __shared__ double s_max[W];
  __shared__ uint64_t s_t[W], s_qh[W], s_ql[W];
  __shared__ int s_hdr[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool resident = a.res.on != 0;
  double* const s_obs = resident ? s_dyn_a + 2 : s_dyn_a;
  int n_obs = p.n_obs, res_guess = 3 + 3 * p.n_obs < 64 ? 3 + 3 * p.n_obs : 64, res_last_op = rr::kResOpNone, steps_done = 0;
  double u0 = p.u0, u1 = p.u1;
  const uint64_t res_deadline = resident ? wall_clock64() + a.res.life_ticks : 0;
#if defined(RR_PLAN_TIMELINE)
  uint64_t atl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RR_ATL(K_) do { __syncthreads(); if (resident && threadIdx.x == 0) { uint64_t t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); atl[(K_)] = t_; } } while (0)
#else
#define RR_ATL(K_) do { } while (0)
#endif
  for (int s = 0;; ++s) {
  const unsigned int step = p.step + (unsigned int)s, rstep = a.plan.rstep + (unsigned int)s;
  const uint64_t n = ctl->n_active;
  const int cur = ctl->cur;
  if (resident) {
    for (uint64_t i = tid; i < n; i += kKldThreads) rr_pf_motion_noise(p.seed, step, p.first_gid + i, p.sigma_v, p.sigma_w, &pre[i], &pre[cap + i]);
    for (uint64_t m = tid; m < a.max_draws; m += kKldThreads) {
      double dummy;
      rr_uniform2(p.seed, RR_STREAM_RESAMPLE, rstep, m, &pre[2 * cap + m], &dummy);
    }
    RR_ATL(0);
    res_last_op = rr::resident_fetch<kKldThreads>(ring, a.res.first_seq + (uint64_t)s, a.res.idle_ticks, res_deadline, a.res.payload_cap,
                                                  res_guess, s_dyn_a, s_hdr);
    if (res_last_op != rr::kResOpStep) break;
    n_obs = (s_hdr[1] - 2) / 3;
    u0 = s_dyn_a[0];
    u1 = s_dyn_a[1];
  } else {
    for (int i = tid; i < 3 * n_obs; i += kKldThreads) s_obs[i] = obs_arg.v[i];
    __syncthreads();
  }
  steps_done = s + 1;
  RR_ATL(1);
  // ---- propagate + weight, in place on the live set (k_propagate_weight<true, true, false>)
  double wmax_local = 0.0;
  for (uint64_t i = tid; i < n; i += kKldThreads) {
    double x = b.x[cur][i], y = b.y[cur][i], yaw = b.yaw[cur][i], v, na, nc;
    if (resident) {
      na = pre[i];
      nc = pre[cap + i];
    } else {
      rr_pf_motion_noise(p.seed, step, p.first_gid + i, p.sigma_v, p.sigma_w, &na, &nc);
    }
    rr_pf_propagate_one(&x, &y, &yaw, &v, u0, u1, p.dt, na, nc);
    b.x[cur][i] = x;
    b.y[cur][i] = y;
    b.yaw[cur][i] = yaw;
    b.v[cur][i] = v;
    const double wgt = p.lik_mode == RR_LIK_PRODUCT ? rr_pf_weight_product(x, y, s_obs, n_obs, p.lik) : rr_pf_weight_fused(x, y, s_obs, n_obs, p.lik);
    w[i] = wgt;
    if (wgt > wmax_local) wmax_local = wgt;  // NaN and negatives drop out
  }
  {
    const double m = rr::wave_max(wmax_local);
    if (lane == 0) s_max[wv] = m;
  }
  __syncthreads();
  double wmax = s_max[0];
  for (int k = 1; k < W; ++k) wmax = s_max[k] > wmax ? s_max[k] : wmax;
  RR_ATL(2);
  // ---- integer image (quantize_reduce_tile; the weights have just been set, so Ctl.weights_uniform does not apply)
  const bool usable = wmax > 0.0 && wmax < INFINITY;
  const int mode = usable ? (int)rr::kImageWeights : a.img.degenerate;
  const int shift = usable ? rr_fix_shift(wmax, n) : 0;
  // ---- inclusive integer CDF, total and sum of squares
  uint64_t carry = 0;
  u128 q2 = {0, 0};
  for (uint64_t base = 0; base < n; base += kKldThreads) {
    const uint64_t i = base + tid;
    const uint64_t q = rr::quantize_at(w, i, n, mode, shift, 0, n);
    u128 sq;
    rr_mul64wide(q, q, &sq.hi, &sq.lo);
    q2 = rr::add128(q2, sq);
    const uint64_t incl = rr::wave_scan_u64(q, lane);
    __syncthreads();  // (s_t of the previous chunk has been read)
    if (lane == 63) s_t[wv] = incl;
    __syncthreads();
    uint64_t off = carry, chunk = 0;
    for (int k = 0; k < W; ++k) {
      if (k < wv) off += s_t[k];
      chunk += s_t[k];
    }
    if (i < n) {
      cdf[i] = off + incl;
      // (store_cdf: every 2^coarse_log2-th entry and the last one -- the table k_kld_draw stages in LDS)
      if ((((i + 1) & ((1ull << coarse_log2) - 1)) == 0) || i == n - 1) coarse[i >> coarse_log2] = off + incl;
    }
    carry += chunk;
  }
  q2 = rr::wave_sum_u128(q2);
  __syncthreads();
  if (lane == 0) {
    s_qh[wv] = q2.hi;
    s_ql[wv] = q2.lo;
  }
  __syncthreads();
  if (tid == 0) {
    u128 qq = {0, 0};
    for (int k = 0; k < W; ++k) qq = rr::add128(qq, u128{s_qh[k], s_ql[k]});
    ctl->weights_uniform = 0;
    ctl->usable = usable ? 1 : 0;
    ctl->image_mode = mode;
    ctl->shift = shift;
    ctl->wmax = wmax;
    PlanArgs pa = a.plan;
    pa.n_global = n;
    pa.rstep = rstep;
    rr::finalize_plan(ctl, carry, 0, carry, qq, pa);  // forced, eager: Ctl.cur flips here, the weights become uniform
  }
  if (front_only) return;  // (the draws, the table and the count follow as launches of their own; never resident)
  __syncthreads();
  RR_ATL(3);
  // ---- the candidate draws in blocks of kKldThreads, as far as the reference's loop would go (:340-352): a block's draws and
  // bins (k_kld_draw; the lower bound over the whole CDF is the index its two-level search finds), its entries in the bin
  // table (k_kld_insert), then the stop rule over the block (k_kld_count: occupied-bin count after every draw, running
  // maximum of the KLD bound) -- the first block that contains the stopping draw is the last one looked at.  A bin's smallest
  // draw index can only come from the blocks seen so far, so the flags of a block are final when it has been inserted: the
  // result is that of all max_particles candidates evaluated at once, for a fifth of the work at the default configuration
  // (5 000 candidates, a tracking filter stops around 300).
  __shared__ uint64_t s_cnt[W];
  __shared__ uint64_t s_req[W];
  __shared__ uint64_t s_stop;
  // The FIRST block of draws keeps its bin table in the LDS (the reference's loop stops after a few hundred draws when the filter
  // tracks, :340-352: the first block is then the only one).  The global table's claim / probe / lower-the-minimum chain is six
  // dependent device-scope round trips (~7 us of the step); the same chain on ds_ atomics is a few hundred cycles.  The flags
  // (is draw m the first of its bin?) do not depend on where the table lives.  Only if the loop goes on are the first block's
  // draws inserted into the global table as well (the later blocks must see their bins).
  constexpr int kLdsSlots = 2 * kKldThreads;
  __shared__ int32_t s_keys[3 * kKldThreads];
  __shared__ unsigned int s_tab[kLdsSlots], s_min[kLdsSlots];
  const int src = ctl->cur ^ 1;
  uint64_t k_carry = 0, req_carry = a.kld.min_particles, seen = 0;
  if (tid == 0) s_stop = ~0ull;
  for (int q = tid; q < kLdsSlots; q += kKldThreads) {
    s_tab[q] = kKldEmpty;
    s_min[q] = kKldEmpty;
  }
  __syncthreads();
  // ... and in SUB-blocks of kKldSub draws (four waves, one per SIMD): a tracking filter stops after 100 - 300 draws, and the
  // sixteen waves of a full block spend 4 us issuing ~500 instructions per draw for 1 024 candidates of which a quarter matter
  constexpr int kKldSub = 256;
  for (uint64_t base = 0; base < a.max_draws;) {
    const bool in_lds = base < (uint64_t)kKldThreads;
    const uint64_t span = in_lds ? (uint64_t)kKldSub : (uint64_t)kKldThreads;
    const uint64_t m = base + tid;
    const bool valid = (uint64_t)tid < span && m < a.max_draws;
    int32_t xb = 0, yb = 0, ab = 0;
    if (valid) {
      const uint64_t target = rr::resample_target(ctl, RR_RESAMPLE_MULTINOMIAL, m, p.seed, rstep, resident ? pre + 2 * cap : nullptr, m);
      uint64_t j = rr_lower_bound_u64(cdf, n, target);
      if (j >= n) j = n - 1;
      idx[m] = (unsigned int)j;
      rr_kld_bin(b.x[src][j], b.y[src][j], b.yaw[src][j], &xb, &yb, &ab);
      if (in_lds) {
        s_keys[3 * m] = xb;
        s_keys[3 * m + 1] = yb;
        s_keys[3 * m + 2] = ab;
      } else {
        keys[3 * m] = xb;
        keys[3 * m + 1] = yb;
        keys[3 * m + 2] = ab;
      }
    }
    __syncthreads();  // (a probing draw compares with the keys of the draw that owns a slot)
    uint64_t flag;
    if (in_lds) {
      unsigned int sl = (unsigned int)(kld_hash(xb, yb, ab) & (uint64_t)(kLdsSlots - 1));
      bool placed = !valid;
      while (!placed) {
        unsigned int o = s_tab[sl];
        if (o == kKldEmpty) {
          o = atomicCAS(&s_tab[sl], kKldEmpty, (unsigned int)m);
          if (o == kKldEmpty) o = (unsigned int)m;
        }
        if (o == (unsigned int)m || (s_keys[3 * o] == xb && s_keys[3 * o + 1] == yb && s_keys[3 * o + 2] == ab)) placed = true;
        else sl = (sl + 1) & (kLdsSlots - 1);
      }
      if (valid) atomicMin(&s_min[sl], (unsigned int)m);
      __syncthreads();
      flag = (valid && s_min[sl] == (unsigned int)m) ? 1ull : 0ull;
    } else {
      kld_insert_one(m, valid, keys, table, minslot, myslot, a.hash_size);
      __syncthreads();
      seen = base + kKldThreads < a.max_draws ? base + kKldThreads : a.max_draws;
      flag = (valid && __hip_atomic_load(&minslot[myslot[m]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)m) ? 1ull : 0ull;
    }
    uint64_t incl = rr::wave_scan_u64(flag, lane);
    if (lane == 63) s_cnt[wv] = incl;
    __syncthreads();
    uint64_t off = k_carry, chunk_total = 0;
    for (int q = 0; q < W; ++q) {
      if (q < wv) off += s_cnt[q];
      chunk_total += s_cnt[q];
    }
    const uint64_t k = off + incl;
    uint64_t req = valid ? rr_kld_required(k, a.kld.min_particles, a.kld.max_particles, a.kld.kld_epsilon, a.kld.kld_z) : 0;
    req = rr::wave_scan_max_u64(req);  // running maximum inside the wave (:350)
    if (lane == 63) s_req[wv] = req;
    __syncthreads();
    uint64_t pre = req_carry, chunk_req = req_carry;
    for (int q = 0; q < W; ++q) {
      if (q < wv) pre = s_req[q] > pre ? s_req[q] : pre;
      chunk_req = s_req[q] > chunk_req ? s_req[q] : chunk_req;
    }
    if (pre > req) req = pre;
    if (valid && rr_kld_stop(m, req, a.kld.min_particles)) atomicMin((unsigned long long*)&s_stop, (unsigned long long)m);
    __syncthreads();
    if (s_stop != ~0ull) break;  // uniform: read after the barrier
    k_carry += chunk_total;
    req_carry = chunk_req;
    base += span;
    if (in_lds && base == (uint64_t)kKldThreads && base < a.max_draws) {
      // the loop leaves the LDS table: the global blocks that follow must find the first 1 024 draws' bins
      const bool v2 = (uint64_t)tid < a.max_draws;
      if (v2) {
        keys[3 * tid] = s_keys[3 * tid];
        keys[3 * tid + 1] = s_keys[3 * tid + 1];
        keys[3 * tid + 2] = s_keys[3 * tid + 2];
      }
      __syncthreads();
      kld_insert_one((uint64_t)tid, v2, keys, table, minslot, myslot, a.hash_size);
      seen = (uint64_t)kKldThreads < a.max_draws ? (uint64_t)kKldThreads : a.max_draws;
    }
    __syncthreads();
  }
  const uint64_t n_new = s_stop == ~0ull ? a.max_draws : s_stop + 1;  // :342: at most max_particles
  __syncthreads();
  RR_ATL(4);
  // the table slots the draws of this step have used go back to empty for the next one (every occupied slot is some draw's)
  for (uint64_t m = tid; m < seen; m += kKldThreads) {
    const unsigned int sl = myslot[m];
    table[sl] = kKldEmpty;
    minslot[sl] = kKldEmpty;
  }
  const int dst = ctl->cur;
  for (uint64_t k = tid; k < n_new; k += kKldThreads) copy_particle(b, dst ^ 1, dst, idx[k], k, false, nullptr);
  if (tid == 0) {
    out[0] = n_new;
    ctl->n_active = n_new;
  }
  RR_ATL(5);
  if (mail || resident) {
  // ---- the mean try_step returns (monte_carlo_localization.rs:299-300 after :359-362: uniform weights over the new set), for
  // the synchronous caller: formed exactly as rr_pf_estimate forms it -- k_moments runs ceil(n / 256) workgroups of four waves
  // with one particle per thread (n <= 262 144), so this workgroup's chunk c stands for workgroups 4c .. 4c + 3, its wave w for
  // wave w % 4 of workgroup 4c + w / 4; the workgroup partials are added in k_moments' order and reduced as k_moments_final
  // does (lane j takes partial j: at most 64 of them for the 16 384 particles this kernel serves) -- and left in the host
  // mailbox / the resident answer.  flags != 0: weights that do not sum to a positive number: the host takes the long way.
  __shared__ double s_mom[W][5];
  __shared__ double s_part[64][5];
  __syncthreads();  // (the gathered set is complete)
  const int n_blocks = (int)((n_new + kBlock - 1) / kBlock);
  const double p0[4] = {b.x[dst][0], b.y[dst][0], b.yaw[dst][0], b.v[dst][0]};
  for (uint64_t base = 0; base < n_new; base += kKldThreads) {
    const uint64_t i = base + tid;
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (i < n_new) {
      const double wi = 1.0;
      const double d0 = b.x[dst][i] - p0[0], d1 = b.y[dst][i] - p0[1], d2 = b.yaw[dst][i] - p0[2], d3 = b.v[dst][i] - p0[3];
      acc[0] += wi;
      acc[1] += wi * d0;
      acc[2] += wi * d1;
      acc[3] += wi * d2;
      acc[4] += wi * d3;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const double sw = rr::wave_sum(acc[k]);
      if (lane == 0) s_mom[wv][k] = sw;
    }
    __syncthreads();
    if (tid < 5 * (kKldThreads / kBlock)) {  // k_moments: a workgroup's four wave sums, in wave order
      const int q = tid / 5, k = tid % 5, blk = (int)(base / kBlock) + q;
      if (blk < n_blocks) {
        double part = 0.0;
        for (int r = 0; r < kBlock / rr::kWave; ++r) part += s_mom[(kBlock / rr::kWave) * q + r][k];
        s_part[blk][k] = part;
      }
    }
    __syncthreads();
  }
  if (wv == 0) {
    double mom[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      double v = 0.0;  // k_moments_final: lane j adds the partials j, j + 64, ... (one here), then the shuffle tree
      if (lane < n_blocks) v += s_part[lane][k];
      mom[k] = rr::wave_sum(v);
    }
    const double Wt = mom[0];
    const bool ok = Wt > 0.0 && Wt < INFINITY;
    const uint64_t flags = (uint64_t)(ok ? 0 : 1) | ((uint64_t)(ctl->grid_timeout != 0) << 1);
    if (resident) {
      const uint64_t seq = a.res.first_seq + (uint64_t)s;
      if (lane < 4) {
        const double e = lane == 0 ? p0[0] + mom[1] / Wt : lane == 1 ? p0[1] + mom[2] / Wt : lane == 2 ? p0[2] + mom[3] / Wt : p0[3] + mom[4] / Wt;
        rr::store_pair_sys(&ring->rsp[lane], (uint64_t)__double_as_longlong(e), seq);
      }
      if (lane == 4) rr::store_pair_sys(&ring->rsp[rr::kResRspFlags], flags, seq);
#if defined(RR_PLAN_TIMELINE)
      if (lane == 0) {
        { uint64_t t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); atl[6] = t_; }
        atl[7] = n_new;
        for (int k = 0; k < 8; ++k) rr::store_pair_sys(&ring->rsp[8 + k], atl[k], seq);
      }
#endif
    } else if (lane == 0) {
      for (int q = 0; q < 4; ++q)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(&mail->est[q]), (uint64_t)__double_as_longlong(p0[q] + mom[1 + q] / Wt), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&mail->flags, flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&mail->seq, mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  }
  if (!resident) break;
  __syncthreads();  // (Ctl.n_active, Ctl.cur and the gathered set are read by the next step)
  }  // for (s)
  if (resident && tid == 0) {  // EXIT marker (resident_core.hpp)
    const uint64_t consumed = a.res.first_seq + (uint64_t)steps_done - 1 + (res_last_op == rr::kResOpQuit ? 1 : 0);
    rr::store_pair_sys(&ring->rsp[rr::kResRspExit], consumed, a.res.launch_id);
  }
}

// the first n_new = kld_out[0] draws become the particle set (set cur^1 -> set cur), and n_new becomes the filter's particle
// count ON THE DEVICE: every later kernel of an adaptive filter reads Ctl.n_active, so the host need not wait for the number
__global__ __launch_bounds__(kBlock) void k_kld_gather_dyn(Bufs b, Ctl* __restrict__ ctl, const unsigned int* __restrict__ idx,
                                                          const uint64_t* __restrict__ kld_out) {
  const uint64_t n_new = kld_out[0];
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const int dst = ctl->cur, src = dst ^ 1;
  if (k < n_new) copy_particle(b, src, dst, idx[k], k, false, nullptr);
  if (k == 0) ctl->n_active = n_new;  // (nothing in this launch reads it)
}

// the first n_new draws become the particle set (set cur^1 -> set cur)
__global__ __launch_bounds__(kBlock) void k_kld_gather(Bufs b, const Ctl* __restrict__ ctl,
                                                      const unsigned int* __restrict__ idx, uint64_t n_new) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n_new) return;
  const int dst = ctl->cur, src = dst ^ 1;
  copy_particle(b, src, dst, idx[k], k, false, nullptr);
}

// =============================================================================================
// host side
// =============================================================================================

struct rr_pf {
  rr_pf_config cfg;
  rr_pf_options opt;
  uint64_t n = 0, n_global = 0;
  uint64_t cap = 0;  // particles the buffers hold (== n unless the filter is KLD-adaptive: max_particles)
  // KLD-adaptive particle count (monte_carlo_localization.rs:322-385)
  bool adaptive = false;
  bool n_dirty = false;  // adaptive: the device (Ctl.n_active) knows a newer particle count than h->n (refresh_count)
  rr_mcl_adaptive kld{};
  int32_t* kld_keys = nullptr;         // [cap][3] bin of every candidate draw
  unsigned int* kld_table = nullptr;   // open-addressing table: draw that claimed the slot
  unsigned int* kld_minslot = nullptr; // smallest draw index with the slot's bin
  unsigned int* kld_myslot = nullptr;  // table slot of every draw
  uint64_t kld_hash_size = 0;
  uint64_t* kld_out = nullptr;         // {new particle count, occupied bins}
  uint64_t* kld_out_host = nullptr;
  hipStream_t stream = nullptr;      // the stream all work is enqueued on
  hipStream_t own_stream = nullptr;  // created with the handle
  bool owns_stream = false;
  bool using_external_stream = false;
  Bufs b{};
  double* slab = nullptr;  // [set][field][n]: x,y,yaw,v of both buffer sets
  double* w = nullptr;
  uint64_t* cdf = nullptr;
  uint64_t* cdf_coarse = nullptr;  // every 2^coarse_log2-th CDF entry (multinomial gather's LDS table)
  int coarse_log2 = 6;  // finest window whose table still fits the LDS (raised at create time for large N)
  uint64_t n_coarse = 0;
  // multinomial search through a guide table over the target space (k_resample_guide_mn; single shard, fused plan;
  // RR_MN_GUIDE=0: the coarse-table search instead); allocated on first use
  unsigned int* guide = nullptr;          // 2^guide_log2 + 2 entries
  unsigned int* guide_markers = nullptr;  // zero between steps
  unsigned int* guide_carry = nullptr;    // one per kResolveSlots buckets; [0] = 1 for good
  int guide_log2 = 0;                     // 0: switched off
  uint64_t* tile_total = nullptr;
  uint64_t* tile_q2 = nullptr;
  unsigned int* idx = nullptr;
  unsigned int* markers = nullptr;  // n_global + kResolveSlots, zero between steps
  unsigned int* carry = nullptr;    // one per kResolveSlots slots
  double* partials = nullptr;
  double* est_partials = nullptr;      // [kFusedMaxTiles][4] per-workgroup sums of the fused per-step estimate
  double* est_partials_host = nullptr; // pinned copy, made when the estimate is read
  double* est_slot_partials = nullptr; // [ceil(cap / kResolveSlots)][waves][4]: the deferred form's sums per slot tile (rr::kEstDeferred)
  double* est_slot_partials_host = nullptr;
  bool est_deferred = false;           // the last plan was asked for the deferred form and nobody has moved the particles yet
  bool est_eager = false;              // rr_pf_step of a multinomial filter: search the draws and add up the estimate right after the plan
  bool est_eager_done = false;         // ... and the launch that did it is in the stream (k_mn_search_est)
  // small particle sets (k_step_small): the step inputs of rr_pf_step_many and its per-step estimates on the device
  bool small_ok = true;  // RR_PF_SMALL=0 at create time: always take the large path
  HostMail* mail = nullptr;  // pinned, host-coherent: where the small kernel leaves the estimate of a synchronous step
  uint64_t mail_seq = 0;
  double* steps_dev = nullptr;
  size_t steps_cap = 0;
  double* est_ring = nullptr;
  size_t est_ring_cap = 0;
  std::vector<double> steps_host;
  // resident service (resident_core.hpp; rr_pf_set_resident): the small filter's step kernel stays on the device between steps
  struct Resident {
    bool enabled = false;
    bool live = false;     // an incarnation was launched and has not been seen to leave
    bool pending = false;  // a step was issued without waiting for its answer (rr_pf_step_async)
    rr::ResidentRing* ring = nullptr;  // pinned, host-coherent
    uint64_t seq = 0;        // last command issued
    uint64_t launch_id = 0;  // of the current / last incarnation
    double idle_us = 0.0, life_us = 100000.0;
    uint64_t launches = 0, steps = 0;
    unsigned int cmd_step = 0, cmd_rstep = 0;  // the step counters the command in flight was issued at (a relaunch starts there)
  } res;
  // k_quantize_plan_mark (K2 + fused plan in one launch): one record per tile, the launch epoch, the largest grid whose
  // workgroups are all resident at once (0: not available), RR_PF_FUSED_PLAN=0 turns it off
  double* packed[2] = {nullptr, nullptr};  // {x, y, yaw, v} mirrors of the two buffer sets (k_step_lazy<PACKED>; lazy multinomial only)
  uint64_t* grid_rec = nullptr;
  unsigned int* grid_ticket = nullptr;
  uint64_t grid_epoch = 0;
  uint64_t grid_capacity = 0;
  uint64_t plan_giveups = 0;  // launches of the one-launch plan that degraded to the serial plan (seen at the last read of Ctl)
  int dev_cus = 0;
  uint64_t shard_capacity = ~0ull;  // the same for k_shard_plan_mark (sharded step over the peer-to-peer transport); ~0: not asked yet
  unsigned int* est_ticket = nullptr;  // arrival counters of its last-workgroup reduction (rr::last_arrival; zero between launches)
  double* scratch_a = nullptr;  // n doubles: explicit noise v / uniforms / AoS staging (5n)
  double* scratch_b = nullptr;  // n doubles: explicit noise w
  double* obs_dev = nullptr;
  size_t obs_cap = 0;
  Ctl* ctl = nullptr;
  Ctl* ctl_host = nullptr;  // pinned
  uint64_t n_tiles = 0;
  unsigned int step = 0, rstep = 0;
  int k1_blocks_per_cu = 8;
  int mn_grid = 256;     // workgroups of the multinomial search kernel (RR_MN_GRID; set at create time from the table size)
  int mn_block = 1024;   // its workgroup size (RR_MN_BLOCK): one large workgroup per CU shares one big LDS table
  bool wmax_live = false;        // Ctl.wmax_bits holds the maximum of the current raw weights
  bool wmax_bits_clean = false;  // Ctl.wmax_bits is known to be zero
  uint64_t last_migrated = 0;
  // sharded multinomial resample: the plan rr_pf_shard_cdf made (select / pack_selected draw from the stream of THAT resample)
  bool shard_plan_valid = false;
  unsigned int shard_plan_rstep = 0;
  int shard_plan_shards = 0;   // as passed to rr_pf_shard_cdf
  int shard_select_shards = 0; // as passed to rr_pf_shard_select (0: not selected yet)
  unsigned int* mn_tile_cnt = nullptr;  // sharded multinomial: selected slots per (destination, tile), scanned in place
  uint64_t mn_tiles = 0;
  rr::P2PState p2p;  // device-initiated exchange over xGMI (rr_pf_p2p_*)
  bool maybe_pending = false;    // a lazy resample plan was launched and nothing has consumed its markers yet
  int pending_kind = kSrcMarkers;  // ... StepSrc: where its sources are (markers / lidx of the multinomial step / window of a shard)
  bool mn_deferred = false;        // a lazy multinomial resample is planned (CDF, guide table) but its draws have not been searched yet
  bool adaptive_small_ok = true;   // RR_MCL_SMALL=0: the adaptive step of a small filter takes the six separate launches
  bool mn_defer_ok = true;         // RR_MN_DEFER=0: always run the search as a launch of its own (k_resample_guide_mn)
  GatherArgs mn_deferred_args{};
  uint64_t slot_pad = 0;           // shard of the peer-to-peer transport: marker position of global slot s = s + slot_pad
  uint64_t window_seq = 0;         // ... and the exchange sequence number of the step whose window resample is pending (its DONE)
  double* rccl_inbox = nullptr;    // RCCL transport: [field][n] particles peers served for this shard's slots (plain device memory)
  bool window_rccl = false;        // the pending window resample came through the RCCL transport (inbox filled in stream order)
  unsigned int* lidx = nullptr;  // source index per slot; kInPlace = a peer stored the particle already (sharded)
  rr_pf_lik lik{};
  std::vector<double> landmarks;
  // profiling
  bool profiling = false;
  bool profile_dispatch_only = false;
  struct Ev { int id; hipEvent_t a, b; };
  std::vector<Ev> events;
  std::vector<hipEvent_t> event_pool;
  uint64_t prof_launches[RR_K_COUNT] = {};
  double prof_ms[RR_K_COUNT] = {};
};

namespace {

const char* kKernelNames[RR_K_COUNT] = {"k_propagate_weight", "k_quantize_reduce", "k_scan_tiles", "k_cdf",
                                        "k_resample_gather",  "k_commit",          "k_moments"};

struct Timed {
  rr_pf* h;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  Timed(rr_pf* h_, int id_) : h(h_), id(id_) {
    if (!h->profiling || h->profile_dispatch_only) return;
    auto take = [&]() {
      hipEvent_t e;
      if (!h->event_pool.empty()) {
        e = h->event_pool.back();
        h->event_pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    a = take();
    b = take();
    (void)hipEventRecord(a, h->stream);
  }
  ~Timed() {
    if (!h->profiling || h->profile_dispatch_only) return;
    (void)hipEventRecord(b, h->stream);
    h->events.push_back({id, a, b});
  }
};

inline unsigned grid_for(uint64_t n, int per) { return (unsigned)((n + per - 1) / per); }

rr_status validate_config(const rr_pf_config* c) {
  // messages are the reference's (particle_filter.rs:81-117)
  if (!c) return fail(RR_INVALID_PARAMETER, "null config");
  if (c->n_particles == 0) return fail(RR_INVALID_PARAMETER, "particle filter requires at least one particle");
  if (!std::isfinite(c->resample_threshold) || c->resample_threshold < 0.0 || c->resample_threshold > 1.0)
    return fail(RR_INVALID_PARAMETER, "particle filter resample_threshold must be within [0.0, 1.0]");
  if (!std::isfinite(c->range_noise) || c->range_noise <= 0.0)
    return fail(RR_INVALID_PARAMETER, "particle filter range_noise must be positive and finite");
  if (!std::isfinite(c->velocity_noise) || c->velocity_noise < 0.0)
    return fail(RR_INVALID_PARAMETER, "particle filter velocity_noise must be non-negative and finite");
  if (!std::isfinite(c->yaw_rate_noise) || c->yaw_rate_noise < 0.0)
    return fail(RR_INVALID_PARAMETER, "particle filter yaw_rate_noise must be non-negative and finite");
  if (!std::isfinite(c->dt) || c->dt <= 0.0)
    return fail(RR_INVALID_PARAMETER, "particle filter dt must be positive and finite");
  return RR_OK;
}

rr_status validate_control(const double u[2]) {  // particle_filter.rs:515-523
  if (!u || !std::isfinite(u[0]) || !std::isfinite(u[1]))
    return fail(RR_INVALID_PARAMETER, "particle filter control input must contain only finite values");
  return RR_OK;
}

rr_status validate_obs(const double* obs, size_t n_obs) {  // particle_filter.rs:538-549
  if (n_obs && !obs) return fail(RR_INVALID_PARAMETER, "null observations");
  for (size_t k = 0; k < n_obs; ++k) {
    const double d = obs[3 * k], x = obs[3 * k + 1], y = obs[3 * k + 2];
    if (!std::isfinite(d) || !std::isfinite(x) || !std::isfinite(y) || d < 0.0)
      return fail(RR_INVALID_PARAMETER, "particle filter observations must have finite, non-negative distances");
  }
  return RR_OK;
}

void set_particle_count(rr_pf* h, uint64_t n);

// keep_lazy: the caller enqueues work that reads the particle count from the device (the asynchronous step of an adaptive
// filter); everybody else gets the host's copy brought up to date first (one small copy + a wait)
rr_status resident_park(rr_pf* h);

// keep_resident: the caller talks to the handle's resident step kernel; everybody else finds the stream idle and the particle
// set in HBM (the kernel is asked to leave first)
rr_status bind(rr_pf* h, bool keep_lazy = false, bool keep_resident = false) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  RR_HIP_TRY(hipSetDevice(h->opt.device));
  if ((h->res.live || h->res.pending) && !keep_resident) {
    rr_status ps = resident_park(h);
    if (ps != RR_OK) return ps;
  }
  if (h->n_dirty && !keep_lazy) {
    RR_HIP_TRY(hipMemcpyAsync(h->kld_out_host, h->kld_out, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
    const uint64_t n_new = h->kld_out_host[0];
    if (n_new == 0 || n_new > h->cap) return fail(RR_RUNTIME_ERROR, "adaptive resample produced an impossible particle count");
    h->n_dirty = false;
    set_particle_count(h, n_new);
  }
  return RR_OK;
}

StepParams make_params(const rr_pf* h, const double u[2], int n_obs) {
  StepParams p{};
  p.n = h->n;
  p.n_global = h->n_global;
  p.first_gid = h->opt.first_global_index;
  p.seed = h->opt.seed;
  p.step = h->step;
  p.rstep = h->rstep;
  p.n_obs = n_obs;
  p.lik_mode = h->opt.likelihood_mode;
  p.u0 = u ? u[0] : 0.0;
  p.u1 = u ? u[1] : 0.0;
  p.dt = h->cfg.dt;
  p.sigma_v = h->cfg.velocity_noise;
  p.sigma_w = h->cfg.yaw_rate_noise;
  p.lik = h->lik;
  p.dyn_n = h->adaptive ? 1 : 0;
  return p;
}

// stage the observation block: inside the launch packet when it fits, else a device buffer
rr_status stage_obs(rr_pf* h, const double* obs, size_t n_obs, ObsArg* arg, bool* kernarg) {
  *kernarg = n_obs <= (size_t)kMaxObsKernarg;
  if (*kernarg) {
    if (n_obs) std::memcpy(arg->v, obs, 3 * n_obs * sizeof(double));
    return RR_OK;
  }
  if (n_obs > h->obs_cap) {
    if (h->obs_dev) RR_HIP_TRY(hipFree(h->obs_dev));
    h->obs_dev = nullptr;
    h->obs_cap = 0;
    RR_HIP_TRY(hipMalloc(&h->obs_dev, 3 * n_obs * sizeof(double)));
    h->obs_cap = n_obs;
  }
  // pageable source: HIP stages it before returning, so the caller's buffer may be reused
  RR_HIP_TRY(hipMemcpyAsync(h->obs_dev, obs, 3 * n_obs * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return RR_OK;
}

template <bool PREDICT, bool WEIGHT, bool EXPLICIT>
rr_status launch_pw(rr_pf* h, const StepParams& p, const ObsArg& arg, bool kernarg) {
  // grid-stride kernel: at most k1_blocks_per_cu workgroups per CU (256 CUs)
  const unsigned grid = std::min<unsigned>(grid_for(h->adaptive ? h->cap : h->n, kBlock), (unsigned)(256 * h->k1_blocks_per_cu));
  const size_t lds = WEIGHT ? 3 * (size_t)p.n_obs * sizeof(double) : 0;
  if (lds > 150 * 1024) return fail(RR_INVALID_PARAMETER, "too many observations for one LDS block (max 6400)");
  if (WEIGHT) {
    // Ctl.wmax_bits must be zero before the weights' maximum is accumulated; the plan kernel of
    // the previous resample pipeline leaves it zeroed, anything else needs the memset
    if (!h->wmax_bits_clean) RR_HIP_TRY(hipMemsetAsync(&h->ctl->wmax_bits, 0, sizeof(uint64_t), h->stream));
    h->wmax_bits_clean = false;
    h->wmax_live = true;
  }
  {
    Timed t(h, RR_K_PROPAGATE_WEIGHT);
    if (kernarg)
      hipLaunchKernelGGL((k_propagate_weight<PREDICT, WEIGHT, EXPLICIT, true>), dim3(grid), dim3(kBlock), lds,
                         h->stream, h->b, h->w, h->ctl, p, arg, (const double*)nullptr, h->scratch_a, h->scratch_b);
    else
      hipLaunchKernelGGL((k_propagate_weight<PREDICT, WEIGHT, EXPLICIT, false>), dim3(grid), dim3(kBlock), lds,
                         h->stream, h->b, h->w, h->ctl, p, arg, (const double*)h->obs_dev, h->scratch_a,
                         h->scratch_b);
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

ImageArgs image_args(const rr_pf* h) {
  ImageArgs a{};
  a.n = h->n;
  a.n_global = h->n_global;
  a.gid0 = h->opt.first_global_index;
  a.degenerate = rr::kDegenerateUniform;
  a.honour_uniform_flag = 1;
  a.dyn_n = h->adaptive ? 1 : 0;
  return a;
}

PlanArgs plan_args(const rr_pf* h, int mode, int scheme, double rho_override) {
  PlanArgs a{};
  a.n_global = h->n_global;
  a.neff_threshold = (double)h->n_global * h->cfg.resample_threshold;  // particle_filter.rs:339
  a.gate = h->opt.resample_gate;
  a.mode = mode;
  a.scheme = scheme;
  a.rho_override = rho_override;
  a.seed = h->opt.seed;
  a.rstep = h->rstep;
  a.set_uniform_on_fire = 1;
  a.lazy_gather = 0;
  return a;
}

// where the maximum to scale by lives: the atomic accumulator after a weight kernel, the
// plan kernel's saved copy once a resample pipeline has consumed (and zeroed) it
const double* wmax_source(const rr_pf* h) {
  return h->wmax_live ? (const double*)&h->ctl->wmax_bits : (const double*)&h->ctl->wmax;
}

void launch_quantize(rr_pf* h, const double* wmax_src, int settle = 0) {
  Timed t(h, RR_K_QUANTIZE_REDUCE);
  hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl,
                     wmax_src, image_args(h), h->tile_total, h->tile_q2, settle);
}

// make a pending lazy resample real (accessors and the non-fused entry points call this first)
void launch_guide_search(rr_pf* h, const double* r_explicit_dev, unsigned int* lidx, const GatherArgs& g);

// the deferred in-step estimate of a resample an accessor made real (the next step's k_step_lazy would have summed it on the way)
static void launch_est_slots(rr_pf* h) {
  if (!h->est_deferred) return;
  h->est_deferred = false;
  hipLaunchKernelGGL(k_est_slots, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->b, (const Ctl*)h->ctl, h->n,
                     h->est_slot_partials);
}

rr_status materialise(rr_pf* h) {
  if (!h->maybe_pending) return RR_OK;
  if (h->pending_kind == kSrcLidx) {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    if (h->mn_deferred) {  // the multinomial search has not run yet (the next step would have done it on the way)
      launch_guide_search(h, (const double*)nullptr, h->lidx, h->mn_deferred_args);
      h->mn_deferred = false;
    }
    hipLaunchKernelGGL(k_gather_lidx, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->ctl, h->lidx, h->n,
                       (const double*)nullptr);
    hipLaunchKernelGGL(k_settle, dim3(1), dim3(1), 0, h->stream, h->ctl);
    launch_est_slots(h);
    RR_HIP_TRY(hipGetLastError());
    h->maybe_pending = false;
    h->pending_kind = kSrcMarkers;
    return RR_OK;
  }
  if (h->pending_kind == kSrcWindow) {  // a shard: the peers' deliveries of the last step must have landed first
    Timed t(h, RR_K_RESAMPLE_GATHER);
    const bool via_p2p = !h->window_rccl;  // (RCCL transport: the inbox was filled in stream order, nothing to wait for)
    hipLaunchKernelGGL(k_resolve_gather_window, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->b, h->ctl,
                       h->markers, h->carry, h->n, h->opt.first_global_index, h->slot_pad,
                       (const double*)(via_p2p ? h->p2p.inbox : h->rccl_inbox), h->idx, via_p2p ? h->window_seq : (uint64_t)0,
                       via_p2p ? h->p2p.peers.timeout_ticks : (uint64_t)0, via_p2p ? h->p2p.err : (int*)nullptr);
    hipLaunchKernelGGL(k_settle, dim3(1), dim3(1), 0, h->stream, h->ctl);
    RR_HIP_TRY(hipGetLastError());
    h->maybe_pending = false;
    h->pending_kind = kSrcMarkers;
    return RR_OK;
  }
  {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    hipLaunchKernelGGL(k_resolve_gather, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->b,
                       h->ctl, h->markers, h->carry, h->idx, (double*)nullptr, h->n, 0, 1);
    hipLaunchKernelGGL(k_settle, dim3(1), dim3(1), 0, h->stream, h->ctl);
    launch_est_slots(h);
  }
  RR_HIP_TRY(hipGetLastError());
  h->maybe_pending = false;
  h->pending_kind = kSrcMarkers;
  return RR_OK;
}

// statistics only (accessors): integer sums into Ctl, no gate decision, nothing consumed
rr_status launch_sums(rr_pf* h, int mode, int scheme, double rho_override) {
  launch_quantize(h, wmax_source(h));
  {
    Timed t(h, RR_K_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 1, plan_args(h, mode, scheme, rho_override), (uint64_t*)nullptr);
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

// the guide table of the multinomial search, made when the first multinomial resample of a single-shard filter asks for it
rr_status ensure_guide(rr_pf* h) {
  if (h->guide_log2 != 0) return RR_OK;  // made already, or switched off
  if (const char* e = std::getenv("RR_MN_GUIDE")) {
    if (std::atoi(e) == 0) {
      h->guide_log2 = -1;
      return RR_OK;
    }
  }
  int lg = 10;
  while ((1ull << lg) < h->cap) ++lg;  // about one bucket per particle: between n/2 and n buckets are in use
  if (const char* e = std::getenv("RR_MN_GUIDE_LOG2")) lg = std::max(10, std::min(28, std::atoi(e)));
  const size_t nb = ((size_t)1 << lg) + rr::kResolveSlots + 2, nc = nb / rr::kResolveSlots + 2;
  const unsigned int one = 1;  // bucket 0 starts at source 0; no source ever writes carry[0]
  const auto make = [&]() -> hipError_t {
    hipError_t e;
    if ((e = hipMalloc(&h->guide, nb * sizeof(unsigned int))) != hipSuccess) return e;
    if ((e = hipMalloc(&h->guide_markers, nb * sizeof(unsigned int))) != hipSuccess) return e;
    if ((e = hipMalloc(&h->guide_carry, nc * sizeof(unsigned int))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(h->guide, 0, nb * sizeof(unsigned int), h->stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(h->guide_markers, 0, nb * sizeof(unsigned int), h->stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(h->guide_carry, 0, nc * sizeof(unsigned int), h->stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(h->guide_carry, &one, sizeof(one), hipMemcpyHostToDevice, h->stream)) != hipSuccess) return e;
    return hipStreamSynchronize(h->stream);  // `one` is a local
  };
  const hipError_t e = make();
  if (e != hipSuccess) {
    (void)hipFree(h->guide);
    (void)hipFree(h->guide_markers);
    (void)hipFree(h->guide_carry);
    h->guide = h->guide_markers = h->guide_carry = nullptr;
    return rr::fail(RR_RUNTIME_ERROR, std::string("guide table of the multinomial search: ") + hipGetErrorString(e));
  }
  h->guide_log2 = lg;
  return RR_OK;
}

// the multinomial draws -> source indices (and, unless lidx is given, the particles themselves)
void launch_guide_resolve(rr_pf* h) {
  hipLaunchKernelGGL(rr::k_guide_resolve, dim3((unsigned)((((size_t)1 << h->guide_log2) + rr::kResolveSlots) / rr::kResolveSlots)),
                     dim3(kBlock), 0, h->stream, h->ctl, h->guide_markers, h->guide_carry, h->guide, h->guide_log2);
}
void launch_guide_search(rr_pf* h, const double* r_explicit_dev, unsigned int* lidx, const GatherArgs& g) {
  hipLaunchKernelGGL(k_resample_guide_mn, dim3(grid_for(g.n_slots, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->ctl, h->cdf,
                     h->guide, h->guide_log2, r_explicit_dev, h->idx, lidx, g);
}
void launch_mn_search(rr_pf* h, bool guide, const double* r_explicit_dev, unsigned int* lidx, const GatherArgs& g) {
  if (guide) {
    launch_guide_resolve(h);
    launch_guide_search(h, r_explicit_dev, lidx, g);
    return;
  }
  hipLaunchKernelGGL(k_resample_gather_mn, dim3(std::min<unsigned>(grid_for(h->n, h->mn_block), (unsigned)h->mn_grid)), dim3(h->mn_block),
                     h->n_coarse * sizeof(uint64_t), h->stream, h->b, h->ctl, h->cdf, h->cdf_coarse, h->coarse_log2, h->n_coarse,
                     r_explicit_dev, h->idx, lidx, g);
}

// The resample pipeline: integer image -> plan (gate) + CDF -> gather.  Every kernel after the
// plan decides on the device whether it has anything to do.  mode 0 = gate, 1 = forced.
rr_status launch_resample(rr_pf* h, int mode, int scheme, double rho_override, const double* r_explicit_dev,
                          bool lazy = false, int settle = 0, int est_mode = rr::kEstOff) {
  // K2 + fused plan in one launch when every tile's workgroup is resident at once (k_quantize_plan_mark)
  const bool one_launch = scheme == RR_RESAMPLE_SYSTEMATIC && h->n_tiles <= h->grid_capacity && h->n == h->n_global &&
                          rr::spin_permit(h->opt.device, h);
  if (!one_launch) launch_quantize(h, wmax_source(h), settle);
  PlanArgs pa = plan_args(h, mode, scheme, rho_override);
  const bool lazy_mn = lazy && scheme == RR_RESAMPLE_MULTINOMIAL && h->lidx && !r_explicit_dev;
  lazy = (lazy && scheme == RR_RESAMPLE_SYSTEMATIC) || lazy_mn;
  pa.lazy_gather = lazy ? 1 : 0;
  const bool fused = h->n_tiles <= (uint64_t)rr::kFusedMaxTiles;
  const bool sys = scheme == RR_RESAMPLE_SYSTEMATIC;
  bool guide = false;
  if (!sys && fused && h->n == h->n_global) {
    if (rr_status st = ensure_guide(h)) return st;
    guide = h->guide_log2 > 0;
  }
  if (!fused) {
    Timed t(h, RR_K_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 1, pa, (uint64_t*)nullptr);
  }
  {
    Timed t(h, RR_K_CDF);
    const dim3 grid((unsigned)h->n_tiles), block(rr::kTileBlock);
    rr::EstArgs ea{};
    if (est_mode != rr::kEstOff && lazy && fused) {  // the mean the reference's try_step returns (rr::EstArgs: in the plan / deferred)
      if (est_mode == rr::kEstDeferred && !h->est_slot_partials)
        RR_HIP_TRY(hipMalloc(&h->est_slot_partials, (size_t)grid_for(h->cap, rr::kResolveSlots) * kEstSlotWords * sizeof(double)));
      for (int k = 0; k < 2; ++k) {
        ea.field[k][0] = h->b.x[k];
        ea.field[k][1] = h->b.y[k];
        ea.field[k][2] = h->b.yaw[k];
        ea.field[k][3] = h->b.v[k];
      }
      ea.partials = h->est_partials;
      ea.ticket = h->est_ticket;
      ea.want = sys ? est_mode : (int)rr::kEstDeferred;
      h->est_deferred = ea.want == rr::kEstDeferred;
    }
    if (sys && fused) {
      if (one_launch && ea.want == rr::kEstDeferred)
        hipLaunchKernelGGL((rr::k_quantize_plan_mark<false, true>), grid, block, 0, h->stream, (const double*)h->w, h->ctl, wmax_source(h),
                           image_args(h), h->grid_rec, h->grid_ticket, ++h->grid_epoch, settle, h->n_tiles, pa, h->markers,
                           h->carry, ea, rr::plan_giveup_ticks());
      else if (one_launch)
        hipLaunchKernelGGL(rr::k_quantize_plan_mark<false>, grid, block, 0, h->stream, (const double*)h->w, h->ctl, wmax_source(h),
                           image_args(h), h->grid_rec, h->grid_ticket, ++h->grid_epoch, settle, h->n_tiles, pa, h->markers,
                           h->carry, ea, rr::plan_giveup_ticks());
      else
        hipLaunchKernelGGL(rr::k_plan_mark, grid, block, 0, h->stream, h->w, h->ctl, image_args(h), h->tile_total,
                           h->tile_q2, h->n_tiles, pa, h->markers, h->carry, ea);
    }
    else if (sys)
      hipLaunchKernelGGL(rr::k_mark, grid, block, 0, h->stream, h->w, h->ctl, image_args(h), h->tile_total, h->markers,
                         h->carry);
    else if (fused)
      hipLaunchKernelGGL(rr::k_plan_cdf, grid, block, 0, h->stream, h->w, h->ctl, image_args(h), h->tile_total,
                         h->tile_q2, h->n_tiles, pa, h->cdf, guide ? (uint64_t*)nullptr : h->cdf_coarse, h->coarse_log2,
                         guide ? h->guide_markers : (unsigned int*)nullptr, h->guide_carry, h->guide_log2, ea);
    else
      hipLaunchKernelGGL(rr::k_cdf, grid, block, 0, h->stream, h->w, h->ctl, image_args(h), h->tile_total, h->cdf,
                         h->cdf_coarse, h->coarse_log2);
  }
  h->wmax_live = false;       // consumed: Ctl.wmax holds the value from now on
  h->wmax_bits_clean = true;  // the plan kernel zeroed the accumulator
  if (lazy_mn) {  // only the source indices; the next k_step_lazy<., true> (or materialise) reads through them
    Timed t(h, RR_K_RESAMPLE_GATHER);
    GatherArgs g{};
    g.n_src = h->n;
    g.n_slots = h->n;
    g.seed = h->opt.seed;
    g.rstep = h->rstep;
    g.scheme = scheme;
    if (guide && h->packed[0] && h->est_eager && h->est_deferred) {
      // the synchronous try_step: the caller waits for the mean of the resampled set, so the draws are searched now (the next
      // step reads through lidx) and the sources' fields added up on the way
      launch_guide_resolve(h);
      hipLaunchKernelGGL(k_mn_search_est, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, (const Ctl*)h->ctl,
                         (const uint64_t*)h->cdf, (const unsigned int*)h->guide, h->guide_log2, g, h->idx, h->lidx,
                         (const double*)h->packed[0], (const double*)h->packed[1], h->est_slot_partials);
      h->est_deferred = false;
      h->est_eager_done = true;
    } else if (guide && h->packed[0] && h->mn_defer_ok) {
      // the draws and their search wait for the kernel that consumes them: the next step's k_step_lazy<kSrcDraw> (or
      // ensure_searched, when an accessor comes first)
      launch_guide_resolve(h);
      h->mn_deferred = true;
      h->mn_deferred_args = g;
    } else {
      launch_mn_search(h, guide, (const double*)nullptr, h->lidx, g);
    }
    h->maybe_pending = true;
    h->pending_kind = kSrcLidx;
  } else if (lazy) {
    h->maybe_pending = true;  // the next k_step_lazy (or materialise) moves the particles
    h->pending_kind = kSrcMarkers;
  } else {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    if (sys) {
      hipLaunchKernelGGL(k_resolve_gather, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->b,
                         h->ctl, h->markers, h->carry, h->idx, (double*)nullptr, h->n, 0, 0);
    } else {
      GatherArgs g{};
      g.n_src = h->n;
      g.first_slot = 0;
      g.n_slots = h->n;
      g.seed = h->opt.seed;
      g.rstep = h->rstep;
      g.scheme = scheme;
      g.to_staging = 0;
      launch_mn_search(h, guide, r_explicit_dev, (unsigned int*)nullptr, g);
    }
  }
  RR_HIP_TRY(hipGetLastError());
  h->rstep += 1;
  return RR_OK;
}

void set_particle_count(rr_pf* h, uint64_t n) {
  h->n = h->n_global = n;
  h->cfg.n_particles = n;
  h->n_tiles = (n + kTile - 1) / kTile;
  h->n_coarse = (n + (1ull << h->coarse_log2) - 1) >> h->coarse_log2;
}

// the host changes an adaptive filter's particle count (rr_pf_set_particles_n): the device's copy follows
rr_status publish_particle_count(rr_pf* h) {
  if (!h->adaptive) return RR_OK;
  h->kld_out_host[1] = h->n;
  RR_HIP_TRY(hipMemcpyAsync(&h->ctl->n_active, &h->kld_out_host[1], sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}

// resample_adaptive, monte_carlo_localization.rs:322-365 (see the kernels above).  r_explicit_dev:
// max_particles uniforms on the device, or nullptr for the engine's Philox stream.
// lazy: do not wait for the new particle count -- every kernel of an adaptive filter reads it from Ctl.n_active, launches are
// sized for the capacity, and the host's copy is refreshed by the next entry point that needs it (bind)
rr_status resample_adaptive(rr_pf* h, const double* r_explicit_dev, bool lazy = false) {
  const uint64_t M = h->kld.max_particles;
  const uint64_t cap_tiles = (h->cap + kTile - 1) / kTile;
  const uint64_t cap_coarse = ((h->cap + (1ull << h->coarse_log2) - 1) >> h->coarse_log2) + 1;
  {
    Timed t(h, RR_K_QUANTIZE_REDUCE);
    hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)cap_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl, wmax_source(h),
                       image_args(h), h->tile_total, h->tile_q2, 0);
  }
  PlanArgs pa = plan_args(h, /*mode=*/1, RR_RESAMPLE_MULTINOMIAL, NAN);
  const bool fused = cap_tiles <= (uint64_t)rr::kFusedMaxTiles;
  if (!fused) return fail(RR_INVALID_PARAMETER, "adaptive filters are limited to 8 388 608 particles");
  {
    Timed t(h, RR_K_CDF);
    hipLaunchKernelGGL(rr::k_plan_cdf, dim3((unsigned)cap_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl, image_args(h), h->tile_total,
                       h->tile_q2, cap_tiles, pa, h->cdf, h->cdf_coarse, h->coarse_log2, (unsigned int*)nullptr, (unsigned int*)nullptr, 0);
  }
  h->wmax_live = false;
  h->wmax_bits_clean = true;
  {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    hipLaunchKernelGGL(k_kld_draw, dim3(grid_for(M, kBlock)), dim3(kBlock), cap_coarse * sizeof(uint64_t), h->stream, h->b,
                       h->ctl, h->cdf, h->cdf_coarse, h->coarse_log2, h->n_coarse, r_explicit_dev, h->idx, h->kld_keys, h->n, M,
                       h->opt.seed, h->rstep, 1);
    hipLaunchKernelGGL(k_kld_insert, dim3(grid_for(M, kBlock)), dim3(kBlock), 0, h->stream, (const int32_t*)h->kld_keys,
                       h->kld_table, h->kld_minslot, h->kld_myslot, M, h->kld_hash_size);
    const int gather_in_count = M <= 16384 ? 1 : 0;
    hipLaunchKernelGGL(k_kld_count, dim3(1), dim3(kKldThreads), 0, h->stream, h->kld_minslot, (const unsigned int*)h->kld_myslot, M, h->kld,
                       h->kld_out, h->kld_table, h->kld_hash_size, h->b, h->ctl, (const unsigned int*)h->idx, gather_in_count);
    if (!gather_in_count)
      hipLaunchKernelGGL(k_kld_gather_dyn, dim3(grid_for(M, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->ctl, (const unsigned int*)h->idx,
                         (const uint64_t*)h->kld_out);
    RR_HIP_TRY(hipGetLastError());
    if (M > kKldWipeInKernel) {  // the bin table of a large adaptive filter: wiped for the next resample by the copy engine / a wide fill
      RR_HIP_TRY(hipMemsetAsync(h->kld_table, 0xff, h->kld_hash_size * sizeof(unsigned int), h->stream));
      RR_HIP_TRY(hipMemsetAsync(h->kld_minslot, 0xff, h->kld_hash_size * sizeof(unsigned int), h->stream));
    }
    h->n_dirty = true;  // weights are uniform 1/n_new from here (Ctl.weights_uniform, :359-362)
  }
  h->rstep += 1;
  if (!lazy) return bind(h);  // the callers that hand the new count back, or go on with host-sized work
  return RR_OK;
}

rr_status fetch_ctl(rr_pf* h) {
  RR_HIP_TRY(hipMemcpyAsync(h->ctl_host, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  rr::spin_release(h->opt.device, h);
  if (h->ctl_host->grid_timeout) {
    // launches of the one-launch plan gave up waiting for workgroups the device did not run concurrently (another process
    // on the GPU) and planned serially instead -- same results, milliseconds instead of microseconds: this handle takes
    // the multi-launch plan from now on (resample_core.hpp, k_quantize_plan_mark)
    h->plan_giveups += (uint64_t)h->ctl_host->grid_timeout;
    h->grid_capacity = 0;
    RR_HIP_TRY(hipMemsetAsync(&h->ctl->grid_timeout, 0, sizeof(int), h->stream));
    h->ctl_host->grid_timeout = 0;
  }
  return h->p2p.check(h->stream);  // a latched peer-wait timeout must not look like a healthy filter
}

// The synchronous entry points wait for the stamp their step's last kernel leaves in the host mailbox: a bounded busy wait
// (a healthy step answers within tens of microseconds; kMailSpinNs of polling, the clock read every 256 polls), then the
// ordinary stream wait.  One place for all of them: same bound, same error, and the device's spinning-kernel slot is
// handed back (spin_release) because the stream is idle once the stamp is there.
constexpr long long kMailSpinNs = 500 * 1000;
rr_status await_mail(rr_pf* h, uint64_t want) {
  const volatile uint64_t* seq = &h->mail->seq;
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) {
      seen = true;
      break;
    }
    if ((spins & 255u) == 255u &&
        std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > kMailSpinNs)
      break;
  }
  if (!seen) {  // slow device / contended queue: wait the ordinary way
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
    if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != want) return fail(RR_RUNTIME_ERROR, "the step's estimate never reached the host mailbox");
  }
  rr::spin_release(h->opt.device, h);  // the stream is idle
  return RR_OK;
}

rr_status compute_moments(rr_pf* h, double est[4], double cov[16]) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int blocks = (int)std::min<uint64_t>(kMomentBlocks, grid_for(h->n, kBlock));
    // the mean alone (rr_pf_estimate, the try_step of filters without an in-step estimate) comes back through the host mailbox:
    // no copy of Ctl, no stream synchronisation
    const bool by_mail = est && !cov && attempt == 0 && !h->p2p.ready && !h->profiling;
    uint64_t want = 0;
    if (by_mail) {
      if (!h->mail) {
        RR_HIP_TRY(hipHostMalloc(&h->mail, sizeof(HostMail), hipHostMallocDefault));
        std::memset(h->mail, 0, sizeof(HostMail));
      }
      want = ++h->mail_seq;
    }
    {
      Timed t(h, RR_K_MOMENTS);
      hipLaunchKernelGGL(k_moments, dim3(blocks), dim3(kBlock), 0, h->stream, h->b, h->w, h->ctl, h->n, attempt,
                         h->partials);
      hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(kNumMoments * 64), 0, h->stream, h->b, h->ctl, h->partials, blocks,
                         by_mail ? h->mail : (HostMail*)nullptr, want);
    }
    RR_HIP_TRY(hipGetLastError());
    if (by_mail) {
      rr_status ms = await_mail(h, want);
      if (ms != RR_OK) return ms;
      if (h->mail->flags == 0) {
        for (int k = 0; k < 4; ++k) est[k] = h->mail->est[k];
        return RR_OK;
      }
      // degenerate weights or a degraded plan to take note of: the long way (Ctl read back; the moments are in it)
    }
    rr_status s = fetch_ctl(h);
    if (s != RR_OK) return s;
    const double W = h->ctl_host->moments[0];
    if (W > 0.0 && std::isfinite(W)) break;
    // sum of weights <= 0: the reference's normalize_weights falls back to uniform weights
    // (particle_filter.rs:433-438); redo with w_i = 1
  }
  const double* m = h->ctl_host->moments;
  const double* p0 = h->ctl_host->shift_point;
  const double W = m[0];
  double d[4] = {m[1] / W, m[2] / W, m[3] / W, m[4] / W};
  if (est)
    for (int k = 0; k < 4; ++k) est[k] = p0[k] + d[k];
  if (cov) {
    const int idx[4][4] = {{5, 6, 7, 8}, {6, 9, 10, 11}, {7, 10, 12, 13}, {8, 11, 13, 14}};
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) cov[4 * r + c] = m[idx[r][c]] / W - d[r] * d[c];
  }
  return RR_OK;
}

hipEvent_t take_event(rr_pf* h) {
  hipEvent_t e;
  if (!h->event_pool.empty()) {
    e = h->event_pool.back();
    h->event_pool.pop_back();
  } else {
    (void)hipEventCreate(&e);
  }
  return e;
}

void drain_events(rr_pf* h) {
  for (auto& e : h->events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      h->prof_ms[e.id] += ms;
      h->prof_launches[e.id] += 1;
    }
    h->event_pool.push_back(e.a);
    h->event_pool.push_back(e.b);
  }
  h->events.clear();
}

rr_status validate_kld(const rr_mcl_adaptive* k) {
  // messages are the reference's (monte_carlo_localization.rs:84-104)
  if (!k) return fail(RR_INVALID_PARAMETER, "null KLD parameters");
  if (k->min_particles == 0) return fail(RR_INVALID_PARAMETER, "MCL min_particles must be greater than zero");
  if (k->max_particles < k->min_particles)
    return fail(RR_INVALID_PARAMETER, "MCL max_particles must be greater than or equal to min_particles");
  if (!std::isfinite(k->kld_epsilon) || k->kld_epsilon <= 0.0)
    return fail(RR_INVALID_PARAMETER, "MCL kld_epsilon must be positive and finite");
  if (!std::isfinite(k->kld_z) || k->kld_z <= 0.0) return fail(RR_INVALID_PARAMETER, "MCL kld_z must be positive and finite");
  return RR_OK;
}

rr_status create_common(const rr_pf_config* cfg_in, const rr_pf_options* opt_in, const double* state, rr_pf** out,
                        const rr_mcl_adaptive* kld = nullptr) {
  if (!out) return fail(RR_INVALID_PARAMETER, "null output handle");
  *out = nullptr;
  if (!cfg_in) return fail(RR_INVALID_PARAMETER, "null config");
  rr_pf_config cfg_v = *cfg_in;
  rr_status s;
  if (kld) {
    if ((s = validate_kld(kld)) != RR_OK) return s;
    cfg_v.n_particles = kld->min_particles;  // try_new :147: the filter starts with min_particles
    if (kld->max_particles >= (1ull << 31)) return fail(RR_INVALID_PARAMETER, "max_particles must be below 2^31");
  }
  const rr_pf_config* cfg = &cfg_v;
  s = validate_config(cfg);
  if (s != RR_OK) return s;
  rr_pf_options opt;
  if (opt_in) opt = *opt_in; else rr_pf_options_default(&opt);
  if (kld) {
    if (opt.resample_scheme != RR_RESAMPLE_MULTINOMIAL || opt.resample_gate != RR_GATE_ALWAYS)
      return fail(RR_INVALID_PARAMETER, "the KLD-adaptive filter resamples multinomially at every step (rr_pf_options_mcl)");
    if (opt.n_global != 0 || opt.first_global_index != 0)
      return fail(RR_INVALID_PARAMETER, "the KLD-adaptive filter cannot be sharded");
  }
  if (state)
    for (int k = 0; k < 4; ++k)
      if (!std::isfinite(state[k]))  // particle_filter.rs:505-513
        return fail(RR_INVALID_PARAMETER, "particle filter state must contain only finite values");
  if (cfg->n_particles >= (1ull << 31)) return fail(RR_INVALID_PARAMETER, "n_particles must be below 2^31 per shard");
  const uint64_t n_global = opt.n_global ? opt.n_global : cfg->n_particles;
  if (n_global >= (1ull << 31)) return fail(RR_INVALID_PARAMETER, "n_global must be below 2^31");
  if (opt.first_global_index + cfg->n_particles > n_global)
    return fail(RR_INVALID_PARAMETER, "shard range exceeds n_global");
  if (opt.resample_scheme != RR_RESAMPLE_MULTINOMIAL && opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC)
    return fail(RR_INVALID_PARAMETER, "unknown resample_scheme");
  if (opt.resample_gate != RR_GATE_NEFF && opt.resample_gate != RR_GATE_ALWAYS)
    return fail(RR_INVALID_PARAMETER, "unknown resample_gate");
  if (opt.likelihood_mode != RR_LIK_FUSED && opt.likelihood_mode != RR_LIK_PRODUCT)
    return fail(RR_INVALID_PARAMETER, "unknown likelihood_mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(RR_RUNTIME_ERROR, "no HIP device available: the engine has no CPU fallback");
  if (opt.device < 0 || opt.device >= ndev) return fail(RR_INVALID_PARAMETER, "device ordinal out of range");
  RR_HIP_TRY(hipSetDevice(opt.device));

  rr_pf* h = new rr_pf();
  h->cfg = *cfg;
  h->opt = opt;
  h->n = cfg->n_particles;
  h->n_global = n_global;
  h->n_tiles = (h->n + kTile - 1) / kTile;
  h->cap = kld ? kld->max_particles : h->n;
  if (kld) {
    h->adaptive = true;
    h->kld = *kld;
    h->opt.record_indices = 1;
    opt.record_indices = 1;
  }
  const uint64_t cap_tiles = (h->cap + kTile - 1) / kTile;
  h->lik = rr_pf_lik_make(cfg->range_noise);
  if (const char* e = std::getenv("RR_MN_GRID")) {
    const int v = std::atoi(e);
    if (v >= 1) h->mn_grid = v;
  }
  if (const char* e = std::getenv("RR_PF_SMALL")) h->small_ok = std::atoi(e) != 0;
  if (const char* e = std::getenv("RR_PF_RESIDENT_US")) {  // resident service from the start (rr_pf_set_resident)
    const double us = std::atof(e);
    if (us > 0.0 && us <= 1e7) {
      h->res.enabled = true;
      h->res.idle_us = us;
      h->res.life_us = std::max(100000.0, 20.0 * us);
    }
  }
  if (const char* e = std::getenv("RR_MN_DEFER")) h->mn_defer_ok = std::atoi(e) != 0;
  if (const char* e = std::getenv("RR_MCL_SMALL")) h->adaptive_small_ok = std::atoi(e) != 0;
  if (const char* e = std::getenv("RR_K1_BLOCKS_PER_CU")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= 64) h->k1_blocks_per_cu = v;
  }
  auto cleanup = [&](rr_status st) {
    rr_pf_destroy(h);
    return st;
  };
#define RR_TRY_OR_CLEAN(expr)                                                                      \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return cleanup(fail(RR_RUNTIME_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  RR_TRY_OR_CLEAN(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  h->owns_stream = true;
  const size_t nb = h->cap * sizeof(double);
  // one slab [set][field][cap] so that peers can map the whole particle state with one IPC handle
  RR_TRY_OR_CLEAN(hipMalloc(&h->slab, 8 * nb));
  for (int k = 0; k < 2; ++k) {
    h->b.x[k] = h->slab + (size_t)(4 * k + 0) * h->cap;
    h->b.y[k] = h->slab + (size_t)(4 * k + 1) * h->cap;
    h->b.yaw[k] = h->slab + (size_t)(4 * k + 2) * h->cap;
    h->b.v[k] = h->slab + (size_t)(4 * k + 3) * h->cap;
  }
  RR_TRY_OR_CLEAN(hipMalloc(&h->w, nb));
  RR_TRY_OR_CLEAN(hipMalloc(&h->cdf, h->cap * sizeof(uint64_t)));
  if (const char* e = std::getenv("RR_MN_COARSE_LOG2")) h->coarse_log2 = std::max(4, std::min(16, std::atoi(e)));
  if (const char* e = std::getenv("RR_MN_BLOCK")) h->mn_block = std::max(64, std::min(1024, std::atoi(e) / 64 * 64));
  while (((h->cap + (1ull << h->coarse_log2) - 1) >> h->coarse_log2) > 18432) h->coarse_log2 += 1;  // <= 144 KB of LDS
  h->n_coarse = (h->n + (1ull << h->coarse_log2) - 1) >> h->coarse_log2;
  {
    const size_t lds = (((h->cap + (1ull << h->coarse_log2) - 1) >> h->coarse_log2) + 1) * sizeof(uint64_t);
    if (lds > 48 * 1024) {  // more dynamic LDS than the default launch limit
      RR_TRY_OR_CLEAN(hipFuncSetAttribute((const void*)k_resample_gather_mn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      RR_TRY_OR_CLEAN(hipFuncSetAttribute((const void*)k_kld_draw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // measured at 1e6 draws (gpurun_out/r02h): 64-entry windows + one 1024-thread workgroup per CU 43.9 us, 256-entry windows +
    // 1024 workgroups of 256 threads 51.1 us -- the search is bound by its random 8-byte requests, a finer table saves two of them
    if (!std::getenv("RR_MN_GRID")) h->mn_grid = 256 * (int)std::max<size_t>(1, std::min<size_t>(4, (144 * 1024) / std::max<size_t>(lds, 1)));
  }
  RR_TRY_OR_CLEAN(hipMalloc(&h->cdf_coarse, (((h->cap + (1ull << h->coarse_log2) - 1) >> h->coarse_log2) + 1) * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->tile_total, cap_tiles * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->tile_q2, 2 * cap_tiles * sizeof(uint64_t)));
  if (opt.record_indices) RR_TRY_OR_CLEAN(hipMalloc(&h->idx, h->cap * sizeof(unsigned int)));
  if (kld) {
    h->kld_hash_size = 2;
    while (h->kld_hash_size < 2 * h->cap) h->kld_hash_size *= 2;  // at most half full
    RR_TRY_OR_CLEAN(hipMalloc(&h->kld_keys, 3 * h->cap * sizeof(int32_t)));
    RR_TRY_OR_CLEAN(hipMalloc(&h->kld_table, h->kld_hash_size * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMalloc(&h->kld_minslot, h->kld_hash_size * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->kld_table, 0xff, h->kld_hash_size * sizeof(unsigned int), h->stream));  // (k_kld_count keeps them clean)
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->kld_minslot, 0xff, h->kld_hash_size * sizeof(unsigned int), h->stream));
    RR_TRY_OR_CLEAN(hipMalloc(&h->kld_myslot, h->cap * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMalloc(&h->kld_out, 2 * sizeof(uint64_t)));
    RR_TRY_OR_CLEAN(hipHostMalloc(&h->kld_out_host, 2 * sizeof(uint64_t)));
  }
  const bool guide_at_create = opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL && !kld && h->n == h->n_global && cap_tiles <= (uint64_t)rr::kFusedMaxTiles &&
                               h->n > kSmallMaxParticles;
  if (opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL && !kld) {  // the fused step resamples lazily through lidx
    RR_TRY_OR_CLEAN(hipMalloc(&h->lidx, h->n * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMemset(h->lidx, 0xff, h->n * sizeof(unsigned int)));
    if (h->n == h->n_global && !std::getenv("RR_MN_NO_PACKED")) {
      for (int k = 0; k < 2; ++k) RR_TRY_OR_CLEAN(hipMalloc(&h->packed[k], 4 * h->n * sizeof(double)));
    }
  }
  {
    // (+ one tile of padding in front of a shard's own block, peer-to-peer transport: rr::resolve_tile_window)
    const size_t nm = (size_t)std::max<uint64_t>(n_global, h->cap) + 3 * (size_t)rr::kResolveSlots;
    RR_TRY_OR_CLEAN(hipMalloc(&h->markers, nm * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMemset(h->markers, 0, nm * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMalloc(&h->carry, (nm / rr::kResolveSlots + 2) * sizeof(unsigned int)));
  }
  RR_TRY_OR_CLEAN(hipMalloc(&h->partials, (size_t)kMomentBlocks * kNumMoments * sizeof(double)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->est_partials, (size_t)rr::kFusedMaxTiles * 4 * sizeof(double)));
  {
    int per_cu = 0, dev_cus = 0;
    RR_TRY_OR_CLEAN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rr::k_quantize_plan_mark<false>, rr::kTileBlock, 0));
    RR_TRY_OR_CLEAN(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    h->grid_capacity = std::min<uint64_t>((uint64_t)per_cu * (uint64_t)dev_cus, (uint64_t)rr::kTileBlock);
    if (const char* e = std::getenv("RR_PF_FUSED_PLAN")) {
      if (std::atoi(e) == 0) h->grid_capacity = 0;
    }
    const size_t rec_bytes = (size_t)(rr::kTileBlock + 1) * rr::kRecWords * sizeof(uint64_t) + 16 * sizeof(uint64_t) +
                             (size_t)rr::kTileBlock * rr::kTimelineWords * sizeof(uint64_t);  // records, heads, (instrumented build) stamps
    RR_TRY_OR_CLEAN(hipMalloc(&h->grid_rec, rec_bytes));
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->grid_rec, 0, rec_bytes, h->stream));
    RR_TRY_OR_CLEAN(hipMalloc(&h->grid_ticket, rr::kTicketWords * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->grid_ticket, 0, rr::kTicketWords * sizeof(unsigned int), h->stream));
  }
  RR_TRY_OR_CLEAN(hipMalloc(&h->est_ticket, rr::kTicketWords * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->est_ticket, 0, rr::kTicketWords * sizeof(unsigned int), h->stream));
  h->slot_pad = (rr::kResolveSlots - h->opt.first_global_index % rr::kResolveSlots) % rr::kResolveSlots;
  RR_TRY_OR_CLEAN(hipMalloc(&h->ctl, sizeof(Ctl)));
  RR_TRY_OR_CLEAN(hipHostMalloc(&h->ctl_host, sizeof(Ctl)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->ctl, 0, sizeof(Ctl), h->stream));
  hipLaunchKernelGGL(k_init, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->w, h->n, h->n_global,
                     h->opt.first_global_index, h->opt.seed, state ? 1 : 0, state ? state[0] : 0.0,
                     state ? state[1] : 0.0, state ? state[2] : 0.0, state ? state[3] : 0.0);
  RR_TRY_OR_CLEAN(hipGetLastError());
  // Particle::new gives every particle w = 1/N (particle_filter.rs:35-43)
  Ctl init{};
  init.weights_uniform = 1;
  init.usable = 0;
  init.image_mode = rr::kImageUniform;
  init.sum = 1.0;
  init.neff = (double)n_global;
  init.n_active = h->n;
  *h->ctl_host = init;
  RR_TRY_OR_CLEAN(hipMemcpyAsync(h->ctl, h->ctl_host, sizeof(Ctl), hipMemcpyHostToDevice, h->stream));
  RR_TRY_OR_CLEAN(hipStreamSynchronize(h->stream));
#undef RR_TRY_OR_CLEAN
  // the guide table of the multinomial search: made here, not inside the first rr_pf_step_async (which promises no host wait)
  if (guide_at_create) {
    const rr_status gs = ensure_guide(h);
    if (gs != RR_OK) return cleanup(gs);
  }
  *out = h;
  return RR_OK;
}

rr_status ensure_scratch(rr_pf* h, size_t doubles_a, size_t doubles_b) {
  // scratch_a doubles as the AoS staging area (5n), scratch_b only ever needs n
  static_assert(sizeof(double) == 8, "");
  if (doubles_a) {
    size_t want = std::max<size_t>(doubles_a, 5 * h->cap);
    if (!h->scratch_a) RR_HIP_TRY(hipMalloc(&h->scratch_a, want * sizeof(double)));
  }
  if (doubles_b && !h->scratch_b) RR_HIP_TRY(hipMalloc(&h->scratch_b, h->cap * sizeof(double)));
  return RR_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
// one launch of k_step_lazy: the template arguments from run-time facts (ea/eb: dispatch timestamps when profiling)
template <bool KA, int SRC, int LIK, bool PK = false, bool EST = false>
static void launch_k1_as(rr_pf* h, unsigned grid, size_t lds, hipEvent_t ea, hipEvent_t eb, const StepParams& p, const ObsArg& arg,
                         unsigned int* markers, const unsigned int* carry, unsigned int* idx_out, const WindowArgs& wa) {
  const double* obs_dev = KA ? nullptr : h->obs_dev;
  if (ea)
    hipExtLaunchKernelGGL((k_step_lazy<KA, SRC, LIK, PK, EST>), dim3(grid), dim3(kBlock), lds, h->stream, ea, eb, 0, h->b, h->w, h->ctl, p, arg,
                          obs_dev, markers, carry, idx_out, wa, h->packed[0], h->packed[1]);
  else
    hipLaunchKernelGGL((k_step_lazy<KA, SRC, LIK, PK, EST>), dim3(grid), dim3(kBlock), lds, h->stream, h->b, h->w, h->ctl, p, arg, obs_dev,
                       markers, carry, idx_out, wa, h->packed[0], h->packed[1]);
}

static void launch_k1(rr_pf* h, bool kernarg, int src, unsigned grid, size_t lds, hipEvent_t ea, hipEvent_t eb,
                      const StepParams& p, const ObsArg& arg, unsigned int* markers, const unsigned int* carry,
                      unsigned int* idx_out, const WindowArgs& wa = WindowArgs{}, bool packed = false) {
  const bool product = p.lik_mode == RR_LIK_PRODUCT;
  const bool est = wa.est_partials != nullptr;  // the builds that add up the deferred estimate (never the window kernels)
#define RR_K1_GO(KA_, SRC_, LIK_)                                                                                             \
  ((est && SRC_ != kSrcWindow) ? launch_k1_as<KA_, SRC_, LIK_, false, (SRC_ != kSrcWindow)>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa) \
                               : launch_k1_as<KA_, SRC_, LIK_>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa))
#define RR_K1_GO_PK(KA_, LIK_)                                                                                         \
  (est ? launch_k1_as<KA_, kSrcLidx, LIK_, true, true>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa)         \
       : launch_k1_as<KA_, kSrcLidx, LIK_, true>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa))
#define RR_K1_GO_DRAW(KA_, LIK_)                                                                                       \
  (est ? launch_k1_as<KA_, kSrcDraw, LIK_, true, true>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa)         \
       : launch_k1_as<KA_, kSrcDraw, LIK_, true>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa))
#define RR_K1_SRC(SRC_)                                                                                       \
  do {                                                                                                        \
    if (kernarg) product ? RR_K1_GO(true, SRC_, RR_LIK_PRODUCT) : RR_K1_GO(true, SRC_, RR_LIK_FUSED);          \
    else product ? RR_K1_GO(false, SRC_, RR_LIK_PRODUCT) : RR_K1_GO(false, SRC_, RR_LIK_FUSED);                \
  } while (0)
  if (src == kSrcDraw) {  // (always with the packed mirror)
    if (kernarg) product ? RR_K1_GO_DRAW(true, RR_LIK_PRODUCT) : RR_K1_GO_DRAW(true, RR_LIK_FUSED);
    else product ? RR_K1_GO_DRAW(false, RR_LIK_PRODUCT) : RR_K1_GO_DRAW(false, RR_LIK_FUSED);
  } else if (src == kSrcLidx && packed) {
    if (kernarg) product ? RR_K1_GO_PK(true, RR_LIK_PRODUCT) : RR_K1_GO_PK(true, RR_LIK_FUSED);
    else product ? RR_K1_GO_PK(false, RR_LIK_PRODUCT) : RR_K1_GO_PK(false, RR_LIK_FUSED);
  } else if (src == kSrcLidx) RR_K1_SRC(kSrcLidx);
  else if (src == kSrcWindow) RR_K1_SRC(kSrcWindow);
  else RR_K1_SRC(kSrcMarkers);
#undef RR_K1_SRC
#undef RR_K1_GO
#undef RR_K1_GO_PK
#undef RR_K1_GO_DRAW
}

// ---- small particle sets: one launch of one workgroup per step, or per K steps (k_step_small)
static size_t small_lds_bytes(uint64_t n, size_t n_obs) { return (3 * n_obs + 5 * (size_t)n + 1) * sizeof(double); }
static bool small_path(const rr_pf* h, size_t n_obs) {
  return h->small_ok && !h->adaptive && h->n == h->n_global && h->n <= kSmallMaxParticles && !h->p2p.ready && !h->using_external_stream &&
         small_lds_bytes(h->n, n_obs) <= 150 * 1024;
}

template <int BLOCK, int R, int LIK>
static rr_status launch_small_as(rr_pf* h, const SmallArgs& a, const ObsArg& arg, size_t lds, double* est_out) {
  static bool raised = false;  // more dynamic LDS than the default launch limit: once per instantiation
  if (lds > 48 * 1024 && !raised) {
    RR_HIP_TRY(hipFuncSetAttribute((const void*)k_step_small<BLOCK, R, LIK>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    raised = true;
  }
  hipLaunchKernelGGL((k_step_small<BLOCK, R, LIK>), dim3(1), dim3(BLOCK), lds, h->stream, h->b, h->w, h->ctl, a, arg,
                     (const double*)h->steps_dev, h->idx, est_out, h->est_partials, h->mail, h->res.ring);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

static void small_args_common(const rr_pf* h, SmallArgs* a, size_t n_obs, size_t K, bool want_est) {
  a->n = h->n;
  a->seed = h->opt.seed;
  a->step0 = h->step;
  a->rstep0 = h->rstep;
  a->n_obs = (int)n_obs;
  a->K = (int)K;
  a->gate = h->opt.resample_gate;
  a->scheme = h->opt.resample_scheme;
  a->neff_threshold = (double)h->n_global * h->cfg.resample_threshold;
  a->dt = h->cfg.dt;
  a->sigma_v = h->cfg.velocity_noise;
  a->sigma_w = h->cfg.yaw_rate_noise;
  a->lik = h->lik;
  a->want_est = want_est ? 1 : 0;
}

static rr_status launch_small(rr_pf* h, const SmallArgs& a, const ObsArg& arg, size_t lds, double* est_out) {
  const bool product = h->opt.likelihood_mode == RR_LIK_PRODUCT;
  // shape of the workgroup: 512 threads x 1 / 2 / 4 consecutive particles.  Measured at 1000 x 4 (step_many, us per step):
  // 512 x 2: 7.2, 1024 x 1: 8.7 -- sixteen waves pay more at the step's dozen barriers than their extra latency hiding
  // brings.  RR_PF_SMALL_BLOCK=1024 selects 1024 threads x 1 / 2 (A/B).
  static const int forced = [] { const char* e = std::getenv("RR_PF_SMALL_BLOCK"); return e ? std::atoi(e) : 0; }();
#define RR_SMALL_GO(B_, R_) (product ? launch_small_as<B_, R_, RR_LIK_PRODUCT>(h, a, arg, lds, est_out) : launch_small_as<B_, R_, RR_LIK_FUSED>(h, a, arg, lds, est_out))
  // the reference's own sizes (100 - 150 particles): two or four waves pay less at the step's barriers and cross-wave sums
  // than eight mostly idle ones (RR_PF_SMALL_BLOCK=512: always 512 threads, for A/B; resident try_step at 100 x 3: 7.1 us against
  // 8.4, and ONE wave with two particles per lane -- no cross-wave hand-over at all -- 8.2: the doubled dependent chain costs more)
  if (h->n <= 128 && forced != 512) return RR_SMALL_GO(128, 1);
  if (h->n <= 256 && forced != 512) return RR_SMALL_GO(256, 1);
  if (h->n <= 512) return RR_SMALL_GO(512, 1);
  if (forced == 1024) return h->n <= 1024 ? RR_SMALL_GO(1024, 1) : RR_SMALL_GO(1024, 2);
  return h->n <= 1024 ? RR_SMALL_GO(512, 2) : RR_SMALL_GO(512, 4);
#undef RR_SMALL_GO
}

// ---- resident service of a small filter (resident_core.hpp): the host side
constexpr size_t kResMaxObs = 128;
constexpr int kResPayloadCap = 2 + 3 * (int)kResMaxObs;
static size_t resident_lds_bytes(uint64_t n) { return ((size_t)kResPayloadCap + 5 * (size_t)n + 1) * sizeof(double); }
static bool small_path(const rr_pf* h, size_t n_obs);
static bool adaptive_one_launch(const rr_pf* h, size_t n_obs) {  // the filters k_mcl_adaptive_small serves in one launch
  return h->adaptive && h->adaptive_small_ok && h->kld.max_particles <= 16384 && n_obs <= (size_t)kMaxObsKernarg && !h->p2p.ready &&
         !h->using_external_stream;
}
static bool resident_path(const rr_pf* h, size_t n_obs) {
  if (!h->res.enabled || h->profiling || n_obs > kResMaxObs) return false;
  if (h->adaptive) return adaptive_one_launch(h, n_obs);
  return small_path(h, n_obs) && resident_lds_bytes(h->n) <= 150 * 1024;
}

// launch an incarnation that waits for command `first_seq` (the particle set is in HBM: nothing of this handle is in flight)
static rr_status resident_launch(rr_pf* h, uint64_t first_seq, unsigned int step0, unsigned int rstep0) {
  rr_status s = h->adaptive ? RR_OK : materialise(h);  // (an adaptive filter never has a lazy resample pending)
  if (s != RR_OK) return s;
  if (!h->res.ring) {
    RR_HIP_TRY(hipHostMalloc(&h->res.ring, sizeof(rr::ResidentRing), hipHostMallocDefault));
    std::memset(h->res.ring, 0, sizeof(rr::ResidentRing));
  }
  rr::ResidentArgs ra{};
  ra.on = 1;
  ra.payload_cap = kResPayloadCap;
  ra.first_seq = first_seq;
  ra.idle_ticks = (uint64_t)(h->res.idle_us * 100.0);
  ra.life_ticks = (uint64_t)(h->res.life_us * 100.0);
  ra.launch_id = ++h->res.launch_id;
  ObsArg arg;
  if (h->adaptive) {  // k_mcl_adaptive_small, resident: the particle count stays on the device (Ctl.n_active)
    if ((s = ensure_scratch(h, 5 * h->cap, 0)) != RR_OK) return s;  // [cap] noise v | [cap] noise w | [cap] uniforms
    const double u[2] = {0.0, 0.0};
    StepParams p = make_params(h, u, /*n_obs hint=*/4);
    p.step = step0;
    AdaptSmallArgs a{};
    a.img = image_args(h);
    a.plan = plan_args(h, /*mode=*/1, RR_RESAMPLE_MULTINOMIAL, NAN);
    a.plan.rstep = rstep0;
    a.kld = h->kld;
    a.max_draws = h->kld.max_particles;
    a.hash_size = h->kld_hash_size;
    a.res = ra;
    hipLaunchKernelGGL(k_mcl_adaptive_small, dim3(1), dim3(kKldThreads), (size_t)kResPayloadCap * sizeof(double), h->stream, h->b, h->w, h->ctl,
                       p, arg, a, h->cdf, h->idx, h->kld_keys, h->kld_table, h->kld_minslot, h->kld_myslot, h->kld_out, h->cdf_coarse,
                       h->coarse_log2, 0, (HostMail*)nullptr, (uint64_t)0, h->res.ring, h->scratch_a, h->cap);
    RR_HIP_TRY(hipGetLastError());
  } else {
    SmallArgs a{};
    small_args_common(h, &a, /*n_obs hint=*/4, /*K=*/0, /*want_est=*/true);
    a.step0 = step0;
    a.rstep0 = rstep0;
    a.res = ra;
    if ((s = launch_small(h, a, arg, resident_lds_bytes(h->n), nullptr)) != RR_OK) return s;
  }
  h->res.live = true;
  h->res.launches += 1;
  h->maybe_pending = false;
  h->pending_kind = kSrcMarkers;
  return RR_OK;
}

// the answer to command `seq` (out may be null: only wait).  An incarnation that left before it took the command (idle / end
// of life) is replaced; the command is still in the ring.
static rr_status resident_await(rr_pf* h, uint64_t seq, double out[4]) {
  rr_pf::Resident& r = h->res;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    uint64_t e[4];
    if (rr::ring_take(&r.ring->rsp[3], seq, &e[3]) && rr::ring_take(&r.ring->rsp[2], seq, &e[2]) && rr::ring_take(&r.ring->rsp[1], seq, &e[1]) &&
        rr::ring_take(&r.ring->rsp[0], seq, &e[0])) {
      r.pending = false;
      if (h->adaptive) {  // (the adaptive kernel vouches for its estimate: flags != 0 => form it the long way)
        uint64_t flags = 0;
        while (!rr::ring_take(&r.ring->rsp[rr::kResRspFlags], seq, &flags)) {
        }
        if (flags != 0) {
          if (!out) return RR_OK;
          rr_status s = bind(h);  // parks the kernel, refreshes the host's particle count
          if (s != RR_OK) return s;
          if ((s = materialise(h)) != RR_OK) return s;
          return compute_moments(h, out, nullptr);
        }
      }
      if (out)
        for (int k = 0; k < 4; ++k) std::memcpy(&out[k], &e[k], sizeof(double));
      return RR_OK;
    }
    uint64_t consumed = 0;
    if (r.live && rr::ring_take(&r.ring->rsp[rr::kResRspExit], r.launch_id, &consumed)) {
      r.live = false;  // this incarnation has left
      if (consumed < seq) {
        rr_status s = resident_launch(h, seq, r.cmd_step, r.cmd_rstep);
        if (s != RR_OK) return s;
      }
      continue;  // (consumed >= seq: the answer is on its way)
    }
    if ((spins & 1023u) == 1023u &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > 2000) {
      (void)hipStreamSynchronize(h->stream);
      r.live = false;
      r.pending = false;
      return fail(RR_RUNTIME_ERROR, "the resident step kernel did not answer");
    }
  }
}

// one step through the resident kernel; out == null: do not wait for the answer
static rr_status resident_step(rr_pf* h, const double control[2], const double* obs, size_t n_obs, double out[4]) {
  rr_pf::Resident& r = h->res;
  rr_status s;
  if (r.pending && (s = resident_await(h, r.seq, nullptr)) != RR_OK) return s;  // one command in flight
  if (!r.ring || !r.live) {  // (also allocates the ring)
    if ((s = resident_launch(h, r.seq + 1, h->step, h->rstep)) != RR_OK) return s;
  }
  const uint64_t seq = ++r.seq;
  r.cmd_step = h->step;
  r.cmd_rstep = h->rstep;
  rr::MailPair* c = r.ring->cmd;
  auto bits_of = [](double v) {
    uint64_t u;
    std::memcpy(&u, &v, sizeof u);
    return u;
  };
  for (size_t i = 0; i < 3 * n_obs; ++i) rr::ring_put(&c[3 + i], bits_of(obs[i]), seq);
  rr::ring_put(&c[2], bits_of(control[1]), seq);
  rr::ring_put(&c[1], bits_of(control[0]), seq);
  rr::ring_put(&c[0], (uint64_t)rr::kResOpStep | ((uint64_t)(2 + 3 * n_obs) << 8), seq);
  h->step += 1;
  h->rstep += 1;
  h->wmax_live = false;
  h->wmax_bits_clean = true;
  if (h->adaptive) h->n_dirty = true;  // the new particle count lives on the device (Ctl.n_active, kld_out)
  r.steps += 1;
  r.pending = true;
  if (!out) return RR_OK;
  return resident_await(h, seq, out);
}

// ask the resident kernel to leave and wait until it has: the particle set, the weights and Ctl are in HBM afterwards
namespace {
rr_status resident_park(rr_pf* h) {
  rr_pf::Resident& r = h->res;
  rr_status s = RR_OK;
  if (r.pending) s = resident_await(h, r.seq, nullptr);
  if (r.live) {
    rr::ring_put(&r.ring->cmd[0], (uint64_t)rr::kResOpQuit, ++r.seq);
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
    r.live = false;
  }
  return s;
}
}  // namespace

// K steps (controls: K x 2, obs: K x n_obs x 3, both validated by the caller) in one launch.  est_out: device, K x 4, or null.
static rr_status step_small(rr_pf* h, const double* controls, const double* obs, size_t n_obs, size_t K, bool want_est, double* est_out,
                            bool to_mailbox = false) {
  rr_status s = materialise(h);
  if (s != RR_OK) return s;
  SmallArgs a{};
  if (to_mailbox) {
    if (!h->mail) {
      RR_HIP_TRY(hipHostMalloc(&h->mail, sizeof(HostMail), hipHostMallocDefault));
      std::memset(h->mail, 0, sizeof(HostMail));
    }
    a.mail_seq = ++h->mail_seq;
  }
  small_args_common(h, &a, n_obs, K, want_est);
  ObsArg arg;
  a.inputs_in_kernarg = (K == 1 && n_obs <= (size_t)kMaxObsKernarg) ? 1 : 0;
  if (a.inputs_in_kernarg) {
    a.u0 = controls[0];
    a.u1 = controls[1];
    if (n_obs) std::memcpy(arg.v, obs, 3 * n_obs * sizeof(double));
  } else {
    const size_t per = 2 + 3 * n_obs;
    h->steps_host.resize(K * per);
    for (size_t k = 0; k < K; ++k) {
      h->steps_host[k * per] = controls[2 * k];
      h->steps_host[k * per + 1] = controls[2 * k + 1];
      if (n_obs) std::memcpy(&h->steps_host[k * per + 2], obs + 3 * n_obs * k, 3 * n_obs * sizeof(double));
    }
    if (K * per > h->steps_cap) {
      if (h->steps_dev) RR_HIP_TRY(hipFree(h->steps_dev));
      h->steps_dev = nullptr;
      h->steps_cap = 0;
      RR_HIP_TRY(hipMalloc(&h->steps_dev, (K * per + K * per / 2) * sizeof(double)));
      h->steps_cap = K * per + K * per / 2;
    }
    // pageable source: HIP stages it before returning, so steps_host may be reused by the next call
    RR_HIP_TRY(hipMemcpyAsync(h->steps_dev, h->steps_host.data(), K * per * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  {
    Timed t(h, RR_K_PROPAGATE_WEIGHT);
    s = launch_small(h, a, arg, small_lds_bytes(h->n, n_obs), est_out);
  }
  if (s != RR_OK) return s;
  h->step += (unsigned int)K;
  h->rstep += (unsigned int)K;
  h->wmax_live = false;       // Ctl.wmax holds the maximum of the last step's weights
  h->wmax_bits_clean = true;  // ... and the accumulator is zero
  h->maybe_pending = false;
  h->pending_kind = kSrcMarkers;
  return RR_OK;
}

extern "C" {

const char* rr_last_error(void) { return rr::last_error_slot().c_str(); }
const char* rr_version(void) { return "rust_robotics_amd 0.1.0 (gfx950)"; }
int rr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

rr_status rr_device_pci_bus_id(int32_t device, char* out, size_t cap) {
  if (!out || cap < 13) return fail(RR_INVALID_PARAMETER, "need room for \"0000:00:00.0\" and its terminator");
  RR_HIP_TRY(hipDeviceGetPCIBusId(out, (int)cap, device));
  for (char* c = out; *c; ++c)
    if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // as sysfs spells it
  return RR_OK;
}

void rr_pf_config_default(rr_pf_config* c) {
  if (!c) return;
  c->n_particles = 100;
  c->resample_threshold = 0.5;
  c->range_noise = 0.2;
  c->velocity_noise = 2.0;
  c->yaw_rate_noise = 40.0 * (RR_PI_HI / 180.0);  // 40.0_f64.to_radians()
  c->dt = 0.1;
}

rr_status rr_pf_config_validate(const rr_pf_config* c) { return validate_config(c); }

void rr_pf_options_default(rr_pf_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->device = 0;
  o->resample_scheme = RR_RESAMPLE_MULTINOMIAL;
  o->resample_gate = RR_GATE_NEFF;
  o->likelihood_mode = RR_LIK_FUSED;
}

void rr_pf_options_mcl(rr_pf_options* o) {
  rr_pf_options_default(o);
  if (o) o->resample_gate = RR_GATE_ALWAYS;
}

rr_status rr_pf_create(const rr_pf_config* cfg, const rr_pf_options* opt, rr_pf** out) {
  return create_common(cfg, opt, nullptr, out);
}

rr_status rr_pf_create_with_state(const rr_pf_config* cfg, const rr_pf_options* opt, const double state[4],
                                  rr_pf** out) {
  if (!state) return fail(RR_INVALID_PARAMETER, "null initial state");
  return create_common(cfg, opt, state, out);
}

void rr_mcl_adaptive_default(rr_mcl_adaptive* k) {
  if (!k) return;
  k->min_particles = 100;  // monte_carlo_localization.rs:68-81
  k->max_particles = 5000;
  k->kld_epsilon = 0.05;
  k->kld_z = 2.326;
}

rr_status rr_mcl_adaptive_validate(const rr_mcl_adaptive* k) { return validate_kld(k); }

rr_status rr_pf_create_adaptive(const rr_pf_config* cfg, const rr_pf_options* opt, const rr_mcl_adaptive* kld,
                                const double* state, rr_pf** out) {
  if (!kld) return fail(RR_INVALID_PARAMETER, "null KLD parameters");
  rr_pf_options o;
  if (opt) o = *opt; else rr_pf_options_mcl(&o);
  return create_common(cfg, &o, state, out, kld);
}

uint64_t rr_pf_particle_capacity(const rr_pf* h) { return h ? h->cap : 0; }

rr_status rr_pf_set_particles_n(rr_pf* h, const double* aos, uint64_t n) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->adaptive) return fail(RR_INVALID_PARAMETER, "only an adaptive filter can change its particle count");
  if (n == 0 || n > h->cap) return fail(RR_INVALID_PARAMETER, "particle count must lie in [1, max_particles]");
  if ((s = materialise(h)) != RR_OK) return s;
  set_particle_count(h, n);
  if ((s = publish_particle_count(h)) != RR_OK) return s;
  return rr_pf_set_particles(h, aos);
}

rr_status rr_pf_resample_adaptive_with_uniforms(rr_pf* h, const double* r, size_t n_r, uint64_t* n_new) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->adaptive) return fail(RR_INVALID_PARAMETER, "not an adaptive filter");
  if (!r || n_r != h->kld.max_particles) return fail(RR_INVALID_PARAMETER, "need exactly max_particles uniforms");
  for (size_t k = 0; k < n_r; ++k)
    if (!(r[k] >= 0.0 && r[k] < 1.0)) return fail(RR_INVALID_PARAMETER, "uniforms must lie in [0, 1)");
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = ensure_scratch(h, h->cap, 0)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(h->scratch_a, r, n_r * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if ((s = resample_adaptive(h, h->scratch_a)) != RR_OK) return s;
  if (n_new) *n_new = h->n;
  return RR_OK;
}

void rr_pf_destroy(rr_pf* h) {
  if (!h) return;
  (void)hipSetDevice(h->opt.device);
  if (h->res.live || h->res.pending) (void)resident_park(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->res.ring) (void)hipHostFree(h->res.ring);
  h->p2p.teardown();
  (void)hipFree(h->slab);
  (void)hipFree(h->w);
  (void)hipFree(h->cdf);
  (void)hipFree(h->cdf_coarse);
  (void)hipFree(h->guide);
  (void)hipFree(h->guide_markers);
  (void)hipFree(h->guide_carry);
  (void)hipFree(h->tile_total);
  (void)hipFree(h->tile_q2);
  (void)hipFree(h->idx);
  (void)hipFree(h->markers);
  (void)hipFree(h->lidx);
  (void)hipFree(h->packed[0]);
  (void)hipFree(h->packed[1]);
  (void)hipFree(h->kld_keys);
  (void)hipFree(h->kld_table);
  (void)hipFree(h->kld_minslot);
  (void)hipFree(h->kld_myslot);
  (void)hipFree(h->kld_out);
  if (h->kld_out_host) (void)hipHostFree(h->kld_out_host);
  (void)hipFree(h->carry);
  (void)hipFree(h->partials);
  (void)hipFree(h->est_partials);
  if (h->est_partials_host) (void)hipHostFree(h->est_partials_host);
  (void)hipFree(h->est_slot_partials);
  if (h->est_slot_partials_host) (void)hipHostFree(h->est_slot_partials_host);
  (void)hipFree(h->steps_dev);
  (void)hipFree(h->est_ring);
  if (h->mail) (void)hipHostFree(h->mail);
  (void)hipFree(h->mn_tile_cnt);
  (void)hipFree(h->est_ticket);
  (void)hipFree(h->rccl_inbox);
  (void)hipFree(h->grid_rec);
  (void)hipFree(h->grid_ticket);
  (void)hipFree(h->scratch_a);
  (void)hipFree(h->scratch_b);
  (void)hipFree(h->obs_dev);
  (void)hipFree(h->ctl);
  if (h->ctl_host) (void)hipHostFree(h->ctl_host);
  for (auto& e : h->events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto e : h->event_pool) (void)hipEventDestroy(e);
  if (h->owns_stream && h->own_stream) (void)hipStreamDestroy(h->own_stream);
  rr::spin_release(h->opt.device, h);
  delete h;
}

rr_status rr_pf_set_landmarks(rr_pf* h, const double* xy, size_t n) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  if (n && !xy) return fail(RR_INVALID_PARAMETER, "null landmarks");
  for (size_t k = 0; k < 2 * n; ++k)
    if (!std::isfinite(xy[k]))  // particle_filter.rs:525-536
      return fail(RR_INVALID_PARAMETER, "particle filter landmarks must contain only finite values");
  h->landmarks.assign(xy, xy + 2 * n);
  return RR_OK;
}

size_t rr_pf_landmark_count(const rr_pf* h) { return h ? h->landmarks.size() / 2 : 0; }

size_t rr_pf_get_landmarks(const rr_pf* h, double* xy_out, size_t cap) {
  if (!h) return 0;
  const size_t cnt = h->landmarks.size() / 2;
  const size_t m = cnt < cap ? cnt : cap;
  if (xy_out && m) std::memcpy(xy_out, h->landmarks.data(), 2 * m * sizeof(double));
  return cnt;
}

rr_status rr_pf_set_range_noise(rr_pf* h, double range_noise) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  if (!std::isfinite(range_noise) || range_noise <= 0.0)
    return fail(RR_INVALID_PARAMETER, "particle filter range_noise must be positive and finite");
  h->cfg.range_noise = range_noise;
  h->lik = rr_pf_lik_make(range_noise);
  return RR_OK;
}

rr_status rr_pf_predict(rr_pf* h, const double control[2]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = validate_control(control)) != RR_OK) return s;
  StepParams p = make_params(h, control, 0);
  ObsArg arg;
  s = launch_pw<true, false, false>(h, p, arg, true);
  h->step += 1;
  return s;
}

rr_status rr_pf_predict_with_noise(rr_pf* h, const double control[2], const double* n_v, const double* n_w) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = validate_control(control)) != RR_OK) return s;
  if (!n_v || !n_w) return fail(RR_INVALID_PARAMETER, "null noise arrays");
  if ((s = ensure_scratch(h, h->n, h->n)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(h->scratch_a, n_v, h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RR_HIP_TRY(hipMemcpyAsync(h->scratch_b, n_w, h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  StepParams p = make_params(h, control, 0);
  ObsArg arg;
  s = launch_pw<true, false, true>(h, p, arg, true);
  h->step += 1;
  return s;
}

rr_status rr_pf_update(rr_pf* h, const double* obs, size_t n_obs) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, nullptr, (int)n_obs);
  return launch_pw<false, true, false>(h, p, arg, kernarg);
}

rr_status rr_pf_resample(rr_pf* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (h->adaptive) return resample_adaptive(h, nullptr);
  return launch_resample(h, 0, h->opt.resample_scheme, NAN, nullptr);
}

// want_estimate: the fused systematic step (and the small-set step, either resampler) also leaves the mean of the particle set
static bool fused_estimate_available(const rr_pf* h) {
  if (small_path(h, 0)) return true;
  if (h->adaptive || h->n_tiles > (uint64_t)rr::kFusedMaxTiles || h->n != h->n_global) return false;
  // systematic: in the plan kernel or deferred; multinomial: deferred only, through the lazy resample (lidx), one GPU
  return h->opt.resample_scheme == RR_RESAMPLE_SYSTEMATIC || (h->lidx != nullptr && !h->p2p.ready);
}

// the adaptive step of a filter of the reference's sizes: one launch of one workgroup (k_mcl_adaptive_small) up to 1 024 candidate
// draws; beyond that the same kernel does propagate + weight + integer image + CDF + plan and the draws / table / count follow
// as three wide launches (k_kld_draw, k_kld_insert, k_kld_count with the gather)
static rr_status step_adaptive_small(rr_pf* h, const StepParams& p, const ObsArg& arg, uint64_t* mail_seq_out) {
  AdaptSmallArgs a{};
  a.img = image_args(h);
  a.plan = plan_args(h, /*mode=*/1, RR_RESAMPLE_MULTINOMIAL, NAN);
  a.kld = h->kld;
  a.max_draws = h->kld.max_particles;
  a.hash_size = h->kld_hash_size;
  const uint64_t M = h->kld.max_particles;
  static const bool hybrid = [] { const char* e = std::getenv("RR_MCL_HYBRID"); return e && std::atoi(e) != 0; }();
  const int front_only = hybrid && M > 1024 ? 1 : 0;  // (A/B: the front half here, the draws / table / count as three wide launches)
  const size_t lds = 3 * (size_t)p.n_obs * sizeof(double);
  HostMail* mail = nullptr;
  uint64_t want = 0;
  if (mail_seq_out && !front_only) {  // a synchronous caller: the estimate comes back through the host mailbox
    if (!h->mail) {
      RR_HIP_TRY(hipHostMalloc(&h->mail, sizeof(HostMail), hipHostMallocDefault));
      std::memset(h->mail, 0, sizeof(HostMail));
    }
    mail = h->mail;
    want = ++h->mail_seq;
    *mail_seq_out = want;
  }
  {
    Timed t(h, RR_K_PROPAGATE_WEIGHT);
    hipLaunchKernelGGL(k_mcl_adaptive_small, dim3(1), dim3(kKldThreads), lds, h->stream, h->b, h->w, h->ctl, p, arg, a, h->cdf, h->idx,
                       h->kld_keys, h->kld_table, h->kld_minslot, h->kld_myslot, h->kld_out, h->cdf_coarse, h->coarse_log2, front_only,
                       mail, want, (rr::ResidentRing*)nullptr, (double*)nullptr, h->cap);
  }
  if (front_only) {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    const uint64_t cap_coarse = ((h->cap + (1ull << h->coarse_log2) - 1) >> h->coarse_log2) + 1;
    hipLaunchKernelGGL(k_kld_draw, dim3(grid_for(M, kBlock)), dim3(kBlock), cap_coarse * sizeof(uint64_t), h->stream, h->b,
                       h->ctl, h->cdf, h->cdf_coarse, h->coarse_log2, h->n_coarse, (const double*)nullptr, h->idx, h->kld_keys, h->n, M,
                       h->opt.seed, h->rstep, 1);
    hipLaunchKernelGGL(k_kld_insert, dim3(grid_for(M, kBlock)), dim3(kBlock), 0, h->stream, (const int32_t*)h->kld_keys,
                       h->kld_table, h->kld_minslot, h->kld_myslot, M, h->kld_hash_size);
    hipLaunchKernelGGL(k_kld_count, dim3(1), dim3(kKldThreads), 0, h->stream, h->kld_minslot, (const unsigned int*)h->kld_myslot, M, h->kld,
                       h->kld_out, h->kld_table, h->kld_hash_size, h->b, h->ctl, (const unsigned int*)h->idx, 1);
  }
  RR_HIP_TRY(hipGetLastError());
  h->step += 1;
  h->rstep += 1;
  h->wmax_live = false;
  h->wmax_bits_clean = true;  // finalize_plan zeroes the accumulator (this launch never used it)
  h->n_dirty = true;          // weights are uniform 1/n_new from here (Ctl.weights_uniform); the count lives on the device
  return RR_OK;
}

// want_estimate: rr::kEstOff, rr::kEstInPlan (the synchronous caller reads it back at once) or rr::kEstDeferred (see rr::EstArgs)
static rr_status step_async_impl(rr_pf* h, const double control[2], const double* obs, size_t n_obs, int want_estimate,
                                 uint64_t* mail_seq_out = nullptr) {
  rr_status s = bind(h, /*keep_lazy=*/h && h->adaptive);  // an adaptive filter steps without knowing its current count on the host
  if (s != RR_OK) return s;
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  if (small_path(h, n_obs)) return step_small(h, control, obs, n_obs, 1, want_estimate != rr::kEstOff, nullptr);
  if (want_estimate && !fused_estimate_available(h))
    return fail(RR_INVALID_PARAMETER, "no in-step estimate for this filter (adaptive, sharded, or beyond 8 388 608 particles)");
  if (want_estimate == rr::kEstDeferred && h->p2p.ready) want_estimate = rr::kEstInPlan;  // (the window kernels do not sum)
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, control, (int)n_obs);
  if (h->adaptive) {  // try_step, monte_carlo_localization.rs:291-300 -- no host synchronisation: the count stays on the device
    if (kernarg && h->adaptive_small_ok && h->kld.max_particles <= 16384) return step_adaptive_small(h, p, arg, mail_seq_out);
    if ((s = launch_pw<true, true, false>(h, p, arg, kernarg)) != RR_OK) return s;
    h->step += 1;
    return resample_adaptive(h, nullptr, /*lazy=*/true);
  }
  const bool multinomial = h->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC;
  if (multinomial && !h->lidx) {
    if ((s = launch_pw<true, true, false>(h, p, arg, kernarg)) != RR_OK) return s;
    h->step += 1;
    return launch_resample(h, 0, h->opt.resample_scheme, NAN, nullptr);
  }
  if (h->maybe_pending && h->pending_kind != (multinomial ? kSrcLidx : kSrcMarkers) && (s = materialise(h)) != RR_OK) return s;
  // systematic: 2 launches per step -- k_step_lazy (propagate + weight, reading through the previous resample's
  // indices) and k_quantize_plan_mark (3 beyond 2^20 particles: k_quantize_reduce, k_plan_mark)
  const size_t lds = 3 * n_obs * sizeof(double);
  if (lds > 150 * 1024) return fail(RR_INVALID_PARAMETER, "too many observations for one LDS block (max 6400)");
  if (!h->wmax_bits_clean) RR_HIP_TRY(hipMemsetAsync(&h->ctl->wmax_bits, 0, sizeof(uint64_t), h->stream));
  h->wmax_bits_clean = false;
  h->wmax_live = true;
  const uint64_t n_rtiles = (h->n + rr::kResolveSlots - 1) / rr::kResolveSlots;
  const unsigned grid = (unsigned)n_rtiles;  // one tile per workgroup
  {
    Timed t(h, RR_K_PROPAGATE_WEIGHT);
    WindowArgs wa_est{};  // the deferred estimate of the step before: summed by this launch as it gathers
    wa_est.est_partials = h->est_deferred ? h->est_slot_partials : nullptr;
    h->est_deferred = false;
    if (multinomial && h->mn_deferred) {  // ... are still to be drawn: this launch does it for its own slots
      WindowArgs wa = wa_est;
      wa.cdf = h->cdf;
      wa.guide = h->guide;
      wa.n_src = h->mn_deferred_args.n_src;
      wa.rstep = h->mn_deferred_args.rstep;
      wa.guide_log2 = h->guide_log2;
      launch_k1(h, kernarg, kSrcDraw, grid, lds, nullptr, nullptr, p, arg, nullptr, nullptr, h->idx, wa, /*packed=*/true);
      h->mn_deferred = false;
    } else if (multinomial) {  // sources of the previous (multinomial) resample are in lidx
      launch_k1(h, kernarg, kSrcLidx, grid, lds, nullptr, nullptr, p, arg, h->lidx, nullptr, nullptr, wa_est,
                /*packed=*/h->packed[0] != nullptr);
    } else {
      hipEvent_t ea = nullptr, eb = nullptr;
      if (h->profiling && h->profile_dispatch_only) {  // timestamps of this dispatch itself: nothing extra in the stream
        ea = take_event(h);
        eb = take_event(h);
        h->events.push_back({RR_K_PROPAGATE_WEIGHT, ea, eb});
      }
      launch_k1(h, kernarg, kSrcMarkers, grid, lds, ea, eb, p, arg, h->markers, h->carry, h->idx, wa_est);
    }
  }
  RR_HIP_TRY(hipGetLastError());
  h->step += 1;
  h->maybe_pending = false;  // consumed (k_quantize_reduce settles Ctl.cur)
  h->pending_kind = kSrcMarkers;
  return launch_resample(h, 0, h->opt.resample_scheme, NAN, nullptr, /*lazy=*/true, /*settle=*/1, want_estimate);
}

rr_status rr_pf_step_async(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  if (h && resident_path(h, n_obs)) {
    rr_status s = bind(h, /*keep_lazy=*/h->adaptive, /*keep_resident=*/true);
    if (s != RR_OK) return s;
    if ((s = validate_control(control)) != RR_OK) return s;
    if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
    return resident_step(h, control, obs, n_obs, nullptr);
  }
  return step_async_impl(h, control, obs, n_obs, rr::kEstOff);
}

// The resident service of a small filter (<= 2048 particles, <= 128 observations per step): idle_us > 0 switches it on --
// rr_pf_step / rr_pf_step_async then talk to ONE kernel that stays on the device, keeps the particles in registers and leaves by
// itself after idle_us without a step (and is started again by the next one); 0 switches it off.  Every other entry point
// asks the kernel to leave first, so results are those of the launched steps, bit for bit.
rr_status rr_pf_set_resident(rr_pf* h, double idle_us) {
  rr_status s = bind(h);  // (parks a live kernel)
  if (s != RR_OK) return s;
  if (!(idle_us >= 0.0) || !(idle_us <= 1e7)) return fail(RR_INVALID_PARAMETER, "resident idle time must lie in [0, 1e7] microseconds");
  h->res.enabled = idle_us > 0.0;
  h->res.idle_us = idle_us;
  h->res.life_us = std::max(100000.0, 20.0 * idle_us);
  return RR_OK;
}

rr_status rr_pf_resident_stats(const rr_pf* h, uint64_t* launches, uint64_t* steps) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  if (launches) *launches = h->res.launches;
  if (steps) *steps = h->res.steps;
  return RR_OK;
}

rr_status rr_pf_step_async_estimate(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  if (h && !fused_estimate_available(h))
    return fail(RR_INVALID_PARAMETER, "the in-step estimate needs the fused step (fixed N, one shard, <= 8 388 608 particles); use "
                                      "rr_pf_step / rr_pf_estimate");
  // Systematic: in the plan kernel.  The deferred form (rr::EstArgs: the resampled set's mean summed by the kernel that moves the
  // particles, the next step's k_step_lazy) was measured against it at 1e6 x 32, round 4: the plan kernel gets 1 us shorter, the
  // step kernel 1.6 us longer (profiles/r04d_deferred_estimate_ab.md) -- RR_PF_EST_DEFER=1 selects it for A/B.  Multinomial: always
  // deferred, the offspring counts of iid draws do not exist before the draws are searched.
  const char* e = std::getenv("RR_PF_EST_DEFER");  // (read per call: the tests switch it within one process)
  const int sys_mode = e && std::atoi(e) != 0 ? (int)rr::kEstDeferred : (int)rr::kEstInPlan;
  const bool small = h && small_path(h, 0);
  const int mode = (h && !small && h->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC) ? (int)rr::kEstDeferred : sys_mode;
  return step_async_impl(h, control, obs, n_obs, mode);
}

rr_status rr_pf_last_step_estimate(rr_pf* h, double out[4]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  // the deferred form, and no step has come since: the resample is made real here (gather + k_est_slots: the same sums the next
  // step would have formed)
  if (h->est_deferred && (s = materialise(h)) != RR_OK) return s;
  if (h->est_deferred) launch_est_slots(h);  // (nothing was pending any more: the live set is the resampled set)
  if (!h->est_partials_host) RR_HIP_TRY(hipHostMalloc(&h->est_partials_host, (size_t)rr::kFusedMaxTiles * 4 * sizeof(double)));
  RR_HIP_TRY(hipMemcpyAsync(h->est_partials_host, h->est_partials, (size_t)h->n_tiles * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  const uint64_t n_slot_tiles = grid_for(h->n, rr::kResolveSlots);
  if (h->est_slot_partials) {
    if (!h->est_slot_partials_host)
      RR_HIP_TRY(hipHostMalloc(&h->est_slot_partials_host, (size_t)grid_for(h->cap, rr::kResolveSlots) * kEstSlotWords * sizeof(double)));
    RR_HIP_TRY(hipMemcpyAsync(h->est_slot_partials_host, h->est_slot_partials, (size_t)n_slot_tiles * kEstSlotWords * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
  }
  if ((s = fetch_ctl(h)) != RR_OK) return s;  // (synchronises the stream)
  if (h->ctl_host->est_step == 0) return fail(RR_INVALID_PARAMETER, "no step has produced an in-step estimate yet");
  // the tiles' partial sums in tile order (a fixed order: the same bits whichever kernel produced them)
  const bool slots = h->ctl_host->est_kind == rr::kEstSlotTiles;
  if (slots && !h->est_slot_partials) return fail(RR_RUNTIME_ERROR, "the deferred estimate's sums are missing");
  const double* part = slots ? h->est_slot_partials_host : h->est_partials_host;
  const uint64_t n_part = slots ? n_slot_tiles * (kBlock / rr::kWave) : h->n_tiles;  // (slot tiles: one entry per wave)
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (slots) {  // est_slots_total: interleaved chunks, then the chunks in order (k_est_mail_any adds them the same way)
    for (int c = 0; c < kEstChunks; ++c) {
      double cs[4] = {0.0, 0.0, 0.0, 0.0};
      for (uint64_t t = (uint64_t)c; t < n_part; t += kEstChunks)
        for (int k = 0; k < 4; ++k) cs[k] += part[4 * t + k];
      for (int k = 0; k < 4; ++k) acc[k] += cs[k];
    }
  } else {
    for (uint64_t t = 0; t < n_part; ++t)
      for (int k = 0; k < 4; ++k) acc[k] += part[4 * t + k];
  }
  for (int k = 0; k < 4; ++k) out[k] = acc[k] / h->ctl_host->est_denom;
  return RR_OK;
}

rr_status rr_pf_synchronize(rr_pf* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  return fetch_ctl(h);  // waits for the stream; also the place where a handle learns that its one-launch plan had to degrade
}

rr_status rr_pf_estimate(rr_pf* h, double out[4]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  return compute_moments(h, out, nullptr);
}

rr_status rr_pf_covariance(rr_pf* h, double out[16]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  return compute_moments(h, nullptr, out);
}

rr_status rr_pf_step_many(rr_pf* h, const double* controls, const double* obs, size_t n_obs, size_t n_steps, double* out_estimates) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (n_steps == 0) return RR_OK;
  if (!controls) return fail(RR_INVALID_PARAMETER, "null controls");
  if (n_steps > (size_t)1 << 20) return fail(RR_INVALID_PARAMETER, "at most 1 048 576 steps per call");
  for (size_t k = 0; k < n_steps; ++k)
    if ((s = validate_control(controls + 2 * k)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs * n_steps)) != RR_OK) return s;
  if (small_path(h, n_obs)) {
    double* ring = nullptr;
    if (out_estimates) {
      if (4 * n_steps > h->est_ring_cap) {
        if (h->est_ring) RR_HIP_TRY(hipFree(h->est_ring));
        h->est_ring = nullptr;
        h->est_ring_cap = 0;
        RR_HIP_TRY(hipMalloc(&h->est_ring, 4 * n_steps * sizeof(double)));
        h->est_ring_cap = 4 * n_steps;
      }
      ring = h->est_ring;
    }
    if ((s = step_small(h, controls, obs, n_obs, n_steps, out_estimates != nullptr, ring)) != RR_OK) return s;
    if (!out_estimates) return RR_OK;
    RR_HIP_TRY(hipMemcpyAsync(out_estimates, ring, 4 * n_steps * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return fetch_ctl(h);  // waits for the stream
  }
  // large filters: the steps one after the other (each of them fills the device on its own)
  for (size_t k = 0; k < n_steps; ++k) {
    const double* o = n_obs ? obs + 3 * n_obs * k : nullptr;
    if (out_estimates) s = rr_pf_step(h, controls + 2 * k, o, n_obs, out_estimates + 4 * k);
    else s = rr_pf_step_async(h, controls + 2 * k, o, n_obs);
    if (s != RR_OK) return s;
  }
  return RR_OK;
}

rr_status rr_pf_step(rr_pf* h, const double control[2], const double* obs, size_t n_obs, double out_state[4]) {
  if (h && out_state && resident_path(h, n_obs)) {
    // try_step of a small filter with the resident service switched on: no launch, no completion signal -- the command goes
    // into pinned memory, the answer comes back the same way (resident_core.hpp)
    rr_status s = bind(h, /*keep_lazy=*/h->adaptive, /*keep_resident=*/true);
    if (s != RR_OK) return s;
    if ((s = validate_control(control)) != RR_OK) return s;
    if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
    return resident_step(h, control, obs, n_obs, out_state);
  }
  if (h && out_state && small_path(h, n_obs)) {
    // try_step of a small filter: one launch; the kernel writes the mean into the host-visible mailbox and the host polls
    // its stamp -- no device-to-host copy, no stream synchronisation in the common case
    rr_status s = bind(h);
    if (s != RR_OK) return s;
    if ((s = validate_control(control)) != RR_OK) return s;
    if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
    if ((s = step_small(h, control, obs, n_obs, 1, true, nullptr, /*to_mailbox=*/true)) != RR_OK) return s;
    if ((s = await_mail(h, h->mail_seq)) != RR_OK) return s;
    for (int k = 0; k < 4; ++k) out_state[k] = h->mail->est[k];
    return RR_OK;
  }
  if (h && out_state && h->opt.resample_scheme == RR_RESAMPLE_SYSTEMATIC && fused_estimate_available(h) && !small_path(h, 0)) {
    // try_step (particle_filter.rs:488-497): the returned mean comes out of the step's own plan kernel -- one
    // 300-byte read-back instead of a gather + a two-kernel moment reduction.  (k_est_mail as the closing act of the plan kernel
    // itself -- device-scope partial sums, a second arrival ticket, the last workgroup adds and posts -- was measured, round 4:
    // 71.9 - 73.7 us per synchronous step against 69.7 - 70.1 with the separate launch, whose dispatch overlaps the plan kernel.)
    rr_status s = step_async_impl(h, control, obs, n_obs, rr::kEstInPlan);
    if (s != RR_OK) return s;
    if (h->p2p.ready) return rr_pf_last_step_estimate(h, out_state);
    if (!h->mail) {
      RR_HIP_TRY(hipHostMalloc(&h->mail, sizeof(HostMail), hipHostMallocDefault));
      std::memset(h->mail, 0, sizeof(HostMail));
    }
    const uint64_t want = ++h->mail_seq;
    hipLaunchKernelGGL(k_est_mail, dim3(1), dim3(256), 0, h->stream, (const Ctl*)h->ctl, (const double*)h->est_partials, h->n_tiles, h->mail, want);
    RR_HIP_TRY(hipGetLastError());
    if ((s = await_mail(h, want)) != RR_OK) return s;
    if (h->mail->flags) return rr_pf_last_step_estimate(h, out_state);  // a degraded plan to take note of (fetch_ctl), or no estimate
    for (int k = 0; k < 4; ++k) out_state[k] = h->mail->est[k];
    return RR_OK;
  }
  if (h && out_state && h->opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL && fused_estimate_available(h) && !small_path(h, 0) &&
      !h->profiling) {
    // try_step of a large multinomial filter (the resampler the reference's localizers use): plan, then ONE launch that searches
    // the draws and adds up the resampled set's mean (k_mn_search_est), then the mailbox -- instead of search + gather + settle +
    // two moment kernels
    h->est_eager = true;
    h->est_eager_done = false;
    rr_status s = step_async_impl(h, control, obs, n_obs, rr::kEstDeferred);
    h->est_eager = false;
    if (s != RR_OK) return s;
    if (!h->est_eager_done) return rr_pf_last_step_estimate(h, out_state);  // (no guide table / packed mirror: the long way)
    if (!h->mail) {
      RR_HIP_TRY(hipHostMalloc(&h->mail, sizeof(HostMail), hipHostMallocDefault));
      std::memset(h->mail, 0, sizeof(HostMail));
    }
    const uint64_t want = ++h->mail_seq;
    hipLaunchKernelGGL(k_est_mail_any, dim3(1), dim3(kEstChunks), 0, h->stream, (const Ctl*)h->ctl, (const double*)h->est_partials, h->n_tiles,
                       (const double*)h->est_slot_partials, (uint64_t)grid_for(h->n, rr::kResolveSlots) * (kBlock / rr::kWave), h->mail, want);
    RR_HIP_TRY(hipGetLastError());
    if ((s = await_mail(h, want)) != RR_OK) return s;
    if (h->mail->flags) return rr_pf_last_step_estimate(h, out_state);
    for (int k = 0; k < 4; ++k) out_state[k] = h->mail->est[k];
    return RR_OK;
  }
  uint64_t want = 0;
  rr_status s = step_async_impl(h, control, obs, n_obs, rr::kEstOff, (h && h->adaptive && out_state) ? &want : nullptr);
  if (s != RR_OK) return s;
  if (!out_state) return rr_pf_synchronize(h);
  if (want) {  // the adaptive step of a small filter has formed the mean itself (k_mcl_adaptive_small): poll the mailbox
    if ((s = await_mail(h, want)) != RR_OK) return s;
    if (h->mail->flags == 0) {
      for (int k = 0; k < 4; ++k) out_state[k] = h->mail->est[k];
      return RR_OK;
    }
  }
  // an adaptive step leaves the new particle count on the device (Ctl.n_active; h->n is stale while n_dirty): the moment kernels
  // below are sized from the host's copy, so bring it up to date first (one 8-byte copy + a wait)
  if ((s = bind(h)) != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;  // the estimate is over the resampled set
  return compute_moments(h, out_state, nullptr);
}

uint64_t rr_pf_particle_count(const rr_pf* h) {
  if (!h) return 0;
  if (h->n_dirty) (void)bind(const_cast<rr_pf*>(h));  // an adaptive filter after asynchronous steps: wait for the device's count
  return h->n;
}

rr_status rr_pf_get_fixed_sums(rr_pf* h, rr_pf_fixed_sums* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = launch_sums(h, 2, h->opt.resample_scheme, NAN)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  const Ctl& c = *h->ctl_host;
  out->usable = c.usable;
  out->shift = c.shift;
  out->total = c.total;
  out->q2_hi = c.q2_hi;
  out->q2_lo = c.q2_lo;
  out->w_max = c.wmax;
  out->sum = c.sum;
  return RR_OK;
}

rr_status rr_pf_n_eff(rr_pf* h, double* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = launch_sums(h, 2, h->opt.resample_scheme, NAN)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  *out = h->ctl_host->neff;
  return RR_OK;
}

rr_status rr_pf_last_resample_fired(rr_pf* h, int32_t* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  *out = h->ctl_host->fired;
  return RR_OK;
}

rr_status rr_pf_get_particles(rr_pf* h, double* out_aos) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!out_aos) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = ensure_scratch(h, 5 * h->n, 0)) != RR_OK) return s;
  if ((s = launch_sums(h, 2, h->opt.resample_scheme, NAN)) != RR_OK) return s;  // refresh Ctl.sum / usable
  hipLaunchKernelGGL(k_pack_aos, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->w, h->ctl, h->n,
                     h->n_global, h->scratch_a);
  RR_HIP_TRY(hipGetLastError());
  RR_HIP_TRY(hipMemcpyAsync(out_aos, h->scratch_a, 5 * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return h->p2p.check(h->stream);
}

rr_status rr_pf_set_particles(rr_pf* h, const double* aos) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!aos) return fail(RR_INVALID_PARAMETER, "null input");
  if ((s = ensure_scratch(h, 5 * h->n, 0)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(h->scratch_a, aos, 5 * h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RR_HIP_TRY(hipMemsetAsync(&h->ctl->wmax_bits, 0, sizeof(uint64_t), h->stream));
  h->wmax_live = true;
  h->wmax_bits_clean = false;
  hipLaunchKernelGGL(k_unpack_aos, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->w, h->ctl, h->n,
                     (const double*)h->scratch_a);
  RR_HIP_TRY(hipGetLastError());
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}

rr_status rr_pf_resample_with_uniforms(rr_pf* h, const double* r, size_t n) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (h->adaptive) return fail(RR_INVALID_PARAMETER, "adaptive filter: use rr_pf_resample_adaptive_with_uniforms");
  if (!r || n != h->n) return fail(RR_INVALID_PARAMETER, "need exactly one uniform per particle");
  for (size_t k = 0; k < n; ++k)
    if (!(r[k] >= 0.0 && r[k] < 1.0)) return fail(RR_INVALID_PARAMETER, "uniforms must lie in [0, 1)");
  if ((s = ensure_scratch(h, h->n, 0)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(h->scratch_a, r, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return launch_resample(h, 1, RR_RESAMPLE_MULTINOMIAL, NAN, h->scratch_a);
}

rr_status rr_pf_resample_systematic(rr_pf* h, double rho) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (h->adaptive) return fail(RR_INVALID_PARAMETER, "adaptive filter: use rr_pf_resample_adaptive_with_uniforms");
  if (!(rho >= 0.0 && rho < 1.0)) return fail(RR_INVALID_PARAMETER, "rho must lie in [0, 1)");
  return launch_resample(h, 1, RR_RESAMPLE_SYSTEMATIC, rho, nullptr);
}

rr_status rr_pf_last_resample_indices(rr_pf* h, uint32_t* out, size_t n) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!h->idx) return fail(RR_INVALID_PARAMETER, "record_indices was not enabled for this filter");
  if (!out || n != h->n) return fail(RR_INVALID_PARAMETER, "need room for one index per particle");
  RR_HIP_TRY(hipMemcpyAsync(out, h->idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}

rr_status rr_pf_get_raw_weights(rr_pf* h, double* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  RR_HIP_TRY(hipMemcpyAsync(out, h->w, h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}

rr_status rr_pf_plan_stats(rr_pf* h, uint64_t* giveups, int32_t* one_launch_enabled) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  if (giveups) *giveups = h->plan_giveups;
  if (one_launch_enabled) *one_launch_enabled = h->grid_capacity != 0 ? 1 : 0;
  return RR_OK;
}

#if defined(RR_PLAN_TIMELINE)
// instrumented build only (tools/resident_timeline.py): the stamps the resident kernel sent with its last answer
rr_status rr_pf_debug_resident_timeline(rr_pf* h, uint64_t out[8]) {
  if (!h || !h->res.ring) return fail(RR_INVALID_PARAMETER, "no resident service");
  for (int k = 7; k >= 0; --k) {  // (the stamps leave the device after the answer: wait for THIS step's)
    uint64_t v = 0;
    for (long spins = 0; spins < 100000000L && !rr::ring_take(&h->res.ring->rsp[8 + k], h->res.seq, &v); ++spins) {
    }
    out[k] = v;
  }
  return RR_OK;
}
// instrumented build only (tools/plan_timeline.py): the stamps of the last k_quantize_plan_mark launch, n_tiles x kTimelineWords
rr_status rr_pf_debug_plan_timeline(rr_pf* h, uint64_t* out, size_t cap_words) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  const size_t words = std::min<size_t>(cap_words, (size_t)h->n_tiles * rr::kTimelineWords);
  RR_HIP_TRY(hipMemcpyAsync(out, h->grid_rec + (rr::kTileBlock + 1) * rr::kRecWords + 16, words * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}
#endif

rr_status rr_pf_get_counters(rr_pf* h, uint32_t* step, uint32_t* resample_step) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  if (step) *step = h->step;
  if (resample_step) *resample_step = h->rstep;
  return RR_OK;
}

rr_status rr_pf_set_stream(rr_pf* h, void* stream) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  if (stream) {
    h->stream = (hipStream_t)stream;
    h->using_external_stream = true;
  } else {
    h->stream = h->own_stream;
    h->using_external_stream = false;
  }
  return RR_OK;
}

static rr_status require_systematic_shard(const rr_pf* h) {
  if (h->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC)
    return fail(RR_INVALID_PARAMETER, "this entry point serves the systematic sharded resample (contiguous served slots); a multinomial "
                                      "shard uses rr_pf_shard_select / rr_pf_shard_pack_selected / rr_pf_shard_adopt_records");
  return RR_OK;
}

rr_status rr_pf_shard_propagate_weight(rr_pf* h, const double control[2], const double* obs, size_t n_obs,
                                       double* d_wmax_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!d_wmax_out) return fail(RR_INVALID_PARAMETER, "null d_wmax_out");
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, control, (int)n_obs);
  if ((s = launch_pw<true, true, false>(h, p, arg, kernarg)) != RR_OK) return s;
  h->step += 1;
  RR_HIP_TRY(hipMemcpyAsync(d_wmax_out, &h->ctl->wmax_bits, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return RR_OK;
}

rr_status rr_pf_shard_quantize(rr_pf* h, const double* d_wmax_global, uint64_t* d_sums_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!d_wmax_global || !d_sums_out) return fail(RR_INVALID_PARAMETER, "null device pointer");
  launch_quantize(h, d_wmax_global);
  {
    Timed t(h, RR_K_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 0, plan_args(h, 0, RR_RESAMPLE_SYSTEMATIC, NAN), d_sums_out);
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status rr_pf_shard_cdf(rr_pf* h, const uint64_t* d_all_sums, int32_t n_shards, int32_t rank) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!d_all_sums || n_shards <= 0 || rank < 0 || rank >= n_shards)
    return fail(RR_INVALID_PARAMETER, "bad shard sums / rank");
  hipLaunchKernelGGL(rr::k_shard_plan, dim3(1), dim3(1), 0, h->stream, h->ctl, d_all_sums, (int)n_shards, (int)rank,
                     plan_args(h, 0, h->opt.resample_scheme, NAN));
  {
    Timed t(h, RR_K_CDF);
    if (h->opt.resample_scheme == RR_RESAMPLE_SYSTEMATIC)
      hipLaunchKernelGGL(rr::k_mark, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl, image_args(h),
                         h->tile_total, h->markers, h->carry);
    else  // multinomial: the local slice of the global CDF is searched per draw (rr_pf_shard_select / _pack_selected)
      hipLaunchKernelGGL(rr::k_cdf, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl, image_args(h),
                         h->tile_total, h->cdf, (uint64_t*)nullptr, 0);
  }
  RR_HIP_TRY(hipGetLastError());
  h->wmax_live = false;
  h->wmax_bits_clean = true;
  h->shard_plan_valid = true;
  h->shard_plan_rstep = h->rstep;
  h->shard_plan_shards = n_shards;
  h->shard_select_shards = 0;
  h->rstep += 1;
  return RR_OK;
}

rr_status rr_pf_shard_get_plan(rr_pf* h, rr_pf_shard_plan* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  const Ctl& c = *h->ctl_host;
  out->fired = c.fired;
  out->usable = c.usable;
  out->total_global = c.total;
  out->base = c.base;
  out->total_local = c.total_local;
  out->rho = c.rho;
  return RR_OK;
}

rr_status rr_pf_shard_gather_slots(rr_pf* h, uint64_t first_slot, uint64_t n_slots, double* d_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = require_systematic_shard(h)) != RR_OK) return s;
  if (n_slots == 0) return RR_OK;
  if (!d_out) return fail(RR_INVALID_PARAMETER, "null d_out");
  if (first_slot + n_slots > h->n_global) return fail(RR_INVALID_PARAMETER, "slot range exceeds n_global");
  (void)first_slot;  // markers are relative to the first slot this shard serves, which is what the caller passes
  {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    hipLaunchKernelGGL(k_resolve_gather, dim3(grid_for(n_slots, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->b,
                       h->ctl, h->markers, h->carry, (unsigned int*)nullptr, d_out, n_slots, 1, 0);
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status rr_pf_shard_adopt(rr_pf* h, const double* d_in) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!d_in) return fail(RR_INVALID_PARAMETER, "null d_in");
  hipLaunchKernelGGL(k_adopt, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->ctl, d_in, h->n,
                     (const double*)nullptr, (uint64_t)0, (uint64_t)0);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

// the same with the self-served slots [self_lo, self_hi) read from d_self (rr_pf_shard_step: no send to oneself)
static rr_status shard_adopt_with_self(rr_pf* h, const double* d_in, const double* d_self, uint64_t self_lo, uint64_t self_hi) {
  hipLaunchKernelGGL(k_adopt, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->ctl, d_in, h->n, d_self, self_lo,
                     self_hi);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

// ---- sharded multinomial resample (see the kernels): select -> counts, pack -> records, adopt
static MnSelectArgs mn_args(const rr_pf* h, int n_shards) {
  MnSelectArgs a{};
  a.n_local = h->n;
  a.n_global = h->n_global;
  a.tiles_per_dest = (h->n + kTile - 1) / kTile;
  a.seed = h->opt.seed;
  a.rstep = h->shard_plan_rstep;  // the resample rr_pf_shard_cdf planned (it has advanced the counter since)
  a.n_shards = n_shards;
  return a;
}

rr_status rr_pf_shard_select(rr_pf* h, int32_t n_shards, uint64_t* d_counts_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (h->opt.resample_scheme != RR_RESAMPLE_MULTINOMIAL) return fail(RR_INVALID_PARAMETER, "rr_pf_shard_select serves multinomial shards");
  if (!d_counts_out || n_shards <= 0 || n_shards > kMaxP2P || h->n_global != h->n * (uint64_t)n_shards)
    return fail(RR_INVALID_PARAMETER, "bad shard count (equal blocks, at most 16 shards)");
  if (!h->shard_plan_valid || h->shard_plan_shards != n_shards)
    return fail(RR_INVALID_PARAMETER, "rr_pf_shard_select needs the plan of a preceding rr_pf_shard_cdf with the same number of shards");
  const MnSelectArgs a = mn_args(h, n_shards);
  const uint64_t n_tiles = a.tiles_per_dest * (uint64_t)n_shards;
  if (n_tiles > h->mn_tiles) {
    if (h->mn_tile_cnt) RR_HIP_TRY(hipFree(h->mn_tile_cnt));
    h->mn_tile_cnt = nullptr;
    h->mn_tiles = 0;
    RR_HIP_TRY(hipMalloc(&h->mn_tile_cnt, n_tiles * sizeof(unsigned int)));
    h->mn_tiles = n_tiles;
  }
  Timed t(h, RR_K_RESAMPLE_GATHER);
  hipLaunchKernelGGL(k_mn_select_count, dim3((unsigned)n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->ctl, a, h->mn_tile_cnt);
  hipLaunchKernelGGL(k_mn_select_scan, dim3(1), dim3(kScanThreads), 0, h->stream, h->mn_tile_cnt, n_tiles, a.tiles_per_dest, (int)n_shards,
                     d_counts_out);
  RR_HIP_TRY(hipGetLastError());
  h->shard_select_shards = n_shards;
  return RR_OK;
}

rr_status rr_pf_shard_pack_selected(rr_pf* h, int32_t n_shards, double* d_send) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (h->opt.resample_scheme != RR_RESAMPLE_MULTINOMIAL) return fail(RR_INVALID_PARAMETER, "rr_pf_shard_pack_selected serves multinomial shards");
  if (!d_send) return fail(RR_INVALID_PARAMETER, "null send buffer");
  if (!h->shard_plan_valid || h->shard_select_shards == 0 || h->shard_select_shards != n_shards)
    return fail(RR_INVALID_PARAMETER, "rr_pf_shard_pack_selected needs a preceding rr_pf_shard_select with the same number of shards");
  const MnSelectArgs a = mn_args(h, n_shards);
  const uint64_t n_tiles = a.tiles_per_dest * (uint64_t)n_shards;
  if (n_tiles > h->mn_tiles) return fail(RR_INVALID_PARAMETER, "call rr_pf_shard_select first");
  Timed t(h, RR_K_RESAMPLE_GATHER);
  hipLaunchKernelGGL(k_mn_select_pack, dim3((unsigned)n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->b, h->ctl, a,
                     (const unsigned int*)h->mn_tile_cnt, (const uint64_t*)h->cdf, h->n, d_send);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status rr_pf_shard_adopt_records(rr_pf* h, const double* d_in, uint64_t n_records) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!d_in) return fail(RR_INVALID_PARAMETER, "null d_in");
  if (n_records != h->n) return fail(RR_INVALID_PARAMETER, "every output slot of the shard needs exactly one record");
  hipLaunchKernelGGL(k_adopt_records, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->b, h->ctl, d_in, h->n);
  RR_HIP_TRY(hipGetLastError());
  h->shard_plan_valid = false;  // this resample is done
  return RR_OK;
}

uint64_t rr_sys_first_slot_above(double rho, uint64_t total_global, uint64_t n_global, uint64_t bound) {
  if (n_global == 0) return 0;
  const rr_sys_plan p = rr_sys_plan_make(rho, total_global, n_global);
  uint64_t lo = 0, hi = n_global;  // target is non-decreasing in the slot index
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (rr_sys_target(p, mid) > bound) hi = mid; else lo = mid + 1;
  }
  return lo;
}

uint64_t rr_sys_segment_matrix(double rho, const uint64_t* totals, int32_t n_shards, uint64_t n_global,
                               uint64_t n_local, int32_t rank, int64_t* out) {
  if (!totals || !out || n_shards <= 0) return 0;
  uint64_t total = 0;
  for (int g = 0; g < n_shards; ++g) total += totals[g];
  const rr_sys_plan p = rr_sys_plan_make(rho, total, n_global ? n_global : 1);
  uint64_t base = 0, first_of_rank = 0;
  for (int src = 0; src < n_shards; ++src) {
    const uint64_t lo = rr_sys_slots_upto_exact(p, total, base);
    const uint64_t hi = rr_sys_slots_upto_exact(p, total, base + totals[src]);
    if (src == rank) first_of_rank = lo;
    base += totals[src];
    for (int dst = 0; dst < n_shards; ++dst) {
      const uint64_t a = std::max<uint64_t>(lo, (uint64_t)dst * n_local);
      const uint64_t b = std::min<uint64_t>(hi, (uint64_t)(dst + 1) * n_local);
      out[(size_t)src * n_shards + dst] = b > a ? (int64_t)(b - a) : 0;
    }
  }
  return first_of_rank;
}

// ---- device-initiated exchange (include/rr_pf.h "peer-to-peer transport")
static rr_status p2p_check_geometry(const rr_pf* h, int n_ranks, int rank) {
  if (n_ranks <= 0 || n_ranks > kMaxP2P || rank < 0 || rank >= n_ranks)
    return fail(RR_INVALID_PARAMETER, "peer-to-peer transport supports 1..16 ranks");
  if (h->n_global != h->n * (uint64_t)n_ranks || h->opt.first_global_index != h->n * (uint64_t)rank)
    return fail(RR_INVALID_PARAMETER, "shard geometry does not match the rank layout (equal blocks, rank * n_local)");
  if (h->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC)
    return fail(RR_INVALID_PARAMETER, "sharded resampling is systematic only");
  return RR_OK;
}

static rr_status p2p_alloc_lidx(rr_pf* h) {
  if (h->lidx) return RR_OK;
  RR_HIP_TRY(hipMalloc(&h->lidx, h->n * sizeof(unsigned int)));
  RR_HIP_TRY(hipMemset(h->lidx, 0xff, h->n * sizeof(unsigned int)));  // kInPlace everywhere
  return RR_OK;
}

rr_status rr_pf_p2p_export(rr_pf* h, uint8_t out[RR_P2P_HANDLE_BYTES]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  return h->p2p.export_handles(h->slab, 5 * h->n, out);  // inbox: 4 fields + the seal plane
}

rr_status rr_pf_p2p_connect(rr_pf* h, const uint8_t* all_handles, int32_t n_ranks, int32_t rank) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!all_handles) return fail(RR_INVALID_PARAMETER, "null handles");
  if ((s = p2p_check_geometry(h, n_ranks, rank)) != RR_OK) return s;
  if ((s = p2p_alloc_lidx(h)) != RR_OK) return s;
  return h->p2p.connect_ipc(h->slab, 5 * h->n, all_handles, n_ranks, rank);
}

rr_status rr_pf_p2p_connect_local(rr_pf* const* handles, int32_t n_ranks) {
  if (!handles || n_ranks <= 0 || n_ranks > kMaxP2P) return fail(RR_INVALID_PARAMETER, "bad handle list");
  rr::P2PState* st[kMaxP2P];
  double* slabs[kMaxP2P];
  size_t inboxes[kMaxP2P];
  int devs[kMaxP2P];
  for (int g = 0; g < n_ranks; ++g) {
    if (!handles[g]) return fail(RR_INVALID_PARAMETER, "null handle");
    rr_status s = p2p_check_geometry(handles[g], n_ranks, g);
    if (s != RR_OK) return s;
    if ((s = bind(handles[g])) != RR_OK) return s;
    if ((s = p2p_alloc_lidx(handles[g])) != RR_OK) return s;
    st[g] = &handles[g]->p2p;
    slabs[g] = handles[g]->slab;
    inboxes[g] = 5 * handles[g]->n;
    devs[g] = handles[g]->opt.device;
  }
  return rr::p2p_link_local(st, slabs, inboxes, devs, n_ranks);
}

// THREE launches: k_step_lazy<kSrcWindow> (propagate + weight; the own slots inside the window this shard served last step
// are resolved from the markers and read in place exactly as on one GPU, the few a peer served come out of the inbox, each
// as soon as its own seal fits) | k_shard_plan_mark (WMAX exchange, integer image, SUMS exchange, gate + plan, markers for the
// served window over the global slot index) | k_push_window (the window's overhang over the own block, resolved and stored
// into the owners' inboxes, every slot sealed -- no acknowledgement round trip, no DONE message).  Shards beyond 2^20
// particles, or several ranks on one device, take the plan as four launches (WMAX exchange | k_quantize_reduce |
// k_scan_exchange | k_mark).  Round 2: 4 launches with a full-size resolve pass (k_resolve_push, 7.9 us at 1e6 particles) and
// a DONE exchange everybody waited in.  (Delivering the overhang from inside the plan kernel, source side, was built and
// measured in round 3: docs/DESIGN_NOTES.md section 5 -- it loses to this on every count.)
rr_status rr_pf_shard_step_p2p_unfused(rr_pf* h, const double control[2], const double* obs, size_t n_obs);

rr_status rr_pf_shard_step_p2p(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->p2p.ready) return fail(RR_INVALID_PARAMETER, "call rr_pf_p2p_connect first");
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  // Ranks that SHARE a device (a test rig; one rank per GPU is the deployment): this step's k_step_lazy waits, inside the
  // kernel, for the particles a peer's k_push_window delivers.  On separate devices that is a wait for another GPU.  On one
  // device the waiting kernels of the sharers can hold every workgroup slot while the push kernel they wait for has not been
  // dispatched yet -- it then never gets a slot, and the waits run into their bound: the round-3 "give-up at 10^6 particles per
  // rank" (run down in round 4 with tools/p2p_shared_device_jump.py: after a resample that moves most of a shard, every slot
  // was found delivered and consistent in memory -- AFTER the waiters had given up and freed the device).  Sharers whose step
  // kernels together can fill the device therefore take the eager form of the step, which has no wait inside a full-size kernel.
  if (h->p2p.n_sharing > 1) {
    if (!h->dev_cus) RR_HIP_TRY(hipDeviceGetAttribute(&h->dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    const uint64_t step_wgs = (h->n + rr::kResolveSlots - 1) / rr::kResolveSlots;
    if (step_wgs * (uint64_t)h->p2p.n_sharing > 3ull * (uint64_t)h->dev_cus) return rr_pf_shard_step_p2p_unfused(h, control, obs, n_obs);
  }
  if (h->maybe_pending && (h->pending_kind != kSrcWindow || h->window_rccl) && (s = materialise(h)) != RR_OK) return s;
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, control, (int)n_obs);
  const size_t lds = 3 * n_obs * sizeof(double);
  if (lds > 150 * 1024) return fail(RR_INVALID_PARAMETER, "too many observations for one LDS block (max 6400)");
  if (!h->wmax_bits_clean) RR_HIP_TRY(hipMemsetAsync(&h->ctl->wmax_bits, 0, sizeof(uint64_t), h->stream));
  WindowArgs wa{};
  wa.inbox = h->p2p.inbox;
  wa.err = h->p2p.err;
  wa.pad = h->slot_pad;
  wa.wait_seq = h->window_seq;  // the step whose resample this launch consumes (its deliveries carry that seal)
  wa.timeout_ticks = h->p2p.peers.timeout_ticks;
  wa.n_ranks = h->p2p.peers.n_ranks;
  const uint64_t seq = ++h->p2p.seq;
  uint64_t* gathered = h->p2p.gathered();
  // A: propagate + weight through the window
  const uint64_t n_rtiles = (h->n + rr::kResolveSlots - 1) / rr::kResolveSlots;
  const unsigned grid = (unsigned)n_rtiles;  // one tile per workgroup
  {
    Timed t(h, RR_K_PROPAGATE_WEIGHT);
    hipEvent_t ea = nullptr, eb = nullptr;
    if (h->profiling && h->profile_dispatch_only) {  // timestamps of this dispatch itself
      ea = take_event(h);
      eb = take_event(h);
      h->events.push_back({RR_K_PROPAGATE_WEIGHT, ea, eb});
    }
    launch_k1(h, kernarg, kSrcWindow, grid, lds, ea, eb, p, arg, h->markers, h->carry, nullptr, wa);
  }
  h->step += 1;
  PlanArgs pa = plan_args(h, 0, RR_RESAMPLE_SYSTEMATIC, NAN);
  pa.lazy_gather = 1;
  if (h->shard_capacity == ~0ull) {  // every workgroup of k_shard_plan_mark resident at once?
    int per_cu = 0, dev_cus = 0;
    RR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rr::k_shard_plan_mark, rr::kTileBlock, 0));
    RR_HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    h->dev_cus = dev_cus;
    h->shard_capacity = h->grid_capacity ? std::min<uint64_t>((uint64_t)per_cu * (uint64_t)dev_cus, (uint64_t)rr::kTileBlock) : 0;
  }
  // Ranks that share this device (a test rig: several shards of one filter on one GPU) run their kernels beside this one's;
  // a plan kernel that spins on every CU would leave a peer's exchange workgroup -- the one it is waiting for -- nowhere to
  // go (seen as stalls of seconds with two 1e6-particle shards on one device).  All sharers together keep to one
  // workgroup per CU.
  const uint64_t fused_cap = h->p2p.n_sharing > 1 ? std::min<uint64_t>(h->shard_capacity, (uint64_t)h->dev_cus / (uint64_t)h->p2p.n_sharing)
                                                  : h->shard_capacity;
  if (h->n_tiles <= fused_cap && rr::spin_permit(h->opt.device, h)) {
    // exchange 1 + B + exchange 2 + C in one launch (k_shard_plan_mark)
    Timed t(h, RR_K_CDF);
    hipLaunchKernelGGL(rr::k_shard_plan_mark, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->p2p.peers, seq,
                       (const double*)h->w, h->ctl, image_args(h), h->grid_rec, h->grid_ticket, ++h->grid_epoch, /*settle=*/1,
                       h->n_tiles, pa, h->markers, h->carry, gathered, h->p2p.err, h->slot_pad);
  } else {
    // exchange 1: global maximum -> Ctl.wmax
    hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PWmax, seq,
                       (const uint64_t*)&h->ctl->wmax_bits, gathered, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
    // B: integer image under the global maximum (settles the resample K1 consumed)
    launch_quantize(h, (const double*)&h->ctl->wmax, /*settle=*/1);
    // tile scan + exchange 2: every rank's sums -> gate, base, plan in Ctl
    {
      Timed t(h, RR_K_SCAN_TILES);
      hipLaunchKernelGGL(rr::k_scan_exchange, dim3(1), dim3(kScanThreads), 0, h->stream, h->p2p.peers, seq, h->tile_total,
                         (const uint64_t*)h->tile_q2, h->n_tiles, gathered, h->ctl, pa, h->p2p.err);
    }
    // C: mark this shard's sources
    {
      Timed t(h, RR_K_CDF);
      hipLaunchKernelGGL(rr::k_mark, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl,
                         image_args(h), h->tile_total, h->markers, h->carry, /*window=*/1, h->slot_pad);
    }
  }
  h->wmax_live = false;
  h->wmax_bits_clean = true;
  h->rstep += 1;
  // D: the overhang of the served window goes to its owners, every slot sealed (k_push_window); a filter of one rank has
  // no peers and its window is its own block
  if (h->p2p.peers.n_ranks > 1) {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    hipLaunchKernelGGL(k_push_window, dim3(kPushGrid), dim3(kBlock), 0, h->stream, h->b, h->ctl, h->markers, h->carry, h->p2p.peers,
                       h->n, h->slot_pad, seq);
  }
  RR_HIP_TRY(hipGetLastError());
  h->maybe_pending = true;
  h->pending_kind = kSrcWindow;
  h->window_seq = seq;
  h->window_rccl = false;
  return RR_OK;
}

// The same step with every phase as its own launch (three exchange kernels, eager gather): kept as
// the plain statement of the protocol and for A/B measurement.
rr_status rr_pf_shard_step_p2p_unfused(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->p2p.ready) return fail(RR_INVALID_PARAMETER, "call rr_pf_p2p_connect first");
  const uint64_t seq = ++h->p2p.seq;
  uint64_t* gathered = h->p2p.gathered();
  uint64_t* local3 = h->p2p.local3();
  // A: propagate + weight; the local maximum stays in Ctl.wmax_bits
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, control, (int)n_obs);
  if ((s = launch_pw<true, true, false>(h, p, arg, kernarg)) != RR_OK) return s;
  h->step += 1;
  PlanArgs pa = plan_args(h, 0, RR_RESAMPLE_SYSTEMATIC, NAN);
  // exchange 1: global maximum -> Ctl.wmax (k_quantize_reduce reads it from there)
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PWmax, seq,
                     (const uint64_t*)&h->ctl->wmax_bits, gathered, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  // B: integer image under the global maximum, local sums -> local3
  launch_quantize(h, (const double*)&h->ctl->wmax);
  {
    Timed t(h, RR_K_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 0, pa, local3);
  }
  // exchange 2: every rank's sums -> plan (gate, base, systematic plan) in Ctl
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PSums, seq,
                     (const uint64_t*)local3, gathered, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  // C: mark this shard's sources
  {
    Timed t(h, RR_K_CDF);
    hipLaunchKernelGGL(rr::k_mark, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl,
                       image_args(h), h->tile_total, h->markers, h->carry);
  }
  h->wmax_live = false;
  h->wmax_bits_clean = true;
  h->rstep += 1;
  // D+E: resolve and store every served slot straight into its owner's slab.  The number of
  // served slots is only known on the device: launch for the worst case this shard could serve
  // (all of them) -- surplus workgroups return at once.
  {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    hipLaunchKernelGGL(k_resolve_gather_p2p, dim3(grid_for(h->n_global, rr::kResolveSlots)), dim3(kBlock), 0, h->stream,
                       h->b, h->ctl, h->markers, h->carry, h->p2p.peers, h->n);
  }
  // exchange 3: every rank has finished writing into everybody's slab
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PDone, seq,
                     (const uint64_t*)local3, gathered, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status rr_pf_p2p_status(rr_pf* h, int32_t* timed_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!timed_out) return fail(RR_INVALID_PARAMETER, "null output");
  return h->p2p.status(h->stream, timed_out);
}

// ---------------------------------------------------------------------------------------------
// native sharded step: RCCL through dlopen
}  // extern "C"

#include <dlfcn.h>

extern "C" {

rr_status rr_comm_unique_id(uint8_t out[RR_COMM_UNIQUE_ID_BYTES]) {
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  rr_status s = rccl_load();
  if (s != RR_OK) return s;
  RR_NCCL_TRY(rccl().GetUniqueId(out));
  return RR_OK;
}

rr_status rr_comm_create(const uint8_t id[RR_COMM_UNIQUE_ID_BYTES], int32_t rank, int32_t n_ranks, int32_t device,
                         rr_comm** out) {
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  *out = nullptr;
  if (!id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(RR_INVALID_PARAMETER, "bad communicator arguments");
  rr_status s = rccl_load();
  if (s != RR_OK) return s;
  RR_HIP_TRY(hipSetDevice(device));
  rr_comm* c = new rr_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->device = device;
  ncclUniqueIdPod uid;
  std::memcpy(uid.internal, id, RR_COMM_UNIQUE_ID_BYTES);
  int e = rccl().CommInitRank(&c->comm, n_ranks, uid, rank);
  if (e != 0) {
    delete c;
    return fail(RR_RUNTIME_ERROR, std::string("ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "error"));
  }
  auto bad = [&](hipError_t err) {
    rr_comm_destroy(c);
    return fail(RR_RUNTIME_ERROR, std::string("communicator scratch: ") + hipGetErrorString(err));
  };
  hipError_t err;
  if ((err = hipMalloc(&c->d_wmax, sizeof(double))) != hipSuccess) return bad(err);
  if ((err = hipMalloc(&c->d_sums, 3 * sizeof(uint64_t))) != hipSuccess) return bad(err);
  if ((err = hipMalloc(&c->d_all, (3 * n_ranks + 1) * sizeof(uint64_t))) != hipSuccess) return bad(err);
  if ((err = hipHostMalloc(&c->h_all, (3 * n_ranks + 1) * sizeof(uint64_t))) != hipSuccess) return bad(err);
  if ((err = hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming)) != hipSuccess) return bad(err);
  if ((err = hipMalloc(&c->d_mom, 21 * (n_ranks + 1) * sizeof(double))) != hipSuccess) return bad(err);
  if ((err = hipHostMalloc(&c->h_mom, 21 * (n_ranks + 1) * sizeof(double))) != hipSuccess) return bad(err);
  c->matrix.assign((size_t)n_ranks * n_ranks, 0);
  *out = c;
  return RR_OK;
}

void rr_comm_destroy(rr_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  (void)hipFree(c->d_wmax);
  (void)hipFree(c->d_sums);
  (void)hipFree(c->d_all);
  if (c->h_all) (void)hipHostFree(c->h_all);
  if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
  (void)hipFree(c->d_send);
  (void)hipFree(c->d_recv);
  (void)hipFree(c->d_fsend);
  (void)hipFree(c->d_frecv);
  (void)hipFree(c->d_cnt);
  if (c->h_cnt) (void)hipHostFree(c->h_cnt);
  (void)hipFree(c->d_mom);
  if (c->h_mom) (void)hipHostFree(c->h_mom);
  delete c;
}

// The systematic sharded step over RCCL, lazy like the peer-to-peer one: k_step_lazy<kSrcWindow> | all-reduce(MAX) |
// k_quantize_reduce_sums (image + this shard's sums by its last workgroup) | all-gather(sums) | k_mark_plan (gate + plan +
// window markers) | D2H of the sums + event | k_pack_window | grouped send/recv of the window's overhang only | k_unpack_inbox.  Own slots inside the own window never
// move (the next step reads them through the markers); the host's one wait -- for the G sums that size the segments --
// is an EVENT recorded right behind the copy, so it overlaps k_mark and k_pack_window instead of draining the stream.
// Round 2 propagated with the non-lazy kernel, gathered ALL served slots into a send buffer and adopted all n slots
// (two 64 MB passes at 1e6 particles) behind a full stream synchronisation: 99 us at world size 1.
// The step is written as phases with the three exchanges between them, so that the same code runs over RCCL
// (rr_pf_shard_step) and over plain device copies between shards of ONE process (rr_pf_shard_step_local: the seam that
// lets a one-GPU box check the segment logic for 2 and 3 shards).
struct WinPlan {
  bool fired = false;
  uint64_t n_send = 0, n_recv = 0, below = 0, self = 0;
};

static double* win_wmax_slot(rr_comm* c) { return reinterpret_cast<double*>(c->d_all + 3 * (size_t)c->n_ranks); }

static rr_status win_ensure(double** buf, size_t* cap, size_t records) {
  if (records <= *cap) return RR_OK;
  if (*buf) RR_HIP_TRY(hipFree(*buf));
  *buf = nullptr;
  *cap = 0;
  const size_t want = records + records / 4 + 1024;
  RR_HIP_TRY(hipMalloc(buf, want * 4 * sizeof(double)));
  *cap = want;
  return RR_OK;
}

static void win_pack(rr_pf* h, rr_comm* c) {
  Timed t(h, RR_K_RESAMPLE_GATHER);
  hipLaunchKernelGGL(k_pack_window, dim3(kPushGrid), dim3(kBlock), 0, h->stream, h->b, h->ctl, h->markers, h->carry, c->rank, h->n, h->slot_pad,
                     c->d_send, (uint64_t)c->cap_send);
}

// A: propagate + weight through the window; the local maximum stays in Ctl.wmax_bits
static rr_status win_phase_a(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  rr_status s;
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  if (h->maybe_pending && !(h->pending_kind == kSrcWindow && h->window_rccl) && (s = materialise(h)) != RR_OK) return s;
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, control, (int)n_obs);
  const size_t lds = 3 * n_obs * sizeof(double);
  if (lds > 150 * 1024) return fail(RR_INVALID_PARAMETER, "too many observations for one LDS block (max 6400)");
  if (!h->rccl_inbox) RR_HIP_TRY(hipMalloc(&h->rccl_inbox, 4 * h->n * sizeof(double)));
  if (!h->wmax_bits_clean) RR_HIP_TRY(hipMemsetAsync(&h->ctl->wmax_bits, 0, sizeof(uint64_t), h->stream));
  WindowArgs wa{};
  wa.inbox = h->rccl_inbox;
  wa.pad = h->slot_pad;
  wa.n_ranks = 0;  // nothing to wait for inside the kernel
  {
    Timed t(h, RR_K_PROPAGATE_WEIGHT);
    launch_k1(h, kernarg, kSrcWindow, (unsigned)((h->n + rr::kResolveSlots - 1) / rr::kResolveSlots), lds, nullptr, nullptr, p, arg,
              h->markers, h->carry, nullptr, wa);
  }
  RR_HIP_TRY(hipGetLastError());
  h->step += 1;
  return RR_OK;
}

// B: the integer image under the GLOBAL maximum (in the slot behind the gathered sums), local sums -> c->d_sums
static rr_status win_phase_b(rr_pf* h, rr_comm* c) {
  Timed t(h, RR_K_QUANTIZE_REDUCE);
  hipLaunchKernelGGL(rr::k_quantize_reduce_sums, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, (const double*)h->w, h->ctl,
                     (const double*)win_wmax_slot(c), image_args(h), h->tile_total, h->tile_q2, /*settle=*/1, h->n_tiles, h->grid_ticket,
                     c->d_sums);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

// C: gate + plan + markers over the global slot index, the sums on their way to the host, the overhang into the send buffer
static rr_status win_phase_c(rr_pf* h, rr_comm* c) {
  const int G = c->n_ranks;
  PlanArgs pa = plan_args(h, 0, RR_RESAMPLE_SYSTEMATIC, NAN);
  pa.lazy_gather = 1;
  {
    Timed t(h, RR_K_CDF);
    hipLaunchKernelGGL(rr::k_mark_plan, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, (const double*)h->w, h->ctl, image_args(h),
                       (const uint64_t*)h->tile_total, (const uint64_t*)c->d_all, G, c->rank, pa, h->markers, h->carry, h->slot_pad);
  }
  RR_HIP_TRY(hipMemcpyAsync(c->h_all, c->d_all, (3 * (size_t)G + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipEventRecord(c->ev_plan, h->stream));
  if (G > 1) {
    rr_status s = win_ensure(&c->d_send, &c->cap_send, std::max<size_t>(65536, h->n / 8));
    if (s != RR_OK) return s;
    win_pack(h, c);
  }
  RR_HIP_TRY(hipGetLastError());
  h->wmax_live = false;
  h->wmax_bits_clean = true;
  h->rstep += 1;
  return RR_OK;
}

// the host's part: gate decision and segment sizes from the G sums (pure integer arithmetic, the same on every rank)
static rr_status win_host(rr_pf* h, rr_comm* c, WinPlan* out) {
  const int G = c->n_ranks, r = c->rank;
  RR_HIP_TRY(hipEventSynchronize(c->ev_plan));
  const double wmax = rr_u2d(c->h_all[3 * (size_t)G]);
  const bool usable = wmax > 0.0 && wmax < INFINITY;
  std::vector<uint64_t> totals(G);
  uint64_t total = 0;
  u128 qq = {0, 0};
  for (int g = 0; g < G; ++g) {
    totals[g] = c->h_all[3 * g];
    total += totals[g];
    qq = rr::add128(qq, u128{c->h_all[3 * g + 1], c->h_all[3 * g + 2]});
  }
  const double neff = (usable && total > 0) ? rr_fix_neff(total, qq.hi, qq.lo) : (double)h->n_global;  // (unusable: the uniform image)
  const double threshold = (double)h->n_global * h->cfg.resample_threshold;
  *out = WinPlan{};
  out->fired = h->opt.resample_gate == RR_GATE_ALWAYS || neff < threshold;
  h->last_migrated = 0;
  if (!out->fired) {
    h->maybe_pending = false;
    h->pending_kind = kSrcMarkers;
    return RR_OK;
  }
  h->maybe_pending = true;
  h->pending_kind = kSrcWindow;
  h->window_rccl = true;
  h->window_seq = 0;
  out->self = h->n;
  if (G == 1) return RR_OK;
  double rho, dummy;
  rr_uniform2(h->opt.seed, RR_STREAM_RESAMPLE, h->rstep - 1, 0, &rho, &dummy);  // (phase C has advanced the counter)
  (void)rr_sys_segment_matrix(rho, totals.data(), G, h->n_global, h->n, r, c->matrix.data());
  const int64_t* M = c->matrix.data();
  uint64_t migrated = 0;
  for (int g = 0; g < G; ++g) {
    if (g != r) {
      out->n_send += (uint64_t)M[(size_t)r * G + g];
      out->n_recv += (uint64_t)M[(size_t)g * G + r];
      if (g < r) out->below += (uint64_t)M[(size_t)g * G + r];
    }
    for (int d = 0; d < G; ++d)
      if (g != d) migrated += (uint64_t)M[(size_t)g * G + d];
  }
  h->last_migrated = migrated;
  out->self = (uint64_t)M[(size_t)r * G + r];
  if (out->n_recv + out->self != h->n) return fail(RR_RUNTIME_ERROR, "segment plan does not cover this shard's slots exactly once");
  rr_status s;
  if (out->n_send > c->cap_send) {  // the overhang did not fit: the kernel left everything in place
    if ((s = win_ensure(&c->d_send, &c->cap_send, out->n_send)) != RR_OK) return s;
    win_pack(h, c);
    RR_HIP_TRY(hipGetLastError());
  }
  return win_ensure(&c->d_recv, &c->cap_recv, out->n_recv);
}

// D: what the peers served for this shard's slots -> the inbox
static rr_status win_phase_d(rr_pf* h, rr_comm* c, const WinPlan& w) {
  if (!w.fired || !w.n_recv) return RR_OK;
  hipLaunchKernelGGL(k_unpack_inbox, dim3(grid_for(w.n_recv, kBlock)), dim3(kBlock), 0, h->stream, (const double*)c->d_recv, w.n_recv, w.below,
                     w.self, h->n, h->rccl_inbox);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

static rr_status shard_step_rccl_window(rr_pf* h, rr_comm* c, const double control[2], const double* obs, size_t n_obs) {
  rr_status s;
  Rccl& R = rccl();
  const int G = c->n_ranks, r = c->rank;
  if ((s = win_phase_a(h, control, obs, n_obs)) != RR_OK) return s;
  RR_NCCL_TRY(R.AllReduce(&h->ctl->wmax_bits, win_wmax_slot(c), 1, kNcclFloat64, kNcclMax, c->comm, h->stream));  // (doubles >= 0)
  if ((s = win_phase_b(h, c)) != RR_OK) return s;
  RR_NCCL_TRY(R.AllGather(c->d_sums, c->d_all, 3, kNcclUint64, c->comm, h->stream));
  if ((s = win_phase_c(h, c)) != RR_OK) return s;
  WinPlan w;
  if ((s = win_host(h, c, &w)) != RR_OK) return s;
  if (w.fired && (w.n_send || w.n_recv)) {
    const int64_t* M = c->matrix.data();
    RR_NCCL_TRY(R.GroupStart());
    uint64_t so = 0, ro = 0;
    for (int g = 0; g < G; ++g) {
      if (g == r) continue;
      const uint64_t ns = (uint64_t)M[(size_t)r * G + g], nr = (uint64_t)M[(size_t)g * G + r];
      if (ns) RR_NCCL_TRY(R.Send(c->d_send + 4 * so, 4 * ns, kNcclFloat64, g, c->comm, h->stream));
      if (nr) RR_NCCL_TRY(R.Recv(c->d_recv + 4 * ro, 4 * nr, kNcclFloat64, g, c->comm, h->stream));
      so += ns;
      ro += nr;
    }
    RR_NCCL_TRY(R.GroupEnd());
  }
  return win_phase_d(h, c, w);
}

rr_status rr_pf_shard_step(rr_pf* h, rr_comm* c, const double control[2], const double* obs, size_t n_obs) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!c) return fail(RR_INVALID_PARAMETER, "null communicator");
  if (h->n_global != h->n * (uint64_t)c->n_ranks || h->opt.first_global_index != h->n * (uint64_t)c->rank)
    return fail(RR_INVALID_PARAMETER, "shard geometry does not match the communicator (equal blocks, rank * n_local)");
  static const bool eager = [] { const char* e = std::getenv("RR_PF_RCCL_EAGER"); return e && std::atoi(e) != 0; }();
  if (h->opt.resample_scheme == RR_RESAMPLE_SYSTEMATIC && !eager) return shard_step_rccl_window(h, c, control, obs, n_obs);
  Rccl& R = rccl();
  // A: propagate + weight, local maximum
  if ((s = rr_pf_shard_propagate_weight(h, control, obs, n_obs, c->d_wmax)) != RR_OK) return s;
  RR_NCCL_TRY(R.AllReduce(c->d_wmax, c->d_wmax, 1, kNcclFloat64, kNcclMax, c->comm, h->stream));
  // B: integer image under the global maximum, local sums
  if ((s = rr_pf_shard_quantize(h, c->d_wmax, c->d_sums)) != RR_OK) return s;
  RR_NCCL_TRY(R.AllGather(c->d_sums, c->d_all, 3, kNcclUint64, c->comm, h->stream));
  // C: plan + local marking; the G totals come back to the host to size the segments
  if ((s = rr_pf_shard_cdf(h, c->d_all, c->n_ranks, c->rank)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(c->h_all, c->d_all, 3 * c->n_ranks * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  rr_pf_shard_plan plan;
  if ((s = rr_pf_shard_get_plan(h, &plan)) != RR_OK) return s;  // synchronises the stream
  if (!plan.fired) {
    h->last_migrated = 0;
    return RR_OK;
  }
  const int G = c->n_ranks, r = c->rank;
  if (h->opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL) {
    // scattered served slots: every rank counts what it serves per destination, the counts are all-gathered into
    // the exchange matrix, records (x, y, yaw, v, local slot) travel in one grouped send/recv
    if (!c->d_cnt) {
      RR_HIP_TRY(hipMalloc(&c->d_cnt, ((size_t)G + (size_t)G * G) * sizeof(uint64_t)));
      RR_HIP_TRY(hipHostMalloc(&c->h_cnt, (size_t)G * G * sizeof(uint64_t)));
    }
    if ((s = rr_pf_shard_select(h, G, c->d_cnt)) != RR_OK) return s;
    RR_NCCL_TRY(R.AllGather(c->d_cnt, c->d_cnt + G, G, kNcclUint64, c->comm, h->stream));
    RR_HIP_TRY(hipMemcpyAsync(c->h_cnt, c->d_cnt + G, (size_t)G * G * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
    uint64_t n_send = 0, n_recv = 0, migrated = 0;
    for (int g = 0; g < G; ++g) {
      n_send += c->h_cnt[(size_t)r * G + g];
      n_recv += c->h_cnt[(size_t)g * G + r];
      for (int d = 0; d < G; ++d)
        if (g != d) migrated += c->h_cnt[(size_t)g * G + d];
    }
    h->last_migrated = migrated;
    if (n_recv != h->n) return fail(RR_RUNTIME_ERROR, "multinomial exchange does not cover this shard's slots exactly once");
    auto ensure = [&](double** buf, size_t* cap, size_t need) -> rr_status {
      if (need <= *cap) return RR_OK;
      if (*buf) RR_HIP_TRY(hipFree(*buf));
      *buf = nullptr;
      *cap = 0;
      RR_HIP_TRY(hipMalloc(buf, (need + need / 4 + 1024) * sizeof(double)));
      *cap = need + need / 4 + 1024;
      return RR_OK;
    };
    if ((s = ensure(&c->d_fsend, &c->cap_fsend, 5 * n_send)) != RR_OK) return s;
    if ((s = ensure(&c->d_frecv, &c->cap_frecv, 5 * h->n)) != RR_OK) return s;
    if ((s = rr_pf_shard_pack_selected(h, G, c->d_fsend)) != RR_OK) return s;
    RR_NCCL_TRY(R.GroupStart());
    uint64_t so = 0, ro = 0;
    for (int g = 0; g < G; ++g) {
      const uint64_t ns = c->h_cnt[(size_t)r * G + g], nr = c->h_cnt[(size_t)g * G + r];
      if (ns) RR_NCCL_TRY(R.Send(c->d_fsend + 5 * so, 5 * ns, kNcclFloat64, g, c->comm, h->stream));
      if (nr) RR_NCCL_TRY(R.Recv(c->d_frecv + 5 * ro, 5 * nr, kNcclFloat64, g, c->comm, h->stream));
      so += ns;
      ro += nr;
    }
    RR_NCCL_TRY(R.GroupEnd());
    return rr_pf_shard_adopt_records(h, c->d_frecv, h->n);
  }
  std::vector<uint64_t> totals(G);
  for (int g = 0; g < G; ++g) totals[g] = c->h_all[3 * g];
  const uint64_t first = rr_sys_segment_matrix(plan.rho, totals.data(), G, h->n_global, h->n, r, c->matrix.data());
  const int64_t* M = c->matrix.data();
  uint64_t n_send = 0, n_recv = 0, migrated = 0;
  for (int g = 0; g < G; ++g) {
    n_send += (uint64_t)M[(size_t)r * G + g];
    n_recv += (uint64_t)M[(size_t)g * G + r];
    for (int d = 0; d < G; ++d)
      if (g != d) migrated += (uint64_t)M[(size_t)g * G + d];
  }
  h->last_migrated = migrated;
  if (n_recv != h->n) return fail(RR_RUNTIME_ERROR, "segment plan does not cover this shard's slots exactly once");
  if (n_send > c->cap_send) {
    if (c->d_send) RR_HIP_TRY(hipFree(c->d_send));
    c->d_send = nullptr;
    c->cap_send = 0;
    RR_HIP_TRY(hipMalloc(&c->d_send, (n_send + n_send / 4 + 1024) * 4 * sizeof(double)));
    c->cap_send = n_send + n_send / 4 + 1024;
  }
  if (h->n > c->cap_recv) {
    if (c->d_recv) RR_HIP_TRY(hipFree(c->d_recv));
    c->d_recv = nullptr;
    RR_HIP_TRY(hipMalloc(&c->d_recv, h->n * 4 * sizeof(double)));
    c->cap_recv = h->n;
  }
  // D: gather what the served slots need into one contiguous buffer, exchange the segments
  if ((s = rr_pf_shard_gather_slots(h, first, n_send, c->d_send)) != RR_OK) return s;
  // (what this rank serves to itself -- nearly everything: systematic resampling moves particles by a boundary's drift --
  // does not go through a send / receive to the same rank: k_adopt reads it where the gather left it)
  RR_NCCL_TRY(R.GroupStart());
  uint64_t so = 0, ro = 0, self_so = 0, self_lo = 0, self_hi = 0;
  for (int g = 0; g < G; ++g) {
    const uint64_t ns = (uint64_t)M[(size_t)r * G + g], nr = (uint64_t)M[(size_t)g * G + r];
    if (g == r) {
      self_so = so;
      self_lo = ro;
      self_hi = ro + nr;  // (ns == nr for g == r)
    } else {
      if (ns) RR_NCCL_TRY(R.Send(c->d_send + 4 * so, 4 * ns, kNcclFloat64, g, c->comm, h->stream));
      if (nr) RR_NCCL_TRY(R.Recv(c->d_recv + 4 * ro, 4 * nr, kNcclFloat64, g, c->comm, h->stream));
    }
    so += ns;
    ro += nr;
  }
  RR_NCCL_TRY(R.GroupEnd());
  // E: adopt
  return shard_adopt_with_self(h, c->d_recv, c->d_send + 4 * self_so, self_lo, self_hi);
}

uint64_t rr_pf_shard_last_migrated(const rr_pf* h) { return h ? h->last_migrated : 0; }

rr_status rr_comm_create_local(int32_t rank, int32_t n_ranks, int32_t device, rr_comm** out) {
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  *out = nullptr;
  if (n_ranks <= 0 || n_ranks > kMaxP2P || rank < 0 || rank >= n_ranks) return fail(RR_INVALID_PARAMETER, "bad rank layout (1..16 ranks)");
  RR_HIP_TRY(hipSetDevice(device));
  rr_comm* c = new rr_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->device = device;
  auto bad = [&](hipError_t err) {
    rr_comm_destroy(c);
    return fail(RR_RUNTIME_ERROR, std::string("communicator scratch: ") + hipGetErrorString(err));
  };
  hipError_t err;
  if ((err = hipMalloc(&c->d_wmax, sizeof(double))) != hipSuccess) return bad(err);
  if ((err = hipMalloc(&c->d_sums, 3 * sizeof(uint64_t))) != hipSuccess) return bad(err);
  if ((err = hipMalloc(&c->d_all, (3 * n_ranks + 1) * sizeof(uint64_t))) != hipSuccess) return bad(err);
  if ((err = hipHostMalloc(&c->h_all, (3 * n_ranks + 1) * sizeof(uint64_t))) != hipSuccess) return bad(err);
  if ((err = hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming)) != hipSuccess) return bad(err);
  c->matrix.assign((size_t)n_ranks * n_ranks, 0);
  *out = c;
  return RR_OK;
}

rr_status rr_pf_shard_step_local(rr_pf* const* hs, rr_comm* const* cs, int32_t n_ranks, const double control[2], const double* obs,
                                 size_t n_obs) {
  if (!hs || !cs || n_ranks <= 0 || n_ranks > kMaxP2P) return fail(RR_INVALID_PARAMETER, "bad shard list");
  const int G = n_ranks;
  rr_status s;
  for (int g = 0; g < G; ++g) {
    if (!hs[g] || !cs[g] || cs[g]->n_ranks != G || cs[g]->rank != g) return fail(RR_INVALID_PARAMETER, "shard / communicator list does not match the rank layout");
    if (hs[g]->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC) return fail(RR_INVALID_PARAMETER, "systematic shards only");
    if (hs[g]->n_global != hs[g]->n * (uint64_t)G || hs[g]->opt.first_global_index != hs[g]->n * (uint64_t)g)
      return fail(RR_INVALID_PARAMETER, "shard geometry does not match the rank layout (equal blocks, rank * n_local)");
  }
  // A, then the all-reduce(MAX) by hand
  uint64_t wmax_bits = 0;
  for (int g = 0; g < G; ++g) {
    if ((s = bind(hs[g])) != RR_OK) return s;
    if ((s = win_phase_a(hs[g], control, obs, n_obs)) != RR_OK) return s;
  }
  for (int g = 0; g < G; ++g) {
    if ((s = bind(hs[g])) != RR_OK) return s;
    uint64_t b = 0;
    RR_HIP_TRY(hipMemcpyAsync(&b, &hs[g]->ctl->wmax_bits, sizeof b, hipMemcpyDeviceToHost, hs[g]->stream));
    RR_HIP_TRY(hipStreamSynchronize(hs[g]->stream));
    if (rr_u2d(b) > rr_u2d(wmax_bits)) wmax_bits = b;
  }
  // B, then the all-gather by hand
  std::vector<uint64_t> all(3 * (size_t)G + 1);
  for (int g = 0; g < G; ++g) {
    if ((s = bind(hs[g])) != RR_OK) return s;
    RR_HIP_TRY(hipMemcpyAsync(win_wmax_slot(cs[g]), &wmax_bits, sizeof wmax_bits, hipMemcpyHostToDevice, hs[g]->stream));
    if ((s = win_phase_b(hs[g], cs[g])) != RR_OK) return s;
    RR_HIP_TRY(hipMemcpyAsync(&all[3 * (size_t)g], cs[g]->d_sums, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, hs[g]->stream));
    RR_HIP_TRY(hipStreamSynchronize(hs[g]->stream));
  }
  std::vector<WinPlan> plans(G);
  for (int g = 0; g < G; ++g) {
    if ((s = bind(hs[g])) != RR_OK) return s;
    RR_HIP_TRY(hipMemcpyAsync(cs[g]->d_all, all.data(), 3 * (size_t)G * sizeof(uint64_t), hipMemcpyHostToDevice, hs[g]->stream));
    if ((s = win_phase_c(hs[g], cs[g])) != RR_OK) return s;
    if ((s = win_host(hs[g], cs[g], &plans[g])) != RR_OK) return s;
    RR_HIP_TRY(hipStreamSynchronize(hs[g]->stream));  // the send buffer is complete
  }
  // the exchange: segment (src -> dst) = M[src][dst] records, the src's send buffer and the dst's receive buffer both in rank order
  for (int d = 0; d < G; ++d) {
    if (!plans[d].fired) continue;
    if ((s = bind(hs[d])) != RR_OK) return s;
    const int64_t* M = cs[d]->matrix.data();
    uint64_t ro = 0;
    for (int g = 0; g < G; ++g) {
      if (g == d) continue;
      const uint64_t nr = (uint64_t)M[(size_t)g * G + d];
      uint64_t so = 0;  // offset of the (g -> d) segment in g's send buffer
      for (int q = 0; q < d; ++q)
        if (q != g) so += (uint64_t)M[(size_t)g * G + q];
      if (nr) RR_HIP_TRY(hipMemcpyAsync(cs[d]->d_recv + 4 * ro, cs[g]->d_send + 4 * so, 4 * nr * sizeof(double), hipMemcpyDeviceToDevice, hs[d]->stream));
      ro += nr;
    }
    if ((s = win_phase_d(hs[d], cs[d], plans[d])) != RR_OK) return s;
    RR_HIP_TRY(hipStreamSynchronize(hs[d]->stream));
  }
  return RR_OK;
}

rr_status rr_pf_shard_estimate(rr_pf* h, rr_comm* c, double est[4], double cov[16]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!c) return fail(RR_INVALID_PARAMETER, "null communicator");
  double e[4], cv[16];
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = compute_moments(h, e, cv)) != RR_OK) return s;
  // weight share of this shard: 1/G after a resample, T_local / T otherwise
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  const Ctl& k = *h->ctl_host;
  double share = 1.0 / c->n_ranks;
  if (!k.weights_uniform && k.total > 0) share = (double)k.total_local / (double)k.total;
  double* rec = c->h_mom + 21 * (size_t)c->n_ranks;
  rec[0] = share;
  std::memcpy(rec + 1, e, sizeof e);
  std::memcpy(rec + 5, cv, sizeof cv);
  double* d_rec = c->d_mom + 21 * (size_t)c->n_ranks;
  RR_HIP_TRY(hipMemcpyAsync(d_rec, rec, 21 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RR_NCCL_TRY(rccl().AllGather(d_rec, c->d_mom, 21, kNcclFloat64, c->comm, h->stream));
  RR_HIP_TRY(hipMemcpyAsync(c->h_mom, c->d_mom, 21 * (size_t)c->n_ranks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  double W = 0.0, mean[4] = {0, 0, 0, 0};
  for (int g = 0; g < c->n_ranks; ++g) {
    const double* a = c->h_mom + 21 * (size_t)g;
    W += a[0];
    for (int q = 0; q < 4; ++q) mean[q] += a[0] * a[1 + q];
  }
  for (int q = 0; q < 4; ++q) mean[q] /= W;
  if (est) std::memcpy(est, mean, sizeof mean);
  if (cov) {
    for (int q = 0; q < 16; ++q) cov[q] = 0.0;
    for (int g = 0; g < c->n_ranks; ++g) {
      const double* a = c->h_mom + 21 * (size_t)g;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) cov[4 * i + j] += a[0] * (a[5 + 4 * i + j] + (a[1 + i] - mean[i]) * (a[1 + j] - mean[j]));
    }
    for (int q = 0; q < 16; ++q) cov[q] /= W;
  }
  return RR_OK;
}

rr_status rr_pf_profile_enable(rr_pf* h, int32_t enable) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  drain_events(h);
  h->profiling = enable != 0;
  h->profile_dispatch_only = enable == 2;  // 2 = only the propagate+weight kernel, timed by its own dispatch packet
  return RR_OK;
}

rr_status rr_pf_profile_read(rr_pf* h, int32_t kernel_id, uint64_t* launches, double* total_ms) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (kernel_id < 0 || kernel_id >= RR_K_COUNT) return fail(RR_INVALID_PARAMETER, "kernel id out of range");
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  drain_events(h);
  if (launches) *launches = h->prof_launches[kernel_id];
  if (total_ms) *total_ms = h->prof_ms[kernel_id];
  return RR_OK;
}

rr_status rr_pf_profile_reset(rr_pf* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  drain_events(h);
  for (int k = 0; k < RR_K_COUNT; ++k) {
    h->prof_launches[k] = 0;
    h->prof_ms[k] = 0.0;
  }
  return RR_OK;
}

const char* rr_pf_kernel_name(int32_t kernel_id) {
  return kernel_id >= 0 && kernel_id < RR_K_COUNT ? kKernelNames[kernel_id] : "";
}

}  // extern "C"
