// resample_core.hpp -- the order-independent integer CDF machinery shared by the PF/MCL engine
// (pf_engine.hip) and the FastSLAM 1.0 engine (fs1_engine.hip): the device control block, the
// quantize-reduce / tile-scan / CDF kernels and the gate decision.
//
// Reference semantics restated here (paths under /root/reference/crates):
//   normalise + N_eff gate   rust_robotics_localization/src/particle_filter.rs:337-345,416-439
//                            rust_robotics_slam/src/fastslam1.rs:186-203,262-265
//   cumulative weights       particle_filter.rs:448-453, fastslam1.rs:213-216
// as the integer image of include/rr_pf_spec.h ("fixed-point CDF").
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>

#include <cmath>

#include "rr_common.hpp"
#include "rr_pf_spec.h"

namespace rr {

constexpr int kBlock = 256;      // propagate / gather / resolve kernels: 4 waves per workgroup
// tile kernels (quantize-reduce, tile scan, CDF, plan + mark): a tile is kTileBlock * kItems = 2048 particles
#ifndef RR_TILE_BLOCK
#define RR_TILE_BLOCK 512
#endif
constexpr int kTileBlock = RR_TILE_BLOCK;
constexpr int kItems = 2048 / RR_TILE_BLOCK;  // particles per thread in the tile kernels
constexpr int kTile = kTileBlock * kItems;  // 2048 particles per scan tile
constexpr int kTileWaves = kTileBlock / 64;
constexpr int kWaveSpan = kTile / kTileWaves;  // consecutive particles owned by one wave in k_quantize_reduce
constexpr int kScanThreads = 1024;
constexpr int kMaxObsKernarg = 96;  // observations that travel inside the launch packet
constexpr int kMomentBlocks = 1024;
constexpr int kNumMoments = 15;  // sum w, 4 first, 10 second moments

// how q_i is formed
enum ImageMode {
  kImageWeights = 0,  // q_i = floor(w_i * 2^shift)
  kImageUniform = 1,  // q_i = 1: PF/MCL fallback when sum w <= 0 (particle_filter.rs:433-438) and
                      // the image of a freshly resampled set
  kImageLast = 2      // q_i = [i is the globally last particle]: FastSLAM with all-zero weights --
                      // the walk of fastslam1.rs:224-226 runs to the last particle
};
// what an unusable weight vector (w_max == 0, inf or NaN) degenerates to
enum DegeneratePolicy { kDegenerateUniform = kImageUniform, kDegenerateLast = kImageLast };

// ---- device-resident control block: everything a later kernel needs to know about an
// earlier one's data-dependent outcome, so the host never has to look.
struct Ctl {
  int cur;              // which of the two buffer sets is live
  int weights_uniform;  // PF: 1 => every particle weighs 1/N (w[] is stale)
  int usable;           // 0 => degenerate raw weights
  int image_mode;       // ImageMode of the current integer image
  int shift;            // fixed-point shift of the current integer image
  int fired;            // last gate decision
  int pending;          // 1 => a fired resample is still only markers: particles not moved yet (lazy gather)
  int grid_timeout;     // set when a workgroup of k_quantize_plan_mark gave up waiting for another one's tile sums
  uint64_t wmax_bits;   // atomic max of the raw weights (bit pattern of a double >= 0)
  uint64_t total;       // T over all shards
  uint64_t total_local;
  uint64_t base;  // CDF base of this shard (sum of the totals of lower-ranked shards)
  uint64_t q2_hi, q2_lo;
  double wmax;  // max used for the current integer image
  double sum;   // T * 2^-shift
  double neff;
  double rho;
  rr_sys_plan plan;
  double moments[kNumMoments];
  double shift_point[4];
  uint64_t best_bits;  // FastSLAM arg-max scratch
  uint64_t best_index;
  uint64_t served_first;  // systematic plan: this shard's sources feed global slots [served_first,
  uint64_t served_count;  //   served_first + served_count) -- computed once by finalize_plan
  double est[4];          // (unused since round 3: the tiles' partial sums are added on the host when the estimate is read)
  uint64_t est_step;      // resample-step counter the in-step estimate belongs to (+1; 0 = none yet)
  double est_denom;       // ... and what the sum of the tiles' partial sums is divided by (N when the resample fired, else T)
  int obs_timeout;        // FastSLAM: a chunk's weight factor did not show up in k_fs1_observe (never seen; reported as an error)
  int est_kind;           // where the in-step estimate's partial sums are: kEstPlanTiles / kEstSlotTiles (see EstArgs)
  uint64_t n_active;      // KLD-adaptive filters: the CURRENT particle count (k_kld_count sets it; the kernels of such a filter read
                          // their n from here instead of their launch packet, so the host does not have to know it to enqueue a step)
};

// q_i of local particle i (global index gid0 + i)
__device__ inline uint64_t quantize_at(const double* __restrict__ w, uint64_t i, uint64_t n, int mode, int shift,
                                       uint64_t gid0, uint64_t n_global) {
  if (i >= n) return 0ull;
  if (mode == kImageWeights) return rr_fix_quantize(w[i], shift);
  if (mode == kImageUniform) return 1ull;
  return gid0 + i == n_global - 1 ? 1ull : 0ull;
}

struct ImageArgs {
  uint64_t n;         // particles of this shard
  uint64_t n_global;  // particles over all shards (sizes the fixed-point headroom)
  uint64_t gid0;      // global index of local particle 0
  int degenerate;     // DegeneratePolicy
  int honour_uniform_flag;  // PF: Ctl.weights_uniform forces the uniform image
  int dyn_n;          // KLD-adaptive filter: n = n_global = Ctl.n_active (launches are sized for the capacity)
};
__device__ inline ImageArgs image_args_now(ImageArgs a, const Ctl* __restrict__ ctl) {
  if (a.dyn_n) a.n = a.n_global = ctl->n_active;
  return a;
}

// ------------------------------------------------------------------------------------------
// K2: per-tile integer totals and sum of squares.  Tile = 2048 particles; each wave owns 512
// consecutive particles as 8 coalesced rows of 64.  wmax_src points at the maximum to scale by
// (Ctl.wmax_bits on one GPU, the all-reduced maximum when sharded).
__device__ inline void quantize_reduce_tile(const double* __restrict__ w, Ctl* __restrict__ ctl,
                                            const double* __restrict__ wmax_src, const ImageArgs& a_in,
                                            uint64_t* __restrict__ tile_total, uint64_t* __restrict__ tile_q2,
                                            int settle) {
  const ImageArgs a = image_args_now(a_in, ctl);
  __shared__ uint64_t s_t[kTileWaves];
  __shared__ uint64_t s_qh[kTileWaves];
  __shared__ uint64_t s_ql[kTileWaves];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double wmax = *wmax_src;
  const bool forced_uniform = a.honour_uniform_flag && ctl->weights_uniform;
  const bool usable = !forced_uniform && wmax > 0.0 && wmax < INFINITY;
  const int mode = usable ? (int)kImageWeights : (forced_uniform ? (int)kImageUniform : a.degenerate);
  const int shift = usable ? rr_fix_shift(wmax, a.n_global) : 0;
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)wv * kWaveSpan;
  uint64_t t = 0;
  u128 q2 = {0, 0};
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    uint64_t q = quantize_at(w, base + r * 64 + lane, a.n, mode, shift, a.gid0, a.n_global);
    t += q;
    u128 sq;
    rr_mul64wide(q, q, &sq.hi, &sq.lo);
    q2 = add128(q2, sq);
  }
  t = wave_sum_u64(t);
  q2 = wave_sum_u128(q2);
  if (lane == 0) {
    s_t[wv] = t;
    s_qh[wv] = q2.hi;
    s_ql[wv] = q2.lo;
  }
  __syncthreads();
  if (tid == 0) {
    uint64_t tt = 0;
    u128 qq = {0, 0};
    for (int k = 0; k < kTileWaves; ++k) {
      tt += s_t[k];
      qq = add128(qq, u128{s_qh[k], s_ql[k]});
    }
    tile_total[blockIdx.x] = tt;
    tile_q2[2 * blockIdx.x] = qq.hi;
    tile_q2[2 * blockIdx.x + 1] = qq.lo;
    if (blockIdx.x == 0) {
      if (settle && ctl->pending) {  // the propagate kernel just before us consumed the lazy gather
        ctl->cur ^= 1;
        ctl->pending = 0;
      }
      ctl->usable = usable ? 1 : 0;
      ctl->image_mode = mode;
      ctl->shift = shift;
      ctl->wmax = wmax;
    }
  }
}

static __global__ __launch_bounds__(kTileBlock) void k_quantize_reduce(const double* __restrict__ w,
                                                                  Ctl* __restrict__ ctl,
                                                                  const double* __restrict__ wmax_src,
                                                                  ImageArgs a,
                                                                  uint64_t* __restrict__ tile_total,
                                                                  uint64_t* __restrict__ tile_q2, int settle) {
  quantize_reduce_tile(w, ctl, wmax_src, a, tile_total, tile_q2, settle);
}

// ------------------------------------------------------------------------------------------
// Gate decision and systematic plan from the GLOBAL integer sums; run by one thread.
//   particle_filter.rs:337-345: fire iff N_eff < N * resample_threshold
//   monte_carlo_localization.rs:298: always;  fastslam1.rs:262-265: fire iff N_eff < NTH
// mode: 0 = decide by gate, 1 = force fire, 2 = statistics only (leave the decision alone).
struct PlanArgs {
  uint64_t n_global;
  double neff_threshold;  // absolute: N * resample_threshold, or NTH
  int gate;               // rr_resample_gate
  int mode;
  int scheme;
  double rho_override;  // NaN => Philox
  uint64_t seed;
  unsigned int rstep;
  int set_uniform_on_fire;  // PF/MCL: weights become 1/N when the resample fires
  int lazy_gather;          // 1 => leave the particles where they are (Ctl.pending); the next propagate kernel
                            // reads them through the resolved indices.  0 => a gather kernel follows: flip now
};

// the systematic plan's one uniform: the caller's override, or the first draw of the resample step's Philox stream.  It depends on
// nothing a plan kernel waits for -- the one-launch plans draw it BEFORE their waits (plan_rho_early) instead of after
__device__ inline double plan_rho(const PlanArgs& a) {
  double rho = a.rho_override;
  if (rho != rho) {
    double dummy;
    rr_uniform2(a.seed, RR_STREAM_RESAMPLE, a.rstep, 0, &rho, &dummy);
  }
  return rho;
}
__device__ inline double plan_rho_early(const PlanArgs& a) {
  // (the same in every lane: it waits in scalar registers, and the empty asm keeps it where it is -- the compiler would sink the
  // draw to its first use, behind the wait)
  const uint64_t bits = rr_d2u(plan_rho(a));
  unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)bits), hi = __builtin_amdgcn_readfirstlane((unsigned int)(bits >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return rr_u2d((uint64_t)lo | ((uint64_t)hi << 32));
}

// rho_known / image_known, shift_known: what the caller already holds of the plan's uniform and of Ctl.image_mode / Ctl.shift (the
// last arrival of a one-launch plan, which has the launch's end waiting for it: no second Philox draw, no round trip to Ctl for two
// words it wrote itself a moment ago); NaN / -1: drawn / read here.
__device__ inline void finalize_plan(Ctl* ctl, uint64_t total_global, uint64_t base, uint64_t total_local, u128 q2,
                                     const PlanArgs& a, double rho_known = NAN, int image_known = -1, int shift_known = 0) {
  ctl->total_local = total_local;
  ctl->total = total_global;
  ctl->base = base;
  ctl->q2_hi = q2.hi;
  ctl->q2_lo = q2.lo;
  const int image_mode = image_known >= 0 ? image_known : ctl->image_mode;
  double neff, sum;
  if (image_mode == kImageWeights && total_global > 0) {
    sum = rr_fix_total_to_double(total_global, image_known >= 0 ? shift_known : ctl->shift);
    neff = rr_fix_neff(total_global, q2.hi, q2.lo);
  } else if (image_mode == kImageUniform) {  // T = N, N_eff = N
    sum = 1.0;
    neff = (double)a.n_global;
  } else {  // all-zero weights: fastslam1.rs:186-193 gives N_eff = 0
    sum = 0.0;
    neff = 0.0;
  }
  ctl->sum = sum;
  ctl->neff = neff;
  if (a.mode == 2) return;
  int fire;
  if (a.mode == 1) fire = 1;
  else fire = a.gate == RR_GATE_ALWAYS ? 1 : (neff < a.neff_threshold);
  ctl->fired = fire;
  ctl->wmax_bits = 0;  // consumed: the next weight kernel accumulates a fresh maximum
  if (fire) {
    // eager: publish the resampled set NOW -- the gather that follows reads buffer set cur^1 and
    // writes set cur (no separate commit launch).  lazy: only mark it pending.
    // PF/MCL weights become uniform either way (particle_filter.rs:468)
    if (a.lazy_gather) ctl->pending = 1;
    else ctl->cur ^= 1;
    if (a.set_uniform_on_fire) ctl->weights_uniform = 1;
  }
  if (fire && a.scheme == RR_RESAMPLE_SYSTEMATIC) {
    const double rho = rho_known == rho_known ? rho_known : plan_rho(a);
    ctl->rho = rho;
    const rr_sys_plan plan = rr_sys_plan_make(rho, total_global, a.n_global);
    ctl->plan = plan;
    if (base == 0 && total_local == total_global && plan.offs < total_global) {
      // One shard holds everything: the served range is all n slots, and no 128-bit arithmetic is needed to know it -- slots_upto(0)
      // = 0, and slots_upto(T) = floor((T n - offs) / T) + 1 capped at n = n because offs = (rho 2^53) T >> 53 < T.  (This thread
      // has the end of the launch waiting for it when it is the last arrival of a one-launch plan: its workgroup finished 1.7 us
      // behind the others, 1.1 us now -- profiles/r05o_wmax_early.md.)
      ctl->served_first = 0;
      ctl->served_count = a.n_global;
    } else {
      const uint64_t first = rr_sys_slots_upto_exact(plan, total_global, base);
      ctl->served_first = first;
      ctl->served_count = rr_sys_slots_upto_exact(plan, total_global, base + total_local) - first;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3: single workgroup.  Exclusive scan of the tile totals (in place) and the local sums.  On one
// GPU (single_shard != 0) it also finalises the plan; when sharded it writes the local sums to
// shard_sums_out (total, q2_hi, q2_lo) for the all-gather and k_shard_plan finishes the job.
// the scan itself, by one workgroup of THREADS threads; thread 0 returns the totals
template <int THREADS>
__device__ inline void scan_tiles_block(uint64_t* __restrict__ tile_total, const uint64_t* __restrict__ tile_q2,
                                        uint64_t n_tiles, uint64_t* total_out, u128* q2_out) {
  __shared__ uint64_t s_w[THREADS / kWave];
  __shared__ uint64_t s_h[THREADS / kWave];
  __shared__ uint64_t s_l[THREADS / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t per = (n_tiles + THREADS - 1) / THREADS;
  const uint64_t lo = (uint64_t)tid * per;
  const uint64_t hi = lo + per < n_tiles ? lo + per : n_tiles;
  uint64_t local = 0;
  u128 q2 = {0, 0};
  for (uint64_t k = lo; k < hi; ++k) {
    local += tile_total[k];
    q2 = add128(q2, u128{tile_q2[2 * k], tile_q2[2 * k + 1]});
  }
  uint64_t incl = wave_scan_u64(local, lane);
  u128 q2w = wave_sum_u128(q2);
  if (lane == 63) s_w[wv] = incl;
  if (lane == 0) {
    s_h[wv] = q2w.hi;
    s_l[wv] = q2w.lo;
  }
  __syncthreads();
  uint64_t wave_off = 0;
  for (int k = 0; k < wv; ++k) wave_off += s_w[k];
  uint64_t run = wave_off + incl - local;  // exclusive prefix of this thread's range
  for (uint64_t k = lo; k < hi; ++k) {
    uint64_t t = tile_total[k];
    tile_total[k] = run;
    run += t;
  }
  if (tid == 0) {
    uint64_t total = 0;
    u128 qq = {0, 0};
    for (int k = 0; k < THREADS / kWave; ++k) {
      total += s_w[k];
      qq = add128(qq, u128{s_h[k], s_l[k]});
    }
    *total_out = total;
    *q2_out = qq;
  }
}

static __global__ __launch_bounds__(kScanThreads) void k_scan_tiles(uint64_t* __restrict__ tile_total,
                                                                   const uint64_t* __restrict__ tile_q2,
                                                                   Ctl* __restrict__ ctl, uint64_t n_tiles,
                                                                   int single_shard, PlanArgs a,
                                                                   uint64_t* __restrict__ shard_sums_out) {
  uint64_t total = 0;
  u128 qq = {0, 0};
  scan_tiles_block<kScanThreads>(tile_total, tile_q2, n_tiles, &total, &qq);
  if (threadIdx.x == 0) {
    if (single_shard) {
      finalize_plan(ctl, total, 0, total, qq, a);
    } else {
      ctl->total_local = total;
      shard_sums_out[0] = total;
      shard_sums_out[1] = qq.hi;
      shard_sums_out[2] = qq.lo;
    }
  }
}

// sharded: combine every shard's (total, q2_hi, q2_lo) in rank order (integer adds: any order
// gives the same bits) and finalise the plan.  One thread.
static __global__ void k_shard_plan(Ctl* __restrict__ ctl, const uint64_t* __restrict__ all_sums, int n_shards,
                                    int rank, PlanArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t total = 0, base = 0;
  u128 qq = {0, 0};
  for (int g = 0; g < n_shards; ++g) {
    if (g == rank) base = total;
    total += all_sums[3 * g];
    qq = add128(qq, u128{all_sums[3 * g + 1], all_sums[3 * g + 2]});
  }
  finalize_plan(ctl, total, base, all_sums[3 * rank], qq, a);
}

// ------------------------------------------------------------------------------------------
// Tile-local scan, blocked layout: thread t of the workgroup owns the 8 CONSECUTIVE particles
// [tile*2048 + 8t, +8).  Each thread prefix-sums its own 8 integer weights serially, one wave64
// scan of the thread totals plus 4 wave totals through LDS gives every thread its exclusive
// offset inside the tile.  (A row-per-wave layout needs one 6-step shuffle chain per row, and
// those chains -- ds_bpermute latency -- were the whole critical path of the resample kernels.)
struct TileScan {
  uint64_t q[kItems];    // integer weights of this thread's particles
  uint64_t c[kItems];    // inclusive prefix inside the thread
  uint64_t thread_off;   // exclusive prefix of this thread inside the tile
};

__device__ inline TileScan tile_scan(const double* __restrict__ w, const ImageArgs& a, int mode, int shift,
                                     uint64_t tile, uint64_t* s_w /* [kTileBlock / kWave] */) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  TileScan t;
  const uint64_t i0 = tile * kTile + (uint64_t)tid * kItems;
  uint64_t run = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    t.q[j] = quantize_at(w, i0 + j, a.n, mode, shift, a.gid0, a.n_global);
    run += t.q[j];
    t.c[j] = run;
  }
  const uint64_t incl = wave_scan_u64(run, lane);
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  uint64_t off = incl - run;
  for (int k = 0; k < wv; ++k) off += s_w[k];
  t.thread_off = off;
  return t;
}

// the sums every workgroup of a fused plan kernel needs: its tile prefix, the grand total, sum q^2
struct TileSums {
  uint64_t pre, tot;
  u128 q2;
};

__device__ inline TileSums tile_sums(const uint64_t* __restrict__ tile_total, const uint64_t* __restrict__ tile_q2,
                                     uint64_t n_tiles, uint64_t* s4 /* [4][kTileBlock / kWave] */) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int W = kTileBlock / kWave;
  uint64_t pre = 0, tot = 0;
  u128 q2 = {0, 0};
  for (uint64_t k = tid; k < n_tiles; k += kTileBlock) {
    const uint64_t t = tile_total[k];
    tot += t;
    if (k < blockIdx.x) pre += t;
    q2 = add128(q2, u128{tile_q2[2 * k], tile_q2[2 * k + 1]});
  }
  pre = wave_sum_u64(pre);
  tot = wave_sum_u64(tot);
  q2 = wave_sum_u128(q2);
  if (lane == 0) {
    s4[wv] = pre;
    s4[W + wv] = tot;
    s4[2 * W + wv] = q2.hi;
    s4[3 * W + wv] = q2.lo;
  }
  __syncthreads();
  TileSums r{0, 0, {0, 0}};
  for (int k = 0; k < W; ++k) {
    r.pre += s4[k];
    r.tot += s4[W + k];
    r.q2 = add128(r.q2, u128{s4[2 * W + k], s4[3 * W + k]});
  }
  return r;
}

__device__ inline int gate_decision(int image_mode, const TileSums& ts, const PlanArgs& pa) {
  double neff;
  if (image_mode == kImageWeights && ts.tot > 0) neff = rr_fix_neff(ts.tot, ts.q2.hi, ts.q2.lo);
  else if (image_mode == kImageUniform) neff = (double)pa.n_global;
  else neff = 0.0;
  return pa.mode == 1 ? 1 : (pa.mode == 2 ? 0 : (pa.gate == RR_GATE_ALWAYS ? 1 : (neff < pa.neff_threshold)));
}

// ------------------------------------------------------------------------------------------
// K4: inclusive integer CDF of this shard (multinomial resampling and FastSLAM read it back):
// cdf[i] = base + tile_offset + within-tile scan.  Reads w (8 B), writes cdf (8 B).
// coarse[k] = cdf[min((k+1) << coarse_log2, n) - 1]: every 2^coarse_log2-th CDF entry, the table the
// multinomial gather stages in LDS (nullptr: not wanted)
__device__ inline void store_cdf(const TileScan& t, uint64_t off, uint64_t i0, uint64_t n, uint64_t* __restrict__ cdf,
                                 uint64_t* __restrict__ coarse, int coarse_log2) {
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const uint64_t i = i0 + j;
    if (i < n) {
      const uint64_t c = off + t.c[j];
      cdf[i] = c;
      if (coarse && ((((i + 1) & ((1ull << coarse_log2) - 1)) == 0) || i == n - 1)) coarse[i >> coarse_log2] = c;
    }
  }
}

// ---- guide table of the multinomial search (index search: bucket the TARGET space, not the index space)
// The draws' targets are uniform on [1, T].  Cut [0, T] into buckets of 2^s targets, s the smallest shift that leaves at
// most 2^guide_log2 buckets, and keep guide[k] = first source i with C_i >= k * 2^s: a draw with target t then lies in
// [guide[t >> s], guide[(t >> s) + 1]] -- every bucket is hit with the same probability and holds ~n / #buckets CDF
// entries on average, a heavy particle spans many buckets and is found WITHOUT reading the CDF at all.  Source i is the
// entry of the bucket run [H_{i-1}, H_i), H_i = (C_i >> s) + 1: the same marker / carry / running-maximum machinery as the
// systematic resample's slot runs (below), with a trivial H.  Bucket 0 always starts at source 0 (carry[0] = 1, set once).
constexpr int kGuideTile = 2 * 256;  // buckets per resolve workgroup (= kResolveSlots, asserted below)

__host__ __device__ inline int guide_shift(uint64_t total, int guide_log2) {
  const int bits = total ? 64 - __builtin_clzll((unsigned long long)total) : 0;  // bit length of T
  return bits > guide_log2 ? bits - guide_log2 : 0;
}

__device__ inline void mark_guide(const TileScan& t, uint64_t off, uint64_t i0, uint64_t n, int s,
                                  unsigned int* __restrict__ markers, unsigned int* __restrict__ carry) {
  uint64_t h_run = (off >> s) + 1;  // buckets settled by the sources before this thread's first
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    if (t.q[j] == 0 || i0 + j >= n) continue;
    const uint64_t h = ((off + t.c[j]) >> s) + 1;
    if (h > h_run) {
      markers[h_run] = (unsigned int)(i0 + j + 1);
      for (uint64_t b = (h_run + kGuideTile - 1) / kGuideTile; b * kGuideTile < h; ++b) carry[b] = (unsigned int)(i0 + j + 1);
      h_run = h;
    }
  }
}

static __global__ __launch_bounds__(kTileBlock) void k_cdf(const double* __restrict__ w, const Ctl* __restrict__ ctl,
                                                      ImageArgs a, const uint64_t* __restrict__ tile_offset,
                                                      uint64_t* __restrict__ cdf, uint64_t* __restrict__ coarse,
                                                      int coarse_log2, unsigned int* __restrict__ guide_markers = nullptr,
                                                      unsigned int* __restrict__ guide_carry = nullptr, int guide_log2 = 0) {
  if (!ctl->fired) return;
  a = image_args_now(a, ctl);
  __shared__ uint64_t s_w[kTileBlock / kWave];
  const TileScan t = tile_scan(w, a, ctl->image_mode, ctl->shift, blockIdx.x, s_w);
  const uint64_t off = ctl->base + tile_offset[blockIdx.x] + t.thread_off;
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
  store_cdf(t, off, i0, a.n, cdf, coarse, coarse_log2);
  // a shard's guide table (multinomial shards over the peer-to-peer transport) buckets the shard's OWN interval of the target space:
  // targets relative to Ctl.base, bucket width from Ctl.total_local -- the single-shard table of k_plan_cdf with base 0
  if (guide_markers)
    mark_guide(t, tile_offset[blockIdx.x] + t.thread_off, i0, a.n, guide_shift(ctl->total_local, guide_log2), guide_markers, guide_carry);
}

// K3+K4 fused (single shard, n_tiles <= kFusedMaxTiles): every workgroup re-derives its tile
// offset and the grand totals from the (L2-resident) tile totals -- integer sums, so every
// workgroup gets the same bits -- takes the gate decision itself, and writes its slice of the
// CDF.  Workgroup 0 publishes the plan.  Saves one launch and one dependent single-block kernel.
constexpr int kFusedMaxTiles = 4096;

// ------------------------------------------------------------------------------------------
// Systematic resampling WITHOUT any search.  Targets are monotone in the slot index, so source j
// (inclusive CDF C_j) feeds the contiguous slot run [H_{j-1}, H_j), H_j = rr_sys_slots_upto(C_j).
// The plan kernel therefore never materialises the CDF: it computes H per source in registers and
//   * writes the marker  markers[H_{j-1} - slot_base] = j + 1   for every source with offspring,
//   * writes             carry[b] = j + 1                        for every slot-tile boundary
//     b * kResolveSlots that falls inside the run (the source of the first slot of slot-tile b).
// The resolve step (engine specific: it also moves the particles) turns markers into indices with
// a running maximum per slot tile seeded by carry[tile] -- markers are increasing in the slot
// index -- and clears them for the next step.  Everything is O(N), balanced on the output side,
// and free of dependent HBM probe chains (the old lower_bound cost 20 of them per slot).
#ifndef RR_RESOLVE_ROWS
#define RR_RESOLVE_ROWS 2
#endif
constexpr int kResolveRows = RR_RESOLVE_ROWS;
constexpr int kResolveSlots = kResolveRows * kBlock;  // slots per resolve workgroup
static_assert(kGuideTile == kResolveSlots, "the guide table is resolved by resolve_tile");

// this thread's 8 consecutive sources: exclusive CDF prefix `off`, inclusive prefixes off + c[j]
// offspring (may be null): number of output slots each of this thread's sources feeds
__device__ inline void mark_sources(const TileScan& t, uint64_t off, uint64_t i0, uint64_t n, const rr_sys_plan plan,
                                    uint64_t total, uint64_t slot_base, unsigned int* __restrict__ markers,
                                    unsigned int* __restrict__ carry, unsigned int* offspring = nullptr) {
  const rr_sys_inv inv = rr_sys_inv_make(plan, total);
  uint64_t h_run = rr_sys_slots_upto(plan, inv, total, off);  // H of the source just before this thread's first
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    if (offspring) offspring[j] = 0;
    if (t.q[j] == 0 || i0 + j >= n) continue;  // zero-weight sources feed no slot: H_j == H_{j-1}
    const uint64_t h = rr_sys_slots_upto(plan, inv, total, off + t.c[j]);
    if (h > h_run) {
      if (offspring) offspring[j] = (unsigned int)(h - h_run);
      const uint64_t lo = h_run - slot_base, hi = h - slot_base;
      markers[lo] = (unsigned int)(i0 + j + 1);
      for (uint64_t b = (lo + kResolveSlots - 1) / kResolveSlots; b * kResolveSlots < hi; ++b)
        carry[b] = (unsigned int)(i0 + j + 1);
      h_run = h;
    }
  }
}

// The estimate the reference's try_step returns (particle_filter.rs:382-396 evaluated at :343 / :332,
// Q15), produced INSIDE the plan kernel: the mean of the particle set as the step leaves it is
//   fired:      sum_j offspring_j * p_j / N     (the resampled set has uniform weights, :468)
//   not fired:  sum_j q_j * p_j / T             (q = the integer image of the weights, T = sum q)
// and this kernel already holds offspring_j / q_j of its 8 sources per thread.  Per-workgroup partial
// sums go to `partials`; the last workgroup to arrive (ticket) adds them in workgroup order -- a
// fixed order, so the result is reproducible -- and publishes Ctl.est.  Lazy-gather plans only
// (Ctl.cur is not touched by this kernel then).
// "Am I the last workgroup of this launch to get here?"  One counter for everybody means one same-address atomic per
// workgroup, and those are carried out one after the other at the memory side (7.5 ns each, rr::atomic_max_u64): two
// levels instead -- kTicketGroups counters on separate cache lines, the last arrival of each group then takes a ticket
// of the final counter.  Call from ONE thread after the workgroup's results are stored; the caller provides the
// release before and the acquire after (as for a single counter).  The overall last arrival zeroes the counters.
constexpr int kTicketGroups = 8;
constexpr int kTicketStride = 32;  // words between counters: 128 B
constexpr int kTicketWords = (kTicketGroups + 1) * kTicketStride;
__device__ inline bool last_arrival(unsigned int* ticket, unsigned int block, unsigned int n_blocks, bool fence = true) {
  const unsigned int g = block % kTicketGroups;
  const unsigned int in_group = (n_blocks - g + kTicketGroups - 1) / kTicketGroups;  // blocks b < n_blocks with b % G == g
  if (atomicAdd(&ticket[g * kTicketStride], 1u) != in_group - 1) return false;
  const unsigned int groups = n_blocks < (unsigned int)kTicketGroups ? n_blocks : (unsigned int)kTicketGroups;
  if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // the group's stores before the final ticket
  if (atomicAdd(&ticket[kTicketGroups * kTicketStride], 1u) != groups - 1) return false;
  for (int k = 0; k <= kTicketGroups; ++k) ticket[k * kTicketStride] = 0;
  return true;
}

// k_quantize_reduce that also leaves this shard's sums (T, sum q^2) for the all-gather: the last workgroup to arrive adds the
// tile totals -- no separate single-workgroup k_scan_tiles launch (7 us at 1e6 particles for 12 KB of input); the tile
// PREFIXES are formed by the consumer (k_mark_plan: every workgroup adds the totals of the tiles before its own).
static __global__ __launch_bounds__(kTileBlock) void k_quantize_reduce_sums(const double* __restrict__ w, Ctl* __restrict__ ctl,
                                                                       const double* __restrict__ wmax_src, ImageArgs a,
                                                                       uint64_t* __restrict__ tile_total, uint64_t* __restrict__ tile_q2,
                                                                       int settle, uint64_t n_tiles, unsigned int* __restrict__ ticket,
                                                                       uint64_t* __restrict__ shard_sums_out) {
  constexpr int W = kTileBlock / kWave;
  __shared__ uint64_t s3[3 * W];
  __shared__ int s_last;
  quantize_reduce_tile(w, ctl, wmax_src, a, tile_total, tile_q2, settle);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = last_arrival(ticket, blockIdx.x, gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  uint64_t t = 0;
  u128 q2 = {0, 0};
  for (uint64_t k = tid; k < n_tiles; k += kTileBlock) {
    t += __builtin_nontemporal_load(&tile_total[k]);
    q2 = add128(q2, u128{__builtin_nontemporal_load(&tile_q2[2 * k]), __builtin_nontemporal_load(&tile_q2[2 * k + 1])});
  }
  t = wave_sum_u64(t);
  q2 = wave_sum_u128(q2);
  if (lane == 0) {
    s3[wv] = t;
    s3[W + wv] = q2.hi;
    s3[2 * W + wv] = q2.lo;
  }
  __syncthreads();
  if (tid == 0) {
    uint64_t tt = 0;
    u128 qq = {0, 0};
    for (int k = 0; k < W; ++k) {
      tt += s3[k];
      qq = add128(qq, u128{s3[W + k], s3[2 * W + k]});
    }
    ctl->total_local = tt;
    shard_sums_out[0] = tt;
    shard_sums_out[1] = qq.hi;
    shard_sums_out[2] = qq.lo;
  }
}

// The in-step estimate (the mean try_step returns, particle_filter.rs:488-497) in two forms:
//   kEstInPlan   the plan kernel adds offspring_j * field_j (fired) or q_j * field_j (gate shut) over its tile: everything is
//                known when the plan kernel ends -- the synchronous rr_pf_step, which reads the value back at once;
//   kEstDeferred fired: nothing here.  The mean of the resampled set is summed by whoever MOVES the particles -- the next
//                step's k_step_lazy, which holds every slot's source fields in registers before it propagates them, or the
//                gather of an accessor that comes first (k_est_slots afterwards) -- per slot tile, in one fixed order.  The plan
//                kernel then neither reads the particle fields (32 MB at 1e6 particles) nor carries the offspring counts to the
//                end; gate shut: as kEstInPlan (the fields are requested once the decision is known).
// Ctl.est_kind says which array the host adds up: kEstPlanTiles -> partials[n_tiles][4], kEstSlotTiles -> the slot tiles' array.
enum : int { kEstOff = 0, kEstInPlan = 1, kEstDeferred = 2 };
enum : int { kEstPlanTiles = 0, kEstSlotTiles = 1 };
struct EstArgs {
  const double* field[2][4];  // x, y, yaw, v of both buffer sets
  double* partials;           // [n_tiles][4]
  unsigned int* ticket;       // (unused since round 3)
  int want;                   // kEstOff / kEstInPlan / kEstDeferred
};

// The particle fields of this thread's kItems CONSECUTIVE sources (the blocked layout of tile_scan: a lane reads 8 * kItems
// contiguous bytes per field, a wave 64 of those back to back -- every fetched line is used in full).  Issued at the very
// start of the plan kernels, before the tile sums are known: the loads depend on nothing the kernel computes, so they are
// in flight while the workgroup waits for the other workgroups' sums and cost the estimate no time of their own.  (Round 2
// read them row-wise AFTER the marks were written and moved the coefficients through LDS: +12 us per step.)
struct EstFields {
  double f[4][kItems];
};
__device__ inline void est_prefetch(const EstArgs& ea, int cur, uint64_t i0, uint64_t n, EstFields& e) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double* __restrict__ p = ea.field[cur][k] + i0;
    if ((kItems % 2) == 0 && i0 + kItems <= n && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {  // uniform except in the last tile
#pragma unroll
      for (int j = 0; j < kItems; j += 2) {
        const double2 v = *reinterpret_cast<const double2*>(p + j);
        e.f[k][j] = v.x;
        e.f[k][j + (kItems > 1 ? 1 : 0)] = v.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < kItems; ++j) e.f[k][j] = (i0 + j < n) ? p[j] : 0.0;
    }
  }
}

// per-workgroup part: this tile's four partial sums -> partials[tile] (thread 0 stores them; nothing is fenced here)
__device__ inline void est_tile_partial(const EstArgs& ea, const TileScan& t, const unsigned int* offspring, int fire, uint64_t i0,
                                        uint64_t n, const EstFields& e, uint64_t tile) {
  __shared__ double s_acc[kTileBlock / kWave][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const double c = (i0 + j < n) ? (fire ? (double)offspring[j] : (double)t.q[j]) : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = rr_fma(c, e.f[k][j], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) s_acc[wv][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = 0.0;
      for (int q = 0; q < kTileBlock / kWave; ++q) v += s_acc[q][k];
      ea.partials[tile * 4 + k] = v;
    }
  }
  __syncthreads();  // (s_acc may be reused by the next tile of the serial plan)
}

// Nothing in the kernel waits for the tiles' partial sums: whoever reads the estimate (rr_pf_last_step_estimate, after the
// stream has drained) adds the n_tiles x 4 numbers on the host, in tile order, and divides by Ctl.est_denom.  Round 2 had
// the last workgroup to arrive do that inside the kernel: a fence, a ticket and a dependent read-back at the tail of
// every step for a number that is read once in a while.  One thread of the launch records which step the sums belong to.
__device__ inline void est_publish(Ctl* __restrict__ ctl, double denom, unsigned int rstep, int kind = kEstPlanTiles) {
  ctl->est_denom = denom;
  ctl->est_step = (uint64_t)rstep + 1;
  ctl->est_kind = kind;
}

// The slot-tile form: a workgroup of kBlock threads has the four fields of its kResolveSlots slots (slot = tile base +
// r * kBlock + tid) in registers.  Per thread the rows in order (est_rows_sum, right after the loads: four doubles live on instead
// of the fields), per wave ONE four-value reduction at the kernel's end (est_wave_store, wave_sum4) -> partials[tile][wave][0..3];
// nothing crosses a wave,
// so no barrier stands between a workgroup's loads and its arithmetic.  Whoever reads the estimate adds the waves' numbers in
// (tile, wave) order.  ONE definition for every kernel that produces these sums, so that they agree to the bit whoever ran.
template <int ROWS>
__device__ inline void est_rows_sum(const double (&f)[4][ROWS], double (&acc)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[k] = 0.0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[k] += f[k][r];
  }
}
template <int BLOCK>
__device__ inline void est_wave_store(const double (&acc)[4], double* __restrict__ partials, uint64_t tile) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double v = wave_sum4(acc);
  if (lane >= 12 && lane < 16) partials[(tile * (BLOCK / kWave) + wv) * 4 + wave_sum4_field(lane)] = v;
}

// multinomial: gate + CDF (+ the guide markers over the target space)
static __global__ __launch_bounds__(kTileBlock) void k_plan_cdf(const double* __restrict__ w, Ctl* __restrict__ ctl,
                                                           ImageArgs a, const uint64_t* __restrict__ tile_total,
                                                           const uint64_t* __restrict__ tile_q2, uint64_t n_tiles,
                                                           PlanArgs pa, uint64_t* __restrict__ cdf,
                                                           uint64_t* __restrict__ coarse, int coarse_log2,
                                                           unsigned int* __restrict__ guide_markers,
                                                           unsigned int* __restrict__ guide_carry, int guide_log2,
                                                           EstArgs ea = EstArgs{}) {
  __shared__ uint64_t s4[4 * (kTileBlock / kWave)];
  __shared__ uint64_t s_w[kTileBlock / kWave];
  a = image_args_now(a, ctl);  // (Ctl.n_active is only rewritten by a later kernel)
  if (a.dyn_n) pa.n_global = a.n_global;
  const TileSums ts = tile_sums(tile_total, tile_q2, n_tiles, s4);
  const int mode = ctl->image_mode;  // written by k_quantize_reduce; nothing below reads what block 0 writes
  const int shift = ctl->shift;
  const int fire = gate_decision(mode, ts, pa);
  if (blockIdx.x == 0 && threadIdx.x == 0) finalize_plan(ctl, ts.tot, 0, ts.tot, ts.q2, pa);
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
  // the in-step estimate of the multinomial scheme is always the deferred form (EstArgs): the offspring counts of iid draws are
  // not known before the draws are searched, which the kernel that moves the particles does
  if (ea.want && blockIdx.x == 0 && threadIdx.x == 0)
    est_publish(ctl, fire ? (double)pa.n_global : (double)ts.tot, pa.rstep, fire ? kEstSlotTiles : kEstPlanTiles);
  if (!fire) {
    if (ea.want) {  // gate shut: the weighted mean over the integer image, as the systematic plan forms it
      EstFields ef;
      est_prefetch(ea, ctl->cur, i0, a.n, ef);
      const TileScan t = tile_scan(w, a, mode, shift, blockIdx.x, s_w);
      est_tile_partial(ea, t, nullptr, 0, i0, a.n, ef, blockIdx.x);
    }
    return;
  }
  const TileScan t = tile_scan(w, a, mode, shift, blockIdx.x, s_w);
  store_cdf(t, ts.pre + t.thread_off, i0, a.n, cdf, coarse, coarse_log2);
  if (guide_markers) mark_guide(t, ts.pre + t.thread_off, i0, a.n, guide_shift(ts.tot, guide_log2), guide_markers, guide_carry);
}

// fused plan + mark (single shard, systematic, n_tiles <= kFusedMaxTiles)
static __global__ __launch_bounds__(kTileBlock) void k_plan_mark(const double* __restrict__ w, Ctl* __restrict__ ctl,
                                                            ImageArgs a, const uint64_t* __restrict__ tile_total,
                                                            const uint64_t* __restrict__ tile_q2, uint64_t n_tiles,
                                                            PlanArgs pa, unsigned int* __restrict__ markers,
                                                            unsigned int* __restrict__ carry, EstArgs ea) {
  __shared__ uint64_t s4[4 * (kTileBlock / kWave)];
  __shared__ uint64_t s_w[kTileBlock / kWave];
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
  EstFields ef;
  if (ea.want == kEstInPlan) est_prefetch(ea, ctl->cur, i0, a.n, ef);  // (lazy plans never flip Ctl.cur in this kernel)
  const TileSums ts = tile_sums(tile_total, tile_q2, n_tiles, s4);
  const int mode = ctl->image_mode;
  const int shift = ctl->shift;
  const int fire = gate_decision(mode, ts, pa);
  // every workgroup derives the same plan (rho comes from the arguments or the Philox stream)
  double rho = pa.rho_override;
  if (rho != rho) {
    double dummy;
    rr_uniform2(pa.seed, RR_STREAM_RESAMPLE, pa.rstep, 0, &rho, &dummy);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) finalize_plan(ctl, ts.tot, 0, ts.tot, ts.q2, pa);
  if (!fire && !ea.want) return;
  const bool est_here = ea.want == kEstInPlan || (ea.want == kEstDeferred && !fire);
  if (ea.want == kEstDeferred && !fire) est_prefetch(ea, ctl->cur, i0, a.n, ef);
  const TileScan t = tile_scan(w, a, mode, shift, blockIdx.x, s_w);
  unsigned int offspring[kItems];
  if (fire) {
    const rr_sys_plan plan = rr_sys_plan_make(rho, ts.tot, pa.n_global);
    mark_sources(t, ts.pre + t.thread_off, i0, a.n, plan, ts.tot, 0, markers, carry, est_here ? offspring : nullptr);
  }
  if (est_here) est_tile_partial(ea, t, offspring, fire, i0, a.n, ef, blockIdx.x);
  if (ea.want && blockIdx.x == 0 && threadIdx.x == 0)
    est_publish(ctl, fire ? (double)pa.n_global : (double)ts.tot, pa.rstep, est_here ? kEstPlanTiles : kEstSlotTiles);
}

// ------------------------------------------------------------------------------------------
// K2 and the fused plan in ONE launch (single shard, systematic, n_tiles <= kTileBlock and every workgroup resident at
// once -- the host checks the occupancy).  A kernel boundary costs ~6 us here (drain, the ~4.5 us floor of any launch,
// the gap), two thirds of what either kernel needs on its own, and the only thing that crosses it is 24 B per tile.
// So: each workgroup quantises its tile (the integer weights stay in registers), stores its tile sums in its own
// record and takes an arrival ticket (two-level, last_arrival); the LAST workgroup to arrive scans the records --
// exclusive prefix per tile, grand totals -- writes them back and raises the launch's flag; every workgroup polls
// that one flag with one thread, then reads its prefix and the totals.  All of this traffic uses device-scope
// (cache-bypassing) loads and stores, so no fence -- no L2 write-back or invalidate -- is needed inside the kernel.
// Timeline at 1e6 particles (489 workgroups, s_memrealtime stamps, us after the first workgroup starts): records stored
// 4.4 - 6.4, last ticket 7.1, flag raised 10.7, sums known everywhere 11.7 - 12.0, markers written 13.7 - 15.6 -- every
// hop through device-scope memory costs ~1.2 us, which is why this is only ~2 us better than two launches and why
// the same trick is not worth it between k_step_lazy and this kernel.
// Reads of Ctl that the settle / finalize writes of workgroup 0 could race with happen before the ticket is taken;
// workgroup 0 writes after it has seen the flag, i.e. after every workgroup has arrived.
constexpr int kRecWords = 4;  // a tile's record: total, q2_hi, q2_lo (+ 1 spare word)
// rec layout: [n_tiles][kRecWords] records, then kRecWords words {grand total, q2_hi, q2_lo, state}
__device__ inline uint64_t ld_dev(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_dev(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_dev(const double* p) { return rr_u2d(ld_dev(reinterpret_cast<const uint64_t*>(p))); }
__device__ inline void st_dev(double* p, double v) { st_dev(reinterpret_cast<uint64_t*>(p), rr_d2u(v)); }

// The launch's state word: 2 * epoch     = RAISED: every workgroup has arrived, the sums are out, go on;
//                          2 * epoch + 1 = GIVEN UP: a waiting workgroup ran out of patience before that.
// Waiting in a kernel for OTHER workgroups of the same kernel only works if all of them are on the device at once.  The
// host launches this kernel when an idle device holds the whole grid (occupancy query, one such kernel per process at a
// time: rr::spin_permit; hipLaunchCooperativeKernel makes the same check and costs 17 us more per launch on this runtime --
// measured, round 4: step 49.8 -> 67.8 us) -- but another PROCESS on the same GPU can keep some workgroups off the CUs while the ones that
// run hold their slots and spin (two such kernels of two processes can do that to each other).  So the wait is bounded
// (`giveup_ticks` of the 100 MHz wall clock, default 2 ms -- a healthy launch waits ~5 us), and when it runs out the
// launch DEGRADES instead of failing: whoever is first moves the state word from an older epoch to RAISED (the last
// arrival) or to GIVEN UP (an impatient waiter) with a compare-and-swap, so all workgroups of the launch see ONE outcome.
// Given up: every waiting workgroup leaves without writing anything -- its record and its ticket are already in -- which
// frees its slots for the workgroups that had not started; each of those posts its record and leaves too; the LAST
// arrival, whoever that is, then does the marking (weights, estimate partials) of EVERY tile itself, one after the other:
// the same integer sums, the same markers, the same bits, ~1-2 ms instead of ~12 us.  It also raises Ctl.grid_timeout,
// which (a) makes every later launch of the kernel skip the wait outright (serial plan at once, no 2 ms each) until (b) the
// host sees it at its next read of Ctl and moves the handle to the multi-launch plan for good.  Nothing is reported as an
// error: the results are those of the undisturbed filter.
// ---- the OUTPUT side of the hand-over (round 5): what the last arrival hands back -- every tile's exclusive prefix and the grand
// totals -- travels as self-vouching pairs {word, epoch ^ mix(word)}.  The last arrival stores them and is done with them: no wait
// for their acknowledgement before a flag; a waiting workgroup polls ITS OWN prefix pair and the three totals (four pairs, one
// round trip per look) and has the numbers the moment they fit -- no flag to see first and no second round trip for what it
// announces.  mix is one-to-one, so a pair of which only half has arrived does not fit, nor does anything an earlier launch
// left (other epoch).  What that buys: the last arrival stores the pairs and moves the state word at once, without waiting
// for the pairs' acknowledgement first; a waiter that sees RAISED reads its pairs and looks again should one be late.
// (Waiters polling the PAIRS instead of the state word -- no flag hop at all -- was built and measured, round 5: +1 us per step with
// shared totals, +4.4 us with one 64-byte record per tile: the polling traffic of 489 workgroups stands in the way of the very
// stores it waits for.  profiles/r05h_plan_handover_ab.md)
struct alignas(16) TagPair {
  uint64_t bits, tag;
};
__device__ inline uint64_t pair_tag(uint64_t epoch, uint64_t word) { return epoch ^ (word * 0x9E3779B97F4A7C15ull); }
__device__ inline void put_pair(TagPair* p, uint64_t word, uint64_t epoch) {
  st_dev(&p->bits, word);
  st_dev(&p->tag, pair_tag(epoch, word));
}
// rec layout in words: [kTileBlock + 1][kRecWords] records and head | 16 spare | (instrumented build) kTileBlock x kTimelineWords
// stamps | pairs: kTileBlock prefixes, then kPlanTotals shared words (totals; the sharded plan's base, gate decision, global maximum)
constexpr int kTimelineWords = 8;
constexpr int kPlanTotals = 8;
constexpr size_t kPlanPairsWord = (size_t)(kTileBlock + 1) * kRecWords + 16 + (size_t)kTileBlock * kTimelineWords;
constexpr size_t kPlanRecBytes = kPlanPairsWord * sizeof(uint64_t) + (size_t)(kTileBlock + kPlanTotals) * sizeof(TagPair);
static_assert(kPlanPairsWord % 2 == 0, "the pairs are 16-byte aligned");
__device__ inline TagPair* plan_pairs(uint64_t* rec) { return reinterpret_cast<TagPair*>(rec + kPlanPairsWord); }
// up to four pairs in one round trip; true when every one of them vouches for `epoch` (word[k] then holds pair k's word)
template <int N>
__device__ inline bool take_pairs(const TagPair* const (&p)[N], uint64_t epoch, uint64_t (&word)[N]) {
  uint64_t tag[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    word[k] = ld_dev(&p[k]->bits);
    tag[k] = ld_dev(&p[k]->tag);
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) ok &= tag[k] == pair_tag(epoch, word[k]);
  return ok;
}

constexpr uint64_t kPlanGiveupTicksDefault = 200000;  // 2 ms
// Instrumented build (make -C csrc timeline: -DRR_PLAN_TIMELINE, a separate .so for tools/plan_timeline.py): thread 0 of
// every workgroup stamps the 100 MHz wall clock at the kernel's stations into the words behind the records:
//   0 start, 1 record stored, 2 ticket taken, 3 sums seen (the pairs fit) / decided, 4 = 3, 5 markers written (6 estimate partial)
#if defined(RR_PLAN_TIMELINE)
#define RR_TL(K_) do { if (threadIdx.x == 0) tl[(uint64_t)blockIdx.x * kTimelineWords + (K_)] = wall_clock64(); } while (0)
#else
#define RR_TL(K_) do { } while (0)
#endif
__device__ inline uint64_t plan_state_decide(uint64_t* state, uint64_t epoch, uint64_t want_bit, bool guess_previous = false) {
  // returns the state of THIS epoch, moving the word there if nobody has yet.  guess_previous (the last arrival, which is on
  // everybody's critical path): try the compare-and-swap straight away against what the previous launch most likely left
  // (RAISED, epoch - 1) instead of loading the word first -- one memory round trip instead of two; a wrong guess costs
  // nothing but the retry, because the failed swap returns the word.
  uint64_t seen = guess_previous ? 2 * (epoch - 1) : ld_dev(state);
  for (;;) {
    if ((seen >> 1) == epoch) return seen;
    const uint64_t mine = 2 * epoch + want_bit;
    const uint64_t was = (uint64_t)atomicCAS((ull*)state, (ull)seen, (ull)mine);
    if (was == seen) return mine;
    seen = was;
  }
}

// integer image + blocked scan of tile `tile` (phase A of the plan kernels; tile_scan with the FastSLAM weights kept)
template <bool FS_WEIGHTS>
__device__ inline void plan_image_tile(typename std::conditional<FS_WEIGHTS, double*, const double*>::type w, const ImageArgs& a,
                                       int mode, int shift, uint64_t tile, uint64_t* s_w, uint64_t* s4, TileScan& t,
                                       double (&w_in)[FS_WEIGHTS ? kItems : 1], uint64_t* tile_total, u128* tile_q2) {
  constexpr int W = kTileBlock / kWave;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t i0 = tile * kTile + (uint64_t)tid * kItems;
  uint64_t run = 0;
  u128 q2 = {0, 0};
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    if (FS_WEIGHTS) {
      w_in[j] = i0 + j < a.n ? w[i0 + j] : 0.0;
      t.q[j] = i0 + j >= a.n ? 0ull
                             : (mode == kImageWeights ? rr_fix_quantize(w_in[j], shift)
                                                      : (mode == kImageUniform ? 1ull : (a.gid0 + i0 + j == a.n_global - 1 ? 1ull : 0ull)));
    } else {
      t.q[j] = quantize_at(w, i0 + j, a.n, mode, shift, a.gid0, a.n_global);
    }
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
    *tile_total = tt;
    *tile_q2 = qq;
  }
}

// phase B of the plan for one tile whose image is in `t` (k_plan_mark's second half): FastSLAM rewrites the weights
// (normalised, or 1/n + markers when the gate fires), PF/MCL marks and adds the tile's share of the estimate
template <bool FS_WEIGHTS, bool DEFER = false>
__device__ inline void plan_apply_tile(typename std::conditional<FS_WEIGHTS, double*, const double*>::type w, const ImageArgs& a,
                                       int mode, int shift, const TileScan& t, const double (&w_in)[FS_WEIGHTS ? kItems : 1],
                                       const TileSums& ts, int fire, double rho, const PlanArgs& pa, uint64_t tile,
                                       unsigned int* __restrict__ markers, unsigned int* __restrict__ carry, const EstArgs& ea,
                                       const EstFields& ef, unsigned int (&offspring)[kItems]) {
  const uint64_t i0 = tile * kTile + (uint64_t)threadIdx.x * kItems;
  if constexpr (FS_WEIGHTS) {
    if (!fire) {
      if (mode != kImageWeights) return;  // all-zero weights stay untouched (fastslam1.rs:198-202)
      const double sum = rr_fix_total_to_double(ts.tot, shift);
#pragma unroll
      for (int j = 0; j < kItems; ++j)
        if (i0 + j < a.n) w[i0 + j] = w_in[j] / sum;
      return;
    }
    const rr_sys_plan plan = rr_sys_plan_make(rho, ts.tot, pa.n_global);
    mark_sources(t, ts.pre + t.thread_off, i0, a.n, plan, ts.tot, 0, markers, carry);
    const double w_new = 1.0 / (double)pa.n_global;
#pragma unroll
    for (int j = 0; j < kItems; ++j)
      if (i0 + j < a.n) w[i0 + j] = w_new;
  } else {
    if (!fire && !ea.want) return;
    if (fire) {
      const rr_sys_plan plan = rr_sys_plan_make(rho, ts.tot, pa.n_global);
      mark_sources(t, ts.pre + t.thread_off, i0, a.n, plan, ts.tot, 0, markers, carry, (!DEFER && ea.want) ? offspring : nullptr);
    }
  }
}

// FS_WEIGHTS (FastSLAM, k_fs1_plan's weight handling): the weights are explicit -- w /= sum when the gate stays shut
// (fastslam1.rs:196-203), w = 1/n when it fires (:228) -- and are rewritten by the threads that read them.
// (amdgpu_waves_per_eu(4): at most 128 VGPRs, i.e. two of these 512-thread workgroups per CU -- the 489 workgroups of a
// 1e6-particle filter must all be on the 256 CUs at once)
// the serial plan of a launch that gave up (k_quantize_plan_mark): every tile by ONE workgroup
template <bool FS_WEIGHTS, bool DEFER>
__device__ __attribute__((noinline)) void plan_serial(typename std::conditional<FS_WEIGHTS, double*, const double*>::type w, ImageArgs a, int mode,
                                                       int shift, int fire, double rho, PlanArgs pa, uint64_t n_tiles, const uint64_t* rec,
                                                       unsigned int* markers, unsigned int* carry, EstArgs ea, bool want_est, int cur_after,
                                                       TileSums ts /* the grand totals; .pre per tile below */, uint64_t* s_w, uint64_t* s4,
                                                       uint64_t* s_tile) {
  const int tid = threadIdx.x;
  TileScan t;
  EstFields ef;
  double w_in[FS_WEIGHTS ? kItems : 1];
  unsigned int offspring[kItems];
  for (uint64_t tile = 0; tile < n_tiles; ++tile) {
    const uint64_t j0 = tile * kTile + (uint64_t)tid * kItems;
    __syncthreads();
    uint64_t tt;
    u128 qq;
    plan_image_tile<FS_WEIGHTS>(w, a, mode, shift, tile, s_w, s4, t, w_in, &tt, &qq);
    if (tid == 0) s_tile[0] = ld_dev(&plan_pairs(const_cast<uint64_t*>(rec))[tile].bits);
    if (want_est) est_prefetch(ea, cur_after, j0, a.n, ef);
    __syncthreads();
    ts.pre = s_tile[0];
    plan_apply_tile<FS_WEIGHTS, DEFER>(w, a, mode, shift, t, w_in, ts, fire, rho, pa, tile, markers, carry, ea, ef, offspring);
    if (want_est) est_tile_partial(ea, t, offspring, fire, j0, a.n, ef, tile);
  }
}

// DEFER (PF / MCL): the estimate in its deferred form (EstArgs) as a build of its own -- no offspring counts, no particle fields
// in flight across the hand-over: the kernel sits at its 128-VGPR cap, and the one build that served both forms spilled
template <bool FS_WEIGHTS, bool DEFER = false>
static __global__ __launch_bounds__(kTileBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_quantize_plan_mark(
    typename std::conditional<FS_WEIGHTS, double*, const double*>::type w, Ctl* __restrict__ ctl,
    const double* __restrict__ wmax_src, ImageArgs a,
    uint64_t* __restrict__ rec, unsigned int* __restrict__ ticket /* kTicketWords, zero between launches */, uint64_t epoch,
    int settle, uint64_t n_tiles, PlanArgs pa, unsigned int* __restrict__ markers, unsigned int* __restrict__ carry,
    EstArgs ea, uint64_t giveup_ticks) {
  constexpr int W = kTileBlock / kWave;
  __shared__ uint64_t s4[4 * W];
  __shared__ uint64_t s_w[W];
  __shared__ uint64_t s_tile[3];
  __shared__ uint64_t s_sum[4];  // this tile's exclusive prefix, T, q2_hi, q2_lo
  __shared__ int s_last;
  __shared__ int s_gaveup;
  const int tid = threadIdx.x;
  TagPair* const pp = plan_pairs(rec);
#if defined(RR_PLAN_TIMELINE)
  uint64_t* const tl = rec + (kTileBlock + 1) * kRecWords + 16;  // (behind the records and the heads: sized by the host)
#endif
  RR_TL(0);
  // ---- A: the integer image of this tile (quantize_reduce_tile's decisions, tile_scan's blocked layout)
  const double wmax = *wmax_src;
  const bool forced_uniform = a.honour_uniform_flag && ctl->weights_uniform;
  const bool usable = !forced_uniform && wmax > 0.0 && wmax < INFINITY;
  const int mode = usable ? (int)kImageWeights : (forced_uniform ? (int)kImageUniform : a.degenerate);
  const int shift = usable ? rr_fix_shift(wmax, a.n_global) : 0;
  const int cur_after = (settle && ctl->pending) ? ctl->cur ^ 1 : ctl->cur;  // what Ctl.cur will be settled to
  const bool serial_only = ctl->grid_timeout != 0;  // an earlier launch gave up: nobody waits any more (see above)
  TileScan t;
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)tid * kItems;
  EstFields ef;
  double w_in[FS_WEIGHTS ? kItems : 1];
  uint64_t* const head = rec + n_tiles * kRecWords;
  {
    uint64_t tt;
    u128 qq;
    plan_image_tile<FS_WEIGHTS>(w, a, mode, shift, blockIdx.x, s_w, s4, t, w_in, &tt, &qq);
    if (tid == 0) {
      uint64_t* r = rec + (uint64_t)blockIdx.x * kRecWords;
      st_dev(&r[0], tt);
      st_dev(&r[1], qq.hi);
      st_dev(&r[2], qq.lo);
      // acknowledged at device scope before the ticket says so.  (The records as self-vouching pairs and the ticket taken at once,
      // the last arrival looking again at a late one, was measured in round 5: the gain of the hand-BACK's pairs, -1 us, was gone
      // again -- twice the loads in the last arrival's scan; profiles/r05h_plan_handover_ab.md)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RR_TL(1);      s_last = last_arrival(ticket, blockIdx.x, (unsigned int)n_tiles, /*fence=*/false) ? 1 : 0;
      s_gaveup = 0;
      RR_TL(2);
    }
  }
  __syncthreads();
  // the particle fields of the estimate: requested HERE -- after this workgroup's record and ticket are out (the
  // s_waitcnt in front of the ticket would otherwise wait for them too and delay every arrival), before the wait for the
  // other workgroups' sums, which is when the memory system has nothing else to do
  const bool last = s_last != 0;
  // (not the last arrival: everybody waits for ITS scan, and its s_waitcnt below would wait for these loads too; it fetches its
  // fields once the flag is up.  Requesting the fields only after the flag for EVERY workgroup was measured too, round 4: the flag
  // then comes 2 us earlier -- the early workgroups' 32 MB of requests no longer stand in the way of the late workgroups' weights and
  // records -- but the sums' read-back and the marking wait behind those requests instead: plan kernel 21.3 -> 22.8 us.)
  if (!FS_WEIGHTS && !DEFER && ea.want && !last) est_prefetch(ea, cur_after, i0, a.n, ef);
  // the plan's one uniform, drawn while this workgroup has nothing to do but wait (the last arrival, for whom everybody waits,
  // draws it once the sums are out)
  double rho = NAN;
  if (!last) rho = plan_rho_early(pa);
  // ---- the last arrival: exclusive prefix per tile and the grand totals, then the state word
  if (last) {
    const int lane = tid & 63, wv = tid >> 6;
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
    __syncthreads();  // (s_w / s4 were read above)
    if (lane == 63) s_w[wv] = inc;
    if (lane == 0) {
      s4[wv] = qw.hi;
      s4[W + wv] = qw.lo;
    }
    __syncthreads();
    uint64_t wave_off = 0;
    for (int k = 0; k < wv; ++k) wave_off += s_w[k];
    const uint64_t my_pre = wave_off + inc - tk;
    if (tid == 0) {
      uint64_t tt = 0;
      u128 qq = {0, 0};
      for (int k = 0; k < W; ++k) {
        tt += s_w[k];
        qq = add128(qq, u128{s4[k], s4[W + k]});
      }
      s_sum[1] = tt;
      s_sum[2] = qq.hi;
      s_sum[3] = qq.lo;
    }
    if ((uint64_t)tid == (uint64_t)blockIdx.x) s_sum[0] = my_pre;
    if ((uint64_t)tid < n_tiles) put_pair(&pp[tid], my_pre, epoch);  // every tile's prefix, vouching for itself
    if (tid == 0) {
      put_pair(&pp[kTileBlock + 0], s_sum[1], epoch);
      put_pair(&pp[kTileBlock + 1], s_sum[2], epoch);
      put_pair(&pp[kTileBlock + 2], s_sum[3], epoch);
    }
    // The state word goes out right behind the pairs -- NOT after their acknowledgement (1.2 us of every workgroup's wait until round
    // 5): a waiter that sees RAISED reads its pairs and, should one not have landed yet, looks again.
    __syncthreads();  // (every thread's pair stores are issued)
    if (tid == 0) s_gaveup = (int)(plan_state_decide(&head[3], epoch, serial_only ? 1 : 0, /*guess_previous=*/true) & 1);
  } else if (tid == 0) {
    // ---- everybody else: one thread waits for the state word of this launch (ONE word for everybody: looking at more than that
    // while waiting -- the pairs themselves, shared or one record per tile -- was measured and is slower: +1 / +4.4 us per step),
    // then takes this tile's prefix and the totals, each of which vouches for itself
    if (serial_only) {
      s_gaveup = 1;
    } else {
      const uint64_t t0 = wall_clock64();
      uint64_t st;
      while (((st = ld_dev(&head[3])) >> 1) != epoch) {
        if (wall_clock64() - t0 > giveup_ticks) {
          st = plan_state_decide(&head[3], epoch, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      s_gaveup = (int)(st & 1);
      if (!(st & 1)) {
        const TagPair* const want[4] = {&pp[blockIdx.x], &pp[kTileBlock + 0], &pp[kTileBlock + 1], &pp[kTileBlock + 2]};
        uint64_t got[4];
        // (stores issued before the state word moved: they land.  The wait is bounded all the same -- ADVICE r5: a stale or damaged
        // record buffer must not hang the device.  The launch's outcome is RAISED by now and cannot be taken back, so on expiry this
        // workgroup takes its numbers from where the last arrival took them: the tiles' records, every one acknowledged at device
        // scope before its ticket -- the same prefix, the same totals -- and raises Ctl.grid_timeout, which moves the handle to the
        // multi-launch plan at the host's next look.)
        const uint64_t tp = wall_clock64();
        bool fit;
        while (!(fit = take_pairs<4>(want, epoch, got))) {
          if (wall_clock64() - tp > giveup_ticks + 100000) break;  // (+ 1 ms: RR_PF_PLAN_TIMEOUT_US = 0, the test hook, means "do not wait for the FLAG")
          __builtin_amdgcn_s_sleep(1);
        }
        if (!fit) {
          uint64_t pre = 0, tot = 0;
          u128 qq = {0, 0};
          for (uint64_t k = 0; k < n_tiles; ++k) {
            const uint64_t* r = rec + k * kRecWords;
            const uint64_t tk = ld_dev(&r[0]);
            if (k < (uint64_t)blockIdx.x) pre += tk;
            tot += tk;
            qq = add128(qq, u128{ld_dev(&r[1]), ld_dev(&r[2])});
          }
          got[0] = pre;
          got[1] = tot;
          got[2] = qq.hi;
          got[3] = qq.lo;
          atomicAdd(&ctl->grid_timeout, 1);
        }
        s_sum[0] = got[0];
        s_sum[1] = got[1];
        s_sum[2] = got[2];
        s_sum[3] = got[3];
      }
    }
  }
  __syncthreads();
  RR_TL(3);
  const bool gaveup = s_gaveup != 0;
  if (gaveup && !last) return;  // record and ticket are in; the last arrival plans this tile as well
  if (!FS_WEIGHTS && !DEFER && ea.want && last && !gaveup) est_prefetch(ea, cur_after, i0, a.n, ef);
  RR_TL(4);
  TileSums ts;
  ts.pre = s_sum[0];
  ts.tot = s_sum[1];
  ts.q2 = u128{s_sum[2], s_sum[3]};
  // ---- what k_quantize_reduce's and k_plan_mark's first threads leave in Ctl: the LAST ARRIVAL, whatever the launch's outcome --
  // every workgroup has arrived by then (and read what it needs of Ctl), and it is the one party that knows the launch's ONE
  // answer (a waiter that went by the pairs does not, and two writers would settle twice)
  if (tid == 0 && last) {
    if (settle && ctl->pending) {
      ctl->cur ^= 1;
      ctl->pending = 0;
    }
    ctl->usable = usable ? 1 : 0;
    ctl->image_mode = mode;
    ctl->shift = shift;
    ctl->wmax = wmax;
    if (gaveup) ctl->grid_timeout += 1;
  }
  if (last) {
    rho = plan_rho(pa);  // (every thread: the marking below needs it; finalize_plan takes it instead of drawing it again)
    if (tid == 0) finalize_plan(ctl, ts.tot, 0, ts.tot, ts.q2, pa, rho, mode, shift);
  }
  // ---- B: plan_mark_tile from here on
  const int fire = gate_decision(mode, ts, pa);
  unsigned int offspring[kItems];
  const double denom = fire ? (double)pa.n_global : (double)ts.tot;
  // the estimate's partial sums are formed here, or (deferred form, gate fired) by whoever moves the particles
  const bool want_est = !FS_WEIGHTS && ea.want && !(DEFER && fire);
  const bool late_fields = want_est && DEFER;  // requested now that the decision is known
  if (DEFER && ea.want && fire && tid == 0 && last) est_publish(ctl, denom, pa.rstep, kEstSlotTiles);
  if (FS_WEIGHTS ? (!fire && mode != kImageWeights) : (!fire && !want_est)) return;  // nothing to write
  if (!gaveup) {
    // One tile -- this workgroup's, its image still in registers.  (Straight-line code: as ONE loop with the serial plan below,
    // the tile's image, weights and fields were loop-carried values of a kernel that sits at its 128-VGPR cap.)
    if (late_fields) est_prefetch(ea, cur_after, i0, a.n, ef);
    plan_apply_tile<FS_WEIGHTS, DEFER>(w, a, mode, shift, t, w_in, ts, fire, rho, pa, blockIdx.x, markers, carry, ea, ef, offspring);
    RR_TL(5);
    if (want_est) est_tile_partial(ea, t, offspring, fire, i0, a.n, ef, blockIdx.x);
    RR_TL(6);
  } else {
    // The launch gave up and this workgroup arrived last: every tile, one after the other (the serial plan) -- the same functions,
    // behind a real call, so that the loop's live values are not the straight path's register pressure.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's own prefix pairs are read back below
    __syncthreads();
    plan_serial<FS_WEIGHTS, DEFER>(w, a, mode, shift, fire, rho, pa, n_tiles, rec, markers, carry, ea, want_est, cur_after, ts, s_w, s4, s_tile);
  }
  if (want_est && tid == 0 && last) est_publish(ctl, denom, pa.rstep, kEstPlanTiles);
}

// sharded: the plan is already in Ctl (k_shard_plan); mark this shard's sources.  tile_offset =
// exclusive tile prefix written by k_scan_tiles.
// window != 0 (peer-to-peer step): marker position of global slot s is s + pad (see resolve_tile_window) instead of
// s - served_first
static __global__ __launch_bounds__(kTileBlock) void k_mark(const double* __restrict__ w, const Ctl* __restrict__ ctl,
                                                       ImageArgs a, const uint64_t* __restrict__ tile_offset,
                                                       unsigned int* __restrict__ markers,
                                                       unsigned int* __restrict__ carry, int window = 0, uint64_t pad = 0) {
  if (!ctl->fired) return;
  __shared__ uint64_t s_w[kTileBlock / kWave];
  const TileScan t = tile_scan(w, a, ctl->image_mode, ctl->shift, blockIdx.x, s_w);
  const rr_sys_plan plan = ctl->plan;
  const uint64_t slot_base = window ? (uint64_t)0 - pad : ctl->served_first;
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
  mark_sources(t, ctl->base + tile_offset[blockIdx.x] + t.thread_off, i0, a.n, plan, ctl->total, slot_base, markers, carry);
}

// k_shard_plan + k_mark in one launch, over RAW tile totals (k_quantize_reduce_sums): every workgroup combines the shards' sums
// itself (G <= 16 triples: the same integer adds on every workgroup, so the same gate decision and plan), adds the totals of
// the tiles before its own, and marks its sources in the window layout; workgroup 0 also leaves the plan in Ctl.
static __global__ __launch_bounds__(kTileBlock) void k_mark_plan(const double* __restrict__ w, Ctl* __restrict__ ctl, ImageArgs a,
                                                            const uint64_t* __restrict__ tile_total, const uint64_t* __restrict__ all_sums,
                                                            int n_shards, int rank, PlanArgs pa, unsigned int* __restrict__ markers,
                                                            unsigned int* __restrict__ carry, uint64_t pad) {
  constexpr int W = kTileBlock / kWave;
  __shared__ uint64_t s_pre[W];
  __shared__ uint64_t s_w[W];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint64_t total = 0, base = 0;
  u128 qq = {0, 0};
  for (int g = 0; g < n_shards; ++g) {
    if (g == rank) base = total;
    total += all_sums[3 * g];
    qq = add128(qq, u128{all_sums[3 * g + 1], all_sums[3 * g + 2]});
  }
  const int mode = ctl->image_mode, shift = ctl->shift;  // written by the quantize kernel; workgroup 0's finalize_plan does not touch them
  TileSums ts;
  ts.pre = 0;
  ts.tot = total;
  ts.q2 = qq;
  const int fire = gate_decision(mode, ts, pa);
  uint64_t pre = 0;
  for (uint64_t k = tid; k < blockIdx.x; k += kTileBlock) pre += tile_total[k];
  pre = wave_sum_u64(pre);
  if (lane == 0) s_pre[wv] = pre;
  __syncthreads();
  pre = 0;
  for (int k = 0; k < W; ++k) pre += s_pre[k];
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) finalize_plan(ctl, total, base, all_sums[3 * rank], qq, pa);
  if (!fire) return;
  double rho = pa.rho_override;
  if (rho != rho) {
    double dummy;
    rr_uniform2(pa.seed, RR_STREAM_RESAMPLE, pa.rstep, 0, &rho, &dummy);
  }
  const rr_sys_plan plan = rr_sys_plan_make(rho, total, pa.n_global);
  const TileScan t = tile_scan(w, a, mode, shift, blockIdx.x, s_w);
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)tid * kItems;
  mark_sources(t, base + pre + t.thread_off, i0, a.n, plan, total, (uint64_t)0 - pad, markers, carry);
}

// markers -> source indices for the kResolveSlots slots of one workgroup; returns this thread's
// indices (slot = tile_base + r*kBlock + tid), clears the markers it consumed
__device__ inline void resolve_tile(unsigned int* __restrict__ markers, const unsigned int* __restrict__ carry,
                                    uint64_t n_slots, uint64_t tile, unsigned int idx[kResolveRows]) {
  __shared__ unsigned int s_m[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t tile_base = tile * kResolveSlots;
  unsigned int run = carry[tile];  // source (+1) of the tile's first slot
  // rows of kBlock consecutive slots: running maximum along the row, carried to the next row
#pragma unroll
  for (int r = 0; r < kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
    unsigned int m = 0;
    if (k < n_slots) {
      m = markers[k];
      if (m) markers[k] = 0;
    }
    m = wave_scan_max_u32(m);
    if (lane == 63) s_m[wv] = m;
    __syncthreads();
    unsigned int pre = run;
    for (int q = 0; q < wv; ++q) pre = s_m[q] > pre ? s_m[q] : pre;
    if (pre > m) m = pre;
    idx[r] = m - 1;
    unsigned int row_max = run;
    for (int q = 0; q < kBlock / kWave; ++q) row_max = s_m[q] > row_max ? s_m[q] : row_max;
    run = row_max;
    __syncthreads();
  }
}

// Sharded step over the peer-to-peer transport: the marker space is the GLOBAL slot index shifted by `pad`, pad chosen so
// that this shard's own block of slots starts on a tile boundary (position P0 = first own global slot + pad, P0 % kResolveSlots
// == 0): own slot k sits at P0 + k, and the consumer -- the next step's k_step_lazy -- resolves the tiles of its own block
// exactly as the unsharded kernel does, no index array in between.  This shard's sources feed the positions
// [p_first, p_end) (Ctl.served_first/count + pad): a window that overlaps the own block almost entirely and sticks out of it
// by the few slots the cumulative weight has drifted across the block boundaries.  Slots of the own block outside the
// window are served by a peer (delivered into this rank's inbox); positions of the window outside the own block belong
// to peers (k_push_window delivers them).  Differences to resolve_tile: markers outside the window are not read (they are
// zero anyway), the carry of a tile is only meaningful when the tile STARTS inside the window (a slot run crosses or begins
// at its first position; the tile that contains p_first starts with no source), and only positions in [clear_lo, clear_hi)
// are consumed -- the own block by the consumer, the rest by the push kernel, which reads own positions of a shared
// boundary tile for its running maximum but leaves them set.
__device__ inline void resolve_tile_window(unsigned int* __restrict__ markers, const unsigned int* __restrict__ carry, uint64_t tile,
                                           uint64_t p_first, uint64_t p_end, uint64_t clear_lo, uint64_t clear_hi,
                                           unsigned int idx[kResolveRows]) {
  __shared__ unsigned int s_m[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t tile_base = tile * kResolveSlots;
  unsigned int run = (tile_base >= p_first && tile_base < p_end) ? carry[tile] : 0u;
#pragma unroll
  for (int r = 0; r < kResolveRows; ++r) {
    const uint64_t k = tile_base + (uint64_t)r * kBlock + tid;
    unsigned int m = 0;
    if (k >= p_first && k < p_end) {
      m = markers[k];
      if (m && k >= clear_lo && k < clear_hi) markers[k] = 0;
    }
    m = wave_scan_max_u32(m);
    if (lane == 63) s_m[wv] = m;
    __syncthreads();
    unsigned int pre = run;
    for (int q = 0; q < wv; ++q) pre = s_m[q] > pre ? s_m[q] : pre;
    if (pre > m) m = pre;
    idx[r] = m - 1;
    unsigned int row_max = run;
    for (int q = 0; q < kBlock / kWave; ++q) row_max = s_m[q] > row_max ? s_m[q] : row_max;
    run = row_max;
    __syncthreads();
  }
}

// guide markers -> guide table (and the markers are cleared for the next step)
static __global__ __launch_bounds__(kBlock) void k_guide_resolve(const Ctl* __restrict__ ctl, unsigned int* __restrict__ markers,
                                                            const unsigned int* __restrict__ carry,
                                                            unsigned int* __restrict__ guide, int guide_log2, int local = 0) {
  if (!ctl->fired) return;
  const uint64_t total = local ? ctl->total_local : ctl->total;  // local: a shard's table over its own interval (k_cdf)
  const uint64_t n_buckets = (total >> guide_shift(total, guide_log2)) + 1;
  if ((uint64_t)blockIdx.x * kResolveSlots >= n_buckets) return;
  unsigned int idx[kResolveRows];
  resolve_tile(markers, carry, n_buckets, blockIdx.x, idx);
#pragma unroll
  for (int r = 0; r < kResolveRows; ++r) {
    const uint64_t k = (uint64_t)blockIdx.x * kResolveSlots + (uint64_t)r * kBlock + threadIdx.x;
    if (k < n_buckets) guide[k] = idx[r];
  }
}

// CDF target of global output slot `slot` under the current plan
__device__ inline uint64_t resample_target(const Ctl* __restrict__ ctl, int scheme, uint64_t slot, uint64_t seed,
                                           unsigned int rstep, const double* __restrict__ r_explicit, uint64_t k) {
  if (scheme == RR_RESAMPLE_SYSTEMATIC) return rr_sys_target(ctl->plan, slot);
  double r, dummy;
  if (r_explicit) r = r_explicit[k];
  else rr_uniform2(seed, RR_STREAM_RESAMPLE, rstep, slot, &r, &dummy);
  return rr_fix_target_multinomial(r, ctl->total);
}

}  // namespace rr
