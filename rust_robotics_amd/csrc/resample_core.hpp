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

#include <hip/hip_runtime.h>

#include <cmath>

#include "rr_common.hpp"
#include "rr_pf_spec.h"

namespace rr {

constexpr int kBlock = 256;
constexpr int kItems = 8;
constexpr int kTile = kBlock * kItems;  // 2048 particles per scan tile
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
};

// ------------------------------------------------------------------------------------------
// K2: per-tile integer totals and sum of squares.  Tile = 2048 particles; each wave owns 512
// consecutive particles as 8 coalesced rows of 64.  wmax_src points at the maximum to scale by
// (Ctl.wmax_bits on one GPU, the all-reduced maximum when sharded).
static __global__ __launch_bounds__(kBlock) void k_quantize_reduce(const double* __restrict__ w,
                                                                  Ctl* __restrict__ ctl,
                                                                  const double* __restrict__ wmax_src,
                                                                  ImageArgs a,
                                                                  uint64_t* __restrict__ tile_total,
                                                                  uint64_t* __restrict__ tile_q2) {
  __shared__ uint64_t s_t[kBlock / kWave];
  __shared__ uint64_t s_qh[kBlock / kWave];
  __shared__ uint64_t s_ql[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double wmax = *wmax_src;
  const bool forced_uniform = a.honour_uniform_flag && ctl->weights_uniform;
  const bool usable = !forced_uniform && wmax > 0.0 && wmax < INFINITY;
  const int mode = usable ? (int)kImageWeights : (forced_uniform ? (int)kImageUniform : a.degenerate);
  const int shift = usable ? rr_fix_shift(wmax, a.n_global) : 0;
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)wv * (kTile / 4);
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
    for (int k = 0; k < kBlock / kWave; ++k) {
      tt += s_t[k];
      qq = add128(qq, u128{s_qh[k], s_ql[k]});
    }
    tile_total[blockIdx.x] = tt;
    tile_q2[2 * blockIdx.x] = qq.hi;
    tile_q2[2 * blockIdx.x + 1] = qq.lo;
    if (blockIdx.x == 0) {
      ctl->usable = usable ? 1 : 0;
      ctl->image_mode = mode;
      ctl->shift = shift;
      ctl->wmax = wmax;
    }
  }
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
};

__device__ inline void finalize_plan(Ctl* ctl, uint64_t total_global, uint64_t base, uint64_t total_local, u128 q2,
                                     const PlanArgs& a) {
  ctl->total_local = total_local;
  ctl->total = total_global;
  ctl->base = base;
  ctl->q2_hi = q2.hi;
  ctl->q2_lo = q2.lo;
  double neff, sum;
  if (ctl->image_mode == kImageWeights && total_global > 0) {
    sum = rr_fix_total_to_double(total_global, ctl->shift);
    neff = rr_fix_neff(total_global, q2.hi, q2.lo);
  } else if (ctl->image_mode == kImageUniform) {  // T = N, N_eff = N
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
  if (fire && a.scheme == RR_RESAMPLE_SYSTEMATIC) {
    double rho = a.rho_override;
    if (rho != rho) {
      double dummy;
      rr_uniform2(a.seed, RR_STREAM_RESAMPLE, a.rstep, 0, &rho, &dummy);
    }
    ctl->rho = rho;
    ctl->plan = rr_sys_plan_make(rho, total_global, a.n_global);
  }
}

// ------------------------------------------------------------------------------------------
// K3: single workgroup.  Exclusive scan of the tile totals (in place) and the local sums.  On one
// GPU (single_shard != 0) it also finalises the plan; when sharded it writes the local sums to
// shard_sums_out (total, q2_hi, q2_lo) for the all-gather and k_shard_plan finishes the job.
static __global__ __launch_bounds__(kScanThreads) void k_scan_tiles(uint64_t* __restrict__ tile_total,
                                                                   const uint64_t* __restrict__ tile_q2,
                                                                   Ctl* __restrict__ ctl, uint64_t n_tiles,
                                                                   int single_shard, PlanArgs a,
                                                                   uint64_t* __restrict__ shard_sums_out) {
  __shared__ uint64_t s_w[kScanThreads / kWave];
  __shared__ uint64_t s_h[kScanThreads / kWave];
  __shared__ uint64_t s_l[kScanThreads / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
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
    for (int k = 0; k < kScanThreads / kWave; ++k) {
      total += s_w[k];
      qq = add128(qq, u128{s_h[k], s_l[k]});
    }
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
// K4: inclusive integer CDF of this shard: cdf[i] = base + tile_offset + within-tile scan.
// Reads w (8 B), writes cdf (8 B).  Skipped when the gate did not fire.
static __global__ __launch_bounds__(kBlock) void k_cdf(const double* __restrict__ w, const Ctl* __restrict__ ctl,
                                                      ImageArgs a, const uint64_t* __restrict__ tile_offset,
                                                      uint64_t* __restrict__ cdf) {
  if (!ctl->fired) return;
  __shared__ uint64_t s_w[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mode = ctl->image_mode;
  const int shift = ctl->shift;
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)wv * (kTile / 4);
  uint64_t vals[kItems];
  uint64_t carry = 0;
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    uint64_t q = quantize_at(w, base + r * 64 + lane, a.n, mode, shift, a.gid0, a.n_global);
    uint64_t incl = wave_scan_u64(q, lane);
    vals[r] = incl + carry;
    carry += shfl_u64(incl, 63);
  }
  if (lane == 0) s_w[wv] = carry;
  __syncthreads();
  uint64_t off = ctl->base + tile_offset[blockIdx.x];
  for (int k = 0; k < wv; ++k) off += s_w[k];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    uint64_t i = base + r * 64 + lane;
    if (i < a.n) cdf[i] = vals[r] + off;
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
