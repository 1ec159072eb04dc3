// fs1_engine.hip -- MI355X (gfx950) FastSLAM 1.0 engine behind include/rr_fastslam1.h.
//
// Replaces the CPU hot path of /root/reference/crates/rust_robotics_slam/src/fastslam1.rs:
//   predict_particle :123-137, update_landmark :140-183 (the N x K double loop of :250-256),
//   normalize_weights / compute_neff :186-203, resample :205-234, get_best_particle :269-274.
// Not a translation of the reference's Vec<Particle>{Vec<Landmark>} layout: every particle's map
// lives in HBM as landmark-major planes plane[(l*6+f)*N + p] (+3 pose planes), so one wavefront
// touching 64 consecutive particles of one landmark field is one coalesced 512-byte access, the
// observation list is wave-uniform (LDS), and the systematic resample is a monotone plane gather.
//
// Kernels of one update (rr_fs1_update_async): 3 launches (round 2: 5)
//   k_fs1_resolve_predict   the last plan's markers -> idx[] (the update reads every observed landmark through it: lazy
//                      gather) and the pose planes moved through it                   48 B / particle
//   k_fs1_observe      (particle, observation chunk): 2x2 EKF per observed landmark,
//                      R 48 B + W 48 B per (particle, landmark) update               <- dominant
//                      the last chunk's workgroup of a particle block forms the weight (product of the chunks' factors in
//                      chunk order) and the block's maximum
//   rr::k_quantize_plan_mark<FS_WEIGHTS>   integer image + gate + w /= sum or slot-run markers + w = 1/n, one launch
//                      (resample_core.hpp; beyond 2^20 particles rr::k_quantize_reduce + k_fs1_plan)
// and of the separate / sharded entry points:
//   k_fs1_predict, k_fs1_resolve   the two halves of k_fs1_resolve_predict on their own
//   rr::k_scan_tiles / k_cdf, k_fs1_normalize, k_fs1_indices(_sharded)   integer CDF, CDF search per output slot
//   k_fs1_gather       out[plane][k] = in[plane][idx[k]] over 3 + 6L planes          16 B / element
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "p2p_core.hpp"
#include "rccl_core.hpp"
#include "resample_core.hpp"
#include "resident_core.hpp"
#include "rr_common.hpp"
#include "rr_fastslam1.h"
#include "rr_fastslam2.h"
#include "rr_pf_spec.h"

using rr::Ctl;
using rr::fail;
using rr::ImageArgs;
using rr::kBlock;
using rr::kScanThreads;
using rr::kTile;
using rr::PlanArgs;

namespace {

constexpr int kMaxChunks = 64;
constexpr int kPlanesPerThread = 8;  // planes one gather thread copies for its output slot
// sharded lazy resample: idx[p] == kInPlace => a peer has delivered slot p's particle into this rank's
// fine-grained inbox (k_fs1_push); the consuming kernels then read slot p of the inbox instead
constexpr unsigned int kInPlace = 0xffffffffu;

struct Planes {
  double* s[2];         // [(3 + 6L) * N]: planes 0..2 = x, y, yaw; plane 3 + l*6 + f = landmark l field f
  const double* inbox;  // sharded: fine-grained mirror of one set where peers deliver cross-rank particles (else null)
};

// LAZY: consume a pending resample -- read the pose of slot p from particle idx[p] of the live
// set and write the predicted pose to slot p of the OTHER set (k_quantize_reduce flips Ctl.cur
// afterwards).  Otherwise in place.
template <bool EXPLICIT, bool LAZY>
__global__ __launch_bounds__(kBlock) void k_fs1_predict(Planes pl, const Ctl* __restrict__ ctl, uint64_t n,
                                                       double u0, double u1, rr_fs1_model m, uint64_t seed,
                                                       unsigned int step, const double* __restrict__ z0,
                                                       const double* __restrict__ z1,
                                                       const unsigned int* __restrict__ idx, uint64_t gid0) {
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const bool pending = LAZY && ctl->pending;
  double* __restrict__ dst = pl.s[pending ? ctl->cur ^ 1 : ctl->cur];
  const unsigned int ji = pending ? idx[p] : (unsigned int)p;
  const bool inplace = pending && ji == kInPlace;
  const double* __restrict__ src = inplace ? pl.inbox : pl.s[ctl->cur];
  const uint64_t j = inplace ? p : ji;
  double x = src[j], y = src[n + j], yaw = src[2 * n + j];
  double a, b;
  if (EXPLICIT) {
    a = z0[p];
    b = z1[p];
  } else {
    rr_fs1_motion_noise(seed, step, gid0 + p, &a, &b);
  }
  rr_fs1_predict_one(&x, &y, &yaw, u0, u1, a, b, m);
  dst[p] = x;
  dst[n + p] = y;
  dst[2 * n + p] = yaw;
}

// observations of an update staged through the first kernel (k_fs1_resolve_predict)
constexpr int kZStagePerThread = 4;              // words one thread of the staging workgroup carries
constexpr int kZStageWords = kZStagePerThread * kBlock;  // 1024 doubles = 341 observations; more: hipMemcpyAsync as before
constexpr int kZRing = 32;                       // pinned slots: the host runs at most this many updates ahead of the device
struct ZStage {
  const double* host;  // pinned, device-visible: 3 * n_z doubles of this update
  double* dev;         // where k_fs1_observe reads them
  uint64_t* done;      // pinned: sequence number of the last slot the device has consumed
  uint64_t seq;
  int words;           // 0: nothing to stage
};

// The same with the resample plan's markers resolved on the way (single GPU, FastSLAM 1.0): the plan kernel leaves markers,
// the first kernel of the NEXT update turns them into source indices (running maximum per 2048 slots, rr::resolve_tile), keeps
// them in idx[] for k_fs1_observe / k_fs1_gather and moves the poses through them -- one launch instead of k_fs1_resolve at
// the end of an update and k_fs1_predict at the start of the next.
__global__ __launch_bounds__(kBlock) void k_fs1_resolve_predict(Planes pl, const Ctl* __restrict__ ctl, uint64_t n, double u0, double u1,
                                                               rr_fs1_model m, uint64_t seed, unsigned int step,
                                                               unsigned int* __restrict__ markers, const unsigned int* __restrict__ carry,
                                                               unsigned int* __restrict__ idx, uint64_t gid0, ZStage zs, int resolve) {
  // The update's observations come along: the host has left them in a pinned slot, the last workgroup reads them over the
  // bus while it does its share of the poses and leaves them in device memory for k_fs1_observe -- no H2D copy operation in
  // the stream (4.7 us blit kernel + a boundary per update at 200 observations; they do not fit a kernel argument).
  const bool stager = zs.words > 0 && blockIdx.x == gridDim.x - 1;
  double zv[kZStagePerThread];
  if (stager) {
#pragma unroll
    for (int k = 0; k < kZStagePerThread; ++k) {
      const int i = k * kBlock + (int)threadIdx.x;
      zv[k] = i < zs.words ? rr::ld_sys(zs.host + i) : 0.0;
    }
  }
  const bool pending = ctl->pending != 0;  // uniform
  unsigned int from[rr::kResolveRows];
  if (pending && resolve) {
    rr::resolve_tile(markers, carry, n, blockIdx.x, from);
  } else if (pending) {  // an accessor has had the markers resolved already (k_fs1_resolve)
#pragma unroll
    for (int r = 0; r < rr::kResolveRows; ++r) {
      const uint64_t p = (uint64_t)blockIdx.x * rr::kResolveSlots + (uint64_t)r * kBlock + threadIdx.x;
      from[r] = p < n ? idx[p] : 0u;
    }
  }
  const double* __restrict__ src = pl.s[ctl->cur];
  double* __restrict__ dst = pl.s[pending ? ctl->cur ^ 1 : ctl->cur];
  double x[rr::kResolveRows], y[rr::kResolveRows], yaw[rr::kResolveRows];
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {  // every row's loads first
    const uint64_t p = (uint64_t)blockIdx.x * rr::kResolveSlots + (uint64_t)r * kBlock + threadIdx.x;
    x[r] = y[r] = yaw[r] = 0.0;
    if (p < n) {
      const uint64_t j = pending ? (uint64_t)from[r] : p;
      if (pending && resolve) idx[p] = from[r];
      x[r] = src[j];
      y[r] = src[n + j];
      yaw[r] = src[2 * n + j];
    }
  }
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t p = (uint64_t)blockIdx.x * rr::kResolveSlots + (uint64_t)r * kBlock + threadIdx.x;
    if (p < n) {
      double a, b;
      rr_fs1_motion_noise(seed, step, gid0 + p, &a, &b);
      rr_fs1_predict_one(&x[r], &y[r], &yaw[r], u0, u1, a, b, m);
      dst[p] = x[r];
      dst[n + p] = y[r];
      dst[2 * n + p] = yaw[r];
    }
  }
  if (stager) {
#pragma unroll
    for (int k = 0; k < kZStagePerThread; ++k) {
      const int i = k * kBlock + (int)threadIdx.x;
      if (i < zs.words) zs.dev[i] = zv[k];
    }
    __syncthreads();  // every thread's loads of the slot have returned (their values have been stored)
    if (threadIdx.x == 0) rr::st_sys_u64(zs.done, zs.seq);  // the host may reuse the slot
  }
}

// FastSLAM 2.0 (fastslam2.rs:339-358): the pose is not pushed through the noisy motion model but
// SAMPLED from the proposal that fuses the motion prior with the first observation of the step --
// per particle a 3x3 prior, a 2x2 innovation covariance, two 3x3 inverses, a Cholesky factor and
// three normals (rr_fs2_predict_one).  Reads the six planes of the first observation's landmark;
// LAZY as in k_fs1_predict.  noise (EXPLICIT): 3 unit normals per particle, [3p + k].
template <bool EXPLICIT, bool LAZY>
__global__ __launch_bounds__(kBlock) void k_fs2_predict(Planes pl, const Ctl* __restrict__ ctl, uint64_t n, double u0,
                                                       double u1, rr_fs2_model m, uint64_t seed, unsigned int step,
                                                       const double* __restrict__ noise,
                                                       const unsigned int* __restrict__ idx, uint64_t gid0, int has_obs,
                                                       double zd, double za, uint64_t id0) {
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const bool pending = LAZY && ctl->pending;
  double* __restrict__ dst = pl.s[pending ? ctl->cur ^ 1 : ctl->cur];
  const unsigned int ji = pending ? idx[p] : (unsigned int)p;
  const bool inplace = pending && ji == kInPlace;
  const double* __restrict__ src = inplace ? pl.inbox : pl.s[ctl->cur];
  const uint64_t j = inplace ? p : ji;
  double pose[3] = {src[j], src[n + j], src[2 * n + j]};
  double lm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (has_obs) {
    const double* in = src + (3 + id0 * 6) * n + j;
#pragma unroll
    for (int f = 0; f < 6; ++f) lm[f] = in[f * n];
  }
  double z[3];
  if (EXPLICIT) {
    z[0] = noise[3 * p];
    z[1] = noise[3 * p + 1];
    z[2] = noise[3 * p + 2];
  } else {
    rr_fs2_noise(seed, step, gid0 + p, z);
  }
  rr_fs2_predict_one(pose, u0, u1, has_obs, zd, za, lm, z, m);
  dst[p] = pose[0];
  dst[n + p] = pose[1];
  dst[2 * n + p] = pose[2];
}

// (particle, observation chunk).  blockIdx.y = chunk.  The chunk's observations are staged in
// LDS and read back with wave-uniform addresses; each update loads the 6 planes of the observed
// landmark for 64 consecutive particles (coalesced), runs the 2x2 EKF of rr_fs1_update_one and
// stores the six fields back (a field the update left alone is rewritten with the same bits).
// SEQ (a landmark id repeats inside the step; host side: one chunk, in place, nothing pending): the
// second update of a landmark must see the first one's result (fastslam1.rs:250-256 runs the
// observations one after the other), so every update loads its planes only after the previous
// update's stores -- no software pipeline, no restrict-qualified alias of the live set.
// VAR: tuning variants of the same arithmetic (RR_FS1_VARIANT; identical results):
//   bit 0  non-temporal stores while a pending resample is being consumed (separate read and write streams)
//   bit 1  non-temporal loads on that path as well
//   bit 2  no software pipeline (loads at the top of each update)
//   bit 3  register budget for 4 waves per SIMD instead of 3
// (The memory pattern alone -- same grid, same pipeline, synthetic arithmetic -- is tools/ubench/plane_layout.hip: 5.2-5.4 TB/s
// whether 0 or 300 FMAs sit between a wave's loads and its stores; this kernel runs at 5.25 TB/s.)
constexpr int kObsNtStore = 1, kObsNtLoad = 2, kObsNoPipe = 4, kObsFourWaves = 8;
constexpr uint64_t kNoFactor = 0x7FF45EA1ED000001ull;  // "no factor here yet": a signalling NaN (k_fs1_observe)
// ASSUMPTION: the closing chunk's workgroups of a particle block may wait for the factors of the block's other
// chunks because those have LOWER workgroup indices in the same one-dimensional launch and the hardware dispatches a grid's
// workgroups in ascending index order (round-robin over the XCDs): whoever is waited for has at least been handed to a CU
// before the waiter exists.  HIP does not promise that order.  The wait is therefore bounded: after kFactorWaitTicks the weight
// of the particle is NaN and Ctl.obs_timeout is latched -- every later accessor and synchronous call of the handle reports
// RR_RUNTIME_ERROR ("a chunk's weight factor did not arrive") instead of returning numbers.  Never observed (the soak runs of
// rounds 2-4: > 10^7 launches); a handle that must not depend on the assumption takes obs_chunks = 1 (rr_fs1_options).
constexpr uint64_t kFactorWaitTicks = 200000000ull;    // 2 s of the 100 MHz wall clock

__global__ __launch_bounds__(kBlock) void k_fs1_no_factors(uint64_t* __restrict__ partial, uint64_t words) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < words; i += (uint64_t)gridDim.x * kBlock) partial[i] = kNoFactor;
}

template <bool LAZY, bool SEQ, int VAR>
__global__ __launch_bounds__(kBlock, (VAR & kObsFourWaves) ? 4 : 1) void k_fs1_observe(
    Planes pl, double* __restrict__ pw, Ctl* __restrict__ ctl, uint64_t n, const double* __restrict__ z, int n_z, int chunk_len,
    int n_chunks, rr_fs1_model m, double* partial /* written by the other chunks while the closing one reads */,
    const unsigned int* __restrict__ idx, unsigned int n_pblocks) {
  extern __shared__ double s_z[];
  __shared__ double s_wmax[kBlock / rr::kWave];
  // Dispatch order (one-dimensional grid of n_pblocks x n_chunks workgroups): chunk by chunk, a chunk's particle blocks side
  // by side -- the order the memory system likes (block by block, all chunks of a block side by side: 20 % slower; bands of
  // blocks, each band chunk by chunk: no better, RR_FS1_BANDS experiment of round 3).  A block's last chunk is therefore
  // dispatched after its other chunks; it forms the block's weights (below).
  const int chunk = (int)(blockIdx.x / n_pblocks);
  const unsigned int pblock = blockIdx.x % n_pblocks;
  const int k0 = chunk * chunk_len;
  const int k1 = min(k0 + chunk_len, n_z);
  for (int i = threadIdx.x; i < 3 * (k1 - k0); i += kBlock) s_z[i] = z[3 * k0 + i];
  __syncthreads();
  const uint64_t p = (uint64_t)pblock * kBlock + threadIdx.x;
  double acc = 0.0;
  if (p < n) {
    // LAZY + pending: the maps of slot p still sit at particle idx[p] of the live set; every observed
    // landmark is read from there and written (all six fields) to slot p of the other set, which
    // folds the resample gather of the observed landmarks into this kernel's own traffic.  The
    // pose was already moved by k_fs1_predict<LAZY>.
    const bool pending = LAZY && ctl->pending;
    const bool nt_st = pending && (VAR & kObsNtStore), nt_ld = pending && (VAR & kObsNtLoad);
    double* __restrict__ dst = pl.s[pending ? ctl->cur ^ 1 : ctl->cur];
    const unsigned int ji = pending ? idx[p] : (unsigned int)p;
    const bool inplace = pending && ji == kInPlace;
    const double* src = inplace ? pl.inbox : pl.s[ctl->cur];
    const uint64_t j = inplace ? p : ji;
    const double px = dst[p], py = dst[n + p], pyaw = dst[2 * n + p];
    acc = chunk == 0 ? pw[p] : 1.0;
    const int nk = k1 - k0;
    auto load6 = [&](const double* in, double* v) {
      if (nt_ld) {
#pragma unroll
        for (int f = 0; f < 6; ++f) v[f] = __builtin_nontemporal_load(in + f * n);
      } else {
#pragma unroll
        for (int f = 0; f < 6; ++f) v[f] = in[f * n];
      }
    };
    auto store6 = [&](double* out, const double* v) {
      if (nt_st) {
#pragma unroll
        for (int f = 0; f < 6; ++f) __builtin_nontemporal_store(v[f], out + f * n);
      } else {
#pragma unroll
        for (int f = 0; f < 6; ++f) out[f * n] = v[f];
      }
    };
    if (SEQ) {
      double* live = pl.s[ctl->cur];
      for (int k = 0; k < nk; ++k) {
        const double zd = s_z[3 * k], za = s_z[3 * k + 1];
        double* io = live + (3 + (uint64_t)s_z[3 * k + 2] * 6) * n + p;
        double e[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) e[f] = io[f * n];
        acc *= rr_fs1_update_one(px, py, pyaw, zd, za, e, m);
#pragma unroll
        for (int f = 0; f < 6; ++f) io[f * n] = e[f];
      }
    } else if (VAR & kObsNoPipe) {
      for (int k = 0; k < nk; ++k) {
        const double zd = s_z[3 * k], za = s_z[3 * k + 1];
        const uint64_t id = (uint64_t)s_z[3 * k + 2];
        double e[6];
        load6(src + (3 + id * 6) * n + j, e);
        acc *= rr_fs1_update_one(px, py, pyaw, zd, za, e, m);
        store6(dst + (3 + id * 6) * n + p, e);
      }
    } else {
      // software pipeline: the six plane loads of observation k+1 are issued before the FP64
      // instructions of update k, so two updates' worth of HBM requests are in flight per wave
      double nxt[6];
      if (nk > 0) load6(src + (3 + (uint64_t)s_z[2] * 6) * n + j, nxt);
      for (int k = 0; k < nk; ++k) {
        const double zd = s_z[3 * k], za = s_z[3 * k + 1];
        const uint64_t id = (uint64_t)s_z[3 * k + 2];
        double e[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) e[f] = nxt[f];
        if (k + 1 < nk) load6(src + (3 + (uint64_t)s_z[3 * k + 5] * 6) * n + j, nxt);
        acc *= rr_fs1_update_one(px, py, pyaw, zd, za, e, m);
        store6(dst + (3 + id * 6) * n + p, e);
      }
    }
  }
  // The weight is the product of the chunks' factors in chunk order (the D-spec's).  The LAST chunk's workgroup of a
  // particle block forms it: the other chunks leave their factor in `partial` and are done -- no wait, no ticket, nothing at
  // their end -- and the last chunk, dispatched after the others, reads the factors back when its own updates are through.
  // A factor that has not landed yet shows as kNoFactor, a signalling-NaN pattern no arithmetic produces (results are quiet
  // NaNs); the reader looks again (bounded: Ctl.obs_timeout) and puts the pattern back for the next update.  No
  // k_fs1_combine launch (round 2: 10.8 - 12.5 us + a launch boundary at 1e5 particles x 25 chunks).
  const bool closing = chunk == n_chunks - 1;  // uniform per workgroup
  if (!closing) {
    if (p < n) rr::st_dev(reinterpret_cast<uint64_t*>(&partial[(uint64_t)chunk * n + p]), rr_d2u(acc));  // device scope: read from any XCD
  } else if (n_chunks > 1) {
    if (p < n) {
      double w = 1.0;
      constexpr int kBatch = 16;  // loads in flight
      for (int c0 = 0; c0 < n_chunks - 1; c0 += kBatch) {
        uint64_t f[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
          f[k] = c0 + k < n_chunks - 1 ? rr::ld_dev(reinterpret_cast<const uint64_t*>(&partial[(uint64_t)(c0 + k) * n + p])) : rr_d2u(1.0);
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          if (c0 + k >= n_chunks - 1) continue;  // a chunk past the end contributes an exact factor 1
          uint64_t* slot = reinterpret_cast<uint64_t*>(&partial[(uint64_t)(c0 + k) * n + p]);
          if (f[k] == kNoFactor) {
            const uint64_t t0 = wall_clock64();
            while ((f[k] = rr::ld_dev(slot)) == kNoFactor) {
              __builtin_amdgcn_s_sleep(8);
              if (wall_clock64() - t0 > kFactorWaitTicks) {
                ctl->obs_timeout = 1;
                break;
              }
            }
          }
          rr::st_dev(slot, kNoFactor);
          w = (c0 + k == 0) ? rr_u2d(f[k]) : w * rr_u2d(f[k]);
        }
      }
      acc = w * acc;  // this chunk's own factor is the last one
      pw[p] = acc;
    }
  } else if (p < n) {
    pw[p] = acc;
  }
  if (closing) {
    double mx = acc > 0.0 ? acc : 0.0;
    mx = rr::wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      double bm = s_wmax[0];
      for (int k = 1; k < kBlock / rr::kWave; ++k) bm = s_wmax[k] > bm ? s_wmax[k] : bm;
      if (bm > 0.0) rr::atomic_max_u64(&ctl->wmax_bits, rr_d2u(bm));
    }
  }
}

// max only (after set_state)
__global__ __launch_bounds__(kBlock) void k_fs1_wmax(const double* __restrict__ pw, Ctl* __restrict__ ctl, uint64_t n) {
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  double w = p < n ? pw[p] : 0.0;
  double mx = w > 0.0 ? w : 0.0;
  mx = rr::wave_max(mx);
  if ((threadIdx.x & 63) == 0 && mx > 0.0) rr::atomic_max_u64(&ctl->wmax_bits, rr_d2u(mx));
}

// fastslam1.rs:196-203 when no resample follows: w /= sum iff the sum is positive
__global__ __launch_bounds__(kBlock) void k_fs1_normalize(double* __restrict__ pw, const Ctl* __restrict__ ctl, uint64_t n) {
  if (ctl->fired || ctl->image_mode != rr::kImageWeights) return;
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p < n) pw[p] = pw[p] / ctl->sum;
}

// Single-GPU plan, fused (the MCL engine's k_plan_mark shape, resample_core.hpp): every workgroup re-derives its tile
// offset and the grand totals from the tile totals, takes the gate decision (fastslam1.rs:262-265) and then either
// normalises its tile's weights (:196-203, gate shut) or marks the slot run each of its sources feeds (:205-234) and
// sets the weights to 1/n (:228).  One launch instead of k_scan_tiles + k_cdf + k_fs1_normalize + k_fs1_indices, no
// CDF array, no per-slot binary search.  k_fs1_resolve turns the markers into idx[] (running maximum per 512 slots).
__global__ __launch_bounds__(rr::kTileBlock) void k_fs1_plan(double* pw /* read (tile_scan) and rewritten: no restrict */, Ctl* __restrict__ ctl, ImageArgs a,
                                                    const uint64_t* __restrict__ tile_total,
                                                    const uint64_t* __restrict__ tile_q2, uint64_t n_tiles, PlanArgs pa,
                                                    unsigned int* __restrict__ markers, unsigned int* __restrict__ carry) {
  __shared__ uint64_t s4[4 * (rr::kTileBlock / rr::kWave)];
  __shared__ uint64_t s_w[rr::kTileBlock / rr::kWave];
  const rr::TileSums ts = rr::tile_sums(tile_total, tile_q2, n_tiles, s4);
  const int mode = ctl->image_mode;
  const int shift = ctl->shift;
  const int fire = rr::gate_decision(mode, ts, pa);
  double rho = pa.rho_override;
  if (rho != rho) {
    double dummy;
    rr_uniform2(pa.seed, RR_STREAM_RESAMPLE, pa.rstep, 0, &rho, &dummy);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rr::finalize_plan(ctl, ts.tot, 0, ts.tot, ts.q2, pa);
  const uint64_t i0 = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * rr::kItems;
  if (!fire) {
    if (mode != rr::kImageWeights) return;  // all-zero weights stay untouched (fastslam1.rs:198-202)
    const double sum = rr_fix_total_to_double(ts.tot, shift);
#pragma unroll
    for (int j = 0; j < rr::kItems; ++j)
      if (i0 + j < a.n) pw[i0 + j] = pw[i0 + j] / sum;
    return;
  }
  const rr_sys_plan plan = rr_sys_plan_make(rho, ts.tot, pa.n_global);
  const rr::TileScan t = rr::tile_scan(pw, a, mode, shift, blockIdx.x, s_w);  // reads this tile's weights ...
  rr::mark_sources(t, ts.pre + t.thread_off, i0, a.n, plan, ts.tot, 0, markers, carry);
  const double w_new = 1.0 / (double)pa.n_global;
#pragma unroll
  for (int j = 0; j < rr::kItems; ++j)
    if (i0 + j < a.n) pw[i0 + j] = w_new;  // ... before the same threads overwrite them
}

__global__ __launch_bounds__(kBlock) void k_fs1_resolve(const Ctl* __restrict__ ctl, unsigned int* __restrict__ markers,
                                                       const unsigned int* __restrict__ carry, uint64_t n,
                                                       unsigned int* __restrict__ idx) {
  if (!ctl->fired) return;
  unsigned int src[rr::kResolveRows];
  rr::resolve_tile(markers, carry, n, blockIdx.x, src);
#pragma unroll
  for (int r = 0; r < rr::kResolveRows; ++r) {
    const uint64_t k = (uint64_t)blockIdx.x * rr::kResolveSlots + (uint64_t)r * kBlock + threadIdx.x;
    if (k < n) idx[k] = src[r];
  }
}

// sharded variants.  Slots this shard serves: [first, first + n_served) with
// first = slots_upto(base), n_served = slots_upto(base + T_local) - first (device-side only).
__device__ inline void served_range(const Ctl* ctl, uint64_t* first, uint64_t* n_served) {
  *first = ctl->served_first;
  *n_served = ctl->served_count;
}

// remote-owned served slots are the two ends of the served range: k in [0, lead) and
// [tail_start, n_served), with lead / tail_start = where this rank's own slots begin / end in
// served-slot coordinates
struct ServedSplit {
  uint64_t first, n_served, lead, tail_start, n_remote;
};
__device__ inline ServedSplit served_split(const Ctl* ctl, uint64_t own_first, uint64_t n_local) {
  ServedSplit v;
  served_range(ctl, &v.first, &v.n_served);
  const uint64_t end = v.first + v.n_served;
  const uint64_t lo = own_first < v.first ? v.first : (own_first > end ? end : own_first);
  const uint64_t own_end = own_first + n_local;
  const uint64_t hi = own_end < v.first ? v.first : (own_end > end ? end : own_end);
  v.lead = lo - v.first;
  v.tail_start = hi - v.first;
  v.n_remote = v.lead + (v.n_served - v.tail_start);
  return v;
}

// workgroups [0, own_blocks): one thread per OWN slot -- its local source if this shard serves
// it, kInPlace if a peer does.  Workgroups beyond: the sources of the served slots that belong to
// peers (ridx, grid-stride; there are few in steady state).
__global__ __launch_bounds__(kBlock) void k_fs1_indices_sharded(const Ctl* __restrict__ ctl,
                                                               const uint64_t* __restrict__ cdf, uint64_t n,
                                                               uint64_t own_first, unsigned int own_blocks,
                                                               unsigned int* __restrict__ idx,
                                                               unsigned int* __restrict__ ridx) {
  if (!ctl->fired) return;
  const rr_sys_plan plan = ctl->plan;
  const ServedSplit v = served_split(ctl, own_first, n);
  if (blockIdx.x < own_blocks) {
    const uint64_t li = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (li >= n) return;
    const uint64_t s = own_first + li;
    idx[li] = (s >= v.first && s < v.first + v.n_served) ? (unsigned int)rr_lower_bound_u64(cdf, n, rr_sys_target(plan, s))
                                                         : kInPlace;
    return;
  }
  const uint64_t stride = (uint64_t)(gridDim.x - own_blocks) * kBlock;
  for (uint64_t r = (uint64_t)(blockIdx.x - own_blocks) * kBlock + threadIdx.x; r < v.n_remote; r += stride) {
    const uint64_t k = r < v.lead ? r : v.tail_start + (r - v.lead);
    ridx[r] = (unsigned int)rr_lower_bound_u64(cdf, n, rr_sys_target(plan, v.first + k));
  }
}

__global__ __launch_bounds__(kBlock) void k_fs1_uniform_weights(const Ctl* __restrict__ ctl, double* __restrict__ pw,
                                                               uint64_t n, uint64_t n_global) {
  if (!ctl->fired) return;
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p < n) pw[p] = 1.0 / (double)n_global;  // fastslam1.rs:228
}

// the served slots that belong to peers: all planes of the source particle into slot li of the
// owner's fine-grained inbox (the owner's idx says kInPlace)
__global__ __launch_bounds__(kBlock) void k_fs1_push(Planes pl, const Ctl* __restrict__ ctl,
                                                    const unsigned int* __restrict__ ridx, uint64_t n_local,
                                                    uint64_t own_first, uint64_t n_planes, rr::P2PPeers peers) {
  if (!ctl->fired) return;
  const ServedSplit v = served_split(ctl, own_first, n_local);
  const int cur = ctl->cur;  // lazy: not flipped yet
  const double* __restrict__ in = pl.s[cur];
  const uint64_t p0 = (uint64_t)blockIdx.y * kPlanesPerThread;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < v.n_remote; r += (uint64_t)gridDim.x * kBlock) {
    const uint64_t k = r < v.lead ? r : v.tail_start + (r - v.lead);
    const uint64_t s = v.first + k;
    const uint64_t d = s / n_local, li = s - d * n_local;
    const uint64_t j = ridx[r];
    double* __restrict__ out = peers.inbox[d];  // fine-grained, [plane][n_local]
    double val[kPlanesPerThread];
#pragma unroll
    for (int q = 0; q < kPlanesPerThread; ++q)
      if (p0 + q < n_planes) val[q] = in[(p0 + q) * n_local + j];
#pragma unroll
    for (int q = 0; q < kPlanesPerThread; ++q)
      if (p0 + q < n_planes) out[(p0 + q) * n_local + li] = val[q];
  }
}

__global__ __launch_bounds__(kBlock) void k_fs1_indices(const Ctl* __restrict__ ctl, const uint64_t* __restrict__ cdf,
                                                       uint64_t n, unsigned int* __restrict__ idx,
                                                       double* __restrict__ pw) {
  if (!ctl->fired) return;
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint64_t target = rr_sys_target(ctl->plan, k);
  idx[k] = (unsigned int)rr_lower_bound_u64(cdf, n, target);
  pw[k] = 1.0 / (double)n;  // fastslam1.rs:228
}

// blockIdx.y selects a group of kPlanesPerThread planes; indices are non-decreasing, so each
// plane is read almost sequentially.  mode 0 (eager): the plan kernel flipped Ctl.cur already --
// read set cur^1, write set cur, runs iff fired.  mode 1 (lazy): a pending resample is being
// consumed -- read set cur, write set cur^1, runs iff pending (a settle flips afterwards).
// plane_list != nullptr restricts the copy to the listed planes (the landmarks a lazy observe
// did not touch).
__global__ __launch_bounds__(kBlock) void k_fs1_gather(Planes pl, const Ctl* __restrict__ ctl,
                                                      const unsigned int* __restrict__ idx, uint64_t n,
                                                      uint64_t n_planes, int lazy,
                                                      const unsigned int* __restrict__ plane_list) {
  if (lazy ? !ctl->pending : !ctl->fired) return;
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const int cur = ctl->cur;
  const double* in = pl.s[lazy ? cur : cur ^ 1];
  double* __restrict__ out = pl.s[lazy ? cur ^ 1 : cur];
  uint64_t j = idx[k];
  if (lazy && idx[k] == kInPlace) {  // sharded: delivered by a peer -- take it out of the inbox
    in = pl.inbox;
    j = k;
  }
  const uint64_t p0 = (uint64_t)blockIdx.y * kPlanesPerThread;
  double v[kPlanesPerThread];
  uint64_t pid[kPlanesPerThread];
#pragma unroll
  for (int q = 0; q < kPlanesPerThread; ++q) {
    pid[q] = p0 + q < n_planes ? (plane_list ? plane_list[p0 + q] : p0 + q) : 0;
    if (p0 + q < n_planes) v[q] = in[pid[q] * n + j];
  }
#pragma unroll
  for (int q = 0; q < kPlanesPerThread; ++q)
    if (p0 + q < n_planes) out[pid[q] * n + k] = v[q];
}

// ---- RCCL / host-orchestrated transport (rr_fs1_shard_update): whole particles that cross ranks travel
// as one contiguous block per (source, destination) pair, laid out [plane][count] so that both the
// packing reads (non-decreasing source indices along a plane) and the unpacking writes are coalesced.
// ChunkTable: start[g] .. start[g + 1] = the block of rank g in remote-slot order (ascending global
// slot, this rank's own slots left out); first_local[g] = local slot the block of SOURCE g starts at.
struct ChunkTable {
  uint64_t start[rr::kMaxP2P + 1];
  uint64_t first_local[rr::kMaxP2P];
  int n_ranks;
};

__device__ inline int chunk_of(const ChunkTable& t, uint64_t r) {
  int g = 0;
  while (g + 1 < t.n_ranks && r >= t.start[g + 1]) ++g;
  return g;
}

// served slots that belong to peers -> send buffer.  ridx[r] = local source of remote slot r (k_fs1_indices_sharded).
__global__ __launch_bounds__(kBlock) void k_fs1_pack(Planes pl, const Ctl* __restrict__ ctl,
                                                    const unsigned int* __restrict__ ridx, uint64_t n_local,
                                                    uint64_t n_planes, ChunkTable t, double* __restrict__ out) {
  if (!ctl->fired) return;
  const double* __restrict__ in = pl.s[ctl->cur];  // lazy: not flipped yet
  const uint64_t n_remote = t.start[t.n_ranks];
  const uint64_t p0 = (uint64_t)blockIdx.y * kPlanesPerThread;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n_remote; r += (uint64_t)gridDim.x * kBlock) {
    const int g = chunk_of(t, r);
    const uint64_t cnt = t.start[g + 1] - t.start[g], k = r - t.start[g];
    const uint64_t j = ridx[r];
    double* __restrict__ o = out + t.start[g] * n_planes + k;
    double val[kPlanesPerThread];
#pragma unroll
    for (int q = 0; q < kPlanesPerThread; ++q)
      if (p0 + q < n_planes) val[q] = in[(p0 + q) * n_local + j];
#pragma unroll
    for (int q = 0; q < kPlanesPerThread; ++q)
      if (p0 + q < n_planes) o[(p0 + q) * cnt] = val[q];
  }
}

// receive buffer -> this rank's inbox (the owner's idx says kInPlace for exactly these slots)
__global__ __launch_bounds__(kBlock) void k_fs1_unpack(double* __restrict__ inbox, const Ctl* __restrict__ ctl,
                                                      uint64_t n_local, uint64_t n_planes, ChunkTable t,
                                                      const double* __restrict__ in) {
  if (!ctl->fired) return;
  const uint64_t n_remote = t.start[t.n_ranks];
  const uint64_t p0 = (uint64_t)blockIdx.y * kPlanesPerThread;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n_remote; r += (uint64_t)gridDim.x * kBlock) {
    const int g = chunk_of(t, r);
    const uint64_t cnt = t.start[g + 1] - t.start[g], k = r - t.start[g];
    const uint64_t li = t.first_local[g] + k;
    const double* __restrict__ i0 = in + t.start[g] * n_planes + k;
#pragma unroll
    for (int q = 0; q < kPlanesPerThread; ++q)
      if (p0 + q < n_planes) inbox[(p0 + q) * n_local + li] = i0[(p0 + q) * cnt];
  }
}

__global__ void k_fs1_settle(Ctl* ctl) {
  if (ctl->pending) {
    ctl->cur ^= 1;
    ctl->pending = 0;
  }
}

// arg max of the weight with ties -> highest index (fastslam1.rs:269-274, Q14): the key (weight bits, index) is order
// preserving for non-negative doubles.  ONE launch: every workgroup leaves its best key, the workgroup that takes the last
// ticket picks the best of those and leaves index, pose and weight of that particle in Ctl and in the handle's
// host-visible mailbox (pinned memory; the host polls its stamp -- no device-to-host copies, no stream synchronisation).
// Round 2: two launches, a read-back of Ctl and four 8-byte copies behind a second synchronisation -- 95 us per call, more
// than the reference spends on a whole update of its 100 particles.
struct BestMail {
  uint64_t index;
  double pose[3];
  double weight;
  uint64_t flags;  // != 0: Ctl holds something the host has to look at (obs_timeout, grid_timeout)
  uint64_t seq;    // stamped last
};
__global__ __launch_bounds__(kBlock) void k_fs1_best(const double* __restrict__ pw, uint64_t n, const double* __restrict__ planes0,
                                                    const double* __restrict__ planes1, Ctl* __restrict__ ctl,
                                                    uint64_t* __restrict__ part_bits, uint64_t* __restrict__ part_idx,
                                                    unsigned int* __restrict__ ticket, BestMail* __restrict__ mail, uint64_t seq,
                                                    const unsigned int* __restrict__ idx) {
  __shared__ uint64_t s_b[kBlock / rr::kWave], s_i[kBlock / rr::kWave];
  __shared__ int s_last;
  auto better = [](uint64_t ob, uint64_t oi, uint64_t bb, uint64_t bi) { return ob > bb || (ob == bb && oi > bi); };
  auto wave_best = [&](uint64_t& bb, uint64_t& bi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ob = rr::shfl_xor_u64(bb, o), oi = rr::shfl_xor_u64(bi, o);
      if (better(ob, oi, bb, bi)) {
        bb = ob;
        bi = oi;
      }
    }
  };
  auto block_best = [&](uint64_t& bb, uint64_t& bi) {  // thread 0 ends up with the workgroup's best
    wave_best(bb, bi);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      s_b[threadIdx.x >> 6] = bb;
      s_i[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 1; k < kBlock / rr::kWave; ++k)
        if (better(s_b[k], s_i[k], bb, bi)) {
          bb = s_b[k];
          bi = s_i[k];
        }
  };
  uint64_t bb = 0, bi = 0;  // (an empty share ranks lowest: key (0, 0))
  for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (uint64_t)gridDim.x * kBlock) {
    const double w = pw[p];
    const uint64_t b = w > 0.0 ? rr_d2u(w) : 0ull;  // NaN / negative weights rank lowest
    if (better(b, p, bb, bi) || (b == bb && p == bi)) {
      bb = b;
      bi = p;
    }
  }
  block_best(bb, bi);
  if (threadIdx.x == 0) {
    rr::st_dev(&part_bits[blockIdx.x], bb);
    rr::st_dev(&part_idx[blockIdx.x], bi);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int before = atomicAdd(ticket, 1u);
    s_last = before == gridDim.x - 1 ? 1 : 0;
    if (s_last) *ticket = 0;
  }
  __syncthreads();
  if (!s_last) return;
  bb = 0;
  bi = 0;
  for (unsigned int k = threadIdx.x; k < gridDim.x; k += kBlock) {
    const uint64_t ob = rr::ld_dev(&part_bits[k]), oi = rr::ld_dev(&part_idx[k]);
    if (better(ob, oi, bb, bi)) {
      bb = ob;
      bi = oi;
    }
  }
  block_best(bb, bi);
  if (threadIdx.x == 0) {
    // a pending (lazy) resample has set the weights already; the particle of slot bi still sits at its source in the live set
    const double* __restrict__ live = ctl->cur ? planes1 : planes0;
    const uint64_t j = (idx && ctl->pending) ? (uint64_t)idx[bi] : bi;
    const double x = live[j], y = live[n + j], yaw = live[2 * n + j], w = pw[bi];
    ctl->best_bits = bb;
    ctl->best_index = bi;
    if (mail) {
      rr::st_sys_u64(&mail->index, bi);
      rr::st_sys(&mail->pose[0], x);
      rr::st_sys(&mail->pose[1], y);
      rr::st_sys(&mail->pose[2], yaw);
      rr::st_sys(&mail->weight, w);
      rr::st_sys_u64(&mail->flags, (uint64_t)(ctl->obs_timeout != 0) | ((uint64_t)(ctl->grid_timeout != 0) << 1));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stamp goes last
      rr::st_sys_u64(&mail->seq, seq);
    }
  }
}

// "everything before me in this stream is done, and this is what Ctl has to report": the stamp of a synchronous call
// (rr_fs1_update, rr_fs1_synchronize) -- the host polls the mailbox instead of copying Ctl back behind a stream synchronisation
__global__ void k_fs1_stamp(const Ctl* __restrict__ ctl, BestMail* __restrict__ mail, uint64_t seq) {
  rr::st_sys_u64(&mail->flags, (uint64_t)(ctl->obs_timeout != 0) | ((uint64_t)(ctl->grid_timeout != 0) << 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  rr::st_sys_u64(&mail->seq, seq);
}

// ------------------------------------------------------------------------------------------
// RESIDENT update of a small particle set (rr_fs1_set_resident; resident_core.hpp): fastslam_update (fastslam1.rs:237-266)
// followed by get_best_particle (:269-274) -- the loop every FastSLAM caller in the reference runs, at its sizes (100
// particles x 8 landmarks: render_gif_slam.rs:166-200) -- for ONE workgroup that stays on the device and takes one update per
// command of the ring: (u0, u1, chunk length, n_z x (distance, angle, landmark id)).  The general path spends three
// launches on an update and a fourth on the best particle (19 + 17 us for a few microseconds of work); here a thread owns a
// particle, the poses, weights and maps stay where they are (HBM / L2: one workgroup reads its own writes after a barrier),
// and the resample gather is EAGER, so an incarnation can leave at any command boundary with nothing pending.
// Same per-element arithmetic (rr_fs1_predict_one, rr_fs1_update_one), the weight as the left-to-right product of the
// observation chunks' factors (choose_chunks' plan, handed over by the host), the same integer image, gate, systematic plan
// and tie rule: bit-identical to the launched path (tests/test_gpu_fs1_resident.py).
struct Fs1SmallArgs {
  uint64_t n, L, n_global, gid0;
  uint64_t seed;
  unsigned int step0, rstep0;
  rr_fs1_model m;
  PlanArgs plan;  // mode 0, N_eff gate against NTH, systematic, eager gather
  rr::ResidentArgs res;
};
constexpr int kFs1ResMaxObs = 64;
constexpr int kFs1ResPayload = 3 + 3 * kFs1ResMaxObs;
constexpr int kFs1RspIndex = 6;  // rsp[0..2] pose, [3] weight, [4] flags, [5] EXIT marker, [6] index of the best particle

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fs1_small(Planes pl, double* __restrict__ pw, Ctl* __restrict__ ctl, Fs1SmallArgs a,
                                                    unsigned int* __restrict__ idx_out, rr::ResidentRing* __restrict__ ring) {
  constexpr int W = BLOCK / rr::kWave;
  __shared__ double s_pay[kFs1ResPayload + 1];
  __shared__ int s_hdr[2];
  __shared__ double s_red[W];
  __shared__ uint64_t s_u[3 * W];
  __shared__ uint64_t s_bb[W], s_bi[W];
  __shared__ unsigned int s_mx[W];
  __shared__ unsigned int s_mark[BLOCK + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t n = a.n, p = (uint64_t)tid;
  const bool mine = p < n;
  int res_guess = 28, res_last_op = rr::kResOpNone, steps_done = 0;
  const uint64_t deadline = wall_clock64() + a.res.life_ticks;
  for (int s = 0;; ++s) {
    // ---- the update's motion noise does not depend on the command: drawn before the wait
    double na = 0.0, nb = 0.0;
    if (mine) rr_fs1_motion_noise(a.seed, a.step0 + (unsigned int)s, a.gid0 + p, &na, &nb);
    double rho, rho_dummy;
    rr_uniform2(a.plan.seed, RR_STREAM_RESAMPLE, a.rstep0 + (unsigned int)s, 0, &rho, &rho_dummy);
    res_last_op = rr::resident_fetch<BLOCK>(ring, a.res.first_seq + (uint64_t)s, a.res.idle_ticks, deadline, kFs1ResPayload, res_guess, s_pay, s_hdr);
    if (res_last_op != rr::kResOpStep) break;
    steps_done = s + 1;
    const int n_z = (s_hdr[1] - 3) / 3;
    const double u0 = s_pay[0], u1 = s_pay[1];
    const int chunk_len = (int)s_pay[2];
    const double* s_z = s_pay + 3;
    const int cur = ctl->cur;
    double* __restrict__ live = pl.s[cur];
    // ---- predict_particle (fastslam1.rs:123-137)
    double px = 0.0, py = 0.0, pyaw = 0.0, w = 0.0;
    if (mine) {
      px = live[p];
      py = live[n + p];
      pyaw = live[2 * n + p];
      rr_fs1_predict_one(&px, &py, &pyaw, u0, u1, na, nb, a.m);
      live[p] = px;
      live[n + p] = py;
      live[2 * n + p] = pyaw;
      // ---- update_landmark per observation (:140-183); the weight: chunk factors multiplied left to right (k_fs1_observe)
      double total_w = pw[p];  // (no observation: the weight stays as it is)
      for (int k0 = 0, c = 0; k0 < n_z; k0 += chunk_len, ++c) {
        double acc = c == 0 ? total_w : 1.0;
        const int k1 = k0 + chunk_len < n_z ? k0 + chunk_len : n_z;
        for (int k = k0; k < k1; ++k) {
          double* io = live + (3 + (uint64_t)s_z[3 * k + 2] * 6) * n + p;
          double e[6];
#pragma unroll
          for (int f = 0; f < 6; ++f) e[f] = io[f * n];
          acc *= rr_fs1_update_one(px, py, pyaw, s_z[3 * k], s_z[3 * k + 1], e, a.m);
#pragma unroll
          for (int f = 0; f < 6; ++f) io[f * n] = e[f];
        }
        total_w = c == 0 ? acc : total_w * acc;
      }
      w = total_w;
    }
    // ---- maximum, integer image, sums (k_quantize_plan_mark<true>'s phase A)
    double wl = w > 0.0 ? w : 0.0;
    wl = rr::wave_max(wl);
    __syncthreads();
    if (lane == 0) s_red[wv] = wl;
    __syncthreads();
    double wmax = s_red[0];
#pragma unroll
    for (int k = 1; k < W; ++k) wmax = s_red[k] > wmax ? s_red[k] : wmax;
    const bool usable = wmax > 0.0 && wmax < INFINITY;
    const int mode = usable ? (int)rr::kImageWeights : (int)rr::kImageLast;
    const int shift = usable ? rr_fix_shift(wmax, a.n_global) : 0;
    const uint64_t q = !mine ? 0ull : (mode == rr::kImageWeights ? rr_fix_quantize(w, shift) : (a.gid0 + p == a.n_global - 1 ? 1ull : 0ull));
    rr::u128 q2;
    rr_mul64wide(q, q, &q2.hi, &q2.lo);
    const uint64_t incl = rr::wave_scan_u64(q, lane);
    q2 = rr::wave_sum_u128(q2);
    __syncthreads();
    if (lane == 63) s_u[wv] = incl;
    if (lane == 0) {
      s_u[W + wv] = q2.hi;
      s_u[2 * W + wv] = q2.lo;
    }
    __syncthreads();
    uint64_t off = incl - q, total = 0;
    rr::u128 qq = {0, 0};
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (k < wv) off += s_u[k];
      total += s_u[k];
      qq = rr::add128(qq, rr::u128{s_u[W + k], s_u[2 * W + k]});
    }
    rr::TileSums ts;
    ts.pre = 0;
    ts.tot = total;
    ts.q2 = qq;
    PlanArgs pa = a.plan;
    pa.rstep = a.rstep0 + (unsigned int)s;
    const int fire = rr::gate_decision(mode, ts, pa);
    if (tid == 0) {  // what the plan kernel's first workgroup leaves in Ctl
      ctl->usable = usable ? 1 : 0;
      ctl->image_mode = mode;
      ctl->shift = shift;
      ctl->wmax = wmax;
      rr::finalize_plan(ctl, total, 0, total, qq, pa);  // eager: Ctl.cur flips here when the gate fires
    }
    int cur_now = cur;
    if (!fire) {  // fastslam1.rs:196-203: w /= sum iff the sum is positive (all-zero weights stay untouched)
      if (mode == rr::kImageWeights) w = w / rr_fix_total_to_double(total, shift);
      if (mine) pw[p] = w;
    } else {  // fastslam1.rs:205-234: systematic walk -> slot-run markers -> running maximum, then every plane moves
      const rr_sys_plan plan = rr_sys_plan_make(rho, total, a.n_global);
      const rr_sys_inv inv = rr_sys_inv_make(plan, total);
      __syncthreads();
      for (uint64_t k = tid; k <= n; k += BLOCK) s_mark[k] = 0;
      __syncthreads();
      if (mine && q != 0) {
        const uint64_t h0 = rr_sys_slots_upto(plan, inv, total, off), h1 = rr_sys_slots_upto(plan, inv, total, off + q);
        if (h1 > h0) s_mark[h0] = (unsigned int)(p + 1);
      }
      __syncthreads();
      const unsigned int mk = mine ? s_mark[p] : 0u;
      const unsigned int mincl = rr::wave_scan_max_u32(mk);
      if (lane == 63) s_mx[wv] = mincl;
      __syncthreads();
      unsigned int pre = 0;
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (k < wv) pre = s_mx[k] > pre ? s_mx[k] : pre;
      const unsigned int src_i = (mincl > pre ? mincl : pre) - 1u;
      double* __restrict__ dst = pl.s[cur ^ 1];
      if (mine) {
        const uint64_t planes = 3 + 6 * a.L;
        for (uint64_t f = 0; f < planes; ++f) dst[f * n + p] = live[f * n + src_i];
        w = 1.0 / (double)a.n_global;
        pw[p] = w;
        if (idx_out) idx_out[p] = src_i;
      }
      cur_now = cur ^ 1;
    }
    // ---- get_best_particle (fastslam1.rs:269-274): arg max of the weight, ties -> the highest index
    uint64_t bb = mine && w > 0.0 ? rr_d2u(w) : 0ull, bi = mine ? p : 0ull;
    auto better = [](uint64_t ob, uint64_t oi, uint64_t b0, uint64_t i0) { return ob > b0 || (ob == b0 && oi > i0); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ob = rr::shfl_xor_u64(bb, o), oi = rr::shfl_xor_u64(bi, o);
      if (better(ob, oi, bb, bi)) {
        bb = ob;
        bi = oi;
      }
    }
    __syncthreads();  // (the gathered planes are complete, s_mark is free)
    if (lane == 0) {
      s_bb[wv] = bb;
      s_bi[wv] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < W; ++k)
        if (better(s_bb[k], s_bi[k], bb, bi)) {
          bb = s_bb[k];
          bi = s_bi[k];
        }
      const double* __restrict__ now = pl.s[cur_now];
      const uint64_t seq = a.res.first_seq + (uint64_t)s;
      ctl->best_bits = bb;
      ctl->best_index = bi;
      rr::store_pair_sys(&ring->rsp[0], rr_d2u(now[bi]), seq);
      rr::store_pair_sys(&ring->rsp[1], rr_d2u(now[n + bi]), seq);
      rr::store_pair_sys(&ring->rsp[2], rr_d2u(now[2 * n + bi]), seq);
      rr::store_pair_sys(&ring->rsp[3], rr_d2u(pw[bi]), seq);
      rr::store_pair_sys(&ring->rsp[rr::kResRspFlags], 0ull, seq);
      rr::store_pair_sys(&ring->rsp[kFs1RspIndex], bi, seq);
    }
    __syncthreads();  // (Ctl.cur and the weights are read by the next command)
  }
  if (tid == 0) {  // EXIT marker: the last command consumed (a quit counts), stamped with the launch id
    const uint64_t consumed = a.res.first_seq + (uint64_t)steps_done - 1 + (res_last_op == rr::kResOpQuit ? 1 : 0);
    rr::store_pair_sys(&ring->rsp[rr::kResRspExit], consumed, a.res.launch_id);
  }
}

// host layouts <-> planes.  tmp holds the AoS image on the device.
__global__ __launch_bounds__(kBlock) void k_fs1_init(Planes pl, double* __restrict__ pw, uint64_t n, uint64_t L,
                                                    double w0, double cov0) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t n_planes = 3 + 6 * L;
  if (t >= n_planes * n) return;
  const uint64_t plane = t / n, p = t % n;
  double v = 0.0;
  if (plane >= 3) {
    const int f = (int)((plane - 3) % 6);
    if (f == 2 || f == 5) v = cov0;
  }
  pl.s[0][t] = v;
  if (plane == 0) pw[p] = w0;
}

// maps AoS [p][l][6] -> planes (dir = 0) or back (dir = 1); one thread per (p, l)
__global__ __launch_bounds__(kBlock) void k_fs1_maps_transpose(double* __restrict__ planes, double* __restrict__ aos,
                                                              uint64_t n, uint64_t L, int dir) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n * L) return;
  const uint64_t l = t / n, p = t % n;  // consecutive threads -> consecutive particles: plane side coalesced
  double* a = aos + (p * L + l) * 6;
  double* b = planes + (3 + l * 6) * n + p;
#pragma unroll
  for (int f = 0; f < 6; ++f) {
    if (dir == 0) b[f * n] = a[f];
    else a[f] = b[f * n];
  }
}

// poses N x (w, x, y, yaw) <-> weight array + pose planes
__global__ __launch_bounds__(kBlock) void k_fs1_poses(double* __restrict__ planes, double* __restrict__ pw,
                                                     double* __restrict__ aos, uint64_t n, int dir) {
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  if (dir == 0) {
    pw[p] = aos[4 * p];
    planes[p] = aos[4 * p + 1];
    planes[n + p] = aos[4 * p + 2];
    planes[2 * n + p] = aos[4 * p + 3];
  } else {
    aos[4 * p] = pw[p];
    aos[4 * p + 1] = planes[p];
    aos[4 * p + 2] = planes[n + p];
    aos[4 * p + 3] = planes[2 * n + p];
  }
}

__global__ void k_fs1_one_landmarks(const double* __restrict__ planes, uint64_t n, uint64_t L, uint64_t p,
                                    double* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 6 * L) return;
  out[t] = planes[(3 + t) * n + p];  // t = l*6 + f
}

}  // namespace

// =============================================================================================
struct rr_fs1 {
  rr_fs1_params prm;
  rr_fs1_options opt;
  int algorithm = 1;                      // 1 = FastSLAM 1.0, 2 = FastSLAM 2.0 (proposal sampling, fastslam2.rs)
  double motion_cov[3] = {0.1, 0.1, 0.01};  // FastSLAM 2.0: MOTION_COV diagonal
  double nonpos_det_w = 1.0;
  uint64_t n = 0, L = 0, n_planes = 0, n_tiles = 0;
  uint64_t n_global = 0, gid0 = 0;  // sharding: particles over all shards, global index of local particle 0
  hipStream_t stream = nullptr;
  Planes pl{};
  double* slab = nullptr;  // [set][plane][n]
  rr::P2PState p2p;
  double* own_inbox = nullptr;  // RCCL transport: plain device mirror of one buffer set (the peer-to-peer transport brings its own)
  int obs_variant = 0;          // RR_FS1_VARIANT: tuning variant of k_fs1_observe (identical results)
  int shard_settle = 0;         // RCCL transport: the local phase consumed a pending resample (k_quantize_reduce flips Ctl.cur)
  uint64_t last_migrated = 0;
  double* pw = nullptr;
  uint64_t* cdf = nullptr;
  uint64_t* tile_total = nullptr;
  uint64_t* tile_q2 = nullptr;
  unsigned int* idx = nullptr;
  // rr::k_quantize_plan_mark<true> (integer image + plan in one launch, resample_core.hpp): tile records, arrival
  // counters, launch epoch, the largest grid whose workgroups are all resident (~0: not asked yet; RR_PF_FUSED_PLAN=0: off)
  uint64_t* grid_rec = nullptr;
  unsigned int* grid_ticket = nullptr;
  uint64_t grid_epoch = 0;
  uint64_t grid_capacity = ~0ull;
  uint64_t plan_giveups = 0;  // launches of the one-launch plan that degraded to the serial plan
  unsigned int* markers = nullptr;  // n + kResolveSlots, zero between resamples (fused single-GPU plan)
  unsigned int* carry = nullptr;    // one per kResolveSlots slots
  unsigned int* ridx = nullptr;  // sharded: sources of the served slots that belong to peers (allocated on connect)
  double* partial = nullptr;  // partial_chunks * n: per-chunk weight products (grown on demand)
  int partial_chunks = 0;
  double* z_dev = nullptr;
  size_t z_cap = 0;
  double* z_ring = nullptr;      // pinned [kZRing][kZStageWords]: observations on their way through k_fs1_resolve_predict
  uint64_t* z_done = nullptr;    // pinned: last ring sequence number the device has consumed
  uint64_t z_seq = 0;
  bool z_staged = false;         // this update's observations are already on their way to z_dev
  double* noise = nullptr;  // 2n
  double* pose_stage = nullptr;  // 4n: AoS (w, x, y, yaw) image for get_state / set_state (the inactive set only holds 3n when L == 0)
  uint64_t* part_bits = nullptr;
  uint64_t* part_idx = nullptr;
  BestMail* best_mail = nullptr;        // pinned: where k_fs1_best leaves the best particle for the host
  unsigned int* best_ticket = nullptr;  // arrivals of k_fs1_best's workgroups
  uint64_t best_seq = 0;
  Ctl* ctl = nullptr;
  Ctl* ctl_host = nullptr;
  unsigned int step = 0, rstep = 0;
  int last_chunks = 1;
  bool wmax_live = false;  // Ctl.wmax_bits holds the maximum of the current weights
  bool wmax_bits_clean = false;  // Ctl.wmax_bits is known to be zero (the last plan kernel consumed and zeroed it)
  bool maybe_pending = false;  // a lazy resample plan was launched; its gather has not been consumed yet
  bool idx_unresolved = false; // ... and its markers have not been turned into idx[] yet (ensure_resolved / k_fs1_resolve_predict)
  unsigned int* plane_list = nullptr;  // device: planes of the landmarks a lazy observe leaves untouched
  std::vector<unsigned int> plane_list_host;
  rr::Profiler prof{RR_FK_COUNT};
  // resident service (k_fs1_small; rr_fs1_set_resident)
  struct Resident {
    bool enabled = false, live = false, pending = false;
    rr::ResidentRing* ring = nullptr;
    uint64_t seq = 0, launch_id = 0;
    double idle_us = 0.0, life_us = 100000.0;
    uint64_t launches = 0, steps = 0;
    unsigned int cmd_step = 0, cmd_rstep = 0;
    bool have_best = false;  // the answer to the last command is the best particle of the CURRENT set
    double best_pose[3] = {0, 0, 0}, best_weight = 0.0;
    uint64_t best_index = 0;
  } res;
};

namespace {

const char* kFkNames[RR_FK_COUNT] = {"k_fs1_predict", "k_fs1_observe", "k_fs1_combine", "k_quantize_reduce",
                                     "k_scan_tiles",  "k_fs1_normalize", "k_cdf",        "k_fs1_indices",
                                     "k_fs1_gather",  "(unused)"};

inline unsigned grid_for(uint64_t n, int per) { return (unsigned)((n + per - 1) / per); }

rr_status fs1_resident_park(rr_fs1* h);

// keep_resident: the caller talks to the handle's resident update kernel; everybody else finds the stream idle (the kernel is
// asked to leave first) and the cached best particle forgotten
rr_status bind(rr_fs1* h, bool keep_resident = false) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  RR_HIP_TRY(hipSetDevice(h->opt.device));
  if (!keep_resident) {
    if (h->res.live || h->res.pending) {
      rr_status ps = fs1_resident_park(h);
      if (ps != RR_OK) return ps;
    }
    h->res.have_best = false;
  }
  return RR_OK;
}

rr_fs1_model host_model(const rr_fs1_params& p, int algorithm = 1, double nonpos_det_w = 1.0) {
  rr_fs1_model m;
  m.dt = p.dt;
  m.q_sqrt0 = rr_sqrt(p.q00);
  m.q_sqrt1 = rr_sqrt(p.q11);
  m.r00 = p.r00;
  m.r11 = p.r11;
  m.init_threshold = p.init_threshold;
  m.init_cov = p.first_obs_cov;
  m.init_test_lt = algorithm == 2 ? 1.0 : 0.0;
  m.nonpos_det_w = nonpos_det_w;
  return m;
}

rr_fs1_model model_of(const rr_fs1* h);

rr_status validate_u(const double u[2]) {
  if (!u || !std::isfinite(u[0]) || !std::isfinite(u[1]))
    return fail(RR_INVALID_PARAMETER, "fastslam control input must contain only finite values");
  return RR_OK;
}

// z rows are (d, angle, id); ids must be integral and inside the map; duplicates force one chunk
rr_status validate_z(const rr_fs1* h, const double* z, size_t n_z, bool* has_duplicates) {
  if (n_z && !z) return fail(RR_INVALID_PARAMETER, "null observations");
  std::vector<char> seen(h->L, 0);
  *has_duplicates = false;
  for (size_t k = 0; k < n_z; ++k) {
    const double d = z[3 * k], a = z[3 * k + 1], id = z[3 * k + 2];
    if (!std::isfinite(d) || !std::isfinite(a) || !std::isfinite(id))
      return fail(RR_INVALID_PARAMETER, "fastslam observations must be finite");
    if (id < 0.0 || id >= (double)h->L || id != std::floor(id))
      return fail(RR_INVALID_PARAMETER, "fastslam observation landmark id out of range");  // the reference would panic on the index
    if (seen[(size_t)id]) *has_duplicates = true;
    seen[(size_t)id] = 1;
  }
  return RR_OK;
}

ImageArgs image_args(const rr_fs1* h) {
  ImageArgs a{};
  a.n = h->n;
  a.n_global = h->n_global;
  a.gid0 = h->gid0;
  a.degenerate = rr::kDegenerateLast;
  a.honour_uniform_flag = 0;
  return a;
}

PlanArgs plan_args(const rr_fs1* h, int mode, double rho_override, bool lazy = false) {
  PlanArgs a{};
  a.n_global = h->n_global;
  a.neff_threshold = h->prm.nth;  // fastslam1.rs:262-265
  a.gate = RR_GATE_NEFF;
  a.mode = mode;
  a.scheme = RR_RESAMPLE_SYSTEMATIC;
  a.rho_override = rho_override;
  a.seed = h->opt.seed;
  a.rstep = h->rstep;
  a.set_uniform_on_fire = 0;  // FastSLAM stores explicit weights (k_fs1_indices writes 1/n)
  a.lazy_gather = lazy ? 1 : 0;
  return a;
}

rr_fs1_model model_of(const rr_fs1* h) { return host_model(h->prm, h->algorithm, h->nonpos_det_w); }

rr_status ensure_z_dev(rr_fs1* h, size_t n_z) {
  if (n_z <= h->z_cap) return RR_OK;
  // (a kernel that still reads the old buffer has been enqueued before this free: hipFree waits for the device)
  if (h->z_dev) RR_HIP_TRY(hipFree(h->z_dev));
  h->z_dev = nullptr;
  h->z_cap = 0;
  const size_t cap = std::max<size_t>(n_z, 64);
  RR_HIP_TRY(hipMalloc(&h->z_dev, 3 * cap * sizeof(double)));
  h->z_cap = cap;
  return RR_OK;
}

// this update's observations into the next pinned slot; the kernel that gets `out` carries them to z_dev
rr_status stage_z(rr_fs1* h, const double* z, size_t n_z, ZStage* out) {
  *out = ZStage{};
  if (n_z == 0 || 3 * n_z > (size_t)kZStageWords) return RR_OK;
  if (rr_status s = ensure_z_dev(h, n_z); s != RR_OK) return s;
  if (!h->z_ring) {
    RR_HIP_TRY(hipHostMalloc(&h->z_ring, (size_t)kZRing * kZStageWords * sizeof(double), hipHostMallocDefault));
    RR_HIP_TRY(hipHostMalloc(&h->z_done, sizeof(uint64_t), hipHostMallocDefault));
    *h->z_done = 0;
    h->z_seq = 0;
  }
  const uint64_t seq = ++h->z_seq;
  // the slot's previous user (seq - kZRing) must have been consumed: the device says so in z_done
  if (seq > (uint64_t)kZRing) {
    volatile uint64_t* done = h->z_done;
    if (*done + kZRing < seq) {
      RR_HIP_TRY(hipStreamSynchronize(h->stream));  // (only when the host is kZRing updates ahead)
      if (*done + kZRing < seq) return fail(RR_RUNTIME_ERROR, "observation staging ring: the device did not consume its slots");
    }
  }
  double* slot = h->z_ring + (size_t)(seq % kZRing) * kZStageWords;
  std::memcpy(slot, z, 3 * n_z * sizeof(double));
  out->host = slot;
  out->dev = h->z_dev;
  out->done = h->z_done;
  out->seq = seq;
  out->words = (int)(3 * n_z);
  return RR_OK;
}

// the markers of the last single-GPU plan -> idx[] (normally the next update's k_fs1_resolve_predict does it on the way)
rr_status ensure_resolved(rr_fs1* h) {
  if (!h->idx_unresolved) return RR_OK;
  rr::ScopedTimer t(h->prof, h->stream, RR_FK_INDICES);
  hipLaunchKernelGGL(k_fs1_resolve, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->ctl, h->markers,
                     h->carry, h->n, h->idx);
  RR_HIP_TRY(hipGetLastError());
  h->idx_unresolved = false;
  return RR_OK;
}

// make a pending lazy resample real: gather every plane, flip the live set
rr_status materialise(rr_fs1* h) {
  if (rr_status rs = ensure_resolved(h); rs != RR_OK) return rs;
  if (!h->maybe_pending) return RR_OK;
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_GATHER);
    hipLaunchKernelGGL(k_fs1_gather, dim3(grid_for(h->n, kBlock), grid_for(h->n_planes, kPlanesPerThread)), dim3(kBlock), 0,
                       h->stream, h->pl, h->ctl, h->idx, h->n, h->n_planes, 1, (const unsigned int*)nullptr);
    hipLaunchKernelGGL(k_fs1_settle, dim3(1), dim3(1), 0, h->stream, h->ctl);
  }
  RR_HIP_TRY(hipGetLastError());
  h->maybe_pending = false;
  return RR_OK;
}

template <bool EXPLICIT, bool LAZY>
rr_status launch_predict(rr_fs1* h, const double u[2], const double* z = nullptr, size_t n_z = 0) {
  if (LAZY && !EXPLICIT && !h->pl.inbox && h->n == h->n_global) {  // one GPU: resolve the last plan's markers on the way (unless an
                                                               // accessor has had that done), carry the observations
    ZStage zs{};
    if (z && (n_z > 0)) {
      if (rr_status st = stage_z(h, z, n_z, &zs); st != RR_OK) return st;
    }
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_PREDICT);
    hipLaunchKernelGGL(k_fs1_resolve_predict, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->pl, h->ctl, h->n,
                       u[0], u[1], model_of(h), h->opt.seed, h->step, h->markers, (const unsigned int*)h->carry, h->idx, h->gid0, zs,
                       h->idx_unresolved ? 1 : 0);
    RR_HIP_TRY(hipGetLastError());
    h->z_staged = zs.words > 0;
    h->idx_unresolved = false;
    h->step += 1;
    return RR_OK;
  }
  if (rr_status rs = ensure_resolved(h); rs != RR_OK) return rs;
  rr::ScopedTimer t(h->prof, h->stream, RR_FK_PREDICT);
  hipLaunchKernelGGL((k_fs1_predict<EXPLICIT, LAZY>), dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pl,
                     h->ctl, h->n, u[0], u[1], model_of(h), h->opt.seed, h->step, (const double*)h->noise,
                     (const double*)(h->noise ? h->noise + h->n : nullptr), (const unsigned int*)h->idx, h->gid0);
  RR_HIP_TRY(hipGetLastError());
  h->step += 1;
  return RR_OK;
}

// FastSLAM 2.0: sample the pose from the proposal built on the first observation (fastslam2.rs:341-347)
template <bool EXPLICIT, bool LAZY>
rr_status launch_propose(rr_fs1* h, const double u[2], const double* z, size_t n_z) {
  rr_fs2_model m;
  m.base = model_of(h);
  m.m0 = h->motion_cov[0];
  m.m1 = h->motion_cov[1];
  m.m2 = h->motion_cov[2];
  if (rr_status rs = ensure_resolved(h); rs != RR_OK) return rs;
  rr::ScopedTimer t(h->prof, h->stream, RR_FK_PREDICT);
  hipLaunchKernelGGL((k_fs2_predict<EXPLICIT, LAZY>), dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pl, h->ctl,
                     h->n, u[0], u[1], m, h->opt.seed, h->step, (const double*)h->noise, (const unsigned int*)h->idx, h->gid0,
                     n_z > 0 ? 1 : 0, n_z ? z[0] : 0.0, n_z ? z[1] : 0.0, n_z ? (uint64_t)z[2] : 0ull);
  RR_HIP_TRY(hipGetLastError());
  h->step += 1;
  return RR_OK;
}

// the pose step of an update: noisy motion model (FastSLAM 1.0) or proposal sampling (2.0)
template <bool LAZY>
rr_status launch_motion(rr_fs1* h, const double u[2], const double* z, size_t n_z) {
  if (h->algorithm == 2) return launch_propose<false, LAZY>(h, u, z, n_z);
  return launch_predict<false, LAZY>(h, u, z, n_z);
}

int choose_chunks(const rr_fs1* h, size_t n_z, bool dup) {
  if (dup || n_z <= 1) return 1;
  int want = h->opt.obs_chunks;
  if (want <= 0) {
    // Short per-wave loops and many more waves than the 3072 the chip holds at 3 waves per SIMD: measured at 200
    // observations (gpurun_out/r02m) 1e5 particles 6 / 13 / 29 chunks 0.395 / 0.374 / 0.364 ms, 125 000 particles
    // 5 / 16 / 25 chunks 0.503 / 0.461 / 0.450 ms, 1e6 particles 1 / 2 / 4 chunks 4.02 / 3.78 / 3.88 ms -- about
    // 45 000 waves per launch, but never fewer than ~7 observations per wave (the pose / index prologue)
    const uint64_t waves = (h->n + 63) / 64;
    want = (int)((45000 + waves - 1) / waves);
    want = std::min<int>(want, (int)std::max<size_t>(1, n_z / 7));
    if (const char* e = std::getenv("RR_FS1_TARGET_WAVES")) want = (int)((std::max(1, std::atoi(e)) + waves - 1) / waves);
  }
  want = std::max(1, std::min<int>({want, kMaxChunks, (int)n_z}));
  const int len = (int)((n_z + want - 1) / want);
  return (int)((n_z + len - 1) / len);
}

// Ctl.wmax_bits must be zero before a kernel accumulates a weight maximum into it; the plan kernel of every
// real (gate or forced) plan leaves it zeroed, anything else needs the memset
rr_status zero_wmax(rr_fs1* h) {
  if (!h->wmax_bits_clean) RR_HIP_TRY(hipMemsetAsync(&h->ctl->wmax_bits, 0, sizeof(uint64_t), h->stream));
  h->wmax_bits_clean = false;
  return RR_OK;
}

rr_status launch_observe(rr_fs1* h, const double* z, size_t n_z, bool dup, bool lazy = false) {
  if (n_z == 0) {  // no observation: weights untouched, but the max must still be known
    if (rr_status zs = zero_wmax(h); zs != RR_OK) return zs;
    hipLaunchKernelGGL(k_fs1_wmax, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
    h->last_chunks = 1;
    h->wmax_live = true;
    return RR_OK;
  }
  if (!h->z_staged) {
    if (rr_status zs = ensure_z_dev(h, n_z); zs != RR_OK) return zs;
    RR_HIP_TRY(hipMemcpyAsync(h->z_dev, z, 3 * n_z * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  h->z_staged = false;
  if (rr_status zs = zero_wmax(h); zs != RR_OK) return zs;
  h->wmax_live = true;
  const int chunks = choose_chunks(h, n_z, dup);
  const int len = (int)((n_z + chunks - 1) / chunks);
  h->last_chunks = chunks;
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_OBSERVE);
    const dim3 grid(grid_for(h->n, kBlock), chunks);
    const size_t lds = 3 * (size_t)len * sizeof(double);
    hipEvent_t ea = nullptr, eb = nullptr;
    if (h->prof.on && h->prof.dispatch_only && !dup) {  // timestamps of this dispatch itself: nothing extra in the stream
      ea = h->prof.take();
      eb = h->prof.take();
      h->prof.events.push_back({RR_FK_OBSERVE, ea, eb});
    }
    const void* kfn = nullptr;
    // (the kernel pointer is chosen first so that the stamped and the plain launch share one argument list)
#define RR_OBS_KERNEL(LAZY_, SEQ_, VAR_) ((const void*)(k_fs1_observe<LAZY_, SEQ_, VAR_>))
    if (dup) kfn = RR_OBS_KERNEL(false, true, 0);  // repeated landmark ids: strictly sequential updates (chunks == 1, in place)
    else if (!lazy) kfn = RR_OBS_KERNEL(false, false, 0);
    else {
      switch (h->obs_variant) {
        case 1: kfn = RR_OBS_KERNEL(true, false, kObsNtStore); break;
        case 2: kfn = RR_OBS_KERNEL(true, false, kObsNtStore | kObsNtLoad); break;
        case 3: kfn = RR_OBS_KERNEL(true, false, kObsNoPipe | kObsFourWaves); break;
        case 4: kfn = RR_OBS_KERNEL(true, false, kObsNoPipe | kObsFourWaves | kObsNtStore); break;
        case 5: kfn = RR_OBS_KERNEL(true, false, kObsFourWaves); break;
        case 6: kfn = RR_OBS_KERNEL(true, false, kObsNoPipe); break;
        default: kfn = RR_OBS_KERNEL(true, false, 0); break;
      }
    }
#undef RR_OBS_KERNEL
    Planes a_pl = h->pl;
    double* a_pw = h->pw;
    Ctl* a_ctl = h->ctl;
    uint64_t a_n = h->n;
    const double* a_z = h->z_dev;
    int a_nz = (int)n_z, a_len = len, a_chunks = chunks;
    rr_fs1_model a_m = model_of(h);
    if (chunks > 1 && chunks > h->partial_chunks) {  // per-chunk weight products
      if (h->partial) RR_HIP_TRY(hipFree(h->partial));
      h->partial = nullptr;
      h->partial_chunks = 0;
      RR_HIP_TRY(hipMalloc(&h->partial, (size_t)chunks * h->n * sizeof(double)));
      h->partial_chunks = chunks;
      hipLaunchKernelGGL(k_fs1_no_factors, dim3(1024), dim3(kBlock), 0, h->stream, reinterpret_cast<uint64_t*>(h->partial),
                         (uint64_t)chunks * h->n);  // every slot reads "no factor yet" between updates
    }
    double* a_partial = h->partial;
    const unsigned int* a_idx = h->idx;
    unsigned int a_pblocks = grid.x;
    const dim3 grid_l((unsigned int)((uint64_t)a_pblocks * (uint64_t)chunks));
    void* args[] = {&a_pl, &a_pw, &a_ctl, &a_n, &a_z, &a_nz, &a_len, &a_chunks, &a_m, &a_partial, &a_idx, &a_pblocks};
    if (ea && !dup) RR_HIP_TRY(hipExtLaunchKernel(kfn, grid_l, dim3(kBlock), args, lds, h->stream, ea, eb, 0));
    else RR_HIP_TRY(hipLaunchKernel(kfn, grid_l, dim3(kBlock), args, lds, h->stream));
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status launch_sums(rr_fs1* h, int mode, double rho_override, bool lazy = false, int settle = 0) {
  if (!h->wmax_live) {  // the last plan kernel consumed the maximum (or the weights were renormalised since)
    if (rr_status zs = zero_wmax(h); zs != RR_OK) return zs;
    hipLaunchKernelGGL(k_fs1_wmax, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
  }
  h->wmax_live = mode == 2;  // statistics leave everything in place; a real plan consumes it
  h->wmax_bits_clean = mode != 2;  // ... and zeroes the accumulator (finalize_plan)
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_QUANTIZE_REDUCE);
    hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl,
                       (const double*)&h->ctl->wmax_bits, image_args(h), h->tile_total, h->tile_q2, settle);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 1, plan_args(h, mode, rho_override, lazy), (uint64_t*)nullptr);
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

// normalise-or-resample after the sums (every kernel decides on the device whether it runs)
rr_status launch_finish(rr_fs1* h, bool lazy = false) {
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_CDF);
    hipLaunchKernelGGL(rr::k_cdf, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl, image_args(h),
                       h->tile_total, h->cdf, (uint64_t*)nullptr, 0);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_NORMALIZE);
    hipLaunchKernelGGL(k_fs1_normalize, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_INDICES);
    hipLaunchKernelGGL(k_fs1_indices, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->ctl, h->cdf, h->n,
                       h->idx, h->pw);
  }
  if (lazy) {
    h->maybe_pending = true;  // the next update reads through idx (or an accessor materialises)
  } else {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_GATHER);
    hipLaunchKernelGGL(k_fs1_gather, dim3(grid_for(h->n, kBlock), grid_for(h->n_planes, kPlanesPerThread)), dim3(kBlock), 0,
                       h->stream, h->pl, h->ctl, h->idx, h->n, h->n_planes, 0, (const unsigned int*)nullptr);
  }
  RR_HIP_TRY(hipGetLastError());
  h->rstep += 1;
  return RR_OK;
}

// gate + normalise-or-resample of a single-GPU update in two launches after k_quantize_reduce (lazy: the particles
// move when the next update, or an accessor, reads them through idx)
rr_status launch_plan_fused(rr_fs1* h, int settle) {
  if (!h->wmax_live) {
    if (rr_status zs = zero_wmax(h); zs != RR_OK) return zs;
    hipLaunchKernelGGL(k_fs1_wmax, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
  }
  h->wmax_live = false;
  h->wmax_bits_clean = true;  // the plan kernel's finalize_plan zeroes the accumulator
  if (h->grid_capacity == ~0ull) {
    int per_cu = 0, dev_cus = 0;
    RR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rr::k_quantize_plan_mark<true>, rr::kTileBlock, 0));
    RR_HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    h->grid_capacity = std::min<uint64_t>((uint64_t)per_cu * (uint64_t)dev_cus, (uint64_t)rr::kTileBlock);
    if (const char* e = std::getenv("RR_PF_FUSED_PLAN")) {
      if (std::atoi(e) == 0) h->grid_capacity = 0;
    }
    if (h->grid_capacity) {
      const size_t rec_bytes = (size_t)(rr::kTileBlock + 1) * rr::kRecWords * sizeof(uint64_t);
      RR_HIP_TRY(hipMalloc(&h->grid_rec, rec_bytes));
      RR_HIP_TRY(hipMemsetAsync(h->grid_rec, 0, rec_bytes, h->stream));
      RR_HIP_TRY(hipMalloc(&h->grid_ticket, rr::kTicketWords * sizeof(unsigned int)));
      RR_HIP_TRY(hipMemsetAsync(h->grid_ticket, 0, rr::kTicketWords * sizeof(unsigned int), h->stream));
    }
  }
  if (h->n_tiles <= h->grid_capacity && rr::spin_permit(h->opt.device, h)) {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_CDF);
    hipLaunchKernelGGL(rr::k_quantize_plan_mark<true>, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl,
                       (const double*)&h->ctl->wmax_bits, image_args(h), h->grid_rec, h->grid_ticket, ++h->grid_epoch, settle,
                       h->n_tiles, plan_args(h, 0, NAN, /*lazy=*/true), h->markers, h->carry, rr::EstArgs{}, rr::plan_giveup_ticks());
  } else {
    {
      rr::ScopedTimer t(h->prof, h->stream, RR_FK_QUANTIZE_REDUCE);
      hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl,
                         (const double*)&h->ctl->wmax_bits, image_args(h), h->tile_total, h->tile_q2, settle);
    }
    {
      rr::ScopedTimer t(h->prof, h->stream, RR_FK_CDF);
      hipLaunchKernelGGL(k_fs1_plan, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl, image_args(h),
                         h->tile_total, h->tile_q2, h->n_tiles, plan_args(h, 0, NAN, /*lazy=*/true), h->markers, h->carry);
    }
  }
  RR_HIP_TRY(hipGetLastError());
  h->maybe_pending = true;
  h->idx_unresolved = true;  // the markers wait for the next update's first kernel (or ensure_resolved)
  h->rstep += 1;
  return RR_OK;
}

bool plan_fusable(const rr_fs1* h) { return h->n_tiles <= (uint64_t)rr::kFusedMaxTiles && h->n == h->n_global; }

// lazy consume, part 3: the landmarks this step does NOT observe still have to move with their
// particle -- gather just their planes (list built on the host from the observation ids)
rr_status launch_rest_gather(rr_fs1* h, const double* z, size_t n_z) {
  std::vector<char> seen(h->L, 0);
  for (size_t k = 0; k < n_z; ++k) seen[(size_t)z[3 * k + 2]] = 1;
  auto& list = h->plane_list_host;
  list.clear();
  for (uint64_t l = 0; l < h->L; ++l)
    if (!seen[l])
      for (int f = 0; f < 6; ++f) list.push_back((unsigned int)(3 + l * 6 + f));
  if (list.empty()) return RR_OK;
  RR_HIP_TRY(hipMemcpyAsync(h->plane_list, list.data(), list.size() * sizeof(unsigned int), hipMemcpyHostToDevice, h->stream));
  rr::ScopedTimer t(h->prof, h->stream, RR_FK_GATHER);
  hipLaunchKernelGGL(k_fs1_gather, dim3(grid_for(h->n, kBlock), grid_for(list.size(), kPlanesPerThread)), dim3(kBlock), 0,
                     h->stream, h->pl, h->ctl, h->idx, h->n, (uint64_t)list.size(), 1, (const unsigned int*)h->plane_list);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status fetch_ctl(rr_fs1* h) {
  RR_HIP_TRY(hipMemcpyAsync(h->ctl_host, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  rr::spin_release(h->opt.device, h);
  if (h->ctl_host->obs_timeout)
    return fail(RR_RUNTIME_ERROR, "k_fs1_observe: a chunk's weight factor did not arrive within 2 s; the weights of this update are not valid");
  if (h->ctl_host->grid_timeout) {
    // launches of the one-launch plan degraded to the serial plan (another process kept workgroups off the device): same
    // results; this handle takes the multi-launch plan from now on (resample_core.hpp, k_quantize_plan_mark)
    h->plan_giveups += (uint64_t)h->ctl_host->grid_timeout;
    h->grid_capacity = 0;
    RR_HIP_TRY(hipMemsetAsync(&h->ctl->grid_timeout, 0, sizeof(int), h->stream));
    h->ctl_host->grid_timeout = 0;
  }
  return h->p2p.check(h->stream);  // a latched peer-wait timeout must not look like a healthy filter
}

rr_status ensure_mailbox(rr_fs1* h) {
  if (h->best_mail) return RR_OK;
  RR_HIP_TRY(hipHostMalloc(&h->best_mail, sizeof(BestMail), hipHostMallocDefault));
  std::memset(h->best_mail, 0, sizeof(BestMail));
  RR_HIP_TRY(hipMalloc(&h->best_ticket, sizeof(unsigned int)));
  RR_HIP_TRY(hipMemsetAsync(h->best_ticket, 0, sizeof(unsigned int), h->stream));
  return RR_OK;
}

// wait for stamp `want` of the mailbox: polls (a healthy device answers within ~15 us), then waits the ordinary way
rr_status await_mailbox(rr_fs1* h, uint64_t want) {
  const volatile uint64_t* seq = &h->best_mail->seq;
  for (long spins = 0; spins < 2000000; ++spins)  // ~ tens of milliseconds
    if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) return RR_OK;
  RR_HIP_TRY(hipStreamSynchronize(h->stream));  // slow device / long queue in front of the kernel
  if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != want) return fail(RR_RUNTIME_ERROR, "the device's stamp never reached the host mailbox");
  return RR_OK;
}

// rr_fs1_synchronize for a filter on one GPU: a one-thread kernel stamps the mailbox behind everything enqueued so far; Ctl is
// only copied back when the stamp says there is something to report
rr_status synchronize_light(rr_fs1* h) {
  if (h->p2p.ready || h->n != h->n_global) return fetch_ctl(h);
  rr_status s = ensure_mailbox(h);
  if (s != RR_OK) return s;
  const uint64_t want = ++h->best_seq;
  hipLaunchKernelGGL(k_fs1_stamp, dim3(1), dim3(1), 0, h->stream, (const Ctl*)h->ctl, h->best_mail, want);
  RR_HIP_TRY(hipGetLastError());
  if ((s = await_mailbox(h, want)) != RR_OK) return s;
  rr::spin_release(h->opt.device, h);  // the stream is idle
  return h->best_mail->flags ? fetch_ctl(h) : RR_OK;
}

rr_status ensure_pose_stage(rr_fs1* h) {
  if (!h->pose_stage) RR_HIP_TRY(hipMalloc(&h->pose_stage, 4 * h->n * sizeof(double)));
  return RR_OK;
}

rr_status ensure_noise(rr_fs1* h) {
  if (!h->noise) RR_HIP_TRY(hipMalloc(&h->noise, 3 * h->n * sizeof(double)));  // FastSLAM 2.0 draws three normals
  return RR_OK;
}

}  // namespace

extern "C" {

void rr_fs1_params_default(rr_fs1_params* p) {
  if (!p) return;
  p->dt = 0.1;
  p->q00 = 0.3;
  p->q11 = 0.0305;
  p->r00 = 0.5;
  p->r11 = 0.0305;
  p->max_range = 20.0;
  p->nth = 100.0 / 1.5;
  p->initial_weight = 1.0 / 100.0;
  p->init_cov = 1000.0;
  p->init_threshold = 100.0;
  p->first_obs_cov = NAN;
}

void rr_fs1_options_default(rr_fs1_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
}

rr_status rr_fs1_create(uint64_t n_particles, uint64_t n_landmarks, const rr_fs1_params* params,
                        const rr_fs1_options* opt_in, rr_fs1** out) {
  if (!out) return fail(RR_INVALID_PARAMETER, "null output handle");
  *out = nullptr;
  if (n_particles == 0) return fail(RR_INVALID_PARAMETER, "fastslam requires at least one particle");
  if (n_particles >= (1ull << 31)) return fail(RR_INVALID_PARAMETER, "n_particles must be below 2^31");
  rr_fs1_params prm;
  if (params) prm = *params; else rr_fs1_params_default(&prm);
  rr_fs1_options opt;
  if (opt_in) opt = *opt_in; else rr_fs1_options_default(&opt);
  if (!(prm.dt > 0.0) || !(prm.q00 >= 0.0) || !(prm.q11 >= 0.0) || !std::isfinite(prm.r00) || !std::isfinite(prm.r11) ||
      !std::isfinite(prm.nth) || !std::isfinite(prm.initial_weight) || !std::isfinite(prm.init_cov))
    return fail(RR_INVALID_PARAMETER, "fastslam parameters must be finite (dt > 0, Q >= 0)");
  if (opt.obs_chunks < 0 || opt.obs_chunks > kMaxChunks) return fail(RR_INVALID_PARAMETER, "obs_chunks out of range");
  {
    const uint64_t ng = opt.n_global ? opt.n_global : n_particles;
    if (ng >= (1ull << 31) || opt.first_global_index + n_particles > ng)
      return fail(RR_INVALID_PARAMETER, "shard range exceeds n_global (< 2^31)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(RR_RUNTIME_ERROR, "no HIP device available: the engine has no CPU fallback");
  if (opt.device < 0 || opt.device >= ndev) return fail(RR_INVALID_PARAMETER, "device ordinal out of range");
  RR_HIP_TRY(hipSetDevice(opt.device));
  rr_fs1* h = new rr_fs1();
  h->prm = prm;
  h->opt = opt;
  h->n = n_particles;
  h->L = n_landmarks;
  h->n_global = opt.n_global ? opt.n_global : n_particles;
  h->gid0 = opt.first_global_index;
  h->n_planes = 3 + 6 * n_landmarks;
  h->n_tiles = (h->n + kTile - 1) / kTile;
  if (const char* e = std::getenv("RR_FS1_VARIANT")) h->obs_variant = std::atoi(e);
  auto cleanup = [&](rr_status st) {
    rr_fs1_destroy(h);
    return st;
  };
#define RR_TRY_OR_CLEAN(expr)                                                                                          \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) return cleanup(fail(RR_RUNTIME_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e)));  \
  } while (0)
  RR_TRY_OR_CLEAN(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t state_bytes = h->n_planes * h->n * sizeof(double);
  RR_TRY_OR_CLEAN(hipMalloc(&h->slab, 2 * state_bytes));  // one slab: peers map the whole state with one IPC handle
  h->pl.s[0] = h->slab;
  h->pl.s[1] = h->slab + h->n_planes * h->n;
  RR_TRY_OR_CLEAN(hipMalloc(&h->pw, h->n * sizeof(double)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->cdf, h->n * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->tile_total, h->n_tiles * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->tile_q2, 2 * h->n_tiles * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->idx, h->n * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->markers, (h->n + rr::kResolveSlots) * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->markers, 0, (h->n + rr::kResolveSlots) * sizeof(unsigned int), h->stream));
  RR_TRY_OR_CLEAN(hipMalloc(&h->carry, (h->n / rr::kResolveSlots + 2) * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->part_bits, 1024 * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->part_idx, 1024 * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->plane_list, (h->n_planes + 1) * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMalloc(&h->ctl, sizeof(Ctl)));
  RR_TRY_OR_CLEAN(hipHostMalloc(&h->ctl_host, sizeof(Ctl)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->ctl, 0, sizeof(Ctl), h->stream));
  hipLaunchKernelGGL(k_fs1_init, dim3(grid_for(h->n_planes * h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pl, h->pw,
                     h->n, h->L, prm.initial_weight, prm.init_cov);
  RR_TRY_OR_CLEAN(hipGetLastError());
  RR_TRY_OR_CLEAN(hipStreamSynchronize(h->stream));
#undef RR_TRY_OR_CLEAN
  *out = h;
  return RR_OK;
}

void rr_fs1_destroy(rr_fs1* h) {
  if (!h) return;
  (void)hipSetDevice(h->opt.device);
  if (h->res.live || h->res.pending) (void)fs1_resident_park(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->res.ring) (void)hipHostFree(h->res.ring);
  h->p2p.teardown();
  (void)hipFree(h->own_inbox);
  (void)hipFree(h->slab);
  (void)hipFree(h->pw);
  (void)hipFree(h->cdf);
  (void)hipFree(h->tile_total);
  (void)hipFree(h->tile_q2);
  (void)hipFree(h->idx);
  (void)hipFree(h->markers);
  (void)hipFree(h->grid_rec);
  (void)hipFree(h->grid_ticket);
  (void)hipFree(h->carry);
  (void)hipFree(h->ridx);
  (void)hipFree(h->partial);
  (void)hipFree(h->z_dev);
  (void)hipHostFree(h->z_ring);
  (void)hipHostFree(h->z_done);
  (void)hipFree(h->noise);
  (void)hipFree(h->pose_stage);
  (void)hipFree(h->part_bits);
  (void)hipFree(h->best_ticket);
  if (h->best_mail) (void)hipHostFree(h->best_mail);
  (void)hipFree(h->part_idx);
  (void)hipFree(h->plane_list);
  (void)hipFree(h->ctl);
  if (h->ctl_host) (void)hipHostFree(h->ctl_host);
  h->prof.destroy();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  rr::spin_release(h->opt.device, h);
  delete h;
}

uint64_t rr_fs1_particle_count(const rr_fs1* h) { return h ? h->n : 0; }
uint64_t rr_fs1_landmark_count(const rr_fs1* h) { return h ? h->L : 0; }

rr_status rr_fs1_predict(rr_fs1* h, const double u[2]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = validate_u(u)) != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  return launch_predict<false, false>(h, u);
}

rr_status rr_fs1_predict_with_noise(rr_fs1* h, const double u[2], const double* z0, const double* z1) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = validate_u(u)) != RR_OK) return s;
  if (!z0 || !z1) return fail(RR_INVALID_PARAMETER, "null noise arrays");
  if ((s = ensure_noise(h)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(h->noise, z0, h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RR_HIP_TRY(hipMemcpyAsync(h->noise + h->n, z1, h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if ((s = materialise(h)) != RR_OK) return s;
  return launch_predict<true, false>(h, u);
}

// ---- FastSLAM 2.0 (include/rr_fastslam2.h)
void rr_fs2_params_default(rr_fs2_params* p) {
  if (!p) return;
  rr_fs1_params_default(&p->base);
  p->base.first_obs_cov = 10.0;  // fastslam2.rs:255
  p->motion_cov[0] = 0.1;        // :30
  p->motion_cov[1] = 0.1;
  p->motion_cov[2] = 0.01;
  p->nonpos_det_weight = 1e-10;  // :289
}

rr_status rr_fs2_create(uint64_t n_particles, uint64_t n_landmarks, const rr_fs2_params* params,
                        const rr_fs1_options* opt, rr_fs2** out) {
  rr_fs2_params prm;
  if (params) prm = *params; else rr_fs2_params_default(&prm);
  for (int k = 0; k < 3; ++k)
    if (!std::isfinite(prm.motion_cov[k]) || prm.motion_cov[k] < 0.0)
      return fail(RR_INVALID_PARAMETER, "fastslam2 motion covariance must be finite and non-negative");
  if (!std::isfinite(prm.nonpos_det_weight)) return fail(RR_INVALID_PARAMETER, "fastslam2 nonpos_det_weight must be finite");
  if (!(prm.base.first_obs_cov == prm.base.first_obs_cov))
    return fail(RR_INVALID_PARAMETER, "fastslam2 initialises a landmark's covariance on its first observation: first_obs_cov must be a number");
  rr_status s = rr_fs1_create(n_particles, n_landmarks, &prm.base, opt, out);
  if (s != RR_OK) return s;
  (*out)->algorithm = 2;
  for (int k = 0; k < 3; ++k) (*out)->motion_cov[k] = prm.motion_cov[k];
  (*out)->nonpos_det_w = prm.nonpos_det_weight;
  return RR_OK;
}

rr_status rr_fs2_update(rr_fs2* h, const double u[2], const double* z, size_t n_z) { return rr_fs1_update(h, u, z, n_z); }
rr_status rr_fs2_update_async(rr_fs2* h, const double u[2], const double* z, size_t n_z) {
  return rr_fs1_update_async(h, u, z, n_z);
}

static rr_status fs2_predict_common(rr_fs2* h, const double u[2], const double* z, size_t n_z, const double* noise) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (h->algorithm != 2) return fail(RR_INVALID_PARAMETER, "not a FastSLAM 2.0 filter");
  if ((s = validate_u(u)) != RR_OK) return s;
  bool dup;
  if ((s = validate_z(h, z, n_z, &dup)) != RR_OK) return s;
  if (noise) {
    if ((s = ensure_noise(h)) != RR_OK) return s;
    RR_HIP_TRY(hipMemcpyAsync(h->noise, noise, 3 * h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  if ((s = materialise(h)) != RR_OK) return s;
  return noise ? launch_propose<true, false>(h, u, z, n_z) : launch_propose<false, false>(h, u, z, n_z);
}

rr_status rr_fs2_predict_with_noise(rr_fs2* h, const double u[2], const double* z, size_t n_z, const double* noise) {
  if (!noise) return fail(RR_INVALID_PARAMETER, "null noise array");
  return fs2_predict_common(h, u, z, n_z, noise);
}

rr_status rr_fs2_predict(rr_fs2* h, const double u[2], const double* z, size_t n_z) {
  return fs2_predict_common(h, u, z, n_z, nullptr);
}

rr_status rr_fs1_observe(rr_fs1* h, const double* z, size_t n_z) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  bool dup;
  if ((s = validate_z(h, z, n_z, &dup)) != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  return launch_observe(h, z, n_z, dup);
}

rr_status rr_fs1_normalize_resample(rr_fs1* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = launch_sums(h, 0, NAN)) != RR_OK) return s;
  return launch_finish(h);
}

rr_status rr_fs1_resample_systematic(rr_fs1* h, double rho) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!(rho >= 0.0 && rho < 1.0)) return fail(RR_INVALID_PARAMETER, "rho must lie in [0, 1)");
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = launch_sums(h, 1, rho)) != RR_OK) return s;
  return launch_finish(h);
}

// ---- resident service: the host side (the protocol is resident_core.hpp's; the PF engine's twin is in pf_engine.hip)
static bool fs1_resident_path(const rr_fs1* h, size_t n_z) {
  static const bool target_waves_env = std::getenv("RR_FS1_TARGET_WAVES") != nullptr;  // (a tuning override of the chunk plan: launched path only)
  // (one workgroup moves every plane of every particle when the gate fires: beyond ~1 MB of maps the launched path's grid-wide
  // gather is faster than the launches it costs)
  return h->res.enabled && h->algorithm == 1 && h->n == h->n_global && !h->pl.inbox && !h->p2p.ready && h->n <= 1024 &&
         h->n * h->n_planes <= 131072 && n_z <= (size_t)kFs1ResMaxObs && !h->prof.on && !target_waves_env;
}

static rr_status fs1_resident_launch(rr_fs1* h, uint64_t first_seq, unsigned int step0, unsigned int rstep0) {
  rr_status s = materialise(h);  // nothing pending, Ctl.cur settled: the kernel gathers eagerly from here on
  if (s != RR_OK) return s;
  if (!h->res.ring) {
    RR_HIP_TRY(hipHostMalloc(&h->res.ring, sizeof(rr::ResidentRing), hipHostMallocDefault));
    std::memset(h->res.ring, 0, sizeof(rr::ResidentRing));
  }
  Fs1SmallArgs a{};
  a.n = h->n;
  a.L = h->L;
  a.n_global = h->n_global;
  a.gid0 = h->gid0;
  a.seed = h->opt.seed;
  a.step0 = step0;
  a.rstep0 = rstep0;
  a.m = model_of(h);
  a.plan = plan_args(h, 0, NAN, /*lazy=*/false);
  a.res.on = 1;
  a.res.payload_cap = kFs1ResPayload;
  a.res.first_seq = first_seq;
  a.res.idle_ticks = (uint64_t)(h->res.idle_us * 100.0);
  a.res.life_ticks = (uint64_t)(h->res.life_us * 100.0);
  a.res.launch_id = ++h->res.launch_id;
  unsigned int* idx_out = h->idx;  // (rr_fs1_last_resample_indices reads it)
  if (h->n <= 128) hipLaunchKernelGGL(k_fs1_small<128>, dim3(1), dim3(128), 0, h->stream, h->pl, h->pw, h->ctl, a, idx_out, h->res.ring);
  else if (h->n <= 256) hipLaunchKernelGGL(k_fs1_small<256>, dim3(1), dim3(256), 0, h->stream, h->pl, h->pw, h->ctl, a, idx_out, h->res.ring);
  else if (h->n <= 512) hipLaunchKernelGGL(k_fs1_small<512>, dim3(1), dim3(512), 0, h->stream, h->pl, h->pw, h->ctl, a, idx_out, h->res.ring);
  else hipLaunchKernelGGL(k_fs1_small<1024>, dim3(1), dim3(1024), 0, h->stream, h->pl, h->pw, h->ctl, a, idx_out, h->res.ring);
  RR_HIP_TRY(hipGetLastError());
  h->res.live = true;
  h->res.launches += 1;
  h->wmax_live = false;
  h->wmax_bits_clean = true;  // (finalize_plan zeroes the accumulator after every update)
  return RR_OK;
}

static rr_status fs1_resident_await(rr_fs1* h, uint64_t seq) {
  rr_fs1::Resident& r = h->res;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    uint64_t e[5], flags;
    if (rr::ring_take(&r.ring->rsp[kFs1RspIndex], seq, &e[4]) && rr::ring_take(&r.ring->rsp[rr::kResRspFlags], seq, &flags) &&
        rr::ring_take(&r.ring->rsp[3], seq, &e[3]) && rr::ring_take(&r.ring->rsp[2], seq, &e[2]) && rr::ring_take(&r.ring->rsp[1], seq, &e[1]) &&
        rr::ring_take(&r.ring->rsp[0], seq, &e[0])) {
      for (int k = 0; k < 3; ++k) std::memcpy(&r.best_pose[k], &e[k], sizeof(double));
      std::memcpy(&r.best_weight, &e[3], sizeof(double));
      r.best_index = e[4];
      r.have_best = true;
      r.pending = false;
      return RR_OK;
    }
    uint64_t consumed = 0;
    if (r.live && rr::ring_take(&r.ring->rsp[rr::kResRspExit], r.launch_id, &consumed)) {
      r.live = false;  // this incarnation has left (idle / end of life); a command it did not take waits for the next one
      if (consumed < seq) {
        rr_status s = fs1_resident_launch(h, seq, r.cmd_step, r.cmd_rstep);
        if (s != RR_OK) return s;
      }
      continue;
    }
    if ((spins & 1023u) == 1023u && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > 2000) {
      (void)hipStreamSynchronize(h->stream);
      r.live = false;
      r.pending = false;
      return fail(RR_RUNTIME_ERROR, "the resident FastSLAM kernel did not answer");
    }
  }
}

// one update through the resident kernel; wait: for the answer (the best particle of the updated set)
static rr_status fs1_resident_update(rr_fs1* h, const double u[2], const double* z, size_t n_z, bool dup, bool wait) {
  rr_fs1::Resident& r = h->res;
  rr_status s;
  if (r.pending && (s = fs1_resident_await(h, r.seq)) != RR_OK) return s;  // one command in flight
  if (!r.ring || !r.live) {
    if ((s = fs1_resident_launch(h, r.seq + 1, h->step, h->rstep)) != RR_OK) return s;
  }
  const uint64_t seq = ++r.seq;
  r.cmd_step = h->step;
  r.cmd_rstep = h->rstep;
  r.have_best = false;
  const int chunks = n_z ? choose_chunks(h, n_z, dup) : 1;
  const int len = n_z ? (int)((n_z + chunks - 1) / chunks) : 1;
  h->last_chunks = chunks;
  auto bits_of = [](double v) {
    uint64_t q;
    std::memcpy(&q, &v, sizeof q);
    return q;
  };
  rr::MailPair* c = r.ring->cmd;
  for (size_t i = 0; i < 3 * n_z; ++i) rr::ring_put(&c[4 + i], bits_of(z[i]), seq);
  rr::ring_put(&c[3], bits_of((double)len), seq);
  rr::ring_put(&c[2], bits_of(u[1]), seq);
  rr::ring_put(&c[1], bits_of(u[0]), seq);
  rr::ring_put(&c[0], (uint64_t)rr::kResOpStep | ((uint64_t)(3 + 3 * n_z) << 8), seq);
  h->step += 1;
  h->rstep += 1;
  h->z_staged = false;
  r.steps += 1;
  r.pending = true;
  return wait ? fs1_resident_await(h, seq) : RR_OK;
}

namespace {
rr_status fs1_resident_park(rr_fs1* h) {
  rr_fs1::Resident& r = h->res;
  rr_status s = RR_OK;
  if (r.pending) s = fs1_resident_await(h, r.seq);
  if (r.live) {
    rr::ring_put(&r.ring->cmd[0], (uint64_t)rr::kResOpQuit, ++r.seq);
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
    r.live = false;
  }
  return s;
}
}  // namespace

// The resident service of a small FastSLAM 1.0 filter (<= 1024 particles, <= 64 observations per update): idle_us > 0 switches
// it on -- rr_fs1_update / rr_fs1_update_async then talk to ONE kernel that stays on the device and answers every update with
// the best particle of the updated set (rr_fs1_best_particle right after it costs nothing); 0 switches it off.
rr_status rr_fs1_set_resident(rr_fs1* h, double idle_us) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!(idle_us >= 0.0) || !(idle_us <= 1e7)) return fail(RR_INVALID_PARAMETER, "resident idle time must lie in [0, 1e7] microseconds");
  h->res.enabled = idle_us > 0.0;
  h->res.idle_us = idle_us;
  h->res.life_us = std::max(100000.0, 20.0 * idle_us);
  return RR_OK;
}

rr_status rr_fs1_resident_stats(const rr_fs1* h, uint64_t* launches, uint64_t* updates) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  if (launches) *launches = h->res.launches;
  if (updates) *updates = h->res.steps;
  return RR_OK;
}

rr_status rr_fs1_update_async(rr_fs1* h, const double u[2], const double* z, size_t n_z) {
  if (h && fs1_resident_path(h, n_z)) {
    rr_status rs = bind(h, /*keep_resident=*/true);
    if (rs != RR_OK) return rs;
    if ((rs = validate_u(u)) != RR_OK) return rs;
    bool rdup;
    if ((rs = validate_z(h, z, n_z, &rdup)) != RR_OK) return rs;
    return fs1_resident_update(h, u, z, n_z, rdup, /*wait=*/false);
  }
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = validate_u(u)) != RR_OK) return s;
  bool dup;
  if ((s = validate_z(h, z, n_z, &dup)) != RR_OK) return s;
  if (dup) {
    // the same landmark twice in one step: the second update must see the first one's result,
    // which the read-through-idx scheme cannot give -- settle first, then update in place
    if ((s = materialise(h)) != RR_OK) return s;
    if ((s = launch_motion<false>(h, u, z, n_z)) != RR_OK) return s;
    if ((s = launch_observe(h, z, n_z, dup)) != RR_OK) return s;
    if (plan_fusable(h)) return launch_plan_fused(h, /*settle=*/0);
    if ((s = launch_sums(h, 0, NAN, /*lazy=*/true, /*settle=*/0)) != RR_OK) return s;
    return launch_finish(h, /*lazy=*/true);
  }
  // lazy: predict and observe read the previous resample's survivors through idx and write the
  // other buffer set; unobserved landmarks are gathered separately; k_quantize_reduce settles
  if ((s = launch_motion<true>(h, u, z, n_z)) != RR_OK) return s;
  if ((s = launch_observe(h, z, n_z, dup, /*lazy=*/true)) != RR_OK) return s;
  if (h->maybe_pending && (s = launch_rest_gather(h, z, n_z)) != RR_OK) return s;
  h->maybe_pending = false;
  if (plan_fusable(h)) return launch_plan_fused(h, /*settle=*/1);
  if ((s = launch_sums(h, 0, NAN, /*lazy=*/true, /*settle=*/1)) != RR_OK) return s;
  return launch_finish(h, /*lazy=*/true);
}

rr_status rr_fs1_synchronize(rr_fs1* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  return synchronize_light(h);  // waits for the stream; also where a handle learns that its one-launch plan had to degrade
}

rr_status rr_fs1_update(rr_fs1* h, const double u[2], const double* z, size_t n_z) {
  if (h && fs1_resident_path(h, n_z)) {
    rr_status rs = bind(h, /*keep_resident=*/true);
    if (rs != RR_OK) return rs;
    if ((rs = validate_u(u)) != RR_OK) return rs;
    bool rdup;
    if ((rs = validate_z(h, z, n_z, &rdup)) != RR_OK) return rs;
    return fs1_resident_update(h, u, z, n_z, rdup, /*wait=*/true);
  }
  rr_status s = rr_fs1_update_async(h, u, z, n_z);
  if (s != RR_OK) return s;
  return rr_fs1_synchronize(h);
}

rr_status rr_fs1_best_particle(rr_fs1* h, double out_pose[3], double* out_weight, uint64_t* out_index) {
  if (h && (h->res.live || h->res.pending || h->res.have_best)) {
    // right after a resident update: its answer IS the best particle of the current set (fastslam1.rs:269-274)
    rr_status rs = bind(h, /*keep_resident=*/true);
    if (rs != RR_OK) return rs;
    if (h->res.pending && (rs = fs1_resident_await(h, h->res.seq)) != RR_OK) return rs;
    if (h->res.have_best) {
      if (out_pose) std::memcpy(out_pose, h->res.best_pose, sizeof(double) * 3);
      if (out_weight) *out_weight = h->res.best_weight;
      if (out_index) *out_index = h->res.best_index;
      return RR_OK;
    }
  }
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  // one GPU: a pending resample stays pending (the pose is read through idx), so the next update keeps its three launches;
  // a shard's idx may point into the inbox: settle first
  const bool through_idx = !h->pl.inbox && h->n == h->n_global;
  if ((s = through_idx ? ensure_resolved(h) : materialise(h)) != RR_OK) return s;
  if ((s = ensure_mailbox(h)) != RR_OK) return s;
  const int blocks = (int)std::min<uint64_t>(256, grid_for(h->n, kBlock));
  const uint64_t want = ++h->best_seq;
  hipLaunchKernelGGL(k_fs1_best, dim3(blocks), dim3(kBlock), 0, h->stream, (const double*)h->pw, h->n, (const double*)h->pl.s[0],
                     (const double*)h->pl.s[1], h->ctl, h->part_bits, h->part_idx, h->best_ticket, h->best_mail, want,
                     (const unsigned int*)(through_idx ? h->idx : nullptr));
  RR_HIP_TRY(hipGetLastError());
  if ((s = await_mailbox(h, want)) != RR_OK) return s;
  if (h->best_mail->flags && (s = fetch_ctl(h)) != RR_OK) return s;  // a latched device-side condition: report it as usual
  if (out_pose) std::memcpy(out_pose, h->best_mail->pose, sizeof(double) * 3);
  if (out_weight) *out_weight = h->best_mail->weight;
  if (out_index) *out_index = h->best_mail->index;
  return RR_OK;
}

rr_status rr_fs1_get_landmarks(rr_fs1* h, uint64_t particle_index, double* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if (particle_index >= h->n) return fail(RR_INVALID_PARAMETER, "particle index out of range");
  if (h->L == 0) return RR_OK;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  const int cur = h->ctl_host->cur;
  double* tmp = h->pl.s[cur ^ 1];  // the inactive buffer set is free between steps
  hipLaunchKernelGGL(k_fs1_one_landmarks, dim3(grid_for(6 * h->L, 256)), dim3(256), 0, h->stream,
                     (const double*)h->pl.s[cur], h->n, h->L, particle_index, tmp);
  RR_HIP_TRY(hipGetLastError());
  RR_HIP_TRY(hipMemcpyAsync(out, tmp, 6 * h->L * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}

rr_status rr_fs1_get_state(rr_fs1* h, double* poses_out, double* maps_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  const int cur = h->ctl_host->cur;
  double* tmp = h->pl.s[cur ^ 1];
  if (poses_out) {
    if ((s = ensure_pose_stage(h)) != RR_OK) return s;
    hipLaunchKernelGGL(k_fs1_poses, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pl.s[cur], h->pw,
                       h->pose_stage, h->n, 1);
    RR_HIP_TRY(hipGetLastError());
    RR_HIP_TRY(hipMemcpyAsync(poses_out, h->pose_stage, 4 * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (maps_out && h->L) {
    hipLaunchKernelGGL(k_fs1_maps_transpose, dim3(grid_for(h->n * h->L, kBlock)), dim3(kBlock), 0, h->stream, h->pl.s[cur],
                       tmp, h->n, h->L, 1);
    RR_HIP_TRY(hipGetLastError());
    RR_HIP_TRY(hipMemcpyAsync(maps_out, tmp, 6 * h->L * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return RR_OK;
}

rr_status rr_fs1_get_poses(rr_fs1* h, double* out) {
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  return rr_fs1_get_state(h, out, nullptr);
}

rr_status rr_fs1_set_state(rr_fs1* h, const double* poses, const double* maps) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  const int cur = h->ctl_host->cur;
  double* tmp = h->pl.s[cur ^ 1];
  if (poses) {
    if ((s = ensure_pose_stage(h)) != RR_OK) return s;
    RR_HIP_TRY(hipMemcpyAsync(h->pose_stage, poses, 4 * h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_fs1_poses, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pl.s[cur], h->pw,
                       h->pose_stage, h->n, 0);
    RR_HIP_TRY(hipGetLastError());
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (maps && h->L) {
    RR_HIP_TRY(hipMemcpyAsync(tmp, maps, 6 * h->L * h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_fs1_maps_transpose, dim3(grid_for(h->n * h->L, kBlock)), dim3(kBlock), 0, h->stream, h->pl.s[cur],
                       tmp, h->n, h->L, 0);
    RR_HIP_TRY(hipGetLastError());
    RR_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  // the weight maximum of the uploaded set (needed if normalize_resample is called next)
  if ((s = zero_wmax(h)) != RR_OK) return s;
  hipLaunchKernelGGL(k_fs1_wmax, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
  RR_HIP_TRY(hipGetLastError());
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  h->wmax_live = true;
  return RR_OK;
}

rr_status rr_fs1_update_host(rr_fs1* h, double* poses, double* maps, const double u[2], const double* z, size_t n_z) {
  if (!poses || (!maps && h && h->L)) return fail(RR_INVALID_PARAMETER, "null state");
  rr_status s = rr_fs1_set_state(h, poses, maps);
  if (s != RR_OK) return s;
  if ((s = rr_fs1_update(h, u, z, n_z)) != RR_OK) return s;
  return rr_fs1_get_state(h, poses, maps);
}

size_t rr_fs1_get_observations(const double x_true[3], const double* landmarks_xy, size_t n_landmarks,
                               const rr_fs1_params* params, uint64_t seed, uint32_t step, double* out, size_t cap) {
  rr_fs1_params prm;
  if (params) prm = *params; else rr_fs1_params_default(&prm);
  if (!x_true || (!landmarks_xy && n_landmarks) || (!out && cap)) return 0;
  const double sr0 = rr_sqrt(prm.r00), sr1 = rr_sqrt(prm.r11);
  size_t cnt = 0;
  for (size_t l = 0; l < n_landmarks; ++l) {
    const double dx = landmarks_xy[2 * l] - x_true[0];
    const double dy = landmarks_xy[2 * l + 1] - x_true[1];
    const double d = rr_sqrt(rr_fma(dy, dy, dx * dx));
    if (d <= prm.max_range) {  // fastslam1.rs:288
      const double angle = rr_normalize_angle(rr_atan2(dy, dx) - x_true[2]);
      double z0, z1;
      rr_normal2(seed, RR_STREAM_SIM, step, l, &z0, &z1);
      if (cnt < cap) {
        out[3 * cnt] = rr_fma(z0, sr0, d);
        out[3 * cnt + 1] = rr_fma(z1, sr1, angle);
        out[3 * cnt + 2] = (double)l;
      }
      ++cnt;
    }
  }
  return cnt;
}

// ---- sharded FastSLAM over the peer-to-peer transport
static rr_status fs1_check_geometry(const rr_fs1* h, int n_ranks, int rank) {
  if (n_ranks <= 0 || n_ranks > rr::kMaxP2P || rank < 0 || rank >= n_ranks)
    return fail(RR_INVALID_PARAMETER, "peer-to-peer transport supports 1..16 ranks");
  if (h->n_global != h->n * (uint64_t)n_ranks || h->gid0 != h->n * (uint64_t)rank)
    return fail(RR_INVALID_PARAMETER, "shard geometry does not match the rank layout (equal blocks, rank * n_local)");
  return RR_OK;
}

static rr_status fs1_alloc_ridx(rr_fs1* h) {
  if (h->ridx) return RR_OK;
  RR_HIP_TRY(hipMalloc(&h->ridx, h->n_global * sizeof(unsigned int)));  // worst case: every slot of every peer
  return RR_OK;
}

rr_status rr_fs1_p2p_export(rr_fs1* h, uint8_t out[RR_P2P_HANDLE_BYTES]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  return h->p2p.export_handles(h->slab, h->n_planes * h->n, out);
}

rr_status rr_fs1_p2p_connect(rr_fs1* h, const uint8_t* all_handles, int32_t n_ranks, int32_t rank) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!all_handles) return fail(RR_INVALID_PARAMETER, "null handles");
  if ((s = fs1_check_geometry(h, n_ranks, rank)) != RR_OK) return s;
  if ((s = fs1_alloc_ridx(h)) != RR_OK) return s;
  if ((s = h->p2p.connect_ipc(h->slab, h->n_planes * h->n, all_handles, n_ranks, rank)) != RR_OK) return s;
  h->pl.inbox = h->p2p.inbox;
  return RR_OK;
}

rr_status rr_fs1_p2p_connect_local(rr_fs1* const* handles, int32_t n_ranks) {
  if (!handles || n_ranks <= 0 || n_ranks > rr::kMaxP2P) return fail(RR_INVALID_PARAMETER, "bad handle list");
  rr::P2PState* st[rr::kMaxP2P];
  double* slabs[rr::kMaxP2P];
  size_t inboxes[rr::kMaxP2P];
  int devs[rr::kMaxP2P];
  for (int g = 0; g < n_ranks; ++g) {
    if (!handles[g]) return fail(RR_INVALID_PARAMETER, "null handle");
    rr_status s = fs1_check_geometry(handles[g], n_ranks, g);
    if (s != RR_OK) return s;
    if ((s = bind(handles[g])) != RR_OK) return s;
    if ((s = fs1_alloc_ridx(handles[g])) != RR_OK) return s;
    st[g] = &handles[g]->p2p;
    slabs[g] = handles[g]->slab;
    inboxes[g] = handles[g]->n_planes * handles[g]->n;
    devs[g] = handles[g]->opt.device;
  }
  rr_status s = rr::p2p_link_local(st, slabs, inboxes, devs, n_ranks);
  if (s != RR_OK) return s;
  for (int g = 0; g < n_ranks; ++g) handles[g]->pl.inbox = handles[g]->p2p.inbox;
  return RR_OK;
}

rr_status rr_fs1_p2p_status(rr_fs1* h, int32_t* timed_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!timed_out) return fail(RR_INVALID_PARAMETER, "null output");
  return h->p2p.status(h->stream, timed_out);
}

rr_status rr_fs1_shard_update_p2p(rr_fs1* h, const double u[2], const double* z, size_t n_z) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->p2p.ready) return fail(RR_INVALID_PARAMETER, "call rr_fs1_p2p_connect first");
  if ((s = validate_u(u)) != RR_OK) return s;
  bool dup;
  if ((s = validate_z(h, z, n_z, &dup)) != RR_OK) return s;
  const uint64_t seq = ++h->p2p.seq;
  uint64_t* gathered = h->p2p.gathered();
  uint64_t* local3 = h->p2p.local3();
  // local part, exactly as rr_fs1_update_async: predict + per-observation EKF, reading the previous
  // resample's survivors through idx (kInPlace = stored by a peer) unless a landmark repeats
  if (dup) {
    if ((s = materialise(h)) != RR_OK) return s;
    if ((s = launch_motion<false>(h, u, z, n_z)) != RR_OK) return s;
    if ((s = launch_observe(h, z, n_z, dup)) != RR_OK) return s;
  } else {
    if ((s = launch_motion<true>(h, u, z, n_z)) != RR_OK) return s;
    if ((s = launch_observe(h, z, n_z, dup, /*lazy=*/true)) != RR_OK) return s;
    if (h->maybe_pending && (s = launch_rest_gather(h, z, n_z)) != RR_OK) return s;
    h->maybe_pending = false;
  }
  PlanArgs pa = plan_args(h, 0, NAN, /*lazy=*/true);
  // exchange 1: global maximum -> Ctl.wmax
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PWmax, seq,
                     (const uint64_t*)&h->ctl->wmax_bits, gathered, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_QUANTIZE_REDUCE);
    hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl,
                       (const double*)&h->ctl->wmax, image_args(h), h->tile_total, h->tile_q2, dup ? 0 : 1);
  }
  // tile scan + exchange 2: every shard's sums -> global totals, gate (N_eff < NTH), plan
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_exchange, dim3(1), dim3(kScanThreads), 0, h->stream, h->p2p.peers, seq, h->tile_total,
                       (const uint64_t*)h->tile_q2, h->n_tiles, gathered, h->ctl, pa, h->p2p.err);
  }
  h->wmax_live = false;
  h->wmax_bits_clean = false;  // a peer wait that gave up skips finalize_plan: do not rely on the zeroed accumulator here
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_CDF);
    hipLaunchKernelGGL(rr::k_cdf, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl, image_args(h),
                       h->tile_total, h->cdf, (uint64_t*)nullptr, 0);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_NORMALIZE);
    hipLaunchKernelGGL(k_fs1_normalize, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
  }
  const unsigned own_blocks = grid_for(h->n, kBlock);
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_INDICES);
    hipLaunchKernelGGL(k_fs1_indices_sharded, dim3(own_blocks + 64), dim3(kBlock), 0, h->stream, h->ctl, h->cdf, h->n, h->gid0,
                       own_blocks, h->idx, h->ridx);
    hipLaunchKernelGGL(k_fs1_uniform_weights, dim3(own_blocks), dim3(kBlock), 0, h->stream, h->ctl, h->pw, h->n, h->n_global);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_GATHER);
    hipLaunchKernelGGL(k_fs1_push, dim3(32, grid_for(h->n_planes, kPlanesPerThread)), dim3(kBlock), 0, h->stream, h->pl, h->ctl,
                       h->ridx, h->n, h->gid0, h->n_planes, h->p2p.peers);
  }
  // exchange 3: everybody has finished writing into everybody's slab
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PDone, seq,
                     (const uint64_t*)local3, gathered, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  RR_HIP_TRY(hipGetLastError());
  h->maybe_pending = true;  // the next update reads through idx (or an accessor materialises)
  h->rstep += 1;
  return RR_OK;
}

// ---- sharded FastSLAM over RCCL (and the same phases for a host-orchestrated transport / the tests):
//   local    predict + per-observation EKF, local weight maximum
//            all-reduce(MAX) of one double
//   quantize integer image under the GLOBAL maximum, local (T, sum q^2)
//            all-gather of 3 x u64 per rank
//   plan     global totals, N_eff gate (fastslam1.rs:262-265), systematic plan (:205-234), local CDF, source of every
//            own slot (kInPlace when a peer serves it), sources of the served slots that belong to peers
//   pack     those particles (3 + 6L planes each) into one block per destination
//            grouped send / recv of the blocks
//   unpack   received blocks into this rank's inbox; the next update reads through idx as on one GPU
rr_status rr_fs1_shard_local(rr_fs1* h, const double u[2], const double* z, size_t n_z, double* d_wmax_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = validate_u(u)) != RR_OK) return s;
  bool dup;
  if ((s = validate_z(h, z, n_z, &dup)) != RR_OK) return s;
  if ((s = fs1_alloc_ridx(h)) != RR_OK) return s;
  if (dup) {
    if ((s = materialise(h)) != RR_OK) return s;
    if ((s = launch_motion<false>(h, u, z, n_z)) != RR_OK) return s;
    if ((s = launch_observe(h, z, n_z, dup)) != RR_OK) return s;
  } else {
    if ((s = launch_motion<true>(h, u, z, n_z)) != RR_OK) return s;
    if ((s = launch_observe(h, z, n_z, dup, /*lazy=*/true)) != RR_OK) return s;
    if (h->maybe_pending && (s = launch_rest_gather(h, z, n_z)) != RR_OK) return s;
    h->maybe_pending = false;
  }
  h->shard_settle = dup ? 0 : 1;
  if (d_wmax_out)  // bit pattern of a non-negative double == the double
    RR_HIP_TRY(hipMemcpyAsync(d_wmax_out, &h->ctl->wmax_bits, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return RR_OK;
}

rr_status rr_fs1_shard_quantize(rr_fs1* h, const double* d_wmax_global, uint64_t* d_sums_out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!d_wmax_global || !d_sums_out) return fail(RR_INVALID_PARAMETER, "null device pointer");
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_QUANTIZE_REDUCE);
    hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl,
                       d_wmax_global, image_args(h), h->tile_total, h->tile_q2, h->shard_settle);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 0, plan_args(h, 0, NAN, /*lazy=*/true), d_sums_out);
  }
  RR_HIP_TRY(hipGetLastError());
  h->shard_settle = 0;
  h->wmax_live = false;
  return RR_OK;
}

rr_status rr_fs1_shard_plan(rr_fs1* h, const uint64_t* d_all_sums, int32_t n_shards, int32_t rank) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!d_all_sums) return fail(RR_INVALID_PARAMETER, "null device pointer");
  if ((s = fs1_check_geometry(h, n_shards, rank)) != RR_OK) return s;
  hipLaunchKernelGGL(rr::k_shard_plan, dim3(1), dim3(64), 0, h->stream, h->ctl, d_all_sums, (int)n_shards, (int)rank,
                     plan_args(h, 0, NAN, /*lazy=*/true));
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_CDF);
    hipLaunchKernelGGL(rr::k_cdf, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl, image_args(h),
                       h->tile_total, h->cdf, (uint64_t*)nullptr, 0);
  }
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_NORMALIZE);
    hipLaunchKernelGGL(k_fs1_normalize, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n);
  }
  const unsigned own_blocks = grid_for(h->n, kBlock);
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_INDICES);
    hipLaunchKernelGGL(k_fs1_indices_sharded, dim3(own_blocks + 64), dim3(kBlock), 0, h->stream, h->ctl, h->cdf, h->n, h->gid0,
                       own_blocks, h->idx, h->ridx);
    hipLaunchKernelGGL(k_fs1_uniform_weights, dim3(own_blocks), dim3(kBlock), 0, h->stream, h->ctl, h->pw, h->n, h->n_global);
  }
  RR_HIP_TRY(hipGetLastError());
  h->maybe_pending = true;  // the next update reads through idx (or an accessor materialises)
  h->wmax_bits_clean = true;  // k_shard_plan's finalize_plan zeroed the accumulator
  h->rstep += 1;
  return RR_OK;
}

rr_status rr_fs1_shard_get_plan(rr_fs1* h, rr_pf_shard_plan* out) {
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

// block tables of one exchange from the segment matrix (row `rank` = what this rank sends, column `rank` = what it receives)
static void fs1_chunk_tables(const int64_t* M, int G, int r, ChunkTable* send, ChunkTable* recv) {
  send->n_ranks = recv->n_ranks = G;
  uint64_t so = 0, ro = 0, li = 0;
  for (int g = 0; g < G; ++g) {
    send->start[g] = so;
    recv->start[g] = ro;
    recv->first_local[g] = li;
    send->first_local[g] = 0;
    li += (uint64_t)M[(size_t)g * G + r];
    if (g == r) continue;  // own slots stay where they are (lazy gather through idx)
    so += (uint64_t)M[(size_t)r * G + g];
    ro += (uint64_t)M[(size_t)g * G + r];
  }
  send->start[G] = so;
  recv->start[G] = ro;
}

rr_status rr_fs1_shard_pack(rr_fs1* h, const int64_t* matrix, int32_t n_shards, int32_t rank, double* d_send) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!matrix) return fail(RR_INVALID_PARAMETER, "null segment matrix");
  if ((s = fs1_check_geometry(h, n_shards, rank)) != RR_OK) return s;
  ChunkTable st, rt;
  fs1_chunk_tables(matrix, n_shards, rank, &st, &rt);
  if (st.start[n_shards] == 0) return RR_OK;
  if (!d_send) return fail(RR_INVALID_PARAMETER, "null send buffer");
  rr::ScopedTimer t(h->prof, h->stream, RR_FK_GATHER);
  const unsigned gx = std::min<unsigned>(grid_for(st.start[n_shards], kBlock), 1024u);
  hipLaunchKernelGGL(k_fs1_pack, dim3(gx, grid_for(h->n_planes, kPlanesPerThread)), dim3(kBlock), 0, h->stream, h->pl, h->ctl,
                     (const unsigned int*)h->ridx, h->n, h->n_planes, st, d_send);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status rr_fs1_shard_unpack(rr_fs1* h, const int64_t* matrix, int32_t n_shards, int32_t rank, const double* d_recv) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!matrix) return fail(RR_INVALID_PARAMETER, "null segment matrix");
  if ((s = fs1_check_geometry(h, n_shards, rank)) != RR_OK) return s;
  ChunkTable st, rt;
  fs1_chunk_tables(matrix, n_shards, rank, &st, &rt);
  if (rt.start[n_shards] == 0) return RR_OK;
  if (!d_recv) return fail(RR_INVALID_PARAMETER, "null receive buffer");
  if (!h->pl.inbox) {  // no peer-to-peer connection: a plain device mirror of one buffer set serves as the inbox
    RR_HIP_TRY(hipMalloc(&h->own_inbox, h->n_planes * h->n * sizeof(double)));
    h->pl.inbox = h->own_inbox;
  }
  rr::ScopedTimer t(h->prof, h->stream, RR_FK_GATHER);
  const unsigned gx = std::min<unsigned>(grid_for(rt.start[n_shards], kBlock), 1024u);
  hipLaunchKernelGGL(k_fs1_unpack, dim3(gx, grid_for(h->n_planes, kPlanesPerThread)), dim3(kBlock), 0, h->stream,
                     const_cast<double*>(h->pl.inbox), h->ctl, h->n, h->n_planes, rt, d_recv);
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

uint64_t rr_fs1_shard_last_migrated(const rr_fs1* h) { return h ? h->last_migrated : 0; }

rr_status rr_fs1_shard_update(rr_fs1* h, rr_comm* c, const double u[2], const double* z, size_t n_z) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!c) return fail(RR_INVALID_PARAMETER, "null communicator");
  if ((s = fs1_check_geometry(h, c->n_ranks, c->rank)) != RR_OK) return s;
  rr::Rccl& R = rr::rccl();
  const int G = c->n_ranks, r = c->rank;
  if ((s = rr_fs1_shard_local(h, u, z, n_z, c->d_wmax)) != RR_OK) return s;
  RR_NCCL_TRY(R.AllReduce(c->d_wmax, c->d_wmax, 1, rr::kNcclFloat64, rr::kNcclMax, c->comm, h->stream));
  if ((s = rr_fs1_shard_quantize(h, c->d_wmax, c->d_sums)) != RR_OK) return s;
  RR_NCCL_TRY(R.AllGather(c->d_sums, c->d_all, 3, rr::kNcclUint64, c->comm, h->stream));
  if ((s = rr_fs1_shard_plan(h, c->d_all, G, r)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(c->h_all, c->d_all, 3 * (size_t)G * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  rr_pf_shard_plan plan;
  if ((s = rr_fs1_shard_get_plan(h, &plan)) != RR_OK) return s;  // synchronises the stream
  h->last_migrated = 0;
  if (!plan.fired || G == 1) return RR_OK;
  std::vector<uint64_t> totals(G);
  for (int g = 0; g < G; ++g) totals[g] = c->h_all[3 * g];
  (void)rr_sys_segment_matrix(plan.rho, totals.data(), G, h->n_global, h->n, r, c->matrix.data());
  const int64_t* M = c->matrix.data();
  uint64_t n_recv = 0, n_send_remote = 0, n_recv_remote = 0, migrated = 0;
  for (int g = 0; g < G; ++g) {
    n_recv += (uint64_t)M[(size_t)g * G + r];
    if (g != r) {
      n_send_remote += (uint64_t)M[(size_t)r * G + g];
      n_recv_remote += (uint64_t)M[(size_t)g * G + r];
    }
    for (int d = 0; d < G; ++d)
      if (g != d) migrated += (uint64_t)M[(size_t)g * G + d];
  }
  h->last_migrated = migrated;
  if (n_recv != h->n) return fail(RR_RUNTIME_ERROR, "segment plan does not cover this shard's slots exactly once");
  auto ensure = [&](double** buf, size_t* cap, size_t need) -> rr_status {
    if (need <= *cap) return RR_OK;
    if (*buf) RR_HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 4096;
    RR_HIP_TRY(hipMalloc(buf, want * sizeof(double)));
    *cap = want;
    return RR_OK;
  };
  if ((s = ensure(&c->d_fsend, &c->cap_fsend, n_send_remote * h->n_planes)) != RR_OK) return s;
  if ((s = ensure(&c->d_frecv, &c->cap_frecv, n_recv_remote * h->n_planes)) != RR_OK) return s;
  if ((s = rr_fs1_shard_pack(h, M, G, r, c->d_fsend)) != RR_OK) return s;
  RR_NCCL_TRY(R.GroupStart());
  uint64_t so = 0, ro = 0;
  for (int g = 0; g < G; ++g) {
    if (g == r) continue;
    const uint64_t ns = (uint64_t)M[(size_t)r * G + g] * h->n_planes, nr = (uint64_t)M[(size_t)g * G + r] * h->n_planes;
    if (ns) RR_NCCL_TRY(R.Send(c->d_fsend + so, ns, rr::kNcclFloat64, g, c->comm, h->stream));
    if (nr) RR_NCCL_TRY(R.Recv(c->d_frecv + ro, nr, rr::kNcclFloat64, g, c->comm, h->stream));
    so += ns;
    ro += nr;
  }
  RR_NCCL_TRY(R.GroupEnd());
  return rr_fs1_shard_unpack(h, M, G, r, c->d_frecv);
}

rr_status rr_fs1_last_resample_fired(rr_fs1* h, int32_t* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  *out = h->ctl_host->fired;
  return RR_OK;
}

rr_status rr_fs1_last_resample_indices(rr_fs1* h, uint32_t* out, size_t n) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out || n != h->n) return fail(RR_INVALID_PARAMETER, "need room for one index per particle");
  if ((s = ensure_resolved(h)) != RR_OK) return s;
  RR_HIP_TRY(hipMemcpyAsync(out, h->idx, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  return RR_OK;
}

rr_status rr_fs1_n_eff(rr_fs1* h, double* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = launch_sums(h, 2, NAN)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  *out = h->ctl_host->neff;
  return RR_OK;
}

rr_status rr_fs1_get_fixed_sums(rr_fs1* h, rr_pf_fixed_sums* out) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  if ((s = launch_sums(h, 2, NAN)) != RR_OK) return s;
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

rr_status rr_fs1_plan_stats(rr_fs1* h, uint64_t* giveups, int32_t* one_launch_enabled) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  if (giveups) *giveups = h->plan_giveups;
  if (one_launch_enabled) *one_launch_enabled = (h->grid_capacity != 0 && h->grid_capacity != ~0ull) ? 1 : 0;
  return RR_OK;
}

rr_status rr_fs1_get_counters(rr_fs1* h, uint32_t* step, uint32_t* resample_step, int32_t* obs_chunks) {
  if (!h) return fail(RR_INVALID_PARAMETER, "null handle");
  if (step) *step = h->step;
  if (resample_step) *resample_step = h->rstep;
  if (obs_chunks) *obs_chunks = h->last_chunks;
  return RR_OK;
}

rr_status rr_fs1_profile_enable(rr_fs1* h, int32_t enable) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  h->prof.drain();
  h->prof.on = enable != 0;
  h->prof.dispatch_only = enable == 2;  // 2 = only k_fs1_observe, timed by its own dispatch packet
  return RR_OK;
}

rr_status rr_fs1_profile_read(rr_fs1* h, int32_t kernel_id, uint64_t* launches, double* total_ms) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (kernel_id < 0 || kernel_id >= RR_FK_COUNT) return fail(RR_INVALID_PARAMETER, "kernel id out of range");
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  h->prof.drain();
  if (launches) *launches = h->prof.launches[kernel_id];
  if (total_ms) *total_ms = h->prof.ms[kernel_id];
  return RR_OK;
}

rr_status rr_fs1_profile_reset(rr_fs1* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  h->prof.reset();
  return RR_OK;
}

const char* rr_fs1_kernel_name(int32_t kernel_id) {
  return kernel_id >= 0 && kernel_id < RR_FK_COUNT ? kFkNames[kernel_id] : "";
}

}  // extern "C"
