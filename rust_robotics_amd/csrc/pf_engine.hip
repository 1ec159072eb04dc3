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
// One translation unit in five files: this one (handle, launch logic, the C ABI of the unsharded filter) #includes, in place,
//   pf_kernels_step.inc      the step kernels of the large filters (k_propagate_weight, k_step_lazy, gathers, moments, mailbox)
//   pf_kernels_small.inc     k_step_small: the one-workgroup step of small sets, K steps per launch, the resident command loop
//   pf_kernels_adaptive.inc  p2p gather kernels, the KLD-adaptive resample (wide + k_mcl_adaptive_small), sharded multinomial
//   pf_sharded_api.inc       the C ABI of the sharded filters (RCCL through dlopen, peer-to-peer, multinomial shards)
// and shares resample_core.hpp / p2p_core.hpp / resident_core.hpp / rr_common.hpp with fs1_engine.hip.
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

#include "pf_kernels_step.inc"
#include "pf_kernels_small.inc"

}  // namespace

#include "pf_kernels_adaptive.inc"

// =============================================================================================
// host side
// =============================================================================================

#if defined(RR_DEBUG_TRACE)
constexpr int kDbgWords = 48;
#endif
struct rr_pf {
#if defined(RR_DEBUG_TRACE)
  uint64_t* dbg_trace = nullptr;  // [dbg_cap][kDbgWords] (instrumented build)
  uint32_t dbg_cap = 0;
#endif
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
  double* est_total_dev = nullptr;     // [4]: the slot tiles' sums added up on the device for an accessor (k_est_slots_total)
  bool est_deferred = false;           // the last plan was asked for the deferred form and nobody has moved the particles yet
  uint64_t shard_est_stamp = 0;        // rr_pf_shard_want_estimate: the resample step (rstep, 1-based) whose sums were asked for last
  bool shard_est = false;              // rr_pf_shard_want_estimate: every window step of this shard leaves its part of the mean
  bool shard_est_ever = false;         // ... has been asked for at some point (k_est_slots then trusts the caller about Ctl.fired)
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
  int spin_gate_no = 0;   // rr::spin_permit's gate: the device, or this shard's own part of its CUs (RR_P2P_CU_PARTITION)
  int cu_part_cus = 0;    // > 0: the handle's stream is confined to that many CUs of its own (p2p_apply_cu_partition)
  bool mn_push_lds_set = false;  // k_mn_push_p2p's dynamic-LDS limit has been raised on this handle's device
  int p2p_last_form = 0;  // rr_pf_p2p_topology: 1 = the last rr_pf_shard_step_p2p took the lazy window step, 2 = the eager step
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
  double* mn_records = nullptr;         // multinomial shards, peer-to-peer: the weighted particles as {x, y, yaw, v} records (k_mn_push_p2p's source)
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
    RR_HIP_TRY(rr::dev_malloc(&h->obs_dev, 3 * n_obs * sizeof(double)));
    h->obs_cap = n_obs;
  }
  // pageable source: HIP stages it before returning, so the caller's buffer may be reused
  RR_HIP_TRY(hipMemcpyAsync(h->obs_dev, obs, 3 * n_obs * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return RR_OK;
}

template <bool PREDICT, bool WEIGHT, bool EXPLICIT>
rr_status launch_pw(rr_pf* h, const StepParams& p, const ObsArg& arg, bool kernarg, double* packed = nullptr) {
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
                         h->stream, h->b, h->w, h->ctl, p, arg, (const double*)nullptr, h->scratch_a, h->scratch_b, packed);
    else
      hipLaunchKernelGGL((k_propagate_weight<PREDICT, WEIGHT, EXPLICIT, false>), dim3(grid), dim3(kBlock), lds,
                         h->stream, h->b, h->w, h->ctl, p, arg, (const double*)h->obs_dev, h->scratch_a,
                         h->scratch_b, packed);
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

// the deferred estimate's per-wave sums, and behind them ONE word: the resample step a shard's sums belong to (WindowArgs.est_stamp)
static size_t est_slot_bytes(const rr_pf* h) { return ((size_t)grid_for(h->cap, rr::kResolveSlots) * kEstSlotWords + 1) * sizeof(double); }
static uint64_t* est_stamp_slot(const rr_pf* h) {
  return h->est_slot_partials ? reinterpret_cast<uint64_t*>(h->est_slot_partials + (size_t)grid_for(h->cap, rr::kResolveSlots) * kEstSlotWords) : nullptr;
}

// make a pending lazy resample real (accessors and the non-fused entry points call this first)
void launch_guide_search(rr_pf* h, const double* r_explicit_dev, unsigned int* lidx, const GatherArgs& g);

// the deferred in-step estimate of a resample an accessor made real (the next step's k_step_lazy would have summed it on the way)
static void launch_est_slots(rr_pf* h) {
  if (!h->est_deferred) return;
  h->est_deferred = false;
  hipLaunchKernelGGL(k_est_slots, dim3(grid_for(h->n, rr::kResolveSlots)), dim3(kBlock), 0, h->stream, h->b, (const Ctl*)h->ctl, h->n,
                     h->est_slot_partials, h->shard_est_ever ? 1 : 0, est_stamp_slot(h), h->shard_est_stamp);
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
    launch_est_slots(h);
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
    if ((e = rr::dev_malloc(&h->guide, nb * sizeof(unsigned int))) != hipSuccess) return e;
    if ((e = rr::dev_malloc(&h->guide_markers, nb * sizeof(unsigned int))) != hipSuccess) return e;
    if ((e = rr::dev_malloc(&h->guide_carry, nc * sizeof(unsigned int))) != hipSuccess) return e;
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
void launch_guide_resolve(rr_pf* h, int local = 0) {
  hipLaunchKernelGGL(rr::k_guide_resolve, dim3((unsigned)((((size_t)1 << h->guide_log2) + rr::kResolveSlots) / rr::kResolveSlots)),
                     dim3(kBlock), 0, h->stream, h->ctl, h->guide_markers, h->guide_carry, h->guide, h->guide_log2, local);
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
                          rr::spin_permit(h->spin_gate_no, h);
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
        RR_HIP_TRY(rr::dev_malloc(&h->est_slot_partials, est_slot_bytes(h)));
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
  rr::spin_release(h->spin_gate_no, h);
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
  rr::spin_release(h->spin_gate_no, h);  // the stream is idle
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
  h->spin_gate_no = h->opt.device & 63;
  const size_t nb = h->cap * sizeof(double);
  // one slab [set][field][cap] so that peers can map the whole particle state with one IPC handle
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->slab, 8 * nb));
  for (int k = 0; k < 2; ++k) {
    h->b.x[k] = h->slab + (size_t)(4 * k + 0) * h->cap;
    h->b.y[k] = h->slab + (size_t)(4 * k + 1) * h->cap;
    h->b.yaw[k] = h->slab + (size_t)(4 * k + 2) * h->cap;
    h->b.v[k] = h->slab + (size_t)(4 * k + 3) * h->cap;
  }
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->w, nb));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->cdf, h->cap * sizeof(uint64_t)));
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
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->cdf_coarse, (((h->cap + (1ull << h->coarse_log2) - 1) >> h->coarse_log2) + 1) * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->tile_total, cap_tiles * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->tile_q2, 2 * cap_tiles * sizeof(uint64_t)));
  if (opt.record_indices) RR_TRY_OR_CLEAN(rr::dev_malloc(&h->idx, h->cap * sizeof(unsigned int)));
  if (kld) {
    h->kld_hash_size = 2;
    while (h->kld_hash_size < 2 * h->cap) h->kld_hash_size *= 2;  // at most half full
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->kld_keys, 3 * h->cap * sizeof(int32_t)));
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->kld_table, h->kld_hash_size * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->kld_minslot, h->kld_hash_size * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->kld_table, 0xff, h->kld_hash_size * sizeof(unsigned int), h->stream));  // (k_kld_count keeps them clean)
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->kld_minslot, 0xff, h->kld_hash_size * sizeof(unsigned int), h->stream));
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->kld_myslot, h->cap * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->kld_out, 2 * sizeof(uint64_t)));
    RR_TRY_OR_CLEAN(hipHostMalloc(&h->kld_out_host, 2 * sizeof(uint64_t)));
  }
  const bool guide_at_create = opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL && !kld && h->n == h->n_global && cap_tiles <= (uint64_t)rr::kFusedMaxTiles &&
                               h->n > kSmallMaxParticles;
  if (opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL && !kld) {  // the fused step resamples lazily through lidx
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->lidx, h->n * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(rr::memset_on(h->stream, h->lidx, 0xff, h->n * sizeof(unsigned int)));
    if (h->n == h->n_global && !std::getenv("RR_MN_NO_PACKED")) {
      for (int k = 0; k < 2; ++k) RR_TRY_OR_CLEAN(rr::dev_malloc(&h->packed[k], 4 * h->n * sizeof(double)));
    }
  }
  {
    // (+ one tile of padding in front of a shard's own block, peer-to-peer transport: rr::resolve_tile_window)
    const size_t nm = (size_t)std::max<uint64_t>(n_global, h->cap) + 3 * (size_t)rr::kResolveSlots;
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->markers, nm * sizeof(unsigned int)));
    // ON THE FILTER'S STREAM (rr::memset_on): a hipMemset goes to the null stream and returns before it has run, and this stream
    // is a non-blocking one -- the first plan then marked into whatever an earlier tenant had left (round 6's "the unsharded
    // reference filter differs between ranks"; with poisoned allocations: wild source indices, a memory fault)
    RR_TRY_OR_CLEAN(rr::memset_on(h->stream, h->markers, 0, nm * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->carry, (nm / rr::kResolveSlots + 2) * sizeof(unsigned int)));
  }
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->partials, (size_t)kMomentBlocks * kNumMoments * sizeof(double)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->est_partials, (size_t)rr::kFusedMaxTiles * 4 * sizeof(double)));
  {
    int per_cu = 0, dev_cus = 0;
    RR_TRY_OR_CLEAN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rr::k_quantize_plan_mark<false>, rr::kTileBlock, 0));
    RR_TRY_OR_CLEAN(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    h->grid_capacity = std::min<uint64_t>((uint64_t)per_cu * (uint64_t)dev_cus, (uint64_t)rr::kTileBlock);
    if (const char* e = std::getenv("RR_PF_FUSED_PLAN")) {
      if (std::atoi(e) == 0) h->grid_capacity = 0;
    }
    const size_t rec_bytes = rr::kPlanRecBytes;  // records, heads, (instrumented build) stamps, the self-vouching pairs
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->grid_rec, rec_bytes));
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->grid_rec, 0, rec_bytes, h->stream));
    RR_TRY_OR_CLEAN(rr::dev_malloc(&h->grid_ticket, rr::kTicketWords * sizeof(unsigned int)));
    RR_TRY_OR_CLEAN(hipMemsetAsync(h->grid_ticket, 0, rr::kTicketWords * sizeof(unsigned int), h->stream));
  }
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->est_ticket, rr::kTicketWords * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->est_ticket, 0, rr::kTicketWords * sizeof(unsigned int), h->stream));
  h->slot_pad = (rr::kResolveSlots - h->opt.first_global_index % rr::kResolveSlots) % rr::kResolveSlots;
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->ctl, sizeof(Ctl)));
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
    if (!h->scratch_a) RR_HIP_TRY(rr::dev_malloc(&h->scratch_a, want * sizeof(double)));
  }
  if (doubles_b && !h->scratch_b) RR_HIP_TRY(rr::dev_malloc(&h->scratch_b, h->cap * sizeof(double)));
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
  const bool est = wa.est_partials != nullptr;  // the builds that add up the deferred estimate
#define RR_K1_GO(KA_, SRC_, LIK_)                                                                                             \
  (est ? launch_k1_as<KA_, SRC_, LIK_, false, true>(h, grid, lds, ea, eb, p, arg, markers, carry, idx_out, wa) \
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
  auto t0 = std::chrono::steady_clock::now();
  const long long patience_ms = 2000 + (long long)(r.life_us / 1000.0);
  int gave_up = 0;
  for (unsigned spins = 0;; ++spins) {
    uint64_t e[4], flags = 0;
    // (the adaptive kernel vouches for its estimate with a flags pair: part of the answer, waited for under the same bound)
    if ((!h->adaptive || rr::ring_take(&r.ring->rsp[rr::kResRspFlags], seq, &flags)) && rr::ring_take(&r.ring->rsp[3], seq, &e[3]) &&
        rr::ring_take(&r.ring->rsp[2], seq, &e[2]) && rr::ring_take(&r.ring->rsp[1], seq, &e[1]) && rr::ring_take(&r.ring->rsp[0], seq, &e[0])) {
      r.pending = false;
      if (h->adaptive) {  // flags != 0 => form the estimate the long way
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
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > patience_ms) {
      // No answer within the bound.  The (re)launched kernel may simply not have had its turn yet -- queued behind another
      // handle's resident kernel on a shared hardware queue, which may live for its whole life_us -- so the bound covers the
      // longest life a resident kernel can have (rr_pf_set_resident caps it).  Past that: wait for the stream, after which the
      // incarnation has either taken the command (the answer is in the ring: the loop's next turn returns it, the step WAS
      // applied and is reported as such) or left without it (EXIT marker, consumed < seq: relaunched by the branch above, the
      // command is still in the ring).  Only a kernel that neither answers nor leaves twice in a row is an error -- never a
      // failure reported for a step the device went on to apply.
      (void)hipStreamSynchronize(h->stream);
      if (++gave_up > 2) {
        r.live = false;
        r.pending = false;
        return fail(RR_RUNTIME_ERROR, "the resident step kernel did not answer");
      }
      t0 = std::chrono::steady_clock::now();
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
      RR_HIP_TRY(rr::dev_malloc(&h->steps_dev, (K * per + K * per / 2) * sizeof(double)));
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
#ifndef RR_SOURCE_SHA16
#define RR_SOURCE_SHA16 "unknown"  // (csrc/Makefile passes the hash of the sources; a build by hand does not know it)
#endif
const char* rr_version(void) { return "rust_robotics_amd 0.1.0 (gfx950; sources " RR_SOURCE_SHA16 ")"; }
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
  (void)hipFree(h->est_total_dev);
  (void)hipFree(h->steps_dev);
  (void)hipFree(h->est_ring);
  if (h->mail) (void)hipHostFree(h->mail);
  (void)hipFree(h->mn_tile_cnt);
  (void)hipFree(h->mn_records);
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
  rr::spin_release(h->spin_gate_no, h);
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
#if defined(RR_DEBUG_TRACE)
    wa_est.trace = h->dbg_trace ? h->dbg_trace + (size_t)(h->step % h->dbg_cap) * kDbgWords : nullptr;
#endif
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
#if defined(RR_DEBUG_TRACE)
  if (h->dbg_trace) {  // the unsharded filter of the hunt: Ctl as the plan leaves it (first 112 bytes: the ints, the integer sums, wmax .. rho)
    const rr_status rs = launch_resample(h, 0, h->opt.resample_scheme, NAN, nullptr, /*lazy=*/true, /*settle=*/1, want_estimate);
    RR_HIP_TRY(hipMemcpyAsync(h->dbg_trace + (size_t)((h->step - 1) % h->dbg_cap) * kDbgWords + 14, h->ctl, 112, hipMemcpyDeviceToDevice, h->stream));
    return rs;
  }
#endif
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
  // (<= 0.5 s idle, hence <= 10 s of life: what a host that waits for an answer has to be prepared to sit out, resident_await)
  // values above 0.5 s are CLAMPED, not rejected (the bound used to be 1e7): 0.5 s of idling means <= 10 s of life, and a host
  // that waits for an answer from a kernel that has died sits out at most 3 x (2 s + life) = 36 s before it is told (resident_await)
  if (!(idle_us >= 0.0)) return fail(RR_INVALID_PARAMETER, "resident idle time must be >= 0 microseconds (values above 5e5 are clamped to 5e5)");
  if (idle_us > 5e5) idle_us = 5e5;
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
  // The DEFERRED form (rr::EstArgs: the resampled set's mean summed by the kernel that moves the particles, i.e. the next step's
  // k_step_lazy<., EST>, or by the accessor's gather + k_est_slots when the value is read before another step has come) for both
  // schemes.  Multinomial: the only form there is (the offspring counts of iid draws do not exist before the draws are searched).
  // Systematic: since round 5, when the EST build of the step kernel was given the register budget of the plain one (64 VGPRs,
  // pf_kernels_step.inc: step_blocks_per_cu) -- 47.4 against the in-plan form's 48.5 us per step at 1e6 x 32, one box, alternating
  // (profiles/r05l_launch_bounds_ab.md; round 4, with the EST build at 93 VGPRs, the deferred form lost by 0.5 us).  What it costs: a
  // caller that reads the value right after the step (no other step in between) pays the accessor's gather, ~25 us at 1e6 particles,
  // where the in-plan form pays a 16 KB copy -- the synchronous rr_pf_step, whose caller always does, keeps the in-plan form;
  // RR_PF_EST_DEFER=0 selects it for rr_pf_step_async_estimate too.  Same value either way to 1e-11 (another order of summation).
  const char* e = std::getenv("RR_PF_EST_DEFER");  // (read per call: the tests switch it within one process)
  const int sys_mode = e && std::atoi(e) == 0 ? (int)rr::kEstInPlan : (int)rr::kEstDeferred;
  const bool small = h && small_path(h, 0);
  const int mode = (h && !small && h->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC) ? (int)rr::kEstDeferred : sys_mode;
  return step_async_impl(h, control, obs, n_obs, mode);
}

// The slot tiles' sums of the deferred estimate, added up ON THE DEVICE by k_est_slots_total (the three-level order of k_est_mail_any:
// the same bits as the synchronous step's mailbox kernel): the host fetches four doubles, and the stamp behind the sums if asked.
static rr_status est_slots_total(rr_pf* h, uint64_t n_part, double acc[4], uint64_t* stamp) {
  if (!h->est_total_dev) RR_HIP_TRY(rr::dev_malloc(&h->est_total_dev, 4 * sizeof(double)));
  hipLaunchKernelGGL(k_est_slots_total, dim3(1), dim3(kEstChunks), 0, h->stream, (const double*)h->est_slot_partials, n_part, h->est_total_dev);
  hipError_t copies = hipGetLastError();
  if (copies == hipSuccess) copies = hipMemcpyAsync(acc, h->est_total_dev, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (copies == hipSuccess && stamp) copies = hipMemcpyAsync(stamp, est_stamp_slot(h), sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream);
  if (copies != hipSuccess) (void)hipStreamSynchronize(h->stream);  // (acc / stamp are the caller's locals: nothing may still be on its way there)
  RR_HIP_TRY(copies);
  return RR_OK;  // (the caller's fetch_ctl waits for the stream)
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
  double slot_total[4] = {0.0, 0.0, 0.0, 0.0};
  if (h->est_slot_partials && (s = est_slots_total(h, n_slot_tiles * (kBlock / rr::kWave), slot_total, nullptr)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;  // (synchronises the stream)
  if (h->ctl_host->est_step == 0) return fail(RR_INVALID_PARAMETER, "no step has produced an in-step estimate yet");
  const bool slots = h->ctl_host->est_kind == rr::kEstSlotTiles;
  if (slots && !h->est_slot_partials) return fail(RR_RUNTIME_ERROR, "the deferred estimate's sums are missing");
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (slots) {
    for (int k = 0; k < 4; ++k) acc[k] = slot_total[k];
  } else {  // the plan tiles' partial sums in est_plan_total's order: interleaved chunks of tiles, then the chunks in order
    const double* part = h->est_partials_host;
    const uint64_t n_part = h->n_tiles;
    double cs[kEstPlanChunks][4] = {};
    for (int c = 0; c < kEstPlanChunks; ++c)
      for (uint64_t t = (uint64_t)c; t < n_part; t += kEstPlanChunks)
        for (int k = 0; k < 4; ++k) cs[c][k] += part[4 * t + k];
    for (int c = 0; c < kEstPlanChunks; ++c)
      for (int k = 0; k < 4; ++k) acc[k] += cs[c][k];
  }
  for (int k = 0; k < 4; ++k) out[k] = acc[k] / h->ctl_host->est_denom;
  return RR_OK;
}

rr_status rr_pf_synchronize(rr_pf* h) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  return fetch_ctl(h);  // waits for the stream; also the place where a handle learns that its one-launch plan had to degrade
}

// rr_pf_warm (include/rr_pf.h): `ms` milliseconds of step-shaped FP64 work on the filter's stream, so that the caller's first
// step runs at the rate of its thousandth (rr::device_warm).  ms == 0: the default, 50 ms.
rr_status rr_pf_warm(rr_pf* h, double ms) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!(ms >= 0.0) || !(ms <= 2000.0)) return fail(RR_INVALID_PARAMETER, "warm-up time must lie in [0, 2000] milliseconds (0: the default, 50)");
  RR_HIP_TRY(rr::device_warm(h->stream, h->opt.device, ms == 0.0 ? 50.0 : ms));
  return RR_OK;
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
        RR_HIP_TRY(rr::dev_malloc(&h->est_ring, 4 * n_steps * sizeof(double)));
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
    // 71.9 - 73.7 us per synchronous step against 69.7 - 70.1 with the separate launch, whose dispatch overlaps the plan kernel.
    // So was the opposite: every workgroup of the plan kernel storing its four sums straight into pinned host memory as
    // self-vouching {bits, seq} pairs, the host polling and adding them -- 2 000 sixteen-byte writes over the link and 489 freshly
    // invalidated lines on the host: 70.5 - 70.9 us against 67.6 - 68.6 with k_est_mail.)
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
    RR_HIP_TRY(rr::dev_malloc(&h->mn_tile_cnt, n_tiles * sizeof(unsigned int)));
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
  // (both schemes: the systematic shards take the window step, the multinomial ones shard_step_p2p_multinomial)
  return RR_OK;
}

static rr_status p2p_alloc_lidx(rr_pf* h) {
  if (h->lidx) return RR_OK;
  RR_HIP_TRY(rr::dev_malloc(&h->lidx, h->n * sizeof(unsigned int)));
  RR_HIP_TRY(rr::memset_on(h->stream, h->lidx, 0xff, h->n * sizeof(unsigned int)));  // kInPlace everywhere
  return RR_OK;
}

// RR_P2P_CU_PARTITION=1 (a test rig, off by default): ranks that SHARE a device each get a stream confined to their own 1/n_sharing
// of its CUs (hipExtStreamCreateWithCUMask; contiguous mask bits, which the driver deals round-robin over the XCDs: every rank
// holds CUs in every XCD).  The eight XCDs' worth of CUs then behave like eight small devices as far as workgroup slots go: a
// consuming kernel that waits inside the kernel for a peer's delivery can no longer sit on the slots the delivering kernel
// needs, so sharers of ANY size take the lazy window step -- the deployment path -- and every one of them may run its own
// one-launch plan (its own spin gate).  This is how BASELINE configs[4] (8 x 2e6 particles) executes as 8 ranks on one GPU
// through exactly the kernels an 8-GPU node would run (tools/world8_one_device.py, tests/test_gpu_world8.py).
static rr_status p2p_apply_cu_partition(rr_pf* h) {
  const char* e = std::getenv("RR_P2P_CU_PARTITION");
  if (!e || std::atoi(e) == 0 || h->p2p.n_sharing <= 1 || h->using_external_stream || !h->owns_stream || h->cu_part_cus) return RR_OK;
  int cus = 0;
  RR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
  const int parts = h->p2p.n_sharing, part = h->p2p.share_ordinal, per = cus / parts;
  if (per < 8 || part >= parts) return RR_OK;  // too many sharers for a useful share: keep the whole-device rules
  std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
  for (int b = part * per; b < (part + 1) * per; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
  hipStream_t s = nullptr;
  RR_HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  RR_HIP_TRY(hipStreamSynchronize(h->own_stream));
  rr::spin_release(h->spin_gate_no, h);
  (void)hipStreamDestroy(h->own_stream);
  h->own_stream = h->stream = s;
  h->cu_part_cus = per;
  h->spin_gate_no = (h->opt.device & 63) + 64 * (1 + (part & 15));
  h->shard_capacity = ~0ull;  // asked again, for the share
  return RR_OK;
}

rr_status rr_pf_p2p_export(rr_pf* h, uint8_t out[RR_P2P_HANDLE_BYTES]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  return h->p2p.export_handles(h->slab, 5 * h->n, out, 8 * h->cap * sizeof(double));  // inbox: 4 fields + the seal plane
}

rr_status rr_pf_p2p_connect(rr_pf* h, const uint8_t* all_handles, int32_t n_ranks, int32_t rank) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!all_handles) return fail(RR_INVALID_PARAMETER, "null handles");
  if ((s = p2p_check_geometry(h, n_ranks, rank)) != RR_OK) return s;
  if ((s = p2p_alloc_lidx(h)) != RR_OK) return s;
  if ((s = h->p2p.connect_ipc(h->slab, 5 * h->n, all_handles, n_ranks, rank)) != RR_OK) return s;
  return p2p_apply_cu_partition(h);
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
  rr_status s = rr::p2p_link_local(st, slabs, inboxes, devs, n_ranks);
  for (int g = 0; g < n_ranks && s == RR_OK; ++g) s = p2p_apply_cu_partition(handles[g]);
  return s;
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

// MULTINOMIAL shards over the peer-to-peer transport (round 6): the resampler the reference's ParticleFilterLocalizer and
// MonteCarloLocalizer really use (particle_filter.rs:441-473, monte_carlo_localization.rs:322-365, :387-392), sharded without a
// collective library and without the host in the step.  iid draws scatter the slots a shard serves over all ranks, so there is
// no window and nothing lazy: propagate + weight (the propagated particles also as 32-byte records) | WMAX exchange | integer image |
// tile scan + SUMS exchange (gate, base, totals) | local slice of the global CDF + the markers of the shard's guide table |
// k_guide_resolve(local) | k_mn_push_p2p_guided: every draw that falls into this shard's CDF interval is searched here (guide pair,
// one or two CDF probes) and its source record stored straight into the owning rank's slab | DONE exchange.  Every shard evaluates all N draws (a Philox block and a
// comparison each), searches N / G of them on average and moves as many particles -- nearly all of them across ranks: that is
// what iid draws cost on any transport.  World size 1, 1e6 x 32: 0.097 ms per step (first form, binary search over the whole CDF and
// four planes per gathered source: 0.21); the RCCL form of the same step (rr_pf_shard_step: count, scan, pack, all-to-all, adopt,
// two host round trips): 0.30.
static rr_status shard_step_p2p_multinomial(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  rr_status s;
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  if ((s = materialise(h)) != RR_OK) return s;
  const uint64_t seq = ++h->p2p.seq;
  ObsArg arg;
  bool kernarg;
  if ((s = stage_obs(h, obs, n_obs, &arg, &kernarg)) != RR_OK) return s;
  StepParams p = make_params(h, control, (int)n_obs);
  // (the propagated particles once more as 32-byte records: what k_mn_push_p2p gathers from -- one cache line per draw, not four)
  if (!h->mn_records) RR_HIP_TRY(rr::dev_malloc(&h->mn_records, 4 * h->cap * sizeof(double)));
  if ((s = ensure_guide(h)) != RR_OK) return s;  // (first step only: allocates and waits once; RR_MN_GUIDE=0: the LDS coarse table instead)
  const bool guided = h->guide_log2 > 0;
  if ((s = launch_pw<true, true, false>(h, p, arg, kernarg, h->mn_records)) != RR_OK) return s;
  h->step += 1;
  PlanArgs pa = plan_args(h, 0, RR_RESAMPLE_MULTINOMIAL, NAN);
  // exchange 1: global maximum -> Ctl.wmax
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PWmax, seq,
                     (const uint64_t*)&h->ctl->wmax_bits, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  // integer image under the global maximum; tile scan + exchange 2: every rank's sums -> gate, base, totals in Ctl
  launch_quantize(h, (const double*)&h->ctl->wmax);
  {
    Timed t(h, RR_K_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_exchange, dim3(1), dim3(kScanThreads), 0, h->stream, h->p2p.peers, seq, h->tile_total,
                       (const uint64_t*)h->tile_q2, h->n_tiles, h->ctl, pa, h->p2p.err);
  }
  {
    Timed t(h, RR_K_CDF);
    hipLaunchKernelGGL(rr::k_cdf, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->w, h->ctl, image_args(h),
                       h->tile_total, h->cdf, h->cdf_coarse, h->coarse_log2,  // (+ the coarse table the unguided push kernel stages in LDS)
                       guided ? h->guide_markers : (unsigned int*)nullptr, h->guide_carry, h->guide_log2);
  }
  if (guided) launch_guide_resolve(h, /*local=*/1);
  h->wmax_live = false;
  h->wmax_bits_clean = false;  // (a peer wait that gave up skips finalize_plan: do not rely on the zeroed accumulator)
  MnSelectArgs a{};
  a.n_local = h->n;
  a.n_global = h->n_global;
  a.tiles_per_dest = (h->n + kTile - 1) / kTile;
  a.seed = h->opt.seed;
  a.rstep = h->rstep;  // the resample being planned: the draws of THIS resample step
  a.n_shards = h->p2p.peers.n_ranks;
  h->rstep += 1;
  {
    Timed t(h, RR_K_RESAMPLE_GATHER);
    const size_t lds = h->n_coarse * sizeof(uint64_t);
    if (guided) {
      hipLaunchKernelGGL(k_mn_push_p2p_guided, dim3((unsigned)std::min<uint64_t>(grid_for(h->n_global, 256 * kMnPushRows), 256ull * 8)), dim3(256), 0, h->stream,
                         (const double*)h->mn_records, h->ctl, a, (const uint64_t*)h->cdf, h->n, (const unsigned int*)h->guide, h->guide_log2, h->p2p.peers);
    } else {  // RR_MN_GUIDE=0: the coarse table of the CDF staged in LDS, one 1024-thread workgroup per CU
      if (!h->mn_push_lds_set) {  // (per handle: the attribute belongs to the handle's device)
        RR_HIP_TRY(hipFuncSetAttribute((const void*)k_mn_push_p2p, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        h->mn_push_lds_set = true;
      }
      hipLaunchKernelGGL(k_mn_push_p2p, dim3((unsigned)std::min<uint64_t>(grid_for(h->n_global, 1024), (uint64_t)h->mn_grid)), dim3(1024), lds, h->stream,
                         (const double*)h->mn_records, h->ctl, a, (const uint64_t*)h->cdf, h->n, (const uint64_t*)h->cdf_coarse, h->coarse_log2,
                         h->n_coarse, h->p2p.peers);
    }
  }
  // exchange 3: every rank has finished writing into everybody's slab
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PDone, seq,
                     (const uint64_t*)h->p2p.local3(), h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  RR_HIP_TRY(hipGetLastError());
  h->p2p_last_form = 2;
  return RR_OK;
}

rr_status rr_pf_shard_step_p2p(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->p2p.ready) return fail(RR_INVALID_PARAMETER, "call rr_pf_p2p_connect first");
  if (h->opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL) return shard_step_p2p_multinomial(h, control, obs, n_obs);
  if ((s = validate_control(control)) != RR_OK) return s;
  if ((s = validate_obs(obs, n_obs)) != RR_OK) return s;
  // Ranks that SHARE a device (a test rig; one rank per GPU is the deployment): this step's k_step_lazy waits, inside the
  // kernel, for the particles a peer's k_push_window delivers.  On separate devices that is a wait for another GPU.  On one
  // device the waiting kernels of the sharers can hold every workgroup slot while the push kernel they wait for has not been
  // dispatched yet -- it then never gets a slot, and the waits run into their bound: the round-3 "give-up at 10^6 particles per
  // rank" (run down in round 4 with tools/p2p_shared_device_jump.py: after a resample that moves most of a shard, every slot
  // was found delivered and consistent in memory -- AFTER the waiters had given up and freed the device).  Sharers whose step
  // kernels together can fill the device therefore take the eager form of the step, which has no wait inside a full-size kernel.
  if (h->p2p.n_sharing > 1 && !h->cu_part_cus) {  // (sharers with a part of the CUs each -- RR_P2P_CU_PARTITION -- cannot do that to each other)
    if (!h->dev_cus) RR_HIP_TRY(hipDeviceGetAttribute(&h->dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    const uint64_t step_wgs = (h->n + rr::kResolveSlots - 1) / rr::kResolveSlots;
    if (step_wgs * (uint64_t)h->p2p.n_sharing > 3ull * (uint64_t)h->dev_cus) {
      h->p2p_last_form = 2;
      return rr_pf_shard_step_p2p_unfused(h, control, obs, n_obs);
    }
  }
  h->p2p_last_form = 1;
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
  // rr_pf_shard_want_estimate: this launch adds up the fields of the sources of its own slots (the step before's resample)
  wa.est_partials = h->est_deferred ? h->est_slot_partials : nullptr;
  wa.est_stamp_slot = h->est_deferred ? est_stamp_slot(h) : nullptr;
  wa.est_stamp = h->shard_est_stamp;
  h->est_deferred = false;
#if defined(RR_DEBUG_TRACE)
  uint64_t* const dbg = h->dbg_trace ? h->dbg_trace + (size_t)(h->step % h->dbg_cap) * kDbgWords : nullptr;
  wa.trace = dbg;
#endif
  const uint64_t seq = ++h->p2p.seq;
  if (h->shard_capacity == ~0ull) {  // every workgroup of k_shard_plan_mark resident at once?
    int per_cu = 0, dev_cus = 0;
    RR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rr::k_shard_plan_mark, rr::kTileBlock, 0));
    RR_HIP_TRY(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
    h->dev_cus = dev_cus;
    if (h->cu_part_cus) dev_cus = h->cu_part_cus;  // the stream's own part of the device
    h->shard_capacity = h->grid_capacity ? std::min<uint64_t>((uint64_t)per_cu * (uint64_t)dev_cus, (uint64_t)rr::kTileBlock) : 0;
  }
  // Ranks that share this device (a test rig: several shards of one filter on one GPU) run their kernels beside this one's;
  // a plan kernel that spins on every CU would leave a peer's exchange workgroup -- the one it is waiting for -- nowhere to
  // go (seen as stalls of seconds with two 1e6-particle shards on one device).  All sharers together keep to one
  // workgroup per CU.
  const uint64_t fused_cap = h->p2p.n_sharing > 1 && !h->cu_part_cus
                                 ? std::min<uint64_t>(h->shard_capacity, (uint64_t)h->dev_cus / (uint64_t)h->p2p.n_sharing)
                                 : h->shard_capacity;
  const bool fused_plan = h->n_tiles <= fused_cap && rr::spin_permit(h->spin_gate_no, h);
  // RR_P2P_WMAX_EARLY=1: the last workgroup of the step kernel sends this shard's weight maximum to the ranks' mailboxes as it
  // finishes (WindowArgs.post_peers) and the one-launch plan's workgroups take the records from their own mailbox, instead of the
  // plan kernel's first workgroup opening with an exchange and a flag.  Measured at world size 1 (round 5,
  // profiles/r05o_wmax_early.md): the plan kernel gets 1.65 us shorter and the step kernel as much longer -- "everybody has
  // finished" costs the step kernel's tail the hand-overs the plan kernel saves -- 52.85 against 52.86 us per step.  Off by default;
  // kept, with its tests, for the first run on a real fabric, where the record's flight would overlap the kernel boundary.
  const char* const early_env = std::getenv("RR_P2P_WMAX_EARLY");  // (read per call: the tests switch it within one process)
  const bool post_wmax = fused_plan && early_env && std::atoi(early_env) != 0;
  if (post_wmax) {
    wa.post_peers = h->p2p.peers_dev;
    wa.post_ticket = h->p2p.post_ticket;
    wa.post_seq = seq;
  }
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
  if (fused_plan) {
    // exchange 1 + B + exchange 2 + C in one launch (k_shard_plan_mark)
    Timed t(h, RR_K_CDF);
    hipLaunchKernelGGL(rr::k_shard_plan_mark, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->p2p.peers, seq,
                       (const double*)h->w, h->ctl, image_args(h), h->grid_rec, h->grid_ticket, ++h->grid_epoch, /*settle=*/1,
                       h->n_tiles, pa, h->markers, h->carry, h->p2p.err, h->slot_pad, post_wmax ? 1 : 0
#if defined(RR_DEBUG_TRACE)
                       , dbg
#endif
                       );
  } else {
    // exchange 1: global maximum -> Ctl.wmax
    hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PWmax, seq,
                       (const uint64_t*)&h->ctl->wmax_bits, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
    // B: integer image under the global maximum (settles the resample K1 consumed)
    launch_quantize(h, (const double*)&h->ctl->wmax, /*settle=*/1);
    // tile scan + exchange 2: every rank's sums -> gate, base, plan in Ctl
    {
      Timed t(h, RR_K_SCAN_TILES);
      hipLaunchKernelGGL(rr::k_scan_exchange, dim3(1), dim3(kScanThreads), 0, h->stream, h->p2p.peers, seq, h->tile_total,
                         (const uint64_t*)h->tile_q2, h->n_tiles, h->ctl, pa, h->p2p.err);
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
  h->est_deferred = h->shard_est;  // (whoever moves the particles next adds up this shard's part of the step's mean)
  if (h->shard_est) h->shard_est_stamp = h->rstep;
  return RR_OK;
}

// The mean try_step returns, for a filter sharded over the peer-to-peer transport: with want != 0 every rr_pf_shard_step_p2p leaves
// THIS shard's part -- the sums of the four fields over the sources of its own output slots (deferred form of the in-step
// estimate: added up by the next step's kernel as it moves the particles, rr::EstArgs) -- and rr_pf_shard_last_estimate_sums
// returns them with the denominator N; the mean is the sum of the shards' sums over N (one all-reduce of four doubles, whenever
// the caller wants the value).  Defined for steps whose resample fired (MonteCarloLocalizer: every step).
rr_status rr_pf_shard_want_estimate(rr_pf* h, int32_t want) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (want && h->opt.resample_scheme != RR_RESAMPLE_SYSTEMATIC)
    return fail(RR_INVALID_PARAMETER, "the sharded estimate serves the systematic scheme (the window steps: rr_pf_shard_step_p2p, rr_pf_shard_step)");
  if (want && !h->est_slot_partials) {
    RR_HIP_TRY(rr::dev_malloc(&h->est_slot_partials, est_slot_bytes(h)));
    RR_HIP_TRY(hipMemsetAsync(est_stamp_slot(h), 0, sizeof(uint64_t), h->stream));  // (no step's sums yet: stamps start at 1)
  }
  h->shard_est = want != 0;  // (want == 0: later steps leave no sums; the last step's stay readable, also after one more step)
  if (want) h->shard_est_ever = true;
  return RR_OK;
}

rr_status rr_pf_shard_last_estimate_sums(rr_pf* h, double out_sums[4], double* out_denom) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out_sums || !out_denom) return fail(RR_INVALID_PARAMETER, "null output");
  if (!h->est_slot_partials || !h->shard_est_ever) return fail(RR_INVALID_PARAMETER, "call rr_pf_shard_want_estimate first");
  if (h->est_deferred && (s = materialise(h)) != RR_OK) return s;  // nobody has moved the particles yet: gather + k_est_slots
  if (h->est_deferred) launch_est_slots(h);
  const uint64_t n_part = (uint64_t)grid_for(h->n, rr::kResolveSlots) * (kBlock / rr::kWave);
  uint64_t stamp = 0;
  if ((s = est_slots_total(h, n_part, out_sums, &stamp)) != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  // The sums carry the resample step they were formed for (written by the kernel that formed them, only when that step's gate
  // fired).  Asked for was step shard_est_stamp -- the last step taken with rr_pf_shard_want_estimate on, however many plain
  // steps followed (the live Ctl.fired belongs to the LATEST step, not to that one).
  if (stamp != h->shard_est_stamp)
    return fail(RR_INVALID_PARAMETER, "the gate of the step whose estimate was asked for stayed shut: no resampled set to take the mean of");
  *out_denom = (double)h->n_global;
  return RR_OK;
}

#if defined(RR_DEBUG_TRACE)
// instrumented build only (make -C csrc trace): what the step kernels saw, kDbgWords words per step in a ring of cap_steps
rr_status rr_pf_debug_trace(rr_pf* h, uint32_t cap_steps) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  RR_HIP_TRY(rr::dev_malloc(&h->dbg_trace, (size_t)cap_steps * kDbgWords * sizeof(uint64_t)));
  RR_HIP_TRY(rr::memset_on(h->stream, h->dbg_trace, 0, (size_t)cap_steps * kDbgWords * sizeof(uint64_t)));
  h->dbg_cap = cap_steps;
  return RR_OK;
}
rr_status rr_pf_debug_trace_read(rr_pf* h, uint64_t* out, uint64_t cap_words) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->dbg_trace) return fail(RR_INVALID_PARAMETER, "no trace");
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  RR_HIP_TRY(hipMemcpy(out, h->dbg_trace, (size_t)std::min<uint64_t>(cap_words, (uint64_t)h->dbg_cap * kDbgWords) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RR_OK;
}
#endif

// Diagnostics (not part of include/rr_pf.h; tools/soak_shard_estimate.py): the per-wave sums of the deferred estimate as they
// stand in device memory once the stream has drained -- a second, independent read of what rr_pf_last_step_estimate /
// rr_pf_shard_last_estimate_sums add up.
rr_status rr_pf_debug_est_slot_partials(rr_pf* h, double* out, uint64_t cap_doubles, uint64_t* n_doubles) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out || !n_doubles) return fail(RR_INVALID_PARAMETER, "null output");
  if (!h->est_slot_partials) return fail(RR_INVALID_PARAMETER, "no deferred estimate has been asked for");
  const uint64_t n = (uint64_t)grid_for(h->n, rr::kResolveSlots) * kEstSlotWords;
  if (n > cap_doubles) return fail(RR_INVALID_PARAMETER, "output too small");
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  RR_HIP_TRY(hipMemcpy(out, h->est_slot_partials, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  *n_doubles = n;
  return RR_OK;
}

// The same step with every phase as its own launch (three exchange kernels, eager gather): kept as
// the plain statement of the protocol and for A/B measurement.
rr_status rr_pf_shard_step_p2p_unfused(rr_pf* h, const double control[2], const double* obs, size_t n_obs) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!h->p2p.ready) return fail(RR_INVALID_PARAMETER, "call rr_pf_p2p_connect first");
  if (h->opt.resample_scheme == RR_RESAMPLE_MULTINOMIAL) return shard_step_p2p_multinomial(h, control, obs, n_obs);  // (its only form)
  const uint64_t seq = ++h->p2p.seq;
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
                     (const uint64_t*)&h->ctl->wmax_bits, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  // B: integer image under the global maximum, local sums -> local3
  launch_quantize(h, (const double*)&h->ctl->wmax);
  {
    Timed t(h, RR_K_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_tiles, dim3(1), dim3(kScanThreads), 0, h->stream, h->tile_total, h->tile_q2, h->ctl,
                       h->n_tiles, 0, pa, local3);
  }
  // exchange 2: every rank's sums -> plan (gate, base, systematic plan) in Ctl
  hipLaunchKernelGGL(rr::k_p2p_exchange, dim3(1), dim3(64), 0, h->stream, h->p2p.peers, (int)rr::kP2PSums, seq,
                     (const uint64_t*)local3, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
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
                     (const uint64_t*)local3, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  if (h->shard_est) {  // rr_pf_shard_want_estimate: the resampled set is in place (eager gather): this shard's part of its mean
    h->est_deferred = true;
    h->shard_est_stamp = h->rstep;
    launch_est_slots(h);
  }
  RR_HIP_TRY(hipGetLastError());
  return RR_OK;
}

rr_status rr_pf_p2p_topology(rr_pf* h, int32_t out[4]) {
  if (!h || !out) return fail(RR_INVALID_PARAMETER, "null argument");
  out[0] = h->p2p.ready ? h->p2p.peers.n_ranks : 0;
  out[1] = h->p2p.n_sharing;
  out[2] = h->cu_part_cus;
  out[3] = h->p2p_last_form;
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

#include "pf_sharded_api.inc"
