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
// One translation unit in two files: the kernels are in fs1_kernels.inc (#included below, in place); this file holds the handle,
// the launch logic and the C ABI.
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
#include "fs1_kernels.inc"
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
  bool cu_partitioned = false;  // RR_P2P_CU_PARTITION: the stream was re-created with this shard's share of the device's CUs
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
  uint64_t obs_giveups = 0;   // host looks at Ctl that found k_fs1_observe's closing chunk had given up waiting (k_fs1_combine repaired it)
  bool obs_two_launch = false;  // ... after which the handle forms the weights in k_fs1_combine for good (no wait inside the observe kernel)
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

int choose_chunks(const rr_fs1* h, size_t n_z, bool dup);

// z rows are (d, angle, id); ids must be integral and inside the map; duplicates force one chunk.  A chunk's observations are staged
// in LDS (24 bytes each, 150 of the CU's 160 KB): a list whose chunk plan does not fit is refused HERE, before anything of the update
// has run (until round 6 the observe launch itself failed -- after the predict -- and left HIP's sticky error for the next call)
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
  if (n_z > 1) {
    const int chunks = choose_chunks(h, n_z, *has_duplicates);
    const size_t len = (n_z + (size_t)chunks - 1) / (size_t)chunks;
    if (3 * len * sizeof(double) > 150 * 1024)
      return fail(RR_INVALID_PARAMETER, *has_duplicates
                                            ? "too many fastslam observations for an update with a repeated landmark id (one sequential chunk, max 6400)"
                                            : "too many fastslam observations per chunk for one LDS block (max 6400 per chunk, 64 chunks)");
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
  RR_HIP_TRY(rr::dev_malloc(&h->z_dev, 3 * cap * sizeof(double)));
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

// how long k_fs1_observe's closing chunk waits for a factor before it leaves the weight to k_fs1_combine, in ticks of the
// 100 MHz wall clock: RR_FS1_FACTOR_WAIT_US, default 2 s (0: never wait -- the test hook that exercises the degrade path)
static uint64_t factor_wait_ticks() {
  static const uint64_t ticks = [] {
    const char* e = std::getenv("RR_FS1_FACTOR_WAIT_US");
    return e ? (uint64_t)(std::max(0.0, std::atof(e)) * 100.0) : kFactorWaitTicksDefault;
  }();
  return ticks;
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
      RR_HIP_TRY(rr::dev_malloc(&h->partial, (size_t)chunks * h->n * sizeof(double)));
      h->partial_chunks = chunks;
      hipLaunchKernelGGL(k_fs1_no_factors, dim3(1024), dim3(kBlock), 0, h->stream, reinterpret_cast<uint64_t*>(h->partial),
                         (uint64_t)chunks * h->n);  // every slot reads "no factor yet" between updates
    }
    double* a_partial = h->partial;
    const unsigned int* a_idx = h->idx;
    unsigned int a_pblocks = grid.x;
    const dim3 grid_l((unsigned int)((uint64_t)a_pblocks * (uint64_t)chunks));
    uint64_t a_wait = h->obs_two_launch ? 0ull : factor_wait_ticks();
    int a_latch = h->obs_two_launch ? 0 : 1;
    void* args[] = {&a_pl, &a_pw, &a_ctl, &a_n, &a_z, &a_nz, &a_len, &a_chunks, &a_m, &a_partial, &a_idx, &a_pblocks, &a_wait, &a_latch};
    if (ea && !dup) RR_HIP_TRY(hipExtLaunchKernel(kfn, grid_l, dim3(kBlock), args, lds, h->stream, ea, eb, 0));
    else RR_HIP_TRY(hipLaunchKernel(kfn, grid_l, dim3(kBlock), args, lds, h->stream));
  }
  if (chunks > 1) {
    // the closing chunk's degrade path (fs1_kernels.inc): a no-op launch unless a closing workgroup gave up waiting (or the
    // handle has left the in-kernel wait after an earlier give-up)
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_COMBINE);
    hipLaunchKernelGGL(k_fs1_combine, dim3(grid_for(h->n, kBlock)), dim3(kBlock), 0, h->stream, h->pw, h->ctl, h->n, h->partial, chunks,
                       h->obs_two_launch ? 1 : 0);
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
      const size_t rec_bytes = rr::kPlanRecBytes;
      RR_HIP_TRY(rr::dev_malloc(&h->grid_rec, rec_bytes));
      RR_HIP_TRY(hipMemsetAsync(h->grid_rec, 0, rec_bytes, h->stream));
      RR_HIP_TRY(rr::dev_malloc(&h->grid_ticket, rr::kTicketWords * sizeof(unsigned int)));
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
  if (h->ctl_host->obs_timeout) {
    // closing workgroups of k_fs1_observe gave up waiting for a chunk's factor (workgroups not dispatched in ascending order, or
    // kept off the device by another process); k_fs1_combine has formed those weights since -- same bits.  This handle forms
    // its weights there from now on (fs1_kernels.inc)
    h->obs_giveups += 1;
    h->obs_two_launch = true;
    RR_HIP_TRY(hipMemsetAsync(&h->ctl->obs_timeout, 0, sizeof(int), h->stream));
    h->ctl_host->obs_timeout = 0;
  }
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
  RR_HIP_TRY(rr::dev_malloc(&h->best_ticket, sizeof(unsigned int)));
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
  if (!h->pose_stage) RR_HIP_TRY(rr::dev_malloc(&h->pose_stage, 4 * h->n * sizeof(double)));
  return RR_OK;
}

rr_status ensure_noise(rr_fs1* h) {
  if (!h->noise) RR_HIP_TRY(rr::dev_malloc(&h->noise, 3 * h->n * sizeof(double)));  // FastSLAM 2.0 draws three normals
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
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->slab, 2 * state_bytes));  // one slab: peers map the whole state with one IPC handle
  h->pl.s[0] = h->slab;
  h->pl.s[1] = h->slab + h->n_planes * h->n;
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->pw, h->n * sizeof(double)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->cdf, h->n * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->tile_total, h->n_tiles * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->tile_q2, 2 * h->n_tiles * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->idx, h->n * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->markers, (h->n + rr::kResolveSlots) * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->markers, 0, (h->n + rr::kResolveSlots) * sizeof(unsigned int), h->stream));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->carry, (h->n / rr::kResolveSlots + 2) * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->part_bits, 1024 * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->part_idx, 1024 * sizeof(uint64_t)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->plane_list, (h->n_planes + 1) * sizeof(unsigned int)));
  RR_TRY_OR_CLEAN(rr::dev_malloc(&h->ctl, sizeof(Ctl)));
  RR_TRY_OR_CLEAN(hipHostMalloc(&h->ctl_host, sizeof(Ctl)));
  RR_TRY_OR_CLEAN(hipMemsetAsync(h->ctl, 0, sizeof(Ctl), h->stream));
  hipLaunchKernelGGL(k_fs1_init, dim3(grid_for(h->n, kBlock), (unsigned)std::min<uint64_t>(h->n_planes, 65535)), dim3(kBlock), 0, h->stream, h->pl, h->pw,
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
  auto t0 = std::chrono::steady_clock::now();
  const long long patience_ms = 2000 + (long long)(r.life_us / 1000.0);  // (see resident_await, pf_engine.hip)
  int gave_up = 0;
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
    if ((spins & 1023u) == 1023u && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > patience_ms) {
      // once the stream has drained the incarnation has either answered (next turn of the loop: the update WAS applied and is
      // reported as such) or left without the command (EXIT marker above: relaunched); an error only after two such rounds
      (void)hipStreamSynchronize(h->stream);
      if (++gave_up > 2) {
        r.live = false;
        r.pending = false;
        return fail(RR_RUNTIME_ERROR, "the resident FastSLAM kernel did not answer");
      }
      t0 = std::chrono::steady_clock::now();
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
  // values above 0.5 s are CLAMPED, not rejected (the bound used to be 1e7): 0.5 s of idling means <= 10 s of life, and a host
  // that waits for an answer from a kernel that has died sits out at most 3 x (2 s + life) = 36 s before it is told (resident_await)
  if (!(idle_us >= 0.0)) return fail(RR_INVALID_PARAMETER, "resident idle time must be >= 0 microseconds (values above 5e5 are clamped to 5e5)");
  if (idle_us > 5e5) idle_us = 5e5;
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

// rr_fs1_warm (include/rr_fastslam1.h): as rr_pf_warm
rr_status rr_fs1_warm(rr_fs1* h, double ms) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!(ms >= 0.0) || !(ms <= 2000.0)) return fail(RR_INVALID_PARAMETER, "warm-up time must lie in [0, 2000] milliseconds (0: the default, 50)");
  RR_HIP_TRY(rr::device_warm(h->stream, h->opt.device, ms == 0.0 ? 50.0 : ms));
  return RR_OK;
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
  RR_HIP_TRY(rr::dev_malloc(&h->ridx, h->n_global * sizeof(unsigned int)));  // worst case: every slot of every peer
  return RR_OK;
}

// RR_P2P_CU_PARTITION=1 (a test rig, as in pf_engine.hip): shards that SHARE a device each get a stream of their own 1/n of its CUs.
// A FastSLAM shard waits for its peers only in the one-workgroup exchange kernels, so it needs no CUs of its own to make progress --
// what it needs is a hardware queue of its own (a spinning exchange kernel must never sit in front of the kernel it waits for), and a
// stream created with a CU mask gets a dedicated one.  Eight shards of one process were seen to give up an exchange without it
// (round 6, tests/test_gpu_world8.py).
static rr_status fs1_apply_cu_partition(rr_fs1* h) {
  const char* e = std::getenv("RR_P2P_CU_PARTITION");
  if (!e || std::atoi(e) == 0 || h->p2p.n_sharing <= 1 || h->cu_partitioned) return RR_OK;
  int cus = 0;
  RR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->opt.device));
  const int parts = h->p2p.n_sharing, part = h->p2p.share_ordinal, per = cus / parts;
  if (per < 8 || part >= parts) return RR_OK;
  std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
  for (int b = part * per; b < (part + 1) * per; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
  hipStream_t s = nullptr;
  RR_HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  RR_HIP_TRY(hipStreamSynchronize(h->stream));
  (void)hipStreamDestroy(h->stream);
  h->stream = s;
  h->cu_partitioned = true;
  return RR_OK;
}

rr_status rr_fs1_p2p_export(rr_fs1* h, uint8_t out[RR_P2P_HANDLE_BYTES]) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!out) return fail(RR_INVALID_PARAMETER, "null output");
  // (no slab: a FastSLAM shard's peers deliver into its INBOX only -- k_fs1_push; at BASELINE configs[3] the slab is 2.4 GB per rank,
  // beyond what hipIpc maps on this driver, the inbox 1.2 GB)
  return h->p2p.export_handles(nullptr, h->n_planes * h->n, out);
}

rr_status rr_fs1_p2p_connect(rr_fs1* h, const uint8_t* all_handles, int32_t n_ranks, int32_t rank) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if (!all_handles) return fail(RR_INVALID_PARAMETER, "null handles");
  if ((s = fs1_check_geometry(h, n_ranks, rank)) != RR_OK) return s;
  if ((s = fs1_alloc_ridx(h)) != RR_OK) return s;
  if ((s = h->p2p.connect_ipc(h->slab, h->n_planes * h->n, all_handles, n_ranks, rank)) != RR_OK) return s;
  h->pl.inbox = h->p2p.inbox;
  return fs1_apply_cu_partition(h);
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
  for (int g = 0; g < n_ranks && s == RR_OK; ++g) s = fs1_apply_cu_partition(handles[g]);
  return s;
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
                     (const uint64_t*)&h->ctl->wmax_bits, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_QUANTIZE_REDUCE);
    hipLaunchKernelGGL(rr::k_quantize_reduce, dim3((unsigned)h->n_tiles), dim3(rr::kTileBlock), 0, h->stream, h->pw, h->ctl,
                       (const double*)&h->ctl->wmax, image_args(h), h->tile_total, h->tile_q2, dup ? 0 : 1);
  }
  // tile scan + exchange 2: every shard's sums -> global totals, gate (N_eff < NTH), plan
  {
    rr::ScopedTimer t(h->prof, h->stream, RR_FK_SCAN_TILES);
    hipLaunchKernelGGL(rr::k_scan_exchange, dim3(1), dim3(kScanThreads), 0, h->stream, h->p2p.peers, seq, h->tile_total,
                       (const uint64_t*)h->tile_q2, h->n_tiles, h->ctl, pa, h->p2p.err);
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
                     (const uint64_t*)local3, h->ctl, &h->ctl->wmax, pa, h->p2p.err);
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
    RR_HIP_TRY(rr::dev_malloc(&h->own_inbox, h->n_planes * h->n * sizeof(double)));
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
    RR_HIP_TRY(rr::dev_malloc(buf, want * sizeof(double)));
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

rr_status rr_fs1_observe_stats(rr_fs1* h, uint64_t* giveups, int32_t* in_kernel_wait_enabled) {
  rr_status s = bind(h);
  if (s != RR_OK) return s;
  if ((s = fetch_ctl(h)) != RR_OK) return s;
  if (giveups) *giveups = h->obs_giveups;
  if (in_kernel_wait_enabled) *in_kernel_wait_enabled = h->obs_two_launch ? 0 : 1;
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
