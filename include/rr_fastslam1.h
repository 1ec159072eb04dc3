/* rr_fastslam1.h -- C ABI of the MI355X FastSLAM 1.0 engine.
 *
 * Drop-in boundary for rust_robotics_slam::fastslam1
 *   (/root/reference/crates/rust_robotics_slam/src/fastslam1.rs): the reference exposes free
 * functions over a caller-owned Vec<Particle> (create_particles :302-306, fastslam_update
 * :237-266, get_best_particle :269-274, get_observations :277-299) and `pub` structs
 * (Landmark :26-31, Particle :44-51).  Here the particle set and every particle's landmark map
 * live on the GPU behind an opaque handle; rr_fs1_update_host is the compatibility shim with the
 * reference's "mutate the caller's vector" shape (upload -> step -> download).
 * Conventions as in rr_pf.h (status codes, rr_last_error, no CPU fallback).
 *
 * Host layouts
 *   poses  : N x (weight, x, y, yaw)                       -- field order of Particle :45-49
 *   maps   : N x L x (x, y, c00, c10, c01, c11)            -- Vec<Landmark> with a column-major
 *                                                             nalgebra Matrix2 (:27-30)
 *   z      : n_z x (distance, angle, landmark_id as double) -- &[(f64, f64, usize)] :240
 * Device layout (HBM): landmark-major planes plane[(l*6 + f) * N + p] so that one wavefront
 * reads 64 consecutive particles of one field with a single coalesced 512-byte access.
 */
#ifndef RR_FASTSLAM1_H
#define RR_FASTSLAM1_H

#include <stddef.h>
#include <stdint.h>

#include "rr_pf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the compile-time constants of fastslam1.rs:13-23 as fields */
typedef struct rr_fs1_params {
  double dt;             /* DT = 0.1 */
  double q00, q11;       /* Q_SIM diagonal: 0.3, 0.0305 (:22) */
  double r00, r11;       /* R_SIM diagonal: 0.5, 0.0305 (:23) */
  double max_range;      /* MAX_RANGE = 20.0 (:14), used by rr_fs1_get_observations */
  double nth;            /* NTH = N_PARTICLE / 1.5 = 66.67 (:18): resample iff N_eff < nth */
  double initial_weight; /* 1 / N_PARTICLE = 0.01 regardless of n (:56, Q10) */
  double init_cov;       /* Landmark::new cov = init_cov_value * I, 1000.0 (:38) */
  double init_threshold; /* cov[(0,0)] > 100.0 => first observation (:143) */
  double first_obs_cov;  /* NaN => the reference leaves cov untouched on first observation (Q11:
                            the EKF branch is then unreachable through updates alone); a finite
                            value c => cov := c * I as fastslam2.rs:254 does */
} rr_fs1_params;

typedef struct rr_fs1_options {
  int32_t device;
  int32_t record_indices; /* keep the last resample's source indices */
  uint64_t seed;          /* Philox key (motion noise, resample offset, simulator) */
  int32_t obs_chunks;     /* 0 = choose automatically; k >= 1 = split a step's observations over k
                             thread groups (partial weight products are combined in chunk order) */
  int32_t reserved;
  /* sharding (rr_fs1_shard_*): global index of this shard's particle 0 and the particle count over
   * all shards; 0 / 0 for a single-GPU filter */
  uint64_t first_global_index;
  uint64_t n_global;
} rr_fs1_options;

typedef struct rr_fs1 rr_fs1; /* opaque */

void rr_fs1_params_default(rr_fs1_params* p);
void rr_fs1_options_default(rr_fs1_options* o);

/* create_particles(n_particles, n_landmarks), fastslam1.rs:302-306 */
rr_status rr_fs1_create(uint64_t n_particles, uint64_t n_landmarks, const rr_fs1_params* params,
                        const rr_fs1_options* opt, rr_fs1** out);
void rr_fs1_destroy(rr_fs1* h);
uint64_t rr_fs1_particle_count(const rr_fs1* h);
uint64_t rr_fs1_landmark_count(const rr_fs1* h);

/* fastslam_update(&mut particles, u, z), :237-266: predict, per-observation EKF update,
 * normalise, N_eff-gated systematic resample.  Waits for completion. */
rr_status rr_fs1_update(rr_fs1* h, const double u[2], const double* z, size_t n_z);
/* the same, enqueued without waiting */
rr_status rr_fs1_update_async(rr_fs1* h, const double u[2], const double* z, size_t n_z);
rr_status rr_fs1_synchronize(rr_fs1* h);
/* as rr_pf_warm (include/rr_pf.h): `ms` milliseconds (0: 50) of step-shaped work on the filter's stream, so that the first
 * fastslam_update (render_gif_slam.rs:166-200 creates the particles and updates at once) runs at the steady rate */
rr_status rr_fs1_warm(rr_fs1* h, double ms);

/* get_best_particle, :269-274: arg max of the weight, ties -> last index */
rr_status rr_fs1_best_particle(rr_fs1* h, double out_pose[3], double* out_weight, uint64_t* out_index);
/* Resident service for the filters the reference's callers run (100 particles x 8 landmarks, render_gif_slam.rs:166-200; its
 * loop: fastslam_update, then get_best_particle, every step): with idle_us > 0, rr_fs1_update / rr_fs1_update_async of a
 * FastSLAM 1.0 filter of up to 1024 particles with at most 1 MB of maps (particles x (3 + 6 landmarks) <= 131 072 values; up to
 * 64 observations per update) launch nothing -- ONE kernel of one
 * workgroup stays on the device, takes each update's control and observations from a pinned command block and answers with
 * the best particle of the updated set in a pinned response block, so the rr_fs1_best_particle that follows an update is
 * free.  It leaves by itself after idle_us microseconds without an update and after max(100 ms, 20 idle_us) in any case (idle_us <= 0.5 s); every
 * other entry point asks it to leave first.  Same bits as the launched update.  0 switches the service off (the default); values
 * above 5e5 are clamped to 5e5, negative or NaN is RR_INVALID_PARAMETER; a host waiting for a kernel that has died gives up after
 * at most 3 x (2 s + life) <= 36 s with RR_RUNTIME_ERROR. */
rr_status rr_fs1_set_resident(rr_fs1* h, double idle_us);
/* incarnations of the resident kernel launched so far and updates served by them */
rr_status rr_fs1_resident_stats(const rr_fs1* h, uint64_t* launches, uint64_t* updates);
/* landmarks of one particle: out = L x (x, y, c00, c10, c01, c11) */
rr_status rr_fs1_get_landmarks(rr_fs1* h, uint64_t particle_index, double* out);
/* out = N x (weight, x, y, yaw) */
rr_status rr_fs1_get_poses(rr_fs1* h, double* out);
/* whole state in host layouts (either pointer may be NULL) */
rr_status rr_fs1_get_state(rr_fs1* h, double* poses_out, double* maps_out);
rr_status rr_fs1_set_state(rr_fs1* h, const double* poses, const double* maps);
/* compatibility shim: caller-owned state in host layouts, one fastslam_update on the GPU */
rr_status rr_fs1_update_host(rr_fs1* h, double* poses, double* maps, const double u[2], const double* z,
                             size_t n_z);

/* get_observations, :277-299 (simulator used by the reference's tests and examples): range gate
 * max_range, noise sqrt(R) * N(0,1) from the engine's Philox stream (seed, step, landmark id).
 * landmarks_xy = L x (x, y); out = cap x (d, angle, id); returns the number of observations. */
size_t rr_fs1_get_observations(const double x_true[3], const double* landmarks_xy, size_t n_landmarks,
                               const rr_fs1_params* params, uint64_t seed, uint32_t step, double* out,
                               size_t cap);

/* ---- parity seams */
/* predict_particle (:123-137) with caller-supplied unit normals z0[N], z1[N] */
rr_status rr_fs1_predict_with_noise(rr_fs1* h, const double u[2], const double* z0, const double* z1);
/* predict with the engine's Philox stream only */
rr_status rr_fs1_predict(rr_fs1* h, const double u[2]);
/* the observation loop (:250-256) only: EKF updates + weight accumulation, no normalise */
rr_status rr_fs1_observe(rr_fs1* h, const double* z, size_t n_z);
/* normalise + gate + resample (:259-265) only */
rr_status rr_fs1_normalize_resample(rr_fs1* h);
/* unconditional systematic resample with rho = r0 * N in [0, 1) (:205-234) */
rr_status rr_fs1_resample_systematic(rr_fs1* h, double rho);
rr_status rr_fs1_last_resample_fired(rr_fs1* h, int32_t* out);
rr_status rr_fs1_last_resample_indices(rr_fs1* h, uint32_t* out, size_t n);
rr_status rr_fs1_n_eff(rr_fs1* h, double* out);
rr_status rr_fs1_get_fixed_sums(rr_fs1* h, rr_pf_fixed_sums* out);
/* observation chunks the last observe used, and the Philox counters */
rr_status rr_fs1_get_counters(rr_fs1* h, uint32_t* step, uint32_t* resample_step, int32_t* obs_chunks);
/* one-launch resample plan of this handle: launches that degraded to the serial plan because the device did not run all
 * of their workgroups at once (rr_pf_plan_stats in rr_pf.h tells the story), and whether the handle still uses it */
rr_status rr_fs1_plan_stats(rr_fs1* h, uint64_t* giveups, int32_t* one_launch_enabled);
/* the observation kernel's in-kernel hand-over of the chunks' weight factors (fastslam1.rs:250-256 `weight *=`, chunked): host looks
 * at the device that found a closing workgroup had given up waiting for a factor -- the weights were then formed by the follow-up
 * kernel, same bits, no error -- and whether the handle still waits inside the kernel (0 after the first give-up) */
rr_status rr_fs1_observe_stats(rr_fs1* h, uint64_t* giveups, int32_t* in_kernel_wait_enabled);

/* ---- sharded FastSLAM: contiguous particle blocks over the GPUs of a node, each
 * particle's whole map moves with it.  Peer-to-peer transport only (include/rr_pf.h "peer-to-peer
 * transport" describes the protocol): the weight maximum and the integer sums are exchanged through
 * the peers' mailboxes, and the systematic resample stores every served slot's 3 + 6L planes
 * straight into the owning shard's state slab over xGMI.  Bit-identical to the unsharded filter. */
rr_status rr_fs1_p2p_export(rr_fs1* h, uint8_t out[RR_P2P_HANDLE_BYTES]);
rr_status rr_fs1_p2p_connect(rr_fs1* h, const uint8_t* all_handles, int32_t n_ranks, int32_t rank);
rr_status rr_fs1_p2p_connect_local(rr_fs1* const* handles, int32_t n_ranks);
/* fastslam_update over all shards, fully asynchronous; every shard calls it with the same u and z */
rr_status rr_fs1_shard_update_p2p(rr_fs1* h, const double u[2], const double* z, size_t n_z);
rr_status rr_fs1_p2p_status(rr_fs1* h, int32_t* timed_out);

/* ---- the same sharded update over RCCL (BASELINE.json configs[3]: "RCCL weight all-reduce + global
 * resample"; replaces fastslam1.rs:186-234 over all shards): all-reduce(MAX) of the weight maximum,
 * all-gather of every shard's integer sums (T, sum q^2), one small D2H of the totals, then a grouped
 * ncclSend / ncclRecv of the whole particles (3 + 6L planes each) whose source lives on another rank,
 * one contiguous [plane][count] block per (source, destination) pair.  Own slots are resampled lazily
 * as on one GPU.  `c` comes from rr_comm_create (include/rr_pf.h); one host synchronisation per update. */
rr_status rr_fs1_shard_update(rr_fs1* h, rr_comm* c, const double u[2], const double* z, size_t n_z);
uint64_t rr_fs1_shard_last_migrated(const rr_fs1* h);
/* The phases of rr_fs1_shard_update for a host-orchestrated transport (any collective library; the
 * CPU tests drive the same sequence with the oracle standing in for the kernels).  d_* are device pointers. */
rr_status rr_fs1_shard_local(rr_fs1* h, const double u[2], const double* z, size_t n_z, double* d_wmax_out);
rr_status rr_fs1_shard_quantize(rr_fs1* h, const double* d_wmax_global, uint64_t* d_sums_out /* [3] */);
rr_status rr_fs1_shard_plan(rr_fs1* h, const uint64_t* d_all_sums /* [n_shards][3] */, int32_t n_shards, int32_t rank);
rr_status rr_fs1_shard_get_plan(rr_fs1* h, rr_pf_shard_plan* out); /* synchronises the stream */
/* matrix = rr_sys_segment_matrix(...)[src * n_shards + dst].  Block of destination g (g != rank, ascending g):
 * matrix[rank][g] particles x (3 + 6L) planes, laid out [plane][count], blocks back to back in d_send;
 * d_recv holds the blocks of the sources g != rank in the same order. */
rr_status rr_fs1_shard_pack(rr_fs1* h, const int64_t* matrix, int32_t n_shards, int32_t rank, double* d_send);
rr_status rr_fs1_shard_unpack(rr_fs1* h, const int64_t* matrix, int32_t n_shards, int32_t rank, const double* d_recv);

/* ---- measurement hooks */
typedef enum rr_fs1_kernel_id {
  RR_FK_PREDICT = 0,
  RR_FK_OBSERVE = 1,
  RR_FK_COMBINE = 2,
  RR_FK_QUANTIZE_REDUCE = 3,
  RR_FK_SCAN_TILES = 4,
  RR_FK_NORMALIZE = 5,
  RR_FK_CDF = 6,
  RR_FK_INDICES = 7,
  RR_FK_GATHER = 8,
  RR_FK_COMMIT = 9,
  RR_FK_COUNT = 10
} rr_fs1_kernel_id;
rr_status rr_fs1_profile_enable(rr_fs1* h, int32_t enable);
rr_status rr_fs1_profile_read(rr_fs1* h, int32_t kernel_id, uint64_t* launches, double* total_ms);
rr_status rr_fs1_profile_reset(rr_fs1* h);
const char* rr_fs1_kernel_name(int32_t kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* RR_FASTSLAM1_H */
