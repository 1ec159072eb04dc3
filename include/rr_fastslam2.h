/* rr_fastslam2.h -- C ABI of the MI355X FastSLAM 2.0 engine.
 *
 * Drop-in boundary for rust_robotics_slam::fastslam2
 *   (/root/reference/crates/rust_robotics_slam/src/fastslam2.rs): free functions over a
 * caller-owned Vec<Particle> -- create_particles :425-429, fastslam2_update :376-383 (the seedable
 * body is fastslam2_update_with_rng :331-374), get_best_particle :385-390, get_observations
 * :419-423 -- with the same `pub` structs as FastSLAM 1.0.
 *
 * FastSLAM 2.0 differs from 1.0 in how the pose is drawn (from a proposal that fuses the motion
 * prior with the step's first observation, compute_proposal :173-216 + sample_pose :219-239) and
 * in three constants of the landmark update (:49-51 initialised iff cov00 < 100, :255 first
 * observation sets cov = 10 I, :289 weight 1e-10 when det S <= 0).  Everything else -- per
 * (particle, observation) 2x2 EKF, normalise, N_eff gate, systematic resample, best particle --
 * is the FastSLAM 1.0 engine, so an rr_fs2 handle IS an rr_fs1 handle: every accessor, seam,
 * measurement hook and the sharded update of rr_fastslam1.h apply to it unchanged
 * (rr_fs1_update on such a handle runs the FastSLAM 2.0 step; rr_fs2_update is the same call).
 */
#ifndef RR_FASTSLAM2_H
#define RR_FASTSLAM2_H

#include "rr_fastslam1.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef rr_fs1 rr_fs2;

typedef struct rr_fs2_params {
  rr_fs1_params base;    /* DT, Q_SIM (used only when a step has no observation, :349-357), R_SIM,
                            MAX_RANGE, NTH, 1/N_PARTICLE, 1000 I, 100; first_obs_cov = 10 (:255) */
  double motion_cov[3];  /* MOTION_COV diagonal 0.1, 0.1, 0.01 (:30) */
  double nonpos_det_weight; /* 1e-10 (:289) */
} rr_fs2_params;

void rr_fs2_params_default(rr_fs2_params* p);
/* create_particles(n_particles, n_landmarks), :425-429 */
rr_status rr_fs2_create(uint64_t n_particles, uint64_t n_landmarks, const rr_fs2_params* params,
                        const rr_fs1_options* opt, rr_fs2** out);
/* fastslam2_update(&mut particles, u, z), :376-383; waits for completion */
rr_status rr_fs2_update(rr_fs2* h, const double u[2], const double* z, size_t n_z);
rr_status rr_fs2_update_async(rr_fs2* h, const double u[2], const double* z, size_t n_z);

/* ---- parity seams */
/* the sampling step only (:339-358) with caller-supplied unit normals, 3 per particle
 * (noise[3p + k]; the third is unused when n_z == 0) */
rr_status rr_fs2_predict_with_noise(rr_fs2* h, const double u[2], const double* z, size_t n_z, const double* noise);
/* the same with the engine's Philox streams */
rr_status rr_fs2_predict(rr_fs2* h, const double u[2], const double* z, size_t n_z);

#ifdef __cplusplus
}
#endif
#endif /* RR_FASTSLAM2_H */
