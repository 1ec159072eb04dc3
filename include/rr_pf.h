/* rr_pf.h -- C ABI of the MI355X particle-filter localization engine.
 *
 * Drop-in boundary for the hot path of
 *   rust_robotics_localization::particle_filter::ParticleFilterLocalizer
 *     (/root/reference/crates/rust_robotics_localization/src/particle_filter.rs:121-573)
 *   rust_robotics_localization::monte_carlo_localization::MonteCarloLocalizer, with a fixed
 *     particle count (min_particles == max_particles: the BASELINE "MCL" configurations) or the
 *     KLD-adaptive count (rr_pf_create_adaptive)
 *     (/root/reference/crates/rust_robotics_localization/src/monte_carlo_localization.rs:136-462)
 * The reference offers no FFI/plugin interface (it is `#![forbid(unsafe_code)]`
 * Rust, lib.rs:1); the seam is its struct + trait surface.  Every entry point
 * below names the reference method it stands in for; a Rust `-sys` crate binds
 * them one to one (INTEGRATION.md shows the extern block and the safe wrapper
 * implementing rust_robotics_core::StateEstimator, traits.rs:31-52).
 *
 * Conventions
 *   - plain C, opaque handle, POD structs, no C++/torch types;
 *   - every fallible call returns rr_status; rr_last_error() returns the message
 *     of the calling thread's last failure.  RR_INVALID_PARAMETER corresponds to
 *     RoboticsError::InvalidParameter (rust_robotics_core/src/error.rs:8-24) and
 *     carries the reference's message text;
 *   - one caller at a time per handle (the reference methods take &mut self);
 *     distinct handles are independent (one HIP stream each);
 *   - host pointers unless a parameter says "device";
 *   - there is NO CPU fallback: creating a handle without a usable HIP device
 *     fails with RR_RUNTIME_ERROR.
 */
#ifndef RR_PF_H
#define RR_PF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rr_status {
  RR_OK = 0,
  RR_INVALID_PARAMETER = 1, /* RoboticsError::InvalidParameter */
  RR_RUNTIME_ERROR = 2      /* HIP / allocation / device failure */
} rr_status;

/* message of the last failing call on this thread ("" if none) */
const char* rr_last_error(void);
/* library version string, e.g. "rust_robotics_amd 0.1.0 (gfx950; sources 3547923a74322cb8)": the hash is the first 16 hex digits of the
 * SHA-256 over the sources the library was built from (csrc/Makefile) */
const char* rr_version(void);
/* number of visible HIP devices (0 if none / runtime unusable) */
int rr_device_count(void);
/* PCI address of a device as sysfs spells it ("0000:c1:00.0"): /sys/bus/pci/devices/<id>/local_cpulist names the host CPUs on
 * the device's NUMA node -- the synchronous step is two trips over the host link (rr_pf_set_resident), a caller that stays
 * on that node saves the socket interconnect on every one of them (nodes/pf_localizer_node pins itself there) */
rr_status rr_device_pci_bus_id(int32_t device, char* out, size_t cap);

/* ParticleFilterConfig, particle_filter.rs:51-65 (field for field) */
typedef struct rr_pf_config {
  uint64_t n_particles;
  double resample_threshold;
  double range_noise;
  double velocity_noise;
  double yaw_rate_noise;
  double dt;
} rr_pf_config;

/* How the engine realises the parts of the step the reference leaves to its
 * (unseedable) RNG and to its two resamplers. */
typedef enum rr_resample_scheme {
  RR_RESAMPLE_MULTINOMIAL = 0, /* particle_filter.rs:441-473, monte_carlo_localization.rs:343-355 */
  RR_RESAMPLE_SYSTEMATIC = 1   /* fastslam1.rs:205-234 (low variance) */
} rr_resample_scheme;
typedef enum rr_resample_gate {
  RR_GATE_NEFF = 0,  /* particle_filter.rs:337-345: iff N_eff < N * resample_threshold */
  RR_GATE_ALWAYS = 1 /* monte_carlo_localization.rs:298: every step */
} rr_resample_gate;
typedef enum rr_likelihood_mode {
  RR_LIK_FUSED = 0,  /* exp(L ln c - sum diff^2 / 2 sigma^2): one exp per particle */
  RR_LIK_PRODUCT = 1 /* prod_l c * exp(-diff_l^2 / 2 sigma^2): the reference's literal form */
} rr_likelihood_mode;

typedef struct rr_pf_options {
  int32_t device;          /* HIP device ordinal */
  int32_t resample_scheme; /* rr_resample_scheme */
  int32_t resample_gate;   /* rr_resample_gate */
  int32_t likelihood_mode; /* rr_likelihood_mode */
  uint64_t seed;           /* Philox key of this filter's noise streams */
  /* sharding (rr_pf_shard_* below); a single-GPU filter uses 0 / 1 / 0 */
  uint64_t first_global_index; /* global index of this shard's particle 0 */
  uint64_t n_global;           /* particles over all shards; 0 => n_particles */
  int32_t record_indices;      /* keep the last resample's source indices for rr_pf_last_resample_indices */
  int32_t reserved;
} rr_pf_options;

typedef struct rr_pf rr_pf; /* opaque: device-resident particle set + stream */

/* The KLD part of MonteCarloLocalizationConfig, monte_carlo_localization.rs:51-82 */
typedef struct rr_mcl_adaptive {
  uint64_t min_particles; /* 100; the filter starts with this many (try_new :147) */
  uint64_t max_particles; /* 5000; capacity of the device buffers */
  double kld_epsilon;     /* 0.05 */
  double kld_z;           /* 2.326 */
} rr_mcl_adaptive;

/* ParticleFilterConfig::default(), particle_filter.rs:67-78 */
void rr_pf_config_default(rr_pf_config* cfg);
/* ParticleFilterConfig::validate(), particle_filter.rs:81-117 (same messages) */
rr_status rr_pf_config_validate(const rr_pf_config* cfg);
/* PF defaults: device 0, multinomial, N_eff gate, fused likelihood, seed 0 */
void rr_pf_options_default(rr_pf_options* opt);
/* MonteCarloLocalizer fixed-N semantics: multinomial, resample every step */
void rr_pf_options_mcl(rr_pf_options* opt);

/* try_new, particle_filter.rs:139-156: all particles at the origin, w = 1/N */
rr_status rr_pf_create(const rr_pf_config* cfg, const rr_pf_options* opt, rr_pf** out);
/* try_with_initial_state, particle_filter.rs:170-199: uniform jitter
 * (+-1, +-1, +-0.25, +-0.5) around state[4] = (x, y, yaw, v) */
rr_status rr_pf_create_with_state(const rr_pf_config* cfg, const rr_pf_options* opt,
                                  const double state[4], rr_pf** out);
void rr_pf_destroy(rr_pf* h);

/* ---- MonteCarloLocalizer with min_particles < max_particles: the particle count adapts every
 * step to the KLD bound over the occupied 0.5 m x 0.5 m x 15 deg bins
 * (resample_adaptive, monte_carlo_localization.rs:322-365; kld_required_particles :367-378).
 * cfg->n_particles is ignored (the filter starts with kld->min_particles); state may be NULL
 * (try_new :142-156) or the initial state of try_with_initial_state (:170-199).  The options must
 * ask for multinomial resampling at every step (rr_pf_options_mcl); sharding is not available.
 * The reference draws particles one at a time until the bound holds; the engine evaluates all
 * max_particles candidate draws at once and keeps the same prefix.  The new particle count stays on
 * the device: rr_pf_step_async does not wait for it (every kernel of such a filter reads its n from
 * device memory, launches are sized for max_particles); the host's copy -- rr_pf_particle_count, the
 * accessors, the synchronous entry points -- is refreshed on demand (one 8-byte copy + a wait). */
void rr_mcl_adaptive_default(rr_mcl_adaptive* kld);
/* MonteCarloLocalizationConfig::validate, :84-131 (same messages) */
rr_status rr_mcl_adaptive_validate(const rr_mcl_adaptive* kld);
rr_status rr_pf_create_adaptive(const rr_pf_config* cfg, const rr_pf_options* opt, const rr_mcl_adaptive* kld,
                                const double* state, rr_pf** out);
/* largest particle count the filter can hold (max_particles; n_particles for a fixed-N filter) */
uint64_t rr_pf_particle_capacity(const rr_pf* h);

/* try_set_landmarks :216-220 (stored only, never read by the update -- Q19);
 * xy = n x (x, y) */
rr_status rr_pf_set_landmarks(rr_pf* h, const double* xy, size_t n);
size_t rr_pf_landmark_count(const rr_pf* h);
/* get_landmarks :239-241; copies min(cap, count) points, returns count */
size_t rr_pf_get_landmarks(const rr_pf* h, double* xy_out, size_t cap);
/* set_range_noise :228-236 */
rr_status rr_pf_set_range_noise(rr_pf* h, double range_noise);

/* try_predict_with_control :255-301; control = (v, yaw_rate) */
rr_status rr_pf_predict(rr_pf* h, const double control[2]);
/* try_update_with_observations :310-334; obs = n_obs x (d, landmark_x, landmark_y) */
rr_status rr_pf_update(rr_pf* h, const double* obs, size_t n_obs);
/* resample :337-345 (gate per options) */
rr_status rr_pf_resample(rr_pf* h);
/* try_step :488-497 = predict + update + resample; writes the estimate */
rr_status rr_pf_step(rr_pf* h, const double control[2], const double* obs, size_t n_obs,
                     double out_state[4]);
/* the same step enqueued on the filter's stream without waiting for it: the
 * form a node uses when it only publishes every k-th estimate, and what
 * bench.py times.  Fused propagate+weight kernel, no host synchronisation. */
rr_status rr_pf_step_async(rr_pf* h, const double control[2], const double* obs, size_t n_obs);
/* rr_pf_step_async that also produces, on the device, the mean try_step returns (:496: the cache refreshed at
 * :343 after a fired resample -- uniform weights over the resampled set -- or at :332 after the update).
 * Systematic scheme: inside the step's own plan kernel, sum_j offspring_j p_j / N resp. sum_j q_j p_j / T over the
 * integer image of the weights.  Multinomial scheme (the resampler the reference's localizers use): the offspring
 * counts of iid draws do not exist before the draws are searched, so the resampled set's mean is added up by the
 * kernel that draws, searches and gathers -- the NEXT step's fused kernel, or rr_pf_last_step_estimate's gather
 * when it is asked first (the same sums in the same order either way).  No extra launch per step, no host
 * synchronisation; rr_pf_last_step_estimate waits for the stream and reads the last step's value.  Fixed N, one
 * shard, up to 8 388 608 particles; rr_pf_step uses the systematic form automatically when it applies. */
rr_status rr_pf_step_async_estimate(rr_pf* h, const double control[2], const double* obs, size_t n_obs);
rr_status rr_pf_last_step_estimate(rr_pf* h, double out[4]);
/* n_steps steps in one call: controls = n_steps x (v, yaw_rate), obs = n_steps x n_obs x (d, landmark_x, landmark_y) (the
 * same number of observations every step).  out_estimates = n_steps x 4 receives what try_step would have returned after
 * each step (the call then waits for the device), or NULL (asynchronous).  For particle sets of up to 2048 particles --
 * every caller in the reference runs 100 - 1200 -- all steps run inside ONE kernel launch of one workgroup, the particles in
 * registers from the first step to the last: a step costs its arithmetic instead of a launch.  The single-step entry points
 * above use the same kernel for such sets (RR_PF_SMALL=0 at create time: never).  Larger filters: a loop over rr_pf_step /
 * rr_pf_step_async.  Results are bit-identical to n_steps single steps. */
rr_status rr_pf_step_many(rr_pf* h, const double* controls, const double* obs, size_t n_obs, size_t n_steps,
                          double* out_estimates);
/* Resident service for the filters the reference's callers really run (100 - 1 200 particles through the synchronous try_step:
 * headless_localizers.rs:39-56, render_gif_particle_filter.rs:77-79, ros2_nodes/ekf_localizer_node/src/main.rs:273): with
 * idle_us > 0, rr_pf_step and rr_pf_step_async of a filter of up to 2048 particles -- or of a KLD-adaptive filter of up to 16 384
 * max_particles -- (and up to 128, adaptive: 96, observations per step) no longer launch anything; other filters keep their launches -- ONE kernel of one workgroup stays on the device with the particles in registers, takes each
 * step's control and observations from a pinned command block and leaves the estimate in a pinned response block the
 * host polls.  A synchronous step then costs its arithmetic plus two trips over the host link instead of a launch and a
 * completion wait.  The kernel leaves by itself after idle_us microseconds without a step (the next step starts it again:
 * nothing is lost, that step just pays a launch) and after max(100 ms, 20 idle_us) in any case (idle_us <= 0.5 s), so work queued behind it is
 * delayed, never blocked; every other entry point of the handle asks it to leave before it touches the particle set.  Same
 * kernel code, same bits as the launched step.  While it runs it holds one workgroup's registers and LDS of one compute unit.
 * idle_us == 0 switches the service off (the default; RR_PF_RESIDENT_US=<us> at create time switches it on).  idle_us above 5e5
 * (0.5 s) is clamped to 5e5; negative or NaN is RR_INVALID_PARAMETER.  Worst-case blocking of a synchronous step: an answer from
 * a live kernel arrives in microseconds; if the kernel has died the host gives up after 3 x (2 s + life) <= 36 s and reports
 * RR_RUNTIME_ERROR. */
rr_status rr_pf_set_resident(rr_pf* h, double idle_us);
/* incarnations of the resident kernel launched so far and steps served by them */
rr_status rr_pf_resident_stats(const rr_pf* h, uint64_t* launches, uint64_t* steps);
/* wait for everything enqueued on the filter's stream */
rr_status rr_pf_synchronize(rr_pf* h);
/* Make the caller's FIRST step as fast as its thousandth (engine extension; the reference's callers -- headless_localizers.rs:39-56
 * -- create a localizer and step it at once): an idle MI355X runs its first ~50 ms of work at reduced clocks and the HIP runtime
 * has one-off costs along a process's first launches (measured, 1e6 x 32: 52.6 us/step right after create against 47.9 after a
 * thousand steps).  rr_pf_warm enqueues `ms` milliseconds (0: the default, 50; at most 2000) of step-shaped FP64 work on the
 * filter's stream and waits for it; the particle set is not touched.  Clocks fall again when the device idles for long. */
rr_status rr_pf_warm(rr_pf* h, double ms);

/* estimate :348-350 -- weighted mean (x, y, yaw, v) of the current particle set */
rr_status rr_pf_estimate(rr_pf* h, double out[4]);
/* calc_covariance :363-365 -- 4x4 row-major */
rr_status rr_pf_covariance(rr_pf* h, double out[16]);
/* particle_count(): config.n_particles, or the current count of an adaptive filter (:318-320) */
uint64_t rr_pf_particle_count(const rr_pf* h);
/* get_particles :244-246; out = N x (x, y, yaw, v, w) with normalised w */
rr_status rr_pf_get_particles(rr_pf* h, double* out_aos);
/* N_eff of the current weights, calc_n_eff :416-423 */
rr_status rr_pf_n_eff(rr_pf* h, double* out);
/* 1 if the most recent resample()/step() actually resampled */
rr_status rr_pf_last_resample_fired(rr_pf* h, int32_t* out);

/* ---- parity seams: what the reference keeps private, exposed so that the
 * engine can be checked against the CPU oracles on identical inputs ---- */
/* overwrite the particle set; aos = N x (x, y, yaw, v, w); w is taken as a raw weight */
rr_status rr_pf_set_particles(rr_pf* h, const double* aos);
/* predict with caller-supplied scaled noise samples instead of the Philox stream
 * (n_v[i], n_w[i] are the values Normal(0, sigma).sample() would have returned) */
rr_status rr_pf_predict_with_noise(rr_pf* h, const double control[2], const double* n_v,
                                   const double* n_w);
/* adaptive filters only: overwrite the particle set with n <= max_particles particles
 * (what the reference's own test does through its private field, :521-545) */
rr_status rr_pf_set_particles_n(rr_pf* h, const double* aos, uint64_t n);
/* adaptive filters only: resample_adaptive with caller-supplied uniforms r[max_particles]
 * (draw m consumes r[m]); *n_new receives the new particle count */
rr_status rr_pf_resample_adaptive_with_uniforms(rr_pf* h, const double* r, size_t n_r, uint64_t* n_new);
/* unconditional multinomial resample with caller-supplied uniforms r[N] in [0,1) */
rr_status rr_pf_resample_with_uniforms(rr_pf* h, const double* r, size_t n);
/* unconditional systematic resample with caller-supplied rho = r0 * N in [0,1) */
rr_status rr_pf_resample_systematic(rr_pf* h, double rho);
/* source index of every output slot of the last resample (needs record_indices) */
rr_status rr_pf_last_resample_indices(rr_pf* h, uint32_t* out, size_t n);
/* raw (unnormalised) weights as the weight kernel wrote them */
rr_status rr_pf_get_raw_weights(rr_pf* h, double* out);
/* integer image of the current weights: shift, T = sum q_i, sum q_i^2 (hi, lo), w_max */
typedef struct rr_pf_fixed_sums {
  int32_t usable; /* 0 => degenerate weights (uniform fallback, :433-438) */
  int32_t shift;
  uint64_t total;
  uint64_t q2_hi, q2_lo;
  double w_max;
  double sum; /* total * 2^-shift */
} rr_pf_fixed_sums;
rr_status rr_pf_get_fixed_sums(rr_pf* h, rr_pf_fixed_sums* out);
/* The fused systematic step plans its resample in ONE launch whose workgroups hand their tile sums to each other inside
 * the kernel; that needs every workgroup on the device at once.  When another process keeps some of them off the CUs the
 * launch does not fail: after RR_PF_PLAN_TIMEOUT_US (default 2000) it plans serially in its last workgroup -- same bits,
 * milliseconds instead of microseconds -- and the handle takes the multi-launch plan from its next host read on.
 * giveups: launches that degraded so far; one_launch_enabled: 0 once the handle has left the one-launch plan
 * (or never used it: RR_PF_FUSED_PLAN=0, more than 2^20 particles).  Synchronises the stream. */
rr_status rr_pf_plan_stats(rr_pf* h, uint64_t* giveups, int32_t* one_launch_enabled);
/* step / resample counters that key the Philox streams */
rr_status rr_pf_get_counters(rr_pf* h, uint32_t* step, uint32_t* resample_step);

/* ---- measurement hooks (bench.py): HIP-event timing of the engine's kernels on
 * the filter's own stream ---- */
typedef enum rr_pf_kernel_id {
  RR_K_PROPAGATE_WEIGHT = 0,
  RR_K_QUANTIZE_REDUCE = 1,
  RR_K_SCAN_TILES = 2,
  RR_K_CDF = 3,
  RR_K_RESAMPLE_GATHER = 4,
  RR_K_COMMIT = 5,
  RR_K_MOMENTS = 6,
  RR_K_COUNT = 7
} rr_pf_kernel_id;
/* enable != 0: bracket every kernel launch with hipEvents from now on */
rr_status rr_pf_profile_enable(rr_pf* h, int32_t enable);
/* accumulated since the last reset: launches and total milliseconds per kernel */
rr_status rr_pf_profile_read(rr_pf* h, int32_t kernel_id, uint64_t* launches, double* total_ms);
rr_status rr_pf_profile_reset(rr_pf* h);
const char* rr_pf_kernel_name(int32_t kernel_id);

/* ---- device self-test of the arithmetic contract (include/rr_detmath.h): evaluates one
 * function of the contract ON THE GPU for n inputs so that tests can assert bit-identity
 * with the host evaluation.  fn: 0 exp(a) 1 log(a) 2 sincos(a)->out0,out1 3 sincos2pi(a)
 * 4 atan2(a,b) 5 sqrt(a) 6 a/b 7 normal2(seed=a[0] bits, stream 3, step b[0] bits, index i)
 * 8 fma(a,b,a).  Host pointers; out1 may be NULL for single-output functions. */
rr_status rr_selftest_math(int32_t device, int32_t fn, size_t n, const double* a, const double* b,
                           double* out0, double* out1);

/* ---- sharded (multi-GPU) operation: one shard per process/GPU.  A shard is an ordinary
 * filter created with options.first_global_index / n_global set; the host side
 * (rust_robotics_amd/sharded.py) runs the RCCL collectives between the phases below.
 * Pointers named d_* are DEVICE pointers on this shard's device.  Systematic resampling only.
 *
 *   A  rr_pf_shard_propagate_weight   fused propagate + weight, local max weight -> d_wmax
 *      -- all-reduce(MAX) of d_wmax over the shards --
 *   B  rr_pf_shard_quantize           integer image under the GLOBAL max, local sums -> d_sums
 *      -- all-gather of d_sums (3 x u64 per shard) --
 *   C  rr_pf_shard_cdf                global totals, gate decision, local slice of the global CDF
 *      -- host: rr_pf_shard_get_plan + rr_sys_first_slot_above give every (source, destination)
 *         segment length; nothing below runs if the gate did not fire --
 *   D  rr_pf_shard_gather_slots       particles that global output slots [first, first+n) copy
 *                                     from THIS shard -> d_out, n x (x, y, yaw, v)
 *      -- all-to-all of the contiguous segments --
 *   E  rr_pf_shard_adopt              the received N x (x, y, yaw, v) becomes the particle set
 */
/* run all subsequent work of this handle on `stream` (a hipStream_t; NULL restores the
 * filter's own stream) so that it is ordered with the caller's collectives */
rr_status rr_pf_set_stream(rr_pf* h, void* stream);
rr_status rr_pf_shard_propagate_weight(rr_pf* h, const double control[2], const double* obs,
                                       size_t n_obs, double* d_wmax_out);
rr_status rr_pf_shard_quantize(rr_pf* h, const double* d_wmax_global, uint64_t* d_sums_out /* [3] */);
rr_status rr_pf_shard_cdf(rr_pf* h, const uint64_t* d_all_sums /* [n_shards][3] */, int32_t n_shards,
                          int32_t rank);
typedef struct rr_pf_shard_plan {
  int32_t fired;
  int32_t usable;
  uint64_t total_global;
  uint64_t base;
  uint64_t total_local;
  double rho;
} rr_pf_shard_plan;
/* synchronises the stream and copies phase C's outcome to the host */
rr_status rr_pf_shard_get_plan(rr_pf* h, rr_pf_shard_plan* out);
rr_status rr_pf_shard_gather_slots(rr_pf* h, uint64_t first_slot, uint64_t n_slots, double* d_out);
rr_status rr_pf_shard_adopt(rr_pf* h, const double* d_in);
/* Multinomial shards (RR_RESAMPLE_MULTINOMIAL: the resampler MonteCarloLocalizer really uses,
 * monte_carlo_localization.rs:322-365,387-392; particle_filter.rs:441-473).  Draw k is output slot k of the
 * GLOBAL index and a pure function of (seed, resample step, k); the shard whose CDF interval contains its
 * target serves it.  After rr_pf_shard_cdf (which materialises the local CDF slice for such shards):
 *   select        -> d_counts_out[n_shards]: slots this shard serves per destination (row `rank` of the matrix)
 *   pack_selected -> d_send: those records, ordered by global slot (= grouped by destination), 5 doubles each:
 *                    x, y, yaw, v, destination-local slot index
 *   adopt_records <- the n_local records of this shard's slots, in any order
 * Bit-identical to the unsharded multinomial filter for any shard count (tests/test_sharded_gloo.py).
 * Order is enforced: select needs the plan of a preceding rr_pf_shard_cdf made for the same n_shards, pack_selected a
 * preceding select with the same n_shards (RR_INVALID_PARAMETER otherwise -- they draw from the Philox stream of THAT resample). */
rr_status rr_pf_shard_select(rr_pf* h, int32_t n_shards, uint64_t* d_counts_out);
rr_status rr_pf_shard_pack_selected(rr_pf* h, int32_t n_shards, double* d_send);
rr_status rr_pf_shard_adopt_records(rr_pf* h, const double* d_in, uint64_t n_records);
/* Host-only integer helper (no GPU needed): the first global output slot i in [0, n_global]
 * whose systematic CDF target exceeds `bound` (n_global if none).  Slots served by a shard
 * with CDF interval (base, base + T] are [first_slot_above(base), first_slot_above(base + T)). */
uint64_t rr_sys_first_slot_above(double rho, uint64_t total_global, uint64_t n_global, uint64_t bound);
/* Host-only: the whole exchange plan of one step.  totals[n_shards] = every shard's T (phase B);
 * out[src * n_shards + dst] = number of global output slots owned by shard dst (equal blocks of
 * n_local slots) whose source particle lives on shard src.  Row `rank` gives the all-to-all send
 * splits, column `rank` the receive splits.  Returns the first global slot shard `rank` serves. */
uint64_t rr_sys_segment_matrix(double rho, const uint64_t* totals, int32_t n_shards, uint64_t n_global,
                               uint64_t n_local, int32_t rank, int64_t* out);

/* ---- peer-to-peer transport: the same sharded step with NO host code and NO collective
 * library inside it.  Every rank maps its peers' particle slab and mailbox (hipIpc handles, or
 * plain pointers when all shards live in one process) and the exchanges become tiny kernels that
 * publish a 32-byte record to every peer over xGMI and wait for theirs (bounded).  At resample
 * time only the slots whose source lives on another rank move -- stored into the owning rank's
 * fine-grained inbox, each with a seal over its fields that the owner's next step checks before it uses them -- the others are read through
 * their source index by the next step, exactly as on one GPU.  Results are bit-identical to the
 * RCCL path and to the unsharded filter. */
#define RR_P2P_HANDLE_BYTES 256
/* this rank's IPC handles (state slab, mailbox, inbox) for the other ranks, followed by the PCI bus id of the exporting
 * device: ranks that find a peer on their own device keep their spinning plan kernels small enough to leave it room */
rr_status rr_pf_p2p_export(rr_pf* h, uint8_t out[RR_P2P_HANDLE_BYTES]);
/* all_handles = n_ranks blobs from rr_pf_p2p_export in rank order (own entry ignored) */
rr_status rr_pf_p2p_connect(rr_pf* h, const uint8_t* all_handles, int32_t n_ranks, int32_t rank);
/* all shards in THIS process (one or several devices with peer access): handles[g] is rank g */
rr_status rr_pf_p2p_connect_local(rr_pf* const* handles, int32_t n_ranks);
/* one sharded step, fully asynchronous (nothing is waited for on the host).  Shards created with RR_RESAMPLE_MULTINOMIAL -- the
 * resampler the reference's ParticleFilterLocalizer and MonteCarloLocalizer really use (particle_filter.rs:441-473,
 * monte_carlo_localization.rs:322-365) -- take the same entry point: iid draws scatter the slots a shard serves over all ranks,
 * so every draw is searched by the shard whose interval of the global CDF holds it and its source is stored straight into the
 * owning rank's slab, with a DONE exchange behind the stores (no window, nothing lazy; rr_pf_shard_want_estimate is not offered) */
rr_status rr_pf_shard_step_p2p(rr_pf* h, const double control[2], const double* obs, size_t n_obs);
/* the same step with an eager gather of ALL slots into their owners' slabs and a separate tile-scan
 * launch: the plain statement of the protocol, kept for A/B measurement */
rr_status rr_pf_shard_step_p2p_unfused(rr_pf* h, const double control[2], const double* obs, size_t n_obs);
/* The mean try_step returns, for a sharded filter (systematic scheme; the peer-to-peer transport and the native RCCL
 * step).  want != 0: every rr_pf_shard_step_p2p / rr_pf_shard_step leaves THIS shard's part of it -- the sums of x, y, yaw, v over the sources of the shard's own output slots, added up by
 * the kernel that moves the particles (the next step's, or the accessor's gather when the value is read first; the deferred
 * form of rr_pf_step_async_estimate).  rr_pf_shard_last_estimate_sums returns the four sums and the denominator N: the mean
 * is the sum of every shard's sums divided by N (one all-reduce of four doubles whenever the caller wants the value).
 * Defined for steps whose resample fired (MonteCarloLocalizer: every step; particle_filter.rs:343). */
rr_status rr_pf_shard_want_estimate(rr_pf* h, int32_t want);
rr_status rr_pf_shard_last_estimate_sums(rr_pf* h, double out_sums[4], double* out_denom);
/* synchronises and reports whether any wait gave up (a peer did not answer within 2 s; every
 * later exchange of the filter then returns at once and the resample is skipped) */
rr_status rr_pf_p2p_status(rr_pf* h, int32_t* timed_out);
/* How this shard is wired (no synchronisation): out[0] = ranks of the filter, out[1] = how many of them live on this shard's
 * device (1 in the deployment; > 1 on a test rig), out[2] = CUs of the device this shard's stream is confined to (0: all of them;
 * > 0 with RR_P2P_CU_PARTITION=1 among sharers of one device: each then runs the lazy window step as on a device of its own),
 * out[3] = form of the last rr_pf_shard_step_p2p: 1 the lazy window step (k_step_lazy<kSrcWindow> | plan | k_push_window -- the
 * deployment path), 2 the eager step (sharers of one device whose step kernels could fill it), 0 none yet. */
rr_status rr_pf_p2p_topology(rr_pf* h, int32_t out[4]);

/* ---- native sharded step: the phases above driven from inside the library with RCCL called
 * directly (librccl is dlopen'ed on first use, so single-GPU users never load it).  One
 * communicator per process/GPU; the 128-byte unique id is created on one rank and handed to the
 * others by whatever launcher the caller has (bench.py: torch.distributed/gloo broadcast). */
typedef struct rr_comm rr_comm;
#define RR_COMM_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId */
rr_status rr_comm_unique_id(uint8_t out[RR_COMM_UNIQUE_ID_BYTES]);
/* ncclCommInitRank on `device` */
rr_status rr_comm_create(const uint8_t id[RR_COMM_UNIQUE_ID_BYTES], int32_t rank, int32_t n_ranks, int32_t device,
                         rr_comm** out);
void rr_comm_destroy(rr_comm* c);
/* Bring-up / test seam: the same sharded step (rr_pf_shard_step, systematic) for n_ranks shards that all live in THIS
 * process, with the three exchanges done as plain device copies instead of RCCL calls -- every kernel and all of the host's
 * segment arithmetic are those of the RCCL transport, so a one-GPU box can check them for 2 and 3 shards (RCCL refuses two
 * ranks on one device).  rr_comm_create_local makes the per-rank scratch without a communicator. */
rr_status rr_comm_create_local(int32_t rank, int32_t n_ranks, int32_t device, rr_comm** out);
rr_status rr_pf_shard_step_local(rr_pf* const* handles, rr_comm* const* comms, int32_t n_ranks, const double control[2],
                                 const double* obs, size_t n_obs);
/* One whole sharded MCL/PF step (systematic resampling) of this rank: phases A-E with
 * all-reduce(MAX) / all-gather / grouped send-recv in between, everything enqueued on the
 * filter's stream; the only host wait is the D2H of the G totals that sizes the segments.
 * Every rank of the communicator must call it with the same control and observations. */
rr_status rr_pf_shard_step(rr_pf* h, rr_comm* c, const double control[2], const double* obs, size_t n_obs);
/* global weighted mean (4) and covariance (16, row-major) over all shards (particle_filter.rs:382-413);
 * collective: every rank must call it */
rr_status rr_pf_shard_estimate(rr_pf* h, rr_comm* c, double est[4], double cov[16]);
/* particles that crossed a shard boundary in the last resample (sum of off-diagonal segment sizes) */
uint64_t rr_pf_shard_last_migrated(const rr_pf* h);

#ifdef __cplusplus
}
#endif
#endif /* RR_PF_H */
