//! Safe wrappers over `rust_robotics_amd-sys` with the method names, argument meaning and error
//! behaviour of the reference types they stand in for:
//!
//! * `ParticleFilterLocalizer`  -- rust_robotics_localization/src/particle_filter.rs:121-573
//! * `MonteCarloLocalizer`      -- rust_robotics_localization/src/monte_carlo_localization.rs:136-462
//!   (fixed particle count when `min_particles == max_particles`, KLD-adaptive otherwise)
//! * `fastslam1::*`, `fastslam2::*` -- rust_robotics_slam/src/fastslam{1,2}.rs free functions
//!
//! NOT COMPILED in the image this repository is built in (it has no Rust toolchain): the file is the
//! binding a maintainer of the reference would add, written against the generated `-sys` crate.
//! The particle set lives on the GPU; `estimate()` / `get_covariance()` refresh a host-side cache
//! (one 300-byte D2H) because the reference's accessors take `&self`.

use nalgebra::{DMatrix, Matrix4, Vector2, Vector4};
use rust_robotics_amd_sys as sys;
use rust_robotics_core::{ControlInput, Obstacles, Point2D, RoboticsError, RoboticsResult, State2D, StateEstimator};
use std::cell::OnceCell;
use std::ffi::CStr;

/// Particle, particle_filter.rs:25-32 (same fields, same order: `rr_pf_get_particles` fills a slice of these directly)
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct Particle {
    pub x: f64,
    pub y: f64,
    pub yaw: f64,
    pub v: f64,
    pub w: f64,
}

pub type PFState = Vector4<f64>;
pub type PFControl = Vector2<f64>;
pub type PFMeasurement = Vec<(f64, f64, f64)>; // (distance, landmark_x, landmark_y)

fn check(status: sys::rr_status) -> RoboticsResult<()> {
    if status == sys::RR_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::rr_last_error()) }.to_string_lossy().into_owned();
    if status == sys::RR_INVALID_PARAMETER {
        Err(RoboticsError::InvalidParameter(msg))
    } else {
        Err(RoboticsError::EstimationError(msg)) // device / runtime failure (rust_robotics_core/src/error.rs:11)
    }
}

/// ParticleFilterConfig, particle_filter.rs:51-78 (field for field)
#[derive(Debug, Clone)]
pub struct ParticleFilterConfig {
    pub n_particles: usize,
    pub resample_threshold: f64,
    pub range_noise: f64,
    pub velocity_noise: f64,
    pub yaw_rate_noise: f64,
    pub dt: f64,
}

impl Default for ParticleFilterConfig {
    fn default() -> Self {
        let mut c = sys::rr_pf_config { n_particles: 0, resample_threshold: 0.0, range_noise: 0.0, velocity_noise: 0.0, yaw_rate_noise: 0.0, dt: 0.0 };
        unsafe { sys::rr_pf_config_default(&mut c) };
        Self { n_particles: c.n_particles as usize, resample_threshold: c.resample_threshold, range_noise: c.range_noise,
               velocity_noise: c.velocity_noise, yaw_rate_noise: c.yaw_rate_noise, dt: c.dt }
    }
}

impl ParticleFilterConfig {
    fn raw(&self) -> sys::rr_pf_config {
        sys::rr_pf_config { n_particles: self.n_particles as u64, resample_threshold: self.resample_threshold, range_noise: self.range_noise,
                            velocity_noise: self.velocity_noise, yaw_rate_noise: self.yaw_rate_noise, dt: self.dt }
    }
    /// particle_filter.rs:81-117 (same messages)
    pub fn validate(&self) -> RoboticsResult<()> {
        check(unsafe { sys::rr_pf_config_validate(&self.raw()) })
    }
}

/// The handle plus the host-side cache the `&self` accessors of the reference need.
pub struct ParticleFilterLocalizer {
    h: *mut sys::rr_pf,
    state_estimate: PFState,
    /// The covariance of the current particle set, computed on the device the first time somebody asks after a change
    /// (`calc_covariance`, `StateEstimator::get_covariance`, both through `&self`) and dropped by everything that moves or
    /// reweights the particles.  The reference recomputes it inside every step (particle_filter.rs:299,332,343); a caller
    /// ported from it therefore never reads the covariance of an earlier step, and a caller that never asks never pays.
    /// `None` inside the cell: the device call failed (a getter must not panic on a device error) --
    /// `get_covariance` then returns `None`, `calc_covariance` a matrix of NaN, `try_calc_covariance` the error itself.
    /// Both getters MAY SYNCHRONISE: the first call after a change waits for the filter's stream and runs two small kernels.
    covariance: OnceCell<Option<(Matrix4<f64>, DMatrix<f64>)>>,
    particles: Vec<Particle>,   // host mirror behind get_particles (filled on demand)
    landmarks: Vec<Point2D>,    // set_landmarks* only stores them, as the reference does (particle_filter.rs:216-220, Q19)
}

unsafe impl Send for ParticleFilterLocalizer {}

impl Drop for ParticleFilterLocalizer {
    fn drop(&mut self) {
        unsafe { sys::rr_pf_destroy(self.h) }
    }
}

fn flatten(obs: &PFMeasurement) -> Vec<f64> {
    obs.iter().flat_map(|&(d, x, y)| [d, x, y]).collect()
}

impl ParticleFilterLocalizer {
    fn options(mcl: bool, seed: u64, device: i32) -> sys::rr_pf_options {
        let mut o: sys::rr_pf_options = unsafe { std::mem::zeroed() };
        unsafe {
            if mcl { sys::rr_pf_options_mcl(&mut o) } else { sys::rr_pf_options_default(&mut o) }
        }
        o.seed = seed;
        o.device = device;
        o
    }

    fn from_handle(h: *mut sys::rr_pf) -> RoboticsResult<Self> {
        let mut s = Self { h, state_estimate: PFState::zeros(), covariance: OnceCell::new(),
                           particles: Vec::new(), landmarks: Vec::new() };
        s.refresh_cache()?;
        Ok(s)
    }

    /// try_new, particle_filter.rs:139-156
    pub fn try_new(config: ParticleFilterConfig) -> RoboticsResult<Self> {
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::rr_pf_create(&config.raw(), &Self::options(false, 0, 0), &mut h) })?;
        Self::from_handle(h)
    }

    /// new, particle_filter.rs:131-136: panics on an invalid configuration like the reference
    pub fn new(config: ParticleFilterConfig) -> Self {
        Self::try_new(config).expect("invalid particle filter configuration")
    }

    /// try_with_initial_state, particle_filter.rs:170-199
    pub fn try_with_initial_state(initial_state: PFState, config: ParticleFilterConfig) -> RoboticsResult<Self> {
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::rr_pf_create_with_state(&config.raw(), &Self::options(false, 0, 0), initial_state.as_ptr(), &mut h) })?;
        Self::from_handle(h)
    }

    /// with_defaults, particle_filter.rs:158-161
    pub fn with_defaults() -> Self {
        Self::new(ParticleFilterConfig::default())
    }

    /// with_initial_state_2d, :202-207
    pub fn with_initial_state_2d(initial_state: State2D, config: ParticleFilterConfig) -> RoboticsResult<Self> {
        Self::try_with_initial_state(PFState::new(initial_state.x, initial_state.y, initial_state.yaw, initial_state.v), config)
    }

    /// try_set_landmarks, :216-220 (stored, never read by the update)
    pub fn try_set_landmarks(&mut self, landmarks: Vec<Point2D>) -> RoboticsResult<()> {
        let flat: Vec<f64> = landmarks.iter().flat_map(|p| [p.x, p.y]).collect();
        check(unsafe { sys::rr_pf_set_landmarks(self.h, flat.as_ptr(), landmarks.len()) })?;
        self.landmarks = landmarks;
        Ok(())
    }

    /// set_landmarks_from_obstacles, :223-225
    pub fn set_landmarks_from_obstacles(&mut self, obstacles: &Obstacles) -> RoboticsResult<()> {
        self.try_set_landmarks(obstacles.points.clone())
    }

    /// get_landmarks, :239-241
    pub fn get_landmarks(&self) -> &[Point2D] {
        &self.landmarks
    }

    /// get_particles, :244-246.  The particles live on the GPU: this call copies them into a host mirror (N x 40 B), which is
    /// why it takes `&mut self` here; a caller that draws the cloud every k-th frame pays for it every k-th frame only.
    pub fn get_particles(&mut self) -> &[Particle] {
        let n = self.particle_count();
        self.particles.resize(n, Particle { x: 0.0, y: 0.0, yaw: 0.0, v: 0.0, w: 0.0 });
        check(unsafe { sys::rr_pf_get_particles(self.h, self.particles.as_mut_ptr() as *mut f64) }).expect("particle read-back failed on the device");
        &self.particles
    }

    /// try_predict_input, :368-372
    pub fn try_predict_input(&mut self, control: ControlInput) -> RoboticsResult<()> {
        self.try_predict_with_control(&PFControl::new(control.v, control.omega))
    }

    /// try_step_state, :374-380 -- the call every caller in the reference makes (headless_localizers.rs:56,
    /// render_gif_particle_filter.rs:77-79).  ONE device round trip: the step's own kernel returns the mean; the covariance
    /// cache is refreshed only when somebody asks for it.
    pub fn try_step_state(&mut self, control: ControlInput, observations: &PFMeasurement) -> RoboticsResult<State2D> {
        let flat = flatten(observations);
        let u = [control.v, control.omega];
        let mut out = [0.0f64; 4];
        check(unsafe { sys::rr_pf_step(self.h, u.as_ptr(), flat.as_ptr(), observations.len(), out.as_mut_ptr()) })?;
        self.state_estimate = PFState::from_column_slice(&out);
        self.covariance.take();  // stale from here on: recomputed at the next calc_covariance / get_covariance
        Ok(State2D::new(out[0], out[1], out[2], out[3]))
    }

    /// state_2d, :353-360
    pub fn state_2d(&self) -> State2D {
        State2D::new(self.state_estimate[0], self.state_estimate[1], self.state_estimate[2], self.state_estimate[3])
    }

    /// Engine extension (rr_pf_set_resident): with `idle_us > 0` the steps of a filter of up to 2048 particles are served by ONE
    /// kernel that stays on the device between them -- a synchronous `try_step_state` then costs its arithmetic plus two trips
    /// over the host link (7 - 9 us at the reference's sizes) instead of a launch and a completion wait (13 - 18 us).
    pub fn set_resident(&mut self, idle_us: f64) -> RoboticsResult<()> {
        check(unsafe { sys::rr_pf_set_resident(self.h, idle_us) })
    }

    /// try_predict_with_control, :255-301
    pub fn try_predict_with_control(&mut self, control: &PFControl) -> RoboticsResult<()> {
        check(unsafe { sys::rr_pf_predict(self.h, control.as_ptr()) })?;
        self.refresh_cache()
    }

    /// try_update_with_observations, :310-334
    pub fn try_update_with_observations(&mut self, observations: &PFMeasurement) -> RoboticsResult<()> {
        let flat = flatten(observations);
        check(unsafe { sys::rr_pf_update(self.h, flat.as_ptr(), observations.len()) })?;
        self.refresh_cache()
    }

    /// resample, :337-345
    pub fn resample(&mut self) {
        check(unsafe { sys::rr_pf_resample(self.h) }).expect("particle filter resample failed on the device");
        self.refresh_cache().expect("particle filter moments failed on the device");
    }

    /// try_step, :488-497 -- one fused launch sequence on the GPU
    pub fn try_step(&mut self, control: &PFControl, observations: &PFMeasurement) -> RoboticsResult<PFState> {
        let flat = flatten(observations);
        let mut out = [0.0f64; 4];
        check(unsafe { sys::rr_pf_step(self.h, control.as_ptr(), flat.as_ptr(), observations.len(), out.as_mut_ptr()) })?;
        self.state_estimate = PFState::from_column_slice(&out);
        self.covariance.take();  // stale from here on: recomputed at the next calc_covariance / get_covariance
        Ok(self.state_estimate)
    }

    /// Compute the covariance now instead of at the first `calc_covariance` / `get_covariance` (two small kernels and a
    /// 128-byte copy); an error of the device surfaces here as a `RoboticsResult` instead of a panic there.
    pub fn refresh_covariance(&mut self) -> RoboticsResult<()> {
        self.covariance.take();
        let c = self.read_covariance()?;
        let _ = self.covariance.set(Some(c));
        Ok(())
    }

    /// Engine extension: enqueue one step without waiting for it; the mean `try_step` would return (:496) is
    /// produced on the device: inside the step's own plan kernel (systematic scheme) or by the kernel that gathers the drawn sources (multinomial).
    pub fn try_step_async(&mut self, control: &PFControl, observations: &PFMeasurement) -> RoboticsResult<()> {
        let flat = flatten(observations);
        self.covariance.take();
        check(unsafe { sys::rr_pf_step_async_estimate(self.h, control.as_ptr(), flat.as_ptr(), observations.len()) })
    }

    /// Engine extension: `controls.len()` steps in one call (`rr_pf_step_many`); `observations[k]` are the observations of step
    /// k (the same number every step).  Returns what `try_step` would have returned after each step.  Up to 2048 particles --
    /// every caller in the reference runs 100 - 1200 -- the whole batch is ONE kernel launch of one workgroup.
    pub fn try_step_many(&mut self, controls: &[PFControl], observations: &[PFMeasurement]) -> RoboticsResult<Vec<PFState>> {
        if controls.len() != observations.len() {
            return Err(RoboticsError::InvalidParameter("one observation list per control".into()));
        }
        let n_obs = observations.first().map_or(0, |o| o.len());
        if observations.iter().any(|o| o.len() != n_obs) {
            return Err(RoboticsError::InvalidParameter("every step of a batch needs the same number of observations".into()));
        }
        let u: Vec<f64> = controls.iter().flat_map(|c| [c[0], c[1]]).collect();
        let flat: Vec<f64> = observations.iter().flat_map(|o| flatten(o)).collect();
        let mut out = vec![0.0f64; 4 * controls.len()];
        check(unsafe { sys::rr_pf_step_many(self.h, u.as_ptr(), flat.as_ptr(), n_obs, controls.len(), out.as_mut_ptr()) })?;
        self.refresh_cache()?;
        Ok(out.chunks_exact(4).map(PFState::from_column_slice).collect())
    }

    /// Wait for the stream and read the mean of the last step enqueued with `try_step_async` (300-byte copy).
    pub fn last_step_estimate(&mut self) -> RoboticsResult<PFState> {
        let mut out = [0.0f64; 4];
        check(unsafe { sys::rr_pf_last_step_estimate(self.h, out.as_mut_ptr()) })?;
        self.state_estimate = PFState::from_column_slice(&out);
        Ok(self.state_estimate)
    }

    /// estimate, :348-350
    pub fn estimate(&self) -> PFState {
        self.state_estimate
    }

    /// Engine extension (`rr_pf_warm`): `ms` milliseconds (0.0: the default, 50) of step-shaped work on the filter's stream, so
    /// that the first `try_step` after construction runs at the rate of the thousandth (an idle MI355X starts at reduced clocks).
    pub fn warm(&mut self, ms: f64) -> RoboticsResult<()> {
        check(unsafe { sys::rr_pf_warm(self.h, ms) })
    }

    /// calc_covariance, :363-365.  May synchronise (first call after a change).  A device error gives a matrix of NaN here
    /// (the reference's signature has no error channel); `try_calc_covariance` reports it.
    pub fn calc_covariance(&self) -> Matrix4<f64> {
        match self.covariance_pair() {
            Some(p) => p.0,
            None => Matrix4::from_element(f64::NAN),
        }
    }

    /// calc_covariance with the device's status: `Err(RoboticsError::...)` instead of NaN when the moment kernels failed
    pub fn try_calc_covariance(&self) -> RoboticsResult<Matrix4<f64>> {
        match self.covariance_pair() {
            Some(p) => Ok(p.0),
            None => self.read_covariance().map(|p| p.0),  // (asks again: returns the device's error message)
        }
    }

    pub fn particle_count(&self) -> usize {
        unsafe { sys::rr_pf_particle_count(self.h) as usize }
    }

    /// set_range_noise, :228-236
    pub fn set_range_noise(&mut self, range_noise: f64) -> RoboticsResult<()> {
        check(unsafe { sys::rr_pf_set_range_noise(self.h, range_noise) })
    }

    /// the mean of the current set now (one small kernel); the covariance is left for whoever asks (covariance_pair)
    fn refresh_cache(&mut self) -> RoboticsResult<()> {
        let mut e = [0.0f64; 4];
        check(unsafe { sys::rr_pf_estimate(self.h, e.as_mut_ptr()) })?;
        self.state_estimate = PFState::from_column_slice(&e);
        self.covariance.take();
        Ok(())
    }

    fn read_covariance(&self) -> RoboticsResult<(Matrix4<f64>, DMatrix<f64>)> {
        let mut c = [0.0f64; 16];
        check(unsafe { sys::rr_pf_covariance(self.h, c.as_mut_ptr()) })?;
        Ok((Matrix4::from_row_slice(&c), DMatrix::from_row_slice(4, 4, &c)))
    }

    /// the cached covariance, computed on first use after a change (`OnceCell::get_or_init` hands out `&` through `&self`)
    fn covariance_pair(&self) -> Option<&(Matrix4<f64>, DMatrix<f64>)> {
        self.covariance.get_or_init(|| self.read_covariance().ok()).as_ref()
    }
}

/// StateEstimator, rust_robotics_core/src/traits.rs:31-52 (dt is ignored as in particle_filter.rs:557-559)
impl StateEstimator for ParticleFilterLocalizer {
    type State = PFState;
    type Measurement = PFMeasurement;
    type Control = PFControl;

    fn predict(&mut self, control: &PFControl, _dt: f64) {
        self.try_predict_with_control(control).expect("invalid particle filter prediction input")
    }
    fn update(&mut self, measurement: &PFMeasurement) {
        self.try_update_with_observations(measurement).expect("invalid particle filter observations");
        self.resample();
    }
    fn get_state(&self) -> &PFState {
        &self.state_estimate
    }
    /// May synchronise (first call after a change); `None` when the device call failed.
    fn get_covariance(&self) -> Option<&DMatrix<f64>> {
        self.covariance_pair().map(|p| &p.1)
    }
}

/// MonteCarloLocalizationConfig, monte_carlo_localization.rs:50-82
#[derive(Debug, Clone)]
pub struct MonteCarloLocalizationConfig {
    pub min_particles: usize,
    pub max_particles: usize,
    pub kld_epsilon: f64,
    pub kld_z: f64,
    pub range_noise: f64,
    pub velocity_noise: f64,
    pub yaw_rate_noise: f64,
    pub dt: f64,
}

impl Default for MonteCarloLocalizationConfig {
    fn default() -> Self {
        Self { min_particles: 100, max_particles: 5000, kld_epsilon: 0.05, kld_z: 2.326, range_noise: 0.2, velocity_noise: 2.0,
               yaw_rate_noise: 40.0_f64.to_radians(), dt: 0.1 }
    }
}

/// MonteCarloLocalizer: the same engine with multinomial resampling at every step; the particle
/// count follows the KLD bound when `min_particles < max_particles` (rr_pf_create_adaptive).
pub struct MonteCarloLocalizer(ParticleFilterLocalizer);

impl MonteCarloLocalizer {
    fn create(config: &MonteCarloLocalizationConfig, state: Option<&PFState>) -> RoboticsResult<Self> {
        let cfg = sys::rr_pf_config { n_particles: config.min_particles as u64, resample_threshold: 1.0, range_noise: config.range_noise,
                                      velocity_noise: config.velocity_noise, yaw_rate_noise: config.yaw_rate_noise, dt: config.dt };
        let opt = ParticleFilterLocalizer::options(true, 0, 0);
        let sp = state.map_or(std::ptr::null(), |s| s.as_ptr());
        let mut h = std::ptr::null_mut();
        if config.min_particles == config.max_particles {
            check(unsafe { if sp.is_null() { sys::rr_pf_create(&cfg, &opt, &mut h) } else { sys::rr_pf_create_with_state(&cfg, &opt, sp, &mut h) } })?;
        } else {
            let kld = sys::rr_mcl_adaptive { min_particles: config.min_particles as u64, max_particles: config.max_particles as u64,
                                             kld_epsilon: config.kld_epsilon, kld_z: config.kld_z };
            check(unsafe { sys::rr_pf_create_adaptive(&cfg, &opt, &kld, sp, &mut h) })?;
        }
        Ok(Self(ParticleFilterLocalizer::from_handle(h)?))
    }
    /// try_new, monte_carlo_localization.rs:142-156
    pub fn try_new(config: MonteCarloLocalizationConfig) -> RoboticsResult<Self> {
        Self::create(&config, None)
    }
    /// try_with_initial_state, :170-199
    pub fn try_with_initial_state(initial_state: PFState, config: MonteCarloLocalizationConfig) -> RoboticsResult<Self> {
        Self::create(&config, Some(&initial_state))
    }
    /// try_step, :291-300
    pub fn try_step(&mut self, control: &PFControl, observations: &PFMeasurement) -> RoboticsResult<PFState> {
        self.0.try_step(control, observations)
    }
    pub fn estimate(&self) -> PFState {
        self.0.estimate()
    }
    /// particle_count, :318-320
    pub fn particle_count(&self) -> usize {
        self.0.particle_count()
    }
    pub fn try_step_state(&mut self, control: ControlInput, observations: &PFMeasurement) -> RoboticsResult<State2D> {
        self.0.try_step_state(control, observations)
    }
    pub fn get_particles(&mut self) -> &[Particle] {
        self.0.get_particles()
    }
    /// rr_pf_set_resident: also for the KLD-adaptive count (the count then never visits the host between steps)
    pub fn set_resident(&mut self, idle_us: f64) -> RoboticsResult<()> {
        self.0.set_resident(idle_us)
    }
}

/// rust_robotics_slam::fastslam1 / fastslam2: the reference's free functions over a caller-owned
/// `Vec<Particle>` on top of the upload -> update -> download shim `rr_fs1_update_host`.
pub mod fastslam {
    use super::{check, sys};
    use nalgebra::{Matrix2, Vector2};
    use rust_robotics_core::RoboticsResult;

    #[derive(Clone)]
    pub struct Landmark { pub x: f64, pub y: f64, pub cov: Matrix2<f64> } // fastslam1.rs:26-31
    #[derive(Clone)]
    pub struct Particle { pub weight: f64, pub x: f64, pub y: f64, pub yaw: f64, pub landmarks: Vec<Landmark> } // :44-51

    /// create_particles, fastslam1.rs:302-306 / fastslam2.rs:425-429
    pub fn create_particles(n_particles: usize, n_landmarks: usize) -> Vec<Particle> {
        let lm = Landmark { x: 0.0, y: 0.0, cov: Matrix2::identity() * 1000.0 };
        vec![Particle { weight: 1.0 / 100.0, x: 0.0, y: 0.0, yaw: 0.0, landmarks: vec![lm; n_landmarks] }; n_particles]
    }

    /// One engine handle per (particle count, landmark count, algorithm); keep it next to the particle vector.
    pub struct Engine { h: *mut sys::rr_fs1, n: usize, l: usize }
    impl Drop for Engine { fn drop(&mut self) { unsafe { sys::rr_fs1_destroy(self.h) } } }

    impl Engine {
        pub fn fastslam1(n: usize, l: usize, seed: u64) -> RoboticsResult<Self> {
            let mut o: sys::rr_fs1_options = unsafe { std::mem::zeroed() };
            unsafe { sys::rr_fs1_options_default(&mut o) };
            o.seed = seed;
            let mut h = std::ptr::null_mut();
            check(unsafe { sys::rr_fs1_create(n as u64, l as u64, std::ptr::null(), &o, &mut h) })?;
            Ok(Self { h, n, l })
        }
        pub fn fastslam2(n: usize, l: usize, seed: u64) -> RoboticsResult<Self> {
            let mut o: sys::rr_fs1_options = unsafe { std::mem::zeroed() };
            unsafe { sys::rr_fs1_options_default(&mut o) };
            o.seed = seed;
            let mut h = std::ptr::null_mut();
            check(unsafe { sys::rr_fs2_create(n as u64, l as u64, std::ptr::null(), &o, &mut h) })?;
            Ok(Self { h, n, l })
        }
        /// fastslam_update (fastslam1.rs:237-266) / fastslam2_update (fastslam2.rs:376-383), by the handle's algorithm
        pub fn update(&mut self, particles: &mut Vec<Particle>, u: Vector2<f64>, z: &[(f64, f64, usize)]) -> RoboticsResult<()> {
            assert_eq!(particles.len(), self.n);
            let mut poses = Vec::with_capacity(4 * self.n);
            let mut maps = Vec::with_capacity(6 * self.n * self.l);
            for p in particles.iter() {
                poses.extend_from_slice(&[p.weight, p.x, p.y, p.yaw]);
                for lm in &p.landmarks { maps.extend_from_slice(&[lm.x, lm.y, lm.cov[(0, 0)], lm.cov[(1, 0)], lm.cov[(0, 1)], lm.cov[(1, 1)]]); }
            }
            let zf: Vec<f64> = z.iter().flat_map(|&(d, a, id)| [d, a, id as f64]).collect();
            check(unsafe { sys::rr_fs1_update_host(self.h, poses.as_mut_ptr(), maps.as_mut_ptr(), u.as_ptr(), zf.as_ptr(), z.len()) })?;
            for (i, p) in particles.iter_mut().enumerate() {
                p.weight = poses[4 * i]; p.x = poses[4 * i + 1]; p.y = poses[4 * i + 2]; p.yaw = poses[4 * i + 3];
                for (l, lm) in p.landmarks.iter_mut().enumerate() {
                    let e = &maps[(i * self.l + l) * 6..][..6];
                    lm.x = e[0]; lm.y = e[1]; lm.cov = Matrix2::new(e[2], e[4], e[3], e[5]);
                }
            }
            Ok(())
        }
    }

    /// The way to run FastSLAM on the GPU: the particle set and every particle's map stay on the device behind the handle;
    /// `update` is fastslam_update (fastslam1.rs:237-266), `best_particle` get_best_particle (:269-274).  With the resident
    /// service switched on (`set_resident`) an update launches nothing and is answered with the best particle, so the loop of
    /// render_gif_slam.rs:172-178 costs 18 us per iteration at 100 particles x 8 landmarks (46 us with launches; the reference on
    /// one host core: 40 us).
    pub struct FastSlam1(Engine);
    pub struct BestParticle { pub x: f64, pub y: f64, pub yaw: f64, pub weight: f64, pub index: usize }
    impl FastSlam1 {
        pub fn new(n_particles: usize, n_landmarks: usize, seed: u64) -> RoboticsResult<Self> {
            Ok(Self(Engine::fastslam1(n_particles, n_landmarks, seed)?))
        }
        pub fn set_resident(&mut self, idle_us: f64) -> RoboticsResult<()> {
            check(unsafe { sys::rr_fs1_set_resident(self.0.h, idle_us) })
        }
        pub fn update(&mut self, u: Vector2<f64>, z: &[(f64, f64, usize)]) -> RoboticsResult<()> {
            let zf: Vec<f64> = z.iter().flat_map(|&(d, a, id)| [d, a, id as f64]).collect();
            check(unsafe { sys::rr_fs1_update(self.0.h, u.as_ptr(), zf.as_ptr(), z.len()) })
        }
        pub fn best_particle(&mut self) -> RoboticsResult<BestParticle> {
            let (mut pose, mut w, mut i) = ([0.0f64; 3], 0.0f64, 0u64);
            check(unsafe { sys::rr_fs1_best_particle(self.0.h, pose.as_mut_ptr(), &mut w, &mut i) })?;
            Ok(BestParticle { x: pose[0], y: pose[1], yaw: pose[2], weight: w, index: i as usize })
        }
        /// the map of one particle: L x (x, y, c00, c10, c01, c11)
        pub fn landmarks_of(&mut self, particle: usize) -> RoboticsResult<Vec<Landmark>> {
            let mut raw = vec![0.0f64; 6 * self.0.l];
            check(unsafe { sys::rr_fs1_get_landmarks(self.0.h, particle as u64, raw.as_mut_ptr()) })?;
            Ok(raw.chunks_exact(6).map(|e| Landmark { x: e[0], y: e[1], cov: Matrix2::new(e[2], e[4], e[3], e[5]) }).collect())
        }
    }

    /// get_best_particle, fastslam1.rs:269-274 (ties -> last)
    pub fn get_best_particle(particles: &[Particle]) -> &Particle {
        particles.iter().max_by(|a, b| a.weight.partial_cmp(&b.weight).unwrap()).unwrap()
    }
}
