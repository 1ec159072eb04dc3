//! The size of BASELINE.json configs[1] from Rust: a fixed-N Monte Carlo localizer (`min_particles == max_particles`,
//! monte_carlo_localization.rs:136-300) with 10^6 particles and 32 landmarks, `try_step` (:291-300) every step, timed.  The
//! wrapper keeps the reference's own resampler -- multinomial draws at every step (:322-365) -- so this is the
//! `mcl_multinomial` leg of `bench.py`, synchronous form: about 116 us per step on one MI355X (2.8e11 particle-landmark updates
//! per second; asynchronous 76-80 us).  The headline of `bench.py` (46-47 us) is the same engine with the systematic scheme of
//! fastslam1.rs:205-234, chosen through `rr_pf_options.resample_scheme`.
//!
//! NOT COMPILED in the image this repository is built in (no Rust toolchain).
//!
//!     RUST_ROBOTICS_AMD_LIB_DIR=$PWD/rust_robotics_amd cargo run --release --example mcl_million \
//!         --manifest-path bindings/rust/rust_robotics_amd/Cargo.toml

use rust_robotics_amd::{MonteCarloLocalizationConfig, MonteCarloLocalizer, PFControl, PFMeasurement, PFState};
use rust_robotics_core::RoboticsResult;
use std::time::Instant;

fn main() -> RoboticsResult<()> {
    let (n, n_landmarks, steps) = (1_000_000usize, 32usize, 200usize);
    // landmarks on a ring of radius 18 m around the origin (any fixed set does; the bench draws a seeded grid)
    let landmarks: Vec<(f64, f64)> = (0..n_landmarks)
        .map(|k| {
            let a = 2.0 * std::f64::consts::PI * (k as f64) / (n_landmarks as f64);
            (18.0 * a.cos(), 18.0 * a.sin())
        })
        .collect();
    let config = MonteCarloLocalizationConfig { min_particles: n, max_particles: n, ..Default::default() };
    let mut mcl = MonteCarloLocalizer::try_with_initial_state(PFState::new(0.0, 0.0, 0.0, 1.0), config)?;
    let control = PFControl::new(1.0, 0.1);
    let (mut x, mut y, mut yaw) = (0.0f64, 0.0f64, 0.0f64);
    let observe = |x: f64, y: f64| -> PFMeasurement {
        landmarks.iter().map(|&(lx, ly)| (((x - lx).powi(2) + (y - ly).powi(2)).sqrt(), lx, ly)).collect()
    };
    // warm-up: 20 synchronous steps (each returns its mean)
    for _ in 0..20 {
        x += 0.1 * yaw.cos();
        y += 0.1 * yaw.sin();
        yaw += 0.01;
        mcl.try_step(&control, &observe(x, y))?;
    }
    let t0 = Instant::now();
    let mut last = PFState::zeros();
    for _ in 0..steps {
        x += 0.1 * yaw.cos();
        y += 0.1 * yaw.sin();
        yaw += 0.01;
        last = mcl.try_step(&control, &observe(x, y))?;
    }
    let per_step = t0.elapsed().as_secs_f64() / steps as f64;
    println!(
        "{} particles x {} landmarks: {:.1} us per synchronous step, {:.3e} particle-landmark updates/s; estimate ({:.3}, {:.3}), truth ({:.3}, {:.3})",
        mcl.particle_count(),
        n_landmarks,
        per_step * 1e6,
        (n * n_landmarks) as f64 / per_step,
        last[0],
        last[1],
        x,
        y
    );
    Ok(())
}
