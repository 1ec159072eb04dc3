//! The particle-filter half of the reference's headless example on the GPU back end: the call pattern of
//! crates/rust_robotics/examples/headless_localizers.rs:39-56 -- `ParticleFilterConfig::default()` (100 particles), landmarks
//! handed over as `Obstacles`, one `try_step_state` per step -- with a scene of this example's own (a robot on the
//! u = (1.0 m/s, 0.1 rad/s) circle, noise-free ranges to three landmarks).  The same loop also runs through the
//! `StateEstimator` trait (rust_robotics_core/src/traits.rs:31-52), which is how generic code in the reference drives a filter.
//!
//! NOT COMPILED in the image this repository is built in (no Rust toolchain); the Python and C++ twins of this program,
//! examples/headless_localizers.py and examples/cpp/headless_localizers.cpp, are compiled and run by the GPU test suite.
//!
//!     make -C rust_robotics_amd/csrc
//!     RUST_ROBOTICS_AMD_LIB_DIR=$PWD/rust_robotics_amd cargo run --release --example headless_pf \
//!         --manifest-path bindings/rust/rust_robotics_amd/Cargo.toml

use rust_robotics_amd::{PFControl, PFMeasurement, ParticleFilterConfig, ParticleFilterLocalizer};
use rust_robotics_core::{ControlInput, Obstacles, Point2D, RoboticsResult, State2D, StateEstimator};

const DT: f64 = 0.1; // ParticleFilterConfig::default().dt

fn advance(truth: State2D, control: ControlInput) -> State2D {
    State2D::new(
        truth.x + control.v * truth.yaw.cos() * DT,
        truth.y + control.v * truth.yaw.sin() * DT,
        truth.yaw + control.omega * DT,
        control.v,
    )
}

fn ranges(truth: &State2D, landmarks: &[Point2D]) -> PFMeasurement {
    landmarks
        .iter()
        .map(|l| (((truth.x - l.x).powi(2) + (truth.y - l.y).powi(2)).sqrt(), l.x, l.y))
        .collect()
}

fn main() -> RoboticsResult<()> {
    let landmarks = vec![Point2D::new(10.0, 0.0), Point2D::new(0.0, 15.0), Point2D::new(-5.0, 20.0)];
    let control = ControlInput::new(1.0, 0.1);

    // ---- the reference's loop: create, set landmarks, try_step_state every step
    let mut pf = ParticleFilterLocalizer::try_new(ParticleFilterConfig::default())?;
    pf.set_landmarks_from_obstacles(&Obstacles::from_points(landmarks.clone()))?;
    pf.warm(0.0)?; // optional: the first steps run at the rate of the thousandth
    pf.set_resident(20_000.0)?; // optional: the step kernel stays on the device between steps (7 us per call instead of 13)
    let mut truth = State2D::new(0.0, 0.0, 0.0, 1.0);
    for step in 0..200 {
        truth = advance(truth, control);
        let est = pf.try_step_state(control, &ranges(&truth, &landmarks))?;
        if step % 40 == 39 {
            let err = ((est.x - truth.x).powi(2) + (est.y - truth.y).powi(2)).sqrt();
            println!("step {:3}: truth ({:6.3}, {:6.3})  estimate ({:6.3}, {:6.3})  error {:.3} m", step + 1, truth.x, truth.y, est.x, est.y, err);
        }
    }
    let cov = pf.try_calc_covariance()?;
    println!("covariance diagonal: {:.4} {:.4} {:.4} {:.4}", cov[(0, 0)], cov[(1, 1)], cov[(2, 2)], cov[(3, 3)]);
    let n = pf.particle_count();
    let heaviest = pf.get_particles().iter().copied().max_by(|a, b| a.w.total_cmp(&b.w)); // (one N x 40 B read-back)
    println!("{} particles, the heaviest of them: {:?}", n, heaviest);

    // ---- the same filter behind the trait object generic code holds
    fn drive<E: StateEstimator<State = rust_robotics_amd::PFState, Measurement = PFMeasurement, Control = PFControl>>(
        e: &mut E,
        landmarks: &[Point2D],
    ) {
        let mut truth = State2D::new(0.0, 0.0, 0.0, 1.0);
        let control = ControlInput::new(1.0, 0.1);
        for _ in 0..50 {
            truth = advance(truth, control);
            e.predict(&PFControl::new(control.v, control.omega), DT); // (dt is ignored, particle_filter.rs:557-559)
            e.update(&ranges(&truth, landmarks));
        }
        let s = e.get_state();
        println!("through StateEstimator: estimate ({:.3}, {:.3}), truth ({:.3}, {:.3})", s[0], s[1], truth.x, truth.y);
    }
    let mut generic = ParticleFilterLocalizer::with_defaults();
    drive(&mut generic, &landmarks);
    Ok(())
}
