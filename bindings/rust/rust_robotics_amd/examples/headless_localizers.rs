//! crates/rust_robotics/examples/headless_localizers.rs with its particle filter taken from the GPU crate: the one
//! edited line is the `use` of `ParticleFilterLocalizer` (source only -- no Rust toolchain in the build image; the C++
//! and Python twins of this file, examples/cpp/headless_localizers.cpp and examples/headless_localizers.py, are the
//! ones the test-suite compiles and runs).
use rust_robotics::localization::{PFMeasurement, ParticleFilterConfig};
use rust_robotics::prelude::*;
use rust_robotics_amd::ParticleFilterLocalizer; // was: rust_robotics::localization::ParticleFilterLocalizer

fn propagate_state(state: &mut State2D, control: ControlInput, dt: f64) {
    state.x += control.v * state.yaw.cos() * dt;
    state.y += control.v * state.yaw.sin() * dt;
    state.yaw += control.omega * dt;
    state.v = control.v;
}

fn build_pf_measurements(state: &State2D, landmarks: &Obstacles) -> PFMeasurement {
    landmarks
        .points
        .iter()
        .map(|landmark| {
            let dx = state.x - landmark.x;
            let dy = state.y - landmark.y;
            ((dx * dx + dy * dy).sqrt(), landmark.x, landmark.y)
        })
        .collect()
}

fn main() -> RoboticsResult<()> {
    let landmarks = Obstacles::from_points(vec![Point2D::new(5.0, 0.0), Point2D::new(0.0, 5.0), Point2D::new(5.0, 5.0)]);
    let mut true_state = State2D::origin();
    let control = ControlInput::new(1.0, 0.1);
    let mut pf = ParticleFilterLocalizer::with_initial_state_2d(State2D::origin(), ParticleFilterConfig::default())?;
    pf.set_landmarks_from_obstacles(&landmarks)?;
    for step in 0..40 {
        propagate_state(&mut true_state, control, 0.1);
        let pf_measurement = build_pf_measurements(&true_state, &landmarks);
        let pf_state = pf.try_step_state(control, &pf_measurement)?;
        if step % 10 == 0 {
            println!("step={step:02} true=({:.2}, {:.2}) pf=({:.2}, {:.2})", true_state.x, true_state.y, pf_state.x, pf_state.y);
        }
    }
    println!("final true=({:.2}, {:.2}) pf=({:.2}, {:.2})", true_state.x, true_state.y, pf.state_2d().x, pf.state_2d().y);
    Ok(())
}
