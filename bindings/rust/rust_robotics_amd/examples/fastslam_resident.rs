//! FastSLAM 1.0 with the particles and their maps resident on the GPU: the call pattern of the reference's SLAM example
//! (crates/rust_robotics/examples/render_gif_slam.rs:166-200 -- `create_particles`, then `fastslam_update` + `get_best_particle`
//! per step) in its two ports, on a scene of this example's own (eight landmarks on a ring, noise-free range / bearing inside
//! the reference's 20 m gate, fastslam1.rs:277-299):
//!   1. the one-line port: the caller keeps its `Vec<Particle>`; every update uploads it, steps, downloads it
//!      (`fastslam::Engine::update` = rr_fs1_update_host) -- correct, and at 100 particles slower than the CPU;
//!   2. the intended port: `fastslam::FastSlam1` keeps everything on the device; with `set_resident` an update launches nothing
//!      and is answered with the best particle (18 us per iteration at 100 x 8; the reference on one host core: 40 us).
//!
//! NOT COMPILED in the image this repository is built in (no Rust toolchain); examples/fastslam_demo.py is the tested twin.
//!
//!     RUST_ROBOTICS_AMD_LIB_DIR=$PWD/rust_robotics_amd cargo run --release --example fastslam_resident \
//!         --manifest-path bindings/rust/rust_robotics_amd/Cargo.toml

use nalgebra::Vector2;
use rust_robotics_amd::fastslam::{create_particles, get_best_particle, Engine, FastSlam1};
use rust_robotics_core::RoboticsResult;

const N_PARTICLE: usize = 100; // fastslam1.rs:17
const DT: f64 = 0.1; // fastslam1.rs:13
const MAX_RANGE: f64 = 20.0; // fastslam1.rs:14

/// (range, bearing, landmark id) of every landmark within MAX_RANGE of the pose, without noise
fn observe(pose: (f64, f64, f64), landmarks: &[(f64, f64)]) -> Vec<(f64, f64, usize)> {
    let wrap = |a: f64| (a + std::f64::consts::PI).rem_euclid(2.0 * std::f64::consts::PI) - std::f64::consts::PI;
    landmarks
        .iter()
        .enumerate()
        .filter_map(|(id, &(lx, ly))| {
            let (dx, dy) = (lx - pose.0, ly - pose.1);
            let d = (dx * dx + dy * dy).sqrt();
            (d <= MAX_RANGE).then(|| (d, wrap(dy.atan2(dx) - pose.2), id))
        })
        .collect()
}

fn main() -> RoboticsResult<()> {
    let landmarks: Vec<(f64, f64)> = (0..8)
        .map(|k| {
            let a = std::f64::consts::FRAC_PI_4 * k as f64;
            (12.0 * a.cos(), 5.0 + 12.0 * a.sin())
        })
        .collect();
    let u = Vector2::new(1.0, 0.1);

    // ---- 1. caller-owned particles (the reference's signature, one handle next to the vector)
    let mut particles = create_particles(N_PARTICLE, landmarks.len());
    let mut engine = Engine::fastslam1(N_PARTICLE, landmarks.len(), 0)?;
    let mut pose = (0.0f64, 0.0f64, 0.0f64);
    for _ in 0..50 {
        pose = (pose.0 + u[0] * DT * pose.2.cos(), pose.1 + u[0] * DT * pose.2.sin(), pose.2 + u[1] * DT);
        engine.update(&mut particles, u, &observe(pose, &landmarks))?;
    }
    let best = get_best_particle(&particles);
    println!("caller-owned vector: best particle at ({:.3}, {:.3}), truth ({:.3}, {:.3})", best.x, best.y, pose.0, pose.1);

    // ---- 2. device-resident particles
    let mut slam = FastSlam1::new(N_PARTICLE, landmarks.len(), 0)?;
    slam.set_resident(20_000.0)?;
    let mut pose = (0.0f64, 0.0f64, 0.0f64);
    for step in 0..300 {
        pose = (pose.0 + u[0] * DT * pose.2.cos(), pose.1 + u[0] * DT * pose.2.sin(), pose.2 + u[1] * DT);
        slam.update(u, &observe(pose, &landmarks))?;
        let best = slam.best_particle()?; // answered by the update itself while the resident service is on
        if step % 100 == 99 {
            println!("step {:3}: best particle #{} at ({:.3}, {:.3}, {:.3}), truth ({:.3}, {:.3}, {:.3})", step + 1, best.index, best.x, best.y, best.yaw, pose.0, pose.1, pose.2);
        }
    }
    let best = slam.best_particle()?;
    for (id, (lm, truth)) in slam.landmarks_of(best.index)?.iter().zip(&landmarks).enumerate() {
        // (the reference never shrinks a landmark's covariance on its first observation -- fastslam1.rs:143-149 -- so under its own
        // settings the map stays where the first observation put it; INTEGRATION.md "FastSLAM 1.0" has the switch that changes that)
        println!("landmark {}: map ({:7.3}, {:7.3})  truth ({:7.3}, {:7.3})  cov[0][0] {:.1}", id, lm.x, lm.y, truth.0, truth.1, lm.cov[(0, 0)]);
    }
    Ok(())
}
