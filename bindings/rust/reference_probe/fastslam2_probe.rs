// Compiled INSIDE rust_robotics_slam::fastslam2 (run_probe.sh appends
//     #[cfg(test)] #[path = ".../fastslam2_probe.rs"] mod reference_probe;
// to a scratch copy of crates/rust_robotics_slam/src/fastslam2.rs): a child module sees its parent's private items, which is the only
// way to reach `fastslam2_update_with_rng` (:331) and `get_observations_with_rng` (:392).
//
// Re-runs the reference's two seeded tests with their own seeds, trajectories and particle counts --
//     test_fastslam2_update_does_not_panic   fastslam2.rs:443-456   StdRng::seed_from_u64(7),  20 particles x 3 landmarks x 5 steps
//     test_landmark_convergence              fastslam2.rs:491-545   StdRng::seed_from_u64(17), 120 particles x 1 landmark x 60 steps
// -- and dumps, per step, the observations the simulator drew and whether the update resampled (read off the generator: the update
// itself is not instrumented), and the FINAL particle set, as JSON with every f64 as bits and as decimal.
// Output directory: $RR_PROBE_OUT (default: the system temp directory).
// Compared with tests/fs2_replay.py (literal restatement on the CPU, and the GPU) by tools/compare_reference_dump.py.
use super::*;
use rand::{rngs::StdRng, RngCore, SeedableRng};
use std::fmt::Write as _;

fn f(x: f64) -> String {
    format!("{{\"bits\": \"{:016x}\", \"value\": {:e}}}", x.to_bits(), x)
}

fn particles_json(particles: &[Particle]) -> String {
    let mut s = String::from("[");
    for (i, p) in particles.iter().enumerate() {
        if i > 0 {
            s.push_str(", ");
        }
        write!(s, "{{\"weight\": {}, \"x\": {}, \"y\": {}, \"yaw\": {}, \"landmarks\": [", f(p.weight), f(p.x), f(p.y), f(p.yaw)).unwrap();
        for (l, lm) in p.landmarks.iter().enumerate() {
            if l > 0 {
                s.push_str(", ");
            }
            // nalgebra Matrix2 is column-major: (0,0), (1,0), (0,1), (1,1) -- the order of include/rr_fastslam1.h's rr_fs1_landmark
            write!(
                s,
                "{{\"x\": {}, \"y\": {}, \"c00\": {}, \"c10\": {}, \"c01\": {}, \"c11\": {}}}",
                f(lm.x), f(lm.y), f(lm.cov[(0, 0)]), f(lm.cov[(1, 0)]), f(lm.cov[(0, 1)]), f(lm.cov[(1, 1)])
            )
            .unwrap();
        }
        s.push_str("]}");
    }
    s.push(']');
    s
}

fn observations_json(z: &[(f64, f64, usize)]) -> String {
    let mut s = String::from("[");
    for (i, &(d, a, id)) in z.iter().enumerate() {
        if i > 0 {
            s.push_str(", ");
        }
        write!(s, "{{\"d\": {}, \"angle\": {}, \"id\": {}}}", f(d), f(a), id).unwrap();
    }
    s.push(']');
    s
}

fn out_path(name: &str) -> std::path::PathBuf {
    let dir = std::env::var("RR_PROBE_OUT").map(std::path::PathBuf::from).unwrap_or_else(|_| std::env::temp_dir());
    std::fs::create_dir_all(&dir).expect("cannot create RR_PROBE_OUT");
    dir.join(name)
}

/// one run: (per-step JSON rows, final particles)
fn run(seed: u64, n_particles: usize, landmarks: &[(f64, f64)], mut x_true: Vector3<f64>, u: Vector2<f64>, steps: usize,
       truth_moves: bool) -> (String, Vec<Particle>) {
    let mut particles = create_particles(n_particles, landmarks.len());
    let mut rng = StdRng::seed_from_u64(seed);
    let mut rows = String::from("[");
    for t in 0..steps {
        if truth_moves {
            x_true = motion_model(x_true, u);
        }
        let z = get_observations_with_rng(&x_true, landmarks, &mut rng);
        // The gate's decision (:368-371) cannot be read from outside the update, and "every weight == 1/n afterwards" is ambiguous
        // (uniform weights normalise to 1/n without a resample).  What is unambiguous is the generator: the update draws three
        // normals per particle (sample_pose_with_rng, :236; two without an observation, :350-354) and then, ONLY IF IT RESAMPLES,
        // one Uniform (:310-311).  Replaying the normals on a copy of the generator taken before the update shows whether the
        // update went one draw further.
        let mut shadow = rng.clone();
        fastslam2_update_with_rng(&mut particles, u, &z, &mut rng);
        let normal = Normal::new(0.0, 1.0).unwrap();
        let per_particle = if z.is_empty() { 2 } else { 3 };
        for _ in 0..(n_particles * per_particle) {
            let _: f64 = normal.sample(&mut shadow);
        }
        let resampled = shadow.next_u64() != rng.clone().next_u64();
        let inv_n = 1.0 / n_particles as f64;
        let all_inv_n = particles.iter().all(|p| p.weight == inv_n);
        if t > 0 {
            rows.push_str(", ");
        }
        write!(rows, "{{\"step\": {}, \"x_true\": [{}, {}, {}], \"z\": {}, \"resampled\": {}, \"all_weights_are_inv_n\": {}, \"neff_after\": {}}}",
               t, f(x_true[0]), f(x_true[1]), f(x_true[2]), observations_json(&z), resampled, all_inv_n, f(compute_neff(&particles))).unwrap();
    }
    rows.push(']');
    (rows, particles)
}

#[test]
fn reference_probe_seed7_update_does_not_panic() {
    // fastslam2.rs:443-456, verbatim inputs
    let landmarks = vec![(10.0, 0.0), (0.0, 10.0), (10.0, 10.0)];
    let (rows, particles) = run(7, 20, &landmarks, Vector3::new(0.0, 0.0, 0.0), Vector2::new(1.0, 0.1), 5, false);
    assert_eq!(particles.len(), 20);
    let json = format!(
        "{{\"probe\": \"fastslam2\", \"test\": \"test_fastslam2_update_does_not_panic\", \"lines\": \"fastslam2.rs:443-456\", \"seed\": 7, \
         \"n_particles\": 20, \"n_landmarks\": 3, \"steps\": 5, \"truth_moves\": false, \"per_step\": {}, \"final_particles\": {}}}\n",
        rows, particles_json(&particles));
    let path = out_path("fastslam2_seed7.json");
    std::fs::write(&path, json).expect("cannot write the dump");
    println!("wrote {}", path.display());
}

#[test]
fn reference_probe_seed17_landmark_convergence() {
    // fastslam2.rs:491-545, verbatim inputs and the test's own closing arithmetic
    let landmarks = vec![(5.0, 5.0)];
    let (rows, particles) = run(17, 120, &landmarks, Vector3::new(0.0, 0.0, PI / 4.0), Vector2::new(0.5, 0.0), 60, true);
    let initialized: Vec<&Particle> = particles.iter().filter(|p| p.landmarks[0].is_initialized()).collect();
    assert!(!initialized.is_empty(), "at least one particle should initialize the landmark");
    let total_weight: f64 = initialized.iter().map(|p| p.weight).sum();
    let (mean_x, mean_y) = if total_weight > 0.0 {
        (initialized.iter().map(|p| p.weight * p.landmarks[0].x).sum::<f64>() / total_weight,
         initialized.iter().map(|p| p.weight * p.landmarks[0].y).sum::<f64>() / total_weight)
    } else {
        (initialized.iter().map(|p| p.landmarks[0].x).sum::<f64>() / initialized.len() as f64,
         initialized.iter().map(|p| p.landmarks[0].y).sum::<f64>() / initialized.len() as f64)
    };
    let lm_err = ((mean_x - 5.0).powi(2) + (mean_y - 5.0).powi(2)).sqrt();
    assert!(lm_err < 6.0, "landmark estimate should converge: err={lm_err}");
    let json = format!(
        "{{\"probe\": \"fastslam2\", \"test\": \"test_landmark_convergence\", \"lines\": \"fastslam2.rs:491-545\", \"seed\": 17, \
         \"n_particles\": 120, \"n_landmarks\": 1, \"steps\": 60, \"truth_moves\": true, \"lm_err\": {}, \"per_step\": {}, \"final_particles\": {}}}\n",
        f(lm_err), rows, particles_json(&particles));
    let path = out_path("fastslam2_seed17.json");
    std::fs::write(&path, json).expect("cannot write the dump");
    println!("wrote {}", path.display());
}
