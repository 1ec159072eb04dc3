//! First 256 draws of `StdRng::seed_from_u64(seed)`, seed = 7 and 17 (rust_robotics_slam/src/fastslam2.rs:449, :496), of every kind the
//! reference's FastSLAM 2.0 path consumes: raw words, `random::<f64>()`, `StandardNormal` (what `Normal::new(0.0, 1.0)` samples,
//! fastslam2.rs:227,350,398), `Uniform::new(0.0, 1.0 / n)` (fastslam2.rs:310).  Each kind starts from a FRESH generator.
//! Output: rng_streams_seed<seed>.json in the directory given as argv[1] (default "."), every f64 as bits (hex) and as decimal.
//! Compared bit for bit with oracle/rand_rs.py by tools/compare_reference_dump.py.
use rand::rngs::StdRng;
use rand::{Rng, RngCore, SeedableRng};
use rand_distr::{Distribution, Normal, StandardNormal, Uniform}; // the reference's own imports (fastslam2.rs:15) + StandardNormal
use std::fmt::Write as _;
use std::fs;

const N: usize = 256;

fn f64_list(v: &[f64]) -> String {
    let mut s = String::from("[");
    for (i, x) in v.iter().enumerate() {
        if i > 0 {
            s.push_str(", ");
        }
        write!(s, "{{\"bits\": \"{:016x}\", \"value\": {:e}}}", x.to_bits(), x).unwrap();
    }
    s.push(']');
    s
}

fn u64_list(v: &[u64]) -> String {
    let mut s = String::from("[");
    for (i, x) in v.iter().enumerate() {
        if i > 0 {
            s.push_str(", ");
        }
        write!(s, "\"{:016x}\"", x).unwrap();
    }
    s.push(']');
    s
}

fn main() {
    let out_dir = std::env::args().nth(1).unwrap_or_else(|| ".".to_string());
    for seed in [7u64, 17u64] {
        let mut rng = StdRng::seed_from_u64(seed);
        let words: Vec<u64> = (0..N).map(|_| rng.next_u64()).collect();

        let mut rng = StdRng::seed_from_u64(seed);
        let unit: Vec<f64> = (0..N).map(|_| rng.random::<f64>()).collect();

        let mut rng = StdRng::seed_from_u64(seed);
        let std_normal: Vec<f64> = (0..N).map(|_| rng.sample::<f64, _>(StandardNormal)).collect();

        // the form the reference writes (fastslam2.rs:227): Normal::new(0.0, 1.0).unwrap().sample(rng)
        let mut rng = StdRng::seed_from_u64(seed);
        let normal = Normal::new(0.0, 1.0).unwrap();
        let normal01: Vec<f64> = (0..N).map(|_| normal.sample(&mut rng)).collect();

        let mut uniforms = String::new();
        for (k, n) in [20usize, 120usize].iter().enumerate() {
            let mut rng = StdRng::seed_from_u64(seed);
            let u = Uniform::new(0.0, 1.0 / *n as f64).expect("valid resampling range"); // fastslam2.rs:310
            let draws: Vec<f64> = (0..N).map(|_| u.sample(&mut rng)).collect();
            if k > 0 {
                uniforms.push_str(", ");
            }
            write!(uniforms, "\"{}\": {}", n, f64_list(&draws)).unwrap();
        }

        let json = format!(
            "{{\"probe\": \"rng_streams\", \"seed\": {}, \"draws\": {}, \"next_u64\": {}, \"random_f64\": {}, \"standard_normal\": {}, \
             \"normal_0_1\": {}, \"uniform_0_inv_n\": {{{}}}}}\n",
            seed,
            N,
            u64_list(&words),
            f64_list(&unit),
            f64_list(&std_normal),
            f64_list(&normal01),
            uniforms
        );
        let path = format!("{}/rng_streams_seed{}.json", out_dir, seed);
        fs::write(&path, json).expect("cannot write the dump");
        println!("wrote {}", path);
    }
}
