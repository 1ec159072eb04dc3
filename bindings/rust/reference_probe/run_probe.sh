#!/usr/bin/env bash
# One command from "a machine with cargo and a checkout of rsasaki0109/rust_robotics" to the first numbers computed by the
# reference itself:   bash bindings/rust/reference_probe/run_probe.sh /path/to/rust_robotics /tmp/rr_probe
# Never writes into the checkout: works on a scratch copy.  Then: python tools/compare_reference_dump.py /tmp/rr_probe [--gpu]
set -euo pipefail
REF=${1:?usage: run_probe.sh <rust_robotics checkout> [output directory]}
OUT=$(mkdir -p "${2:-/tmp/rr_probe}" && cd "${2:-/tmp/rr_probe}" && pwd)
HERE=$(cd "$(dirname "$0")" && pwd)
command -v cargo >/dev/null || { echo "cargo not found: this kit needs a Rust toolchain (the engine's build image has none)"; exit 2; }
[ -f "$REF/crates/rust_robotics_slam/src/fastslam2.rs" ] || { echo "$REF does not look like a rust_robotics checkout"; exit 2; }

# 1. the generators' streams (stand-alone crate; rand / rand_distr at the reference's pins)
( cd "$HERE/rng_streams" && cargo run --release --quiet -- "$OUT" )

# 2. the seeded FastSLAM 2.0 tests, from inside the reference's own module
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$REF/Cargo.toml" "$REF/crates" "$WORK/"
[ -f "$REF/Cargo.lock" ] && cp "$REF/Cargo.lock" "$WORK/"      # the reference's pinned versions
[ -d "$REF/vendor" ] && cp -r "$REF/vendor" "$WORK/"
for f in README.md LICENSE LICENSE-MIT rust-toolchain.toml .cargo; do [ -e "$REF/$f" ] && cp -r "$REF/$f" "$WORK/" || true; done
for extra in ros2_nodes examples benches xtask; do [ -e "$REF/$extra" ] && cp -r "$REF/$extra" "$WORK/" || true; done  # whatever the workspace manifest lists
cat >> "$WORK/crates/rust_robotics_slam/src/fastslam2.rs" <<RS

#[cfg(test)]
#[path = "$HERE/fastslam2_probe.rs"]
mod reference_probe;
RS
( cd "$WORK" && RR_PROBE_OUT="$OUT" cargo test -p rust_robotics_slam --lib reference_probe -- --nocapture --test-threads 1 )
ls -la "$OUT"
echo "next: python tools/compare_reference_dump.py $OUT   (add --gpu on an MI355X box)"
