#!/usr/bin/env python3
"""Port of the particle-filter part of crates/rust_robotics/examples/headless_localizers.rs:9-95: same landmarks,
control, 40 steps and print-outs; the only change is where ``ParticleFilterLocalizer`` is imported from.
(The EKF / UKF localizers of that example are outside this engine's scope.)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rust_robotics_amd.core import ControlInput, Obstacles, Point2D, State2D  # noqa: E402
from rust_robotics_amd.localization import ParticleFilterConfig, ParticleFilterLocalizer  # noqa: E402


def propagate_state(state: State2D, control: ControlInput, dt: float) -> None:  # :9-14
    state.x += control.v * math.cos(state.yaw) * dt
    state.y += control.v * math.sin(state.yaw) * dt
    state.yaw += control.omega * dt
    state.v = control.v


def build_pf_measurements(state: State2D, landmarks: Obstacles):  # :16-26
    return [(math.hypot(state.x - lm.x, state.y - lm.y), lm.x, lm.y) for lm in landmarks.points]


def main() -> int:
    landmarks = Obstacles.from_points([Point2D(5.0, 0.0), Point2D(0.0, 5.0), Point2D(5.0, 5.0)])  # :29-33
    true_state = State2D()
    control = ControlInput(1.0, 0.1)
    pf = ParticleFilterLocalizer.with_initial_state_2d(State2D(), ParticleFilterConfig())  # :39-43
    pf.set_landmarks_from_obstacles(landmarks)
    for step in range(40):  # :45-70
        propagate_state(true_state, control, 0.1)
        pf_state = pf.try_step_state(control, build_pf_measurements(true_state, landmarks))
        if step % 10 == 0:
            print(f"step={step:02d} true=({true_state.x:.2f}, {true_state.y:.2f}) pf=({pf_state.x:.2f}, {pf_state.y:.2f})")
    f = pf.state_2d()
    print(f"final true=({true_state.x:.2f}, {true_state.y:.2f}) pf=({f.x:.2f}, {f.y:.2f})")
    err = math.hypot(f.x - true_state.x, f.y - true_state.y)
    print(f"HEADLESS_OK error={err:.3f}")
    return 0 if err < 1.0 else 1


if __name__ == "__main__":
    sys.exit(main())
