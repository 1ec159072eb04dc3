// Port of the particle-filter part of crates/rust_robotics/examples/headless_localizers.rs:9-95 to the C++ wrapper
// (include/rust_robotics.hpp): same landmarks, same control, same 40 steps, same print-outs -- the only edit a
// caller makes is the type's namespace.  (The EKF / UKF localizers of that example are outside this engine's scope.)
//   g++ -std=c++17 -I include examples/cpp/headless_localizers.cpp -L rust_robotics_amd -lrust_robotics_amd -Wl,-rpath,$PWD/rust_robotics_amd
#include <cmath>
#include <cstdio>

#include "rust_robotics.hpp"

using namespace rr;

static void propagate_state(State2D& state, const ControlInput& control, double dt) {  // :9-14
  state.x += control.v * std::cos(state.yaw) * dt;
  state.y += control.v * std::sin(state.yaw) * dt;
  state.yaw += control.omega * dt;
  state.v = control.v;
}

static PFMeasurement build_pf_measurements(const State2D& state, const Obstacles& landmarks) {  // :16-26
  PFMeasurement m;
  for (const auto& lm : landmarks.points) {
    const double dx = state.x - lm.x, dy = state.y - lm.y;
    m.emplace_back(std::sqrt(dx * dx + dy * dy), lm.x, lm.y);
  }
  return m;
}

int main() {
  const Obstacles landmarks = Obstacles::from_points({{5.0, 0.0}, {0.0, 5.0}, {5.0, 5.0}});  // :29-33
  State2D true_state{};
  const ControlInput control{1.0, 0.1};
  auto pf = ParticleFilterLocalizer::with_initial_state_2d(State2D{}, ParticleFilterConfig{});  // :39-43
  pf.set_landmarks_from_obstacles(landmarks);
  for (int step = 0; step < 40; ++step) {  // :45-70
    propagate_state(true_state, control, 0.1);
    const State2D pf_state = pf.try_step_state(control, build_pf_measurements(true_state, landmarks));
    if (step % 10 == 0)
      std::printf("step=%02d true=(%.2f, %.2f) pf=(%.2f, %.2f)\n", step, true_state.x, true_state.y, pf_state.x, pf_state.y);
  }
  const State2D f = pf.state_2d();
  std::printf("final true=(%.2f, %.2f) pf=(%.2f, %.2f)\n", true_state.x, true_state.y, f.x, f.y);
  const double err = std::hypot(f.x - true_state.x, f.y - true_state.y);
  std::printf("HEADLESS_OK error=%.3f\n", err);
  return err < 1.0 ? 0 : 1;
}
