"""``rust_robotics_slam`` modules on the hot path (only fastslam1; lib.rs:4-19 lists the rest)."""
from . import fastslam1  # noqa: F401
