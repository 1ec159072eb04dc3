"""``rust_robotics_slam`` modules on the hot path (fastslam1, fastslam2; lib.rs:4-19 lists the rest)."""
from . import fastslam1, fastslam2  # noqa: F401
