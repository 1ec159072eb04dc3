"""rust_robotics_amd -- MI355X-native engine for rust_robotics' sampling-based localization hot path.

Layout: ``csrc/`` holds the hand-written HIP kernels and the C ABI
(``include/rr_pf.h``, ``include/rr_fastslam1.h``); ``localization`` and ``slam.fastslam1`` mirror
the reference's Rust API names over that ABI; ``sharded`` runs one shard per GPU with RCCL
collectives between the phases of a step.  Importing the package does not load the shared
library; the first engine call does, and fails loudly if it has not been built.
"""
from .core import ControlInput, Obstacles, Point2D, RoboticsError, State2D  # noqa: F401

__version__ = "0.1.0"
