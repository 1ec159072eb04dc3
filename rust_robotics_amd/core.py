"""Value types, trait and error of ``rust_robotics_core`` that the hot path's API uses.

Mirrors /root/reference/crates/rust_robotics_core/src:
  types.rs:17-20 Point2D, :141-146 State2D (+ to_vector :170-172), :189-192 ControlInput
  (+ to_vector :203-205), :344-346 Obstacles; traits.rs:31-52 StateEstimator;
  error.rs:8-24 RoboticsError.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Protocol, Sequence, Tuple

import numpy as np


class RoboticsError(Exception):
    """error.rs:8-24; ``kind`` is the enum variant name."""

    def __init__(self, kind: str, message: str):
        super().__init__(f"{kind}: {message}")
        self.kind = kind
        self.message = message

    @classmethod
    def invalid_parameter(cls, message: str) -> "RoboticsError":
        return cls("InvalidParameter", message)

    @classmethod
    def runtime(cls, message: str) -> "RoboticsError":
        return cls("NumericalError", message)


@dataclass
class Point2D:
    x: float = 0.0
    y: float = 0.0


@dataclass
class State2D:
    x: float = 0.0
    y: float = 0.0
    yaw: float = 0.0
    v: float = 0.0

    def to_vector(self) -> np.ndarray:
        return np.array([self.x, self.y, self.yaw, self.v], dtype=np.float64)


@dataclass
class ControlInput:
    v: float = 0.0
    omega: float = 0.0

    def to_vector(self) -> np.ndarray:
        return np.array([self.v, self.omega], dtype=np.float64)


@dataclass
class Obstacles:
    points: List[Point2D] = field(default_factory=list)

    @classmethod
    def from_points(cls, points: Sequence[Point2D]) -> "Obstacles":
        return cls(list(points))


class StateEstimator(Protocol):
    """traits.rs:31-52"""

    def predict(self, control, dt: float) -> None: ...

    def update(self, measurement) -> None: ...

    def get_state(self) -> np.ndarray: ...

    def get_covariance(self) -> Optional[np.ndarray]: ...
