"""Utilities on the surface (gpflow/utilities): bijectors, traversal helpers, to_default_float."""
from __future__ import annotations

import numpy as np

from .base import (multiple_assign, parameter_dict, positive, read_values, set_trainable,  # noqa: F401
                   triangular)
from .posteriors import assert_params_false  # noqa: F401


def to_default_float(x):
    return np.asarray(x, dtype=np.float64)


def triangular_size(n: int) -> int:
    return n * (n + 1) // 2
