"""Global defaults (mirror of gpflow/config/__config__.py:95-109, 229-385): default_float is float64,
default_jitter 1e-6, likelihood positive minimum 1e-6, softplus positive bijector.  Same environment
variables (GPFLOW_FLOAT, GPFLOW_JITTER, GPFLOW_POSITIVE_BIJECTOR, GPFLOW_POSITIVE_MINIMUM,
GPFLOW_LIKELIHOOD_POSITIVE_MINIMUM).  The device path computes in fp64 only, so float32 is rejected."""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass, replace
from typing import Iterator, Optional

import numpy as np


def _env(name, default):
    return os.getenv(name, default)


def _float_type():
    v = _env("GPFLOW_FLOAT", "float64")
    if v in (np.float64, "float64"):
        return np.float64
    raise TypeError("gpflow_amd computes in float64 only (GPFLOW_FLOAT must be float64)")


@dataclass(frozen=True)
class Config:
    int: type = np.int32
    float: type = np.float64
    jitter: float = 1e-6
    positive_bijector: str = "softplus"
    positive_minimum: float = 0.0
    likelihood_positive_minimum: float = 1e-6
    summary_fmt: Optional[str] = "fancy_grid"


def _from_env() -> Config:
    try:
        jitter = float(_env("GPFLOW_JITTER", 1e-6))
    except ValueError:
        raise TypeError("Config cannot set the jitter value with non float type.")
    bij = str(_env("GPFLOW_POSITIVE_BIJECTOR", "softplus")).lower()
    if bij not in ("softplus", "exp"):
        raise ValueError(f"Config cannot set the positive bijector '{bij}'")
    return Config(float=_float_type(), jitter=jitter, positive_bijector=bij,
                  positive_minimum=float(_env("GPFLOW_POSITIVE_MINIMUM", 0.0)),
                  likelihood_positive_minimum=float(_env("GPFLOW_LIKELIHOOD_POSITIVE_MINIMUM", 1e-6)))


_config = _from_env()


def config() -> Config:
    return _config


def set_config(new: Config) -> None:
    global _config
    _config = new


def default_int() -> type:
    return _config.int


def default_float() -> type:
    return _config.float


def default_jitter() -> float:
    return _config.jitter


def default_positive_bijector() -> str:
    return _config.positive_bijector


def default_positive_minimum() -> float:
    return _config.positive_minimum


def default_likelihood_positive_minimum() -> float:
    return _config.likelihood_positive_minimum


def set_default_float(value_type) -> None:
    if value_type not in (np.float64, float, "float64"):
        raise TypeError("gpflow_amd computes in float64 only")
    set_config(replace(_config, float=np.float64))


def set_default_jitter(value: float) -> None:
    if not isinstance(value, (float, int)) or isinstance(value, bool):
        raise TypeError("Expected float32 or float64 scalar value")
    if value < 0:
        raise ValueError("Jitter must be non-negative")
    set_config(replace(_config, jitter=float(value)))


def set_default_positive_bijector(value: str) -> None:
    v = value.lower()
    if v not in ("softplus", "exp"):
        raise ValueError(f"`{value}` not in set of valid bijectors: ['exp', 'softplus']")
    set_config(replace(_config, positive_bijector=v))


def set_default_positive_minimum(value: float) -> None:
    if not isinstance(value, (float, int)) or isinstance(value, bool):
        raise TypeError("Expected float32 or float64 scalar value")
    if value < 0:
        raise ValueError("Positive minimum must be non-negative")
    set_config(replace(_config, positive_minimum=float(value)))


def set_default_likelihood_positive_minimum(value: float) -> None:
    if not isinstance(value, (float, int)) or isinstance(value, bool):
        raise TypeError("Expected float32 or float64 scalar value")
    if value < 0:
        raise ValueError("Likelihood positive minimum must be non-negative")
    set_config(replace(_config, likelihood_positive_minimum=float(value)))


@contextlib.contextmanager
def as_context(temporary_config: Optional[Config] = None) -> Iterator[None]:
    """Temporarily replace the global config (gpflow/config/__config__.py:376-385)."""
    current = config()
    temporary_config = replace(current) if temporary_config is None else temporary_config
    try:
        set_config(temporary_config)
        yield
    finally:
        set_config(current)
