"""Stand-in for `check_shapes`: every decorator / checker is the identity (shape checking is documentation + assertions in
the reference; it never changes a value)."""
from typing import Any


def check_shapes(*specs: Any, **kw: Any):
    return lambda f: f


def inherit_check_shapes(f):
    return f


def check_shape(x, spec=None, *a, **k):
    return x


def get_check_shapes(f):
    return None


def get_shape(x, context=None):
    import numpy as np
    return tuple(np.shape(x))


def register_get_shape(*types):
    return lambda f: f


def disable_check_shapes():
    class _C:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
    return _C()


def get_enable_check_shapes():
    return False


def set_enable_check_shapes(*a, **k):
    pass


class Shape(tuple):
    pass


class ErrorContext:
    def __init__(self, *a, **k):
        pass


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any()


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    return _Any
