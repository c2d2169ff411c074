"""Small re-implementation of the part of `multipledispatch` that gpflow.utilities.multipledispatch builds on:
type-signature registry, most-specific-first ordering, variadic signatures are not supported (GPflow registers none on the
hot path)."""
from . import dispatcher, variadic  # noqa: F401
from .dispatcher import Dispatcher  # noqa: F401
