"""gpflow/mean_functions.py: the historical name of gpflow/functions.py -- re-exports."""
from .functions import (Additive, Constant, Function, Identity, Linear, MeanFunction, Polynomial, Product,  # noqa: F401
                        Zero)
