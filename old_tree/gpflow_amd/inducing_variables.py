"""Inducing variables on the path (gpflow/inducing_variables/inducing_variables.py:63-88,
multioutput/inducing_variables.py:98-175)."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from .base import Module, Parameter


class InducingVariables(Module):
    @property
    def num_inducing(self) -> int:
        raise NotImplementedError


class InducingPointsBase(InducingVariables):
    def __init__(self, Z, name=None):
        if not isinstance(Z, Parameter):
            Z = Parameter(np.asarray(Z, dtype=np.float64) if not hasattr(Z, "detach") else Z)
        self.Z = Z
        self.name = name

    @property
    def num_inducing(self) -> int:
        return int(self.Z.shape[0])

    def __len__(self) -> int:
        return self.num_inducing


class InducingPoints(InducingPointsBase):
    """Real-space inducing points."""


class MultioutputInducingVariables(InducingVariables):
    @property
    def inducing_variables(self):
        raise NotImplementedError


class SharedIndependentInducingVariables(MultioutputInducingVariables):
    """One set of inducing points shared by all latent GPs (multioutput/inducing_variables.py:169-175)."""

    def __init__(self, inducing_variable: InducingVariables):
        self.inducing_variable = inducing_variable

    @property
    def num_inducing(self) -> int:
        return self.inducing_variable.num_inducing

    @property
    def inducing_variables(self):
        return (self.inducing_variable,)


class SeparateIndependentInducingVariables(MultioutputInducingVariables):
    """One set of inducing points per latent GP (multioutput/inducing_variables.py:98-166)."""

    def __init__(self, inducing_variable_list: Sequence[InducingVariables]):
        self.inducing_variable_list = list(inducing_variable_list)

    @property
    def num_inducing(self) -> int:
        return self.inducing_variable_list[0].num_inducing

    @property
    def inducing_variables(self):
        return tuple(self.inducing_variable_list)


def inducingpoint_wrapper(inducing_variable) -> InducingVariables:
    """gpflow/models/util.py:31-38"""
    if not isinstance(inducing_variable, InducingVariables):
        inducing_variable = InducingPoints(inducing_variable)
    return inducing_variable
