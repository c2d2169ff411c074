"""training_loss / training_loss_closure (gpflow/models/training_mixins.py:43-147).  `compile` is
accepted for signature compatibility; there is no tracing compiler here (the device work is already
fused C-ABI calls)."""
from __future__ import annotations

from typing import Callable, Iterator, Union


class InternalDataTrainingLossMixin:
    def training_loss(self):
        return self._training_loss()

    def training_loss_closure(self, *, compile: bool = True) -> Callable[[], object]:
        return self.training_loss


class ExternalDataTrainingLossMixin:
    def training_loss(self, data):
        return self._training_loss(data)

    def training_loss_closure(self, data: Union[tuple, Iterator], *, compile: bool = True):
        if isinstance(data, tuple):
            def closure():
                return self._training_loss(data)
        else:
            def closure():
                return self._training_loss(next(data))
        return closure


def training_loss(model, data=None):
    """gpflow/models/util.py / training_mixins helpers"""
    if isinstance(model, ExternalDataTrainingLossMixin):
        return model.training_loss(data)
    return model.training_loss()


def training_loss_closure(model, data=None, **kwargs):
    if isinstance(model, ExternalDataTrainingLossMixin):
        return model.training_loss_closure(data, **kwargs)
    return model.training_loss_closure(**kwargs)
