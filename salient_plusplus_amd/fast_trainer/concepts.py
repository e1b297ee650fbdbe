"""Names for the callables the training and evaluation loops exchange (the reference keeps plain
``Callable`` aliases in fast_trainer/concepts.py; these are structural protocols with the same
names, so annotations written against the reference keep type-checking)."""
from typing import Any, List, Optional, Protocol

import torch

from .samplers import PreparedBatch
from .transferers import DeviceIterator


class TrainCore(Protocol):
    """Runs forward/backward on one batch and returns the loss (or any per-batch result)."""

    def __call__(self, model: torch.nn.Module, batch: PreparedBatch, *args: Any, **kwargs: Any) -> Any: ...


class TrainCallback(Protocol):
    """Told about the batches of a step (one per device) and their results."""

    def __call__(self, batches: List[PreparedBatch], results: List[Any]) -> None: ...


class TestCallback(Protocol):
    """Told about every evaluated batch; returns nothing."""

    def __call__(self, batch: PreparedBatch) -> None: ...


class TrainImpl(Protocol):
    """An epoch driver such as ``serial_train``."""

    def __call__(self, model: torch.nn.Module, train_core: TrainCore, devit: DeviceIterator,
                 optimizer: torch.optim.Optimizer, cb: Optional[TrainCallback]) -> None: ...
