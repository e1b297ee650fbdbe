"""Type aliases for the callables the training and evaluation loops take (annotation-only)."""
from typing import Any, Callable, List

from .samplers import PreparedBatch

TrainCore = Callable[..., Any]                                      # (model, batch, ...) -> per-batch result
TrainCallback = Callable[[List[PreparedBatch], List[Any]], None]    # batches of a step (one per device), their results
TestCallback = Callable[[PreparedBatch], None]
TrainImpl = Callable[..., None]                                     # an epoch driver such as serial_train
