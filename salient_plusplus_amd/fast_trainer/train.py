"""Training-side consumers of the data path: ``barebones_train_core`` (one batch: forward, nll loss,
backward, optimiser step), ``make_eval_and_loss`` and ``serial_train`` (one epoch over a
``DeviceIterator``), with the signatures of the reference's fast_trainer/train.py (:15-71, :74-78,
:343-380).

Left out on purpose: ``serial_train_ns`` (the PyG NeighborSampler variant, switched off in the
reference itself, train.py:145-147) and ``data_parallel_train`` (single-process multi-GPU, asserted
off under DDP, ddp.py:299)."""
import contextlib
from typing import Optional

import torch
import torch.nn.functional as F

from .concepts import TrainCallback, TrainCore
from .samplers import PreparedBatch
from .transferers import DeviceIterator
from .utils import runtime_stats_cuda


def barebones_train_core(model: torch.nn.Module, batch: PreparedBatch, preload_hook=None, optimizer=None, sync=True):
    """``sync=True``: gradients are reduced as usual and the optimiser steps here.  ``sync=False``:
    forward and backward run under ``model.no_sync()`` (DDP gradient accumulation) and nothing steps."""
    guard = contextlib.nullcontext() if sync else model.no_sync()
    with guard:
        log_probs = model(batch.x, batch.adjs)
        loss = F.nll_loss(log_probs, batch.y.squeeze(-1))      # labels arrive as [batch, 1]
        loss.backward()
    if sync:
        optimizer.step()
    return loss


def make_eval_and_loss(module, train_core):
    """Adapter for ``parallel_apply``-style callers that pass a batch as its four fields."""
    return lambda *fields, **_ignored: train_core(module, PreparedBatch(*fields))


def _world_mean(value):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(value)
        return value / float(torch.distributed.get_world_size())
    return value


def serial_train(model: torch.nn.Module, train_core: TrainCore, devit: DeviceIterator,
                 optimizer: torch.optim.Optimizer, lr_scheduler, cb: Optional[TrainCallback] = None,
                 dataset=None, devices=None) -> None:
    """One epoch.  Note the reference's behaviour, kept for drop-in equality of results: the optimiser
    steps inside ``train_core`` AND once more after it (train.py:53 and :371)."""
    if not (dataset is None or devices is None):
        raise AssertionError("serial_train_ns  disabled because of dataset.x_cpu/x_gpu split.")
    model.train()
    for (batch,) in devit:
        optimizer.zero_grad()
        result = train_core(model, batch, preload_hook=devit, optimizer=optimizer, sync=True)
        optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step(_world_mean(result).cpu())
        if cb is not None:
            cb([batch], [result])
    report = getattr(devit, "print_stats", None)
    if report is not None:
        report()


__all__ = ["barebones_train_core", "make_eval_and_loss", "serial_train", "runtime_stats_cuda"]
