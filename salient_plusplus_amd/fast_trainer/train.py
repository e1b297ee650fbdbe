"""Consumers of the data path on the training side (reference: fast_trainer/train.py:15-71
``barebones_train_core``, ``:74-78`` ``make_eval_and_loss``, ``:343-380`` ``serial_train``).

Only the loops that drive a ``DeviceIterator`` of this repository are provided: the PyG
NeighborSampler variant (``serial_train_ns``) is disabled in the reference itself (train.py:145-147)
and the single-process multi-GPU ``data_parallel_train`` is asserted off under DDP (ddp.py:299)."""
from typing import Optional

import torch
import torch.nn.functional as F

from .concepts import TrainCallback, TrainCore
from .samplers import PreparedBatch
from .transferers import DeviceIterator
from .utils import runtime_stats_cuda


def barebones_train_core(model: torch.nn.Module, batch: PreparedBatch, preload_hook=None, optimizer=None, sync=True):
    """Forward, nll loss on ``y.squeeze(-1)``, backward and -- with ``sync`` -- the optimiser step
    (train.py:15-71; without ``sync`` the backward runs under ``model.no_sync()``)."""
    if sync:
        out = model(batch.x, batch.adjs)
        loss = F.nll_loss(out, batch.y.squeeze(-1))
        loss.backward()
        optimizer.step()
    else:
        with model.no_sync():
            out = model(batch.x, batch.adjs)
            loss = F.nll_loss(out, batch.y.squeeze(-1))
            loss.backward()
    return loss


def make_eval_and_loss(module, train_core):
    def eval_and_loss(*args, **_):
        return train_core(module, PreparedBatch(*args))

    return eval_and_loss


def serial_train(model: torch.nn.Module, train_core: TrainCore, devit: DeviceIterator,
                 optimizer: torch.optim.Optimizer, lr_scheduler, cb: Optional[TrainCallback] = None,
                 dataset=None, devices=None) -> None:
    """One epoch over ``devit`` (train.py:343-380).  As in the reference the optimiser steps both
    inside ``train_core`` (when it is ``barebones_train_core``) and once more here (:371)."""
    if dataset is not None and devices is not None:
        raise AssertionError("serial_train_ns  disabled because of dataset.x_cpu/x_gpu split.")   # train.py:145-147
    model.train()
    iterator = iter(devit)
    while True:
        try:
            inp, = next(iterator)
        except StopIteration:
            break
        optimizer.zero_grad()
        result = train_core(model, inp, preload_hook=devit, optimizer=optimizer, sync=True)
        optimizer.step()
        if lr_scheduler is not None:
            world_size = 1.0
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.all_reduce(result)
                world_size = 1.0 * torch.distributed.get_world_size()
            lr_scheduler.step(result.cpu() / world_size)
        if cb is not None:
            cb([inp], [result])
    if hasattr(devit, "print_stats"):
        devit.print_stats()


__all__ = ["barebones_train_core", "make_eval_and_loss", "serial_train", "runtime_stats_cuda"]
