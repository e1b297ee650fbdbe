"""Evaluation over a ``DeviceIterator`` -- the inference-time consumer of the data path
(``batchwise_test`` of the reference's fast_trainer/test.py; the launcher uses fanout [20,20,20])."""
import torch

from .concepts import TestCallback
from .transferers import DeviceIterator


@torch.no_grad()
def batchwise_test(model: torch.nn.Module, num_batches: int, devit: DeviceIterator, cb: TestCallback = None):
    """Counts correct top-1 predictions over every batch of ``devit`` (single device).

    Returns ``(correct, evaluated)``.  The per-batch counts stay on the device until the end, so the
    loop never waits for the GPU."""
    model.eval()
    if len(devit.devices) != 1:
        raise ValueError("batchwise_test evaluates on exactly one device")
    device = torch.device(devit.devices[0])
    hits = torch.zeros(max(1, num_batches), dtype=torch.long, device=device)
    evaluated = seen = 0
    for (batch,) in devit:
        predicted = model(batch.x, batch.adjs).argmax(dim=-1).reshape(-1)
        hits[seen % hits.numel()] += (predicted == batch.y.reshape(-1)).sum()
        evaluated += batch.batch_size
        seen += 1
        if cb is not None:
            cb(batch)
    return int(hits.sum().item()), evaluated
