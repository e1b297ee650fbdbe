"""Batchwise evaluation loop (reference: fast_trainer/test.py:8-33): the inference-time consumer of
the data path (fanout [20,20,20] in the reference's launcher)."""
import torch

from .concepts import TestCallback
from .transferers import DeviceIterator


@torch.no_grad()
def batchwise_test(model: torch.nn.Module, num_batches: int, devit: DeviceIterator, cb: TestCallback = None):
    """-> (number of correct predictions, number of evaluated seeds)."""
    model.eval()
    device, = devit.devices
    on_gpu = torch.device(device).type == "cuda"
    results = torch.empty(num_batches, dtype=torch.long, pin_memory=on_gpu)
    total = 0
    for i, inputs in enumerate(devit):
        inp, = inputs
        out = model(inp.x, inp.adjs)
        out = out.argmax(dim=-1, keepdim=True).reshape(-1)
        correct = (out == inp.y.reshape(-1)).sum()
        results[i].copy_(correct, non_blocking=True)
        total += inp.batch_size
        if cb is not None:
            cb(inp)
    if on_gpu:
        torch.cuda.current_stream(device).synchronize()
    return results.sum().item(), total
