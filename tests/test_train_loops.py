"""CPU: the training / evaluation loops that consume a DeviceIterator (fast_trainer/train.py,
test.py) driven by a stub iterator and a plain torch model -- host logic only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Devit:
    def __init__(self, batches):
        self.devices = ["cpu"]
        self.batches = batches
        self.printed = 0

    def __iter__(self):
        return iter([[b] for b in self.batches])

    def print_stats(self):
        self.printed += 1


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 3)

    def forward(self, x, adjs):
        return torch.log_softmax(self.lin(x[:adjs]), dim=-1)


def _batches(n):
    from salient_plusplus_amd.fast_trainer.samplers import PreparedBatch
    g = torch.Generator().manual_seed(0)
    out = []
    for k in range(n):
        x = torch.randn((10, 4), generator=g)
        y = (x[:6].sum(-1) > 0).long().unsqueeze(-1)
        out.append(PreparedBatch(x, y, 6, slice(6 * k, 6 * k + 6)))
    return out


def test_serial_train_steps_and_callbacks():
    from salient_plusplus_amd.fast_trainer.train import barebones_train_core, make_eval_and_loss, serial_train
    torch.manual_seed(0)
    model = _Model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    w0 = model.lin.weight.detach().clone()
    seen = []
    devit = _Devit(_batches(5))
    serial_train(model, barebones_train_core, devit, opt, None, cb=lambda inp, res: seen.append((inp[0].batch_size, float(res[0].detach()))))
    assert len(seen) == 5 and all(bs == 6 for bs, _ in seen) and devit.printed == 1
    assert not torch.equal(model.lin.weight, w0) and model.training
    # the reference steps the optimiser twice per batch (train.py:53 and :371): one batch with SGD
    # moves the weights by 2 * lr * grad
    m2 = _Model()
    m2.load_state_dict({"lin.weight": w0.clone(), "lin.bias": torch.zeros(3)})
    m3 = _Model()
    m3.load_state_dict(m2.state_dict())
    b = _batches(1)[0]
    serial_train(m2, barebones_train_core, _Devit([b]), torch.optim.SGD(m2.parameters(), lr=0.1), None)
    loss = torch.nn.functional.nll_loss(m3(b.x, b.adjs), b.y.squeeze(-1))
    loss.backward()
    torch.testing.assert_close(m2.lin.weight, m3.lin.weight - 2 * 0.1 * m3.lin.weight.grad)
    f = make_eval_and_loss(m3, lambda mod, batch: batch.batch_size)
    assert f(b.x, b.y, b.adjs, b.idx_range) == 6


def test_batchwise_test_counts():
    from salient_plusplus_amd.fast_trainer.test import batchwise_test
    torch.manual_seed(1)
    model = _Model()
    bs = _batches(4)
    want = sum(int((model(b.x, b.adjs).argmax(-1) == b.y.reshape(-1)).sum()) for b in bs)
    touched = []
    correct, total = batchwise_test(model, 4, _Devit(bs), cb=lambda inp: touched.append(inp.idx_range.start))
    assert (correct, total) == (want, 24) and touched == [0, 6, 12, 18] and not model.training
