"""Observability hooks with the region names the reference uses (fast_trainer/utils.py:123-249):
named regions bracketed by event pairs and summed per epoch.  Disabled by default (zero overhead);
``runtime_stats_cuda.enable()`` turns the event recording on."""
import time
from collections import defaultdict
from typing import Callable, Dict, List, NamedTuple, Optional

import torch


class TimerResult(NamedTuple):
    name: str
    nanos: int


class Timer:
    def __init__(self, name: str, fn: Optional[Callable[[TimerResult], None]] = None):
        self.name, self._fn = name, fn

    def __enter__(self):
        self._t0 = time.perf_counter_ns()
        return self

    def __exit__(self, *a):
        res = TimerResult(self.name, time.perf_counter_ns() - self._t0)
        if self._fn is not None:
            self._fn(res)
        return False


class RuntimeStatisticsCUDA:
    """start_region/end_region pairs measured with device events (utils.py:123-249)."""

    def __init__(self, name: str = "SALIENT"):
        self.name = name
        self.enabled = False
        self._open: Dict[str, "torch.cuda.Event"] = {}
        self._pairs: Dict[str, List] = defaultdict(list)
        self._last = None
        self.epoch_totals: List[Dict[str, float]] = []

    def enable(self, on: bool = True):
        self.enabled = bool(on) and torch.cuda.is_available()

    def get_last_event(self):
        return self._last

    def start_region(self, region: str, use_event=None):
        if not self.enabled:
            return
        ev = use_event
        if ev is None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        self._open[region] = ev

    def end_region(self, region: str, use_event=None):
        if not self.enabled or region not in self._open:
            return
        ev = use_event
        if ev is None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        self._pairs[region].append((self._open.pop(region), ev))
        self._last = ev

    def start_epoch(self):
        self._pairs.clear()
        self._open.clear()

    def end_epoch(self):
        if not self.enabled:
            return
        torch.cuda.synchronize()
        tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self._pairs.items()}
        self.epoch_totals.append(tot)
        self._pairs.clear()

    def report_stats(self, display_keys=None) -> str:
        rows = self.epoch_totals[1:] if len(self.epoch_totals) > 1 else self.epoch_totals   # first epoch dropped
        keys = sorted({k for r in rows for k in r})
        lines = [f"[{self.name}] region: mean ms per epoch over {len(rows)} epoch(s)"]
        for k in keys:
            if display_keys and k not in display_keys:
                continue
            vals = [r.get(k, 0.0) for r in rows]
            lines.append(f"  {display_keys[k] if display_keys else k}: {sum(vals) / max(1, len(vals)):.3f}")
        return "\n".join(lines)

    def clear_stats(self):
        self.epoch_totals.clear()


runtime_stats_cuda = RuntimeStatisticsCUDA()

_runtime_stats: Dict[str, List[float]] = defaultdict(list)


def append_runtime_stats(name: str, value: float):
    _runtime_stats[name].append(value)


def get_runtime_stats():
    return _runtime_stats
