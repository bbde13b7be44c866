"""Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline)."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch


class KernelTimer:
    enabled = False
    _records = defaultdict(list)

    @classmethod
    def reset(cls):
        cls._records = defaultdict(list)

    @classmethod
    @contextmanager
    def span(cls, name: str):
        if not cls.enabled:
            yield
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # torch's current stream == the stream the C-ABI call launches on
        yield
        e1.record()
        cls._records[name].append((e0, e1))

    @classmethod
    def summary(cls) -> dict:
        """name -> (launches, total ms, mean ms); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, evs in cls._records.items():
            tot = sum(a.elapsed_time(b) for a, b in evs)
            out[name] = (len(evs), tot, tot / max(len(evs), 1))
        return out
