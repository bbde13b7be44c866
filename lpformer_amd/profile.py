"""Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline)."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch


class _NoOp:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOOP = _NoOp()


class KernelTimer:
    enabled = False
    _records = defaultdict(list)

    @classmethod
    def reset(cls):
        cls._records = defaultdict(list)

    @classmethod
    def span(cls, name: str):
        """Context manager around one kernel launch; a shared no-op object when timing is off (the launch path calls
        this ~15 times per step)."""
        return cls._timed(name) if cls.enabled else _NOOP

    @classmethod
    @contextmanager
    def _timed(cls, name: str):
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
