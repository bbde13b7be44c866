"""Deterministic parameter values for the golden fixtures (TEST INFRASTRUCTURE ONLY).

The golden generator (tests/golden/make_golden.py) loads these values into the reference model before
recording its outputs; the tests regenerate the identical values from (name, shape, seed) instead of
storing megabytes of random weights in the fixtures.  numpy's PCG64 stream is stable across versions.
"""
import zlib

import numpy as np


def make_param(name: str, shape, seed: int) -> np.ndarray:
    """Value of parameter `name` with `shape` for fixture `seed` (float32)."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        if name.endswith("weight"):  # LayerNorm gamma
            return (1.0 + 0.25 * rng.standard_normal(shape)).astype(np.float32)
        return (0.2 * rng.standard_normal(shape)).astype(np.float32)  # biases, LN beta
    fan = shape[-1] + shape[-2]
    bound = np.sqrt(6.0 / fan)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def make_state(shapes: dict, seed: int) -> dict:
    """{name: ndarray} for an ordered {name: shape} mapping."""
    return {k: make_param(k, v, seed) for k, v in shapes.items()}
