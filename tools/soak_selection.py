"""Soak run (GPU box): the random-config parity check of tests/test_gpu_random_sweep.py over many more seeds and shapes
than the test suite carries.  Usage: python tools/soak_selection.py [n_cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import test_gpu_random_sweep as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
bad = 0
t0 = time.time()
for k in range(n_cases):
    seed = 100 + k
    n = int(rng.integers(200, 3000))
    edges = int(n * rng.uniform(1.0, 14.0))
    gamma = float(rng.uniform(2.02, 3.0))
    dim = int(rng.choice([32, 64, 128, 256]))
    layers = int(rng.integers(1, 4))
    th1 = float(rng.choice([0.0, 1e-5, 1e-4, 1e-3, 1e-2]))
    thn = float(rng.choice([1e-3, 5e-3, 1e-2, 1.0]))
    thc = float(rng.choice([0.0, 0.0, 1e-3]))
    if thn == 1.0 and th1 >= 1.0:
        th1 = 1e-2
    eps = float(rng.choice([5e-5, 1e-4, 2e-4, 1e-3]))
    case = (seed, n, edges, gamma, dim, layers, bool(rng.integers(0, 2)), (thc, th1, thn), eps, bool(rng.integers(0, 2)))
    try:
        T.test_random_config_matches_oracle(case)
        print("ok  ", case, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", case, repr(e)[:300], flush=True)
print(f"{n_cases - bad} / {n_cases} cases passed in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
