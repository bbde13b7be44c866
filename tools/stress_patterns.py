"""The pattern-table stress sweep as a report (the same 48 cases run as tests: tests/test_gpu_timed_path.py::
test_pattern_table_stress): select4 -> rows4 -> tail against the oracle over random graphs, widths 128 / 256, mask modes,
PE weights scaled so that the table's coverage goes from ~100 % to a few percent."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_timed_path import STRESS, stress_case  # noqa: E402

worst = 0.0
for k, (dim, gain, th, seed) in enumerate(STRESS, 1):
    err, raw, left, cov, mode, n_sel = stress_case(k, dim, gain, th, seed)
    worst = max(worst, err)
    print(f"{k:3d} D={dim} gain={gain:5.1f} th={th} mode={mode:5s} entries={n_sel:6d} flips raw {raw:6.2f} left {left:5.2f} "
          f"covered {[None if c is None else round(c, 3) for c in cov]} rel err {err:.2e}", flush=True)
    assert err <= 1e-4, "MISMATCH"
print("worst relative logit error", worst)
