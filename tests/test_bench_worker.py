"""bench.py's CPU-baseline leg without a GPU: the worker processes (oracle/bench_worker.py) score a sample through
memory-mapped inputs and return exactly what a direct call of the oracle returns."""
import numpy as np

from lpformer_amd import data as D, graph
from lpformer_amd.ppr import calc_ppr
from oracle import fixture_weights
from oracle import lpformer_oracle as O


def test_cpu_workers_equal_direct_oracle_call():
    import bench
    from tests.golden_util import Fixture
    fx = Fixture("lp_all_d64")
    n = fx.n
    r, c, v = fx.ppr_coo
    ppr = graph.csr_from_coo(r, c, v, n)
    mask = graph.mask_csr(fx.edge_index, n)
    P = fx.params
    cfg = dict(fx.cfg)
    x_node = fx["x_node"]
    batch = fx["batch"]
    sample = np.ascontiguousarray(np.concatenate([batch, batch[::-1]], axis=1))
    got, secs = bench.run_cpu_workers(sample, x_node, mask, ppr, P, cfg, n_proc=3, chunk=50)
    want = O.forward(sample, None, None, (mask.rowptr, mask.col.astype(np.int64)),
                     (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, cfg, x_node=x_node)["logit"]
    assert secs > 0 and np.abs(got - want).max() <= 2e-6   # (chunked calls: BLAS sums in another order)
    assert np.abs(got[:batch.shape[1]] - fx["logit"]).max() <= 2e-5   # ... and the reference's own logits
