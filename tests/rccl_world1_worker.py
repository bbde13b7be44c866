"""First contact with RCCL on a one-GPU box (tests/test_gpu_dist.py::test_rccl_world_of_one): a process group with
backend "nccl" (= RCCL on ROCm) and ONE rank, ``lpformer_amd.dist.FORCE_COLLECTIVES`` on, so that ``allgather_rows``,
``measure_allgather_gbps``, ``max_over_ranks`` and the three encoder layouts of ``set_row_shard`` go through the
library's real collective calls (ncclAllGather / ncclAllReduce on the device) instead of returning early.  Everything
must equal the unsharded result bit for bit.  Exit code 0 = every comparison held and librccl is mapped."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import lpformer_amd  # noqa: E402
from lpformer_amd import data as D  # noqa: E402
from lpformer_amd import dist as LD  # noqa: E402


def main():
    os.environ["LPF_DIST_FORCE_INIT"] = "1"
    rank, world, local = LD.init_from_env("nccl")
    assert (rank, world) == (0, 1) and torch.distributed.get_backend() == "nccl"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    LD.FORCE_COLLECTIVES = True
    bad = []
    # the collective wrappers by themselves: even and ragged row counts
    for n, d in ((1024, 128), (1501, 64)):
        local_rows = torch.randn(n, d, device=dev)
        full = LD.allgather_rows(local_rows, n)
        if full.data_ptr() == local_rows.data_ptr() or not torch.equal(full, local_rows):
            bad.append(f"allgather_rows({n}, {d}) did not go through the collective or changed the rows")
    sc = torch.randn(777, device=dev)
    if not torch.equal(LD.gather_scores(sc, 777), sc):
        bad.append("gather_scores")
    if LD.max_over_ranks(3.25, dev) != 3.25:
        bad.append("max_over_ranks")
    gbps = LD.measure_allgather_gbps(235_868, 128, dev, reps=3)
    if not (0.0 < gbps < float("inf")):
        bad.append(f"measure_allgather_gbps returned {gbps}")
    # the three encoder layouts with the collectives in place
    n, dim, layers = 1501, 64, 3
    ei, w = D.chung_lu_graph(n, 6000, seed=3, max_weight=4)
    x = np.random.default_rng(0).standard_normal((n, 48)).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, ppr=lpformer_amd.calc_ppr(ei, n, 0.15, 1e-3))
    args = D.train_args_for(dict(thresholds=(0.0, 1e-3, 1e-2), dim=dim, gnn_layers=layers, residual=True))
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(args, data, device=dev).to(dev).eval()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
    batch = torch.from_numpy(D.sample_pairs(ei, n, 1024, seed=1)).to(dev)
    LD.FORCE_COLLECTIVES = False
    want_x = model.propagate().clone()
    want_s = model.score_pairs(batch, want_x, score, logits=True).clone()
    assert model.check_selection()
    LD.FORCE_COLLECTIVES = True
    for mode in ("replicated", "sharded", "gather_once"):
        model.set_row_shard(0, 1, mode)
        h = model.propagate()
        if not torch.equal(h, want_x):
            bad.append(f"{mode}: encoder output differs by {(h - want_x).abs().max().item():.3e}")
        got = model.score_pairs(batch, h, score, logits=True)
        if not model.check_selection() or not torch.equal(got, want_s):
            bad.append(f"{mode}: scores differ by {(got - want_s).abs().max().item():.2e}")
    maps = open("/proc/self/maps").read()
    if "librccl" not in maps and "libnccl" not in maps:
        bad.append("librccl is not mapped into the process")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    if bad:
        print("; ".join(bad), flush=True)
        sys.exit(1)
    print(f"rccl world-1 ok: all-gather of a [235868, 128] fp32 matrix at {gbps:.0f} GB/s (one rank: a device copy)",
          flush=True)


if __name__ == "__main__":
    main()
