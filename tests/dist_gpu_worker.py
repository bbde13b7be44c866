"""One rank of the multi-rank encoder test (tests/test_gpu_dist.py launches WORLD_SIZE of these as child processes,
all on cuda:0, process group backend gloo): every encoder layout of ``LinkTransformer.set_row_shard`` under a REAL process
group against the unsharded encoder computed in the same process, and the pair stage on this rank's shard of a batch.
Exit code 0 = every comparison held."""
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
    rank, world, _ = LD.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n, dim, layers = (1501, 64, 3) if os.environ.get("LPF_TEST_RAGGED") == "1" else (1600, 128, 2)
    ei, w = D.chung_lu_graph(n, 6000, seed=3, max_weight=4)
    x = np.random.default_rng(0).standard_normal((n, 48)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, 1e-3)
    data = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    args = D.train_args_for(dict(thresholds=(0.0, 1e-3, 1e-2), dim=dim, gnn_layers=layers, residual=(layers == 3)))
    torch.manual_seed(0)                                  # same weights on every rank
    model = lpformer_amd.LinkTransformer(args, data, device=dev).to(dev).eval()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
    batch = torch.from_numpy(D.sample_pairs(ei, n, 1024, seed=1)).to(dev)
    want_x = model.propagate().clone()                    # world = 1 path
    want_s = model.score_pairs(batch, want_x, score, logits=True).clone()
    assert model.check_selection()
    mine = LD.shard_pairs(batch, world, rank).contiguous()
    lo, hi = LD.row_range(batch.shape[1], world, rank)
    bad = []
    for mode in ("replicated", "sharded", "gather_once"):
        model.set_row_shard(rank, world, mode)
        h = model.propagate()
        if not torch.equal(h, want_x):
            bad.append(f"{mode}: encoder output differs by {(h - want_x).abs().max().item():.3e}")
        got = model.score_pairs(mine, h, score, logits=True)
        # (a sub-batch is not bitwise the full batch: the one-pass attention merges the pieces of a pair's segment in an
        #  order that depends on where the segment falls in the tile grid -- DESIGN.md section 2 -- hence 2e-6)
        if not model.check_selection() or (got - want_s[lo:hi]).abs().max().item() > 2e-6:
            bad.append(f"{mode}: scores of this rank's pairs differ by {(got - want_s[lo:hi]).abs().max().item():.2e}")
        allsc = LD.gather_scores(got, batch.shape[1])
        if allsc.shape != want_s.shape or (allsc - want_s).abs().max().item() > 2e-6:
            bad.append(f"{mode}: gathered scores differ")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    if bad:
        print(f"[rank {rank}] " + "; ".join(bad), flush=True)
        sys.exit(1)
    print(f"[rank {rank}] ok", flush=True)


if __name__ == "__main__":
    main()
