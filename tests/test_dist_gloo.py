"""world_size-2 gloo tests (CPU) of the multi-GPU layout logic: row blocks, all-gather assembly, pair sharding,
max-over-ranks timing.  The kernels themselves need a GPU; what is checked here is the N>1 plumbing around them."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lpformer_amd import dist as LD


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, bs, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = LD.init_from_env("gloo")
    assert (r, w) == (rank, world)
    full = torch.arange(n * d, dtype=torch.float32).reshape(n, d)
    lo, hi = LD.row_range(n, world, rank)
    got = LD.allgather_rows(full[lo:hi].clone(), n)
    ok_rows = torch.equal(got, full)
    batch = torch.stack([torch.arange(bs), torch.arange(bs) + 1000])
    mine = LD.shard_pairs(batch, world, rank)
    scores = mine[0].float() * 2.0  # stand-in for the per-pair score
    allsc = LD.gather_scores(scores, bs)
    ok_scores = torch.equal(allsc, batch[0].float() * 2.0)
    tmax = LD.max_over_ranks(1.0 + rank)
    q.put((rank, ok_rows, ok_scores, mine.shape[1], tmax))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bs", [(10, 8), (11, 7)])  # divisible and ragged splits
def test_two_rank_layout(n, bs):
    world, d = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, d, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)
    assert sum(r[3] for r in res) == bs          # every pair scored exactly once
    assert all(r[4] == 2.0 for r in res)         # max over ranks


def test_row_range_partitions():
    for n in (0, 1, 7, 64, 235_868):
        for world in (1, 2, 3, 8):
            spans = [LD.row_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
