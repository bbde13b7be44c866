"""Multi-GPU layout for the scoring path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL).

The reference is single-process, single-device (src/run.py:21,94); this layout is new.

* Graph structures (CSR adjacency, PPR) and weights are replicated on every GPU (they fit 288 GB many times over).
* The encoder is row-sharded: rank r owns a contiguous block of node rows.  Each GCN layer needs the previous
  layer's full activations for its neighbour gather, so a layer is: GEMM + fused SpMM on the local rows, then one
  ``all_gather`` of the layer output.  The last of them is "the" all-gather of node embeddings; afterwards every
  rank holds X_node and derives Z = X W_rx^T + b_r locally.
* Candidate pairs are independent units: a batch is split by pair index, no collective on the pair path
  (weak scaling: per-GPU batch fixed).

xGMI is point-to-point (7 links x ~153 GB/s per GPU); RCCL picks the algorithm, the message is one contiguous
[rows, D] fp32 block per rank, so the all-gather is a single collective per layer (no bucketing needed).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


# Test knob: with a process group of ONE rank still issue the (then trivial) collectives and walk the sharded encoder
# branches -- a 1-GPU box can load librccl and run the very call path the N > 1 runs take (tests/test_gpu_dist.py).
FORCE_COLLECTIVES = False


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default group if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("LPF_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
        if backend is None:  # (LPF_DIST_BACKEND=gloo: functional runs of the N > 1 path with several ranks on ONE GPU)
            backend = os.environ.get("LPF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def row_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row block of rank `rank` (first n % world ranks get one extra row)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pairs(batch: torch.Tensor, world: int, rank: int) -> torch.Tensor:
    """Columns [lo, hi) of a [2, BS] candidate batch owned by `rank` (contiguous split by pair index)."""
    lo, hi = row_range(batch.shape[1], world, rank)
    return batch[:, lo:hi]


def allgather_rows(local: torch.Tensor, n: int, group=None) -> torch.Tensor:
    """Assemble the full [n, D] matrix from every rank's row block (blocks follow ``row_range``)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES):
        return local
    world = dist.get_world_size(group)
    d = local.shape[1]
    full = torch.empty(n, d, dtype=local.dtype, device=local.device)
    sizes = [row_range(n, world, r) for r in range(world)]
    if n % world == 0:
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)  # one ncclAllGather
    else:  # ragged split: pad every block to the largest one, still a single collective
        rows = max(hi - lo for lo, hi in sizes)
        padded = torch.zeros(rows, d, dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
        buf = torch.empty(world * rows, d, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        for r, (lo, hi) in enumerate(sizes):
            full[lo:hi] = buf[r * rows:r * rows + (hi - lo)]
    return full


def gather_scores(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Concatenate per-rank score vectors in rank order (only needed when one rank wants the whole batch)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    return allgather_rows(local.reshape(-1, 1), total, group).reshape(-1)


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVES):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def measure_allgather_gbps(n: int, d: int, device, reps: int = 3, group=None) -> float:
    """Payload rate (GB/s of the assembled [n, d] fp32 matrix per second) of ``allgather_rows`` on this process
    group, measured: the one number the encoder plan needs."""
    import time
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
        return float("inf")
    rank = dist.get_rank(group)
    lo, hi = row_range(n, world, rank)
    local = torch.zeros(hi - lo, d, dtype=torch.float32, device=device)
    allgather_rows(local, n, group)  # warm-up (connection set-up)
    if local.is_cuda:
        torch.cuda.synchronize()
    dist.barrier(group)
    t0 = time.perf_counter()
    for _ in range(reps):
        allgather_rows(local, n, group)
    if local.is_cuda:
        torch.cuda.synchronize()
    dt = max_over_ranks((time.perf_counter() - t0) / reps, device if local.is_cuda else None)
    return n * d * 4 / dt / 1e9


def encoder_plan(encoder_ms_one_gpu: float, n: int, d: int, n_layers: int, world: int, allgather_gbps: float,
                 last_agg_ms: float | None = None, node_keys_ms: float = 0.0, replicated_ms: float = 0.0) -> dict:
    """Cost model for the encoder + the per-node attention projection (Z) on ``world`` GPUs (all times in ms):
         replicated:  every rank runs all of it, no exchange           = enc + keys
         sharded:     1/world of every layer + (L + 1) all-gathers of an [n, d] fp32 matrix, Z / Y on every rank
                                                                       = enc / world + (L + 1) * ag + keys
         gather_once: layers 1..L-1 replicated, the last layer's aggregation and Z / Y on the rank's rows, ONE
                      all-gather of [X | Z] (2 d floats per node)  = enc - (1 - 1/world) * (last_agg + keys) + 2 * ag
    ``last_agg_ms``: the row-shardable part of the last layer on one GPU -- its aggregation (SpMM + epilogue), or the
    whole layer when it runs as one fused launch (default: 0.7 * enc / L).  A sharded encoder whose layers are all
    fused needs L all-gathers, not L + 1 (layer 0 reads the replicated features): the estimate is on the safe side;
    ``node_keys_ms``: the per-node projection(s) ``gather_once`` runs on the rank's row block (Z);  ``replicated_ms``:
    per-node work EVERY rank does in full whatever the layout (the query table Y = X W_l^T + b_l with
    ``query_from = "table"``: only [X | Z] travels in the one all-gather) -- a constant of all three estimates.
    Returns the estimates and the cheapest mode.
    The reference has no counterpart (single device)."""
    if last_agg_ms is None:
        last_agg_ms = 0.7 * encoder_ms_one_gpu / max(n_layers, 1)
    repl = encoder_ms_one_gpu + node_keys_ms + replicated_ms
    if world <= 1:
        return {"mode": "replicated", "replicated_ms": repl, "sharded_ms": repl, "gather_once_ms": repl,
                "allgather_ms": 0.0}
    ag_ms = n * d * 4 / (allgather_gbps * 1e9) * 1e3
    sharded = encoder_ms_one_gpu / world + (n_layers + 1) * ag_ms + node_keys_ms + replicated_ms
    once = (encoder_ms_one_gpu - (1.0 - 1.0 / world) * (last_agg_ms + node_keys_ms) + node_keys_ms + 2.0 * ag_ms +
            replicated_ms)
    best = min((repl, "replicated"), (sharded, "sharded"), (once, "gather_once"))[1]
    return {"mode": best, "replicated_ms": repl, "sharded_ms": sharded, "gather_once_ms": once, "allgather_ms": ag_ms}
