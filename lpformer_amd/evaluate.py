"""Evaluation sweep over many candidate pairs: the reference's ``test_edge`` /
``test_heart_negatives`` / ``test_edge_citation2`` loops (src/train/testing.py:14-121) restructured for the device:

* the encoder runs ONCE per sweep (``test_edge`` re-runs it for every batch through ``model(edge)``,
  testing.py:87 -> link_transformer.py:100);
* batches are issued round-robin over a few HIP streams, so the selection kernels of one batch run underneath the
  matrix-core kernels of the previous one (per-stream workspaces in ``LinkTransformer``);
* a very long sweep (hundreds of batches: HeaRT / citation2 negatives) replays RECORDED steps
  (``lpformer_amd.PlannedScorer``, one per stream): ``score_pairs`` spends ~0.13 ms of host time per batch between its
  launches, more than the device needs at D = 64;
* scores stay on the device -- no ``.cpu()`` per batch (testing.py:88,117); the ranking metrics below
  (src/train/evaluation.py:23-50 and the OGB ``hits@K`` rule) are a few reductions over them.

The metric helpers are plain tensor reductions over at most a few million scores (host-layer plumbing, any device).
"""
from __future__ import annotations

from typing import Optional

import torch


def _as_2xp(edges: torch.Tensor) -> torch.Tensor:
    edges = torch.as_tensor(edges)
    if edges.dim() != 2 or 2 not in edges.shape:
        raise ValueError("edges must be [P, 2] (the reference's split layout) or [2, P]")
    return edges.t() if edges.shape[1] == 2 and edges.shape[0] != 2 else edges


# Full batches per stream from which recording the step pays.  Measured (tools/sweep_rate.py, 8 streams): recording the
# eight plans costs ~10 ms per sweep; a replayed batch saves 0.015 ms (collab-like: the device needs 0.19 of the 0.206 ms
# an eager batch takes), 0.04 ms (ppa-like), 0.08 ms (citation2-like) -- break-even at 700 / 260 / 130 batches.  Sweeps
# of that length are the HeaRT and citation2 negatives (testing.py:95-121), not the few batches of an ogbl-collab split.
PLAN_MIN_BATCHES = 64


def _planned_sweep(model, score_func, batch, out, h, batch_size, n_full, test_set, streams, logits) -> bool:
    """The first ``n_full`` full batches of a sweep through recorded steps, one ``PlannedScorer`` per stream.  Returns
    False (nothing written) when the step of this configuration cannot be recorded."""
    from .graphed import PlannedScorer
    nl = max(1, min(streams, n_full))
    main = torch.cuda.current_stream(model.device)
    lanes = model.lanes(nl)   # the model's persistent streams: every sweep finds the workspaces of the one before
    try:
        plans = [PlannedScorer(model, score_func, h, batch[:, k * batch_size:(k + 1) * batch_size], test_set=test_set,
                               logits=logits, adopt_input=True, stream=lanes[k]) for k in range(nl)]
    except RuntimeError:
        return False
    jobs = [[] for _ in plans]
    for p in plans:
        p.stream.wait_stream(main)   # h, batch and out are ready
    for i in range(n_full):
        lo, p = i * batch_size, plans[i % nl]
        jobs[i % nl].append(lo)
        res = p(batch[:, lo:lo + batch_size], validate=False, ordered=False)
        with torch.cuda.stream(p.stream):
            out[lo:lo + batch_size].copy_(res, non_blocking=True)
    # as below: one status read per stream at the end; a stream that reports an overflow scores its batches again, one at
    # a time with the status checked after each (check() re-records with a workspace sized for the offending batch)
    for p, lane_jobs in zip(plans, jobs):
        if p.check():
            continue
        for lo in lane_jobs:
            for _attempt in range(4):
                res = p(batch[:, lo:lo + batch_size], validate=False, ordered=False)
                with torch.cuda.stream(p.stream):
                    out[lo:lo + batch_size].copy_(res, non_blocking=True)
                if p.check():
                    break
            else:
                raise RuntimeError("score_edges: the selection workspace could not be sized")
    for p in plans:
        main.wait_stream(p.stream)
    return True


@torch.no_grad()
def score_edges(model, score_func, edges, batch_size: int = 32768, *, h: Optional[torch.Tensor] = None,
                test_set: bool = False, streams: int = 4, logits: bool = False, plans: Optional[bool] = None) -> torch.Tensor:
    """Probabilities (or pre-sigmoid logits) for every pair of ``edges``, as one device tensor of shape [P].

    Same arithmetic per pair as ``score_func(model(edge, test_set=test_set))`` of the reference loop; ``h`` (the encoder
    output, ``model.propagate(test_set=...)``) is computed once if not given.  ``plans``: replay recorded steps for the
    full batches (default: when the sweep has at least ``PLAN_MIN_BATCHES`` of them per stream); the scores are bitwise
    the ones of the eager loop."""
    dev = model.device
    batch = _as_2xp(edges).to(dev)
    if batch.dtype != torch.int64:
        batch = batch.long()
    batch = batch.contiguous()
    total = batch.shape[1]
    if h is None:
        h = model.propagate(test_set=test_set)
    out = torch.empty(total, dtype=torch.float32, device=dev)
    if total == 0:
        return out
    main = torch.cuda.current_stream(dev)
    start = 0
    n_full = total // batch_size
    if getattr(model, "_multi_head", False):
        # (num_heads > 1 / two attention layers: layer by layer, head by head through train.py's pair_stage -- no recorded
        #  plan exists for it, and its selection reads its status back per batch: one lane)
        plans, streams = False, 1
    if plans is None:
        plans = n_full >= PLAN_MIN_BATCHES * max(1, min(streams, n_full))
    if plans and n_full > 0 and _planned_sweep(model, score_func, batch, out, h, batch_size, n_full, test_set, streams,
                                               logits):
        start = n_full * batch_size
        if start == total:
            return out
    lanes = model.lanes(max(1, min(streams, (total - start + batch_size - 1) // batch_size)))  # persistent: workspaces are per stream
    for s in lanes:
        s.wait_stream(main)  # h, batch and out are ready
    jobs = [[] for _ in lanes]
    for i, lo in enumerate(range(start, total, batch_size)):
        hi = min(lo + batch_size, total)
        jobs[i % len(lanes)].append((lo, hi))
        with torch.cuda.stream(lanes[i % len(lanes)]):
            out[lo:hi] = model.score_pairs(batch[:, lo:hi], h, score_func, test_set=test_set, logits=logits)
    # The selection of a batch is sized from EARLIER batches of its lane and nothing is read back while the sweep is
    # queued: a batch that outgrows its workspace (hub-heavy negatives after a sparse start) comes back as NaN and
    # leaves a sticky status on the lane.  Read every lane's status once, at the end, and score the batches of a lane
    # that reports an overflow again, one at a time with the status checked after each (the first of them re-sizes
    # the workspace) -- never a NaN, or a silently wrong metric, out of this function.
    for lane, lane_jobs in zip(lanes, jobs):
        if model.check_selection(lane):
            continue
        with torch.cuda.stream(lane):
            for lo, hi in lane_jobs:
                for _attempt in range(4):
                    out[lo:hi] = model.score_pairs(batch[:, lo:hi], h, score_func, test_set=test_set, logits=logits)
                    if model.check_selection(lane):
                        break
                else:
                    raise RuntimeError("score_edges: the selection workspace could not be sized")
    for s in lanes:
        main.wait_stream(s)
    return out


@torch.no_grad()
def score_negatives(model, score_func, negatives, batch_size: int = 32768, **kw) -> torch.Tensor:
    """HeaRT-style negatives [P, K, 2] -> scores [P, K] (``test_heart_negatives``, testing.py:95-121)."""
    negatives = torch.as_tensor(negatives)
    p, k = negatives.shape[0], negatives.shape[1]
    return score_edges(model, score_func, negatives.reshape(-1, 2), batch_size, **kw).view(p, k)


def hits_at_k(pos: torch.Tensor, neg: torch.Tensor, k: int) -> float:
    """OGB ``hits@K`` (what ``evaluate_hits`` asks the ogb Evaluator for, evaluation.py:7-18): the fraction of
    positive scores strictly above the K-th largest negative score; 1.0 when there are fewer than K negatives."""
    pos, neg = pos.reshape(-1), neg.reshape(-1)
    if neg.numel() < k:
        return 1.0
    kth = torch.topk(neg, k).values[-1]
    return float((pos > kth).float().mean().item()) if pos.numel() else float("nan")


def ranking_metrics(pos: torch.Tensor, neg: torch.Tensor) -> dict:
    """``evaluate_mrr`` (evaluation.py:23-50): per positive, rank among its own K negatives as the mean of the
    optimistic and the pessimistic rank; MRR and Hits@{10,50,100} averaged over the positives.
    pos [P], neg [P, K]."""
    pos = pos.reshape(-1, 1)
    optimistic = (neg >= pos).sum(dim=1)
    pessimistic = (neg > pos).sum(dim=1)
    rank = 0.5 * (optimistic + pessimistic).to(torch.float32) + 1.0
    return {"Hits@10": float((rank <= 10).float().mean().item()), "Hits@50": float((rank <= 50).float().mean().item()),
            "Hits@100": float((rank <= 100).float().mean().item()), "MRR": float((1.0 / rank).mean().item())}
