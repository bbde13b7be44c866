"""One scoring step captured in a HIP graph.

``LinkTransformer.score_pairs`` issues its launches (2 selection kernels, the q gather and the elementwise branch on a
side stream, the one-pass attention + its fix-up, the merged dense tail) without any host synchronisation or
per-launch host value, so a whole step can be captured once and replayed: the launch-bound part of small batches
(about nine launches) collapses into one graph launch.  Inputs and outputs are static tensors owned by the scorer.

The reference has nothing to mirror here (its loop is eager PyTorch, src/train/testing.py:86-117).
"""
from __future__ import annotations

import torch


class GraphedScorer:
    """``scorer(batch) -> scores`` for candidate batches of ONE fixed size against a fixed encoder output ``h``.

    The selection workspace is sized from ``example_batch`` (twice its entry counts); a later batch that does not fit
    comes back as NaN and raises the sticky status that ``model.check_selection(scorer.stream)`` reports."""

    def __init__(self, model, score_func, h: torch.Tensor, example_batch: torch.Tensor, test_set: bool = False,
                 logits: bool = False):
        self.model, self.h = model, h
        dev = model.device
        example = model._prep_batch(example_batch)
        self.batch = example.clone()
        self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(2):  # sizes the per-stream workspaces and fills every parameter-derived cache
                model.score_pairs(self.batch, h, score_func, test_set=test_set, logits=logits)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = model.score_pairs(self.batch, h, score_func, test_set=test_set, logits=logits)

    def __call__(self, batch: torch.Tensor) -> torch.Tensor:
        """Scores of ``batch`` ([2, BS] node ids, same BS as the example); the result tensor is reused by the next
        call."""
        if batch.shape != self.batch.shape:
            raise ValueError(f"this graph was captured for batches of shape {tuple(self.batch.shape)}")
        self.batch.copy_(batch, non_blocking=True)
        self.graph.replay()
        return self.out
