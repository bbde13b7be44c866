"""One scoring step captured in a HIP graph.

``LinkTransformer.score_pairs`` issues its launches (2 selection kernels, the q gather and the elementwise branch on a
side stream, the one-pass attention, the merged dense tail) without any host synchronisation or per-launch host value,
so a whole step can be captured once and replayed: the launch-bound part of small batches collapses into one graph
launch.  Inputs and outputs are static tensors owned by the scorer.

A captured graph holds RAW POINTERS.  Everything they point at is either allocated during capture (the graph's own
memory pool) or held by this object: the scorer runs on a stream of its own, so the model's per-stream workspaces
behind that stream are used by nobody else, and it keeps strong references to the per-encoder-output tables (Z and
the bf16 copy), the folded weight tables and the selection workspace it was captured with -- the model replacing its
caches (another encoder output, an optimiser step) cannot free memory a replay reads.  Replays are refused to go stale:
``__call__`` compares the version key of every parameter with the one of the capture and captures again when it
differs; ``check()`` reports a selection overflow and re-captures with a workspace sized for the batch that overflowed.

The reference has nothing to mirror here (its loop is eager PyTorch, src/train/testing.py:86-117).
"""
from __future__ import annotations

import torch


class GraphedScorer:
    """``scorer(batch) -> scores`` for candidate batches of ONE fixed size against a fixed encoder output ``h``.

    The selection workspace is sized from ``example_batch`` (twice its entry counts); a later batch that does not fit
    comes back as NaN: call ``scorer.check()`` before using the scores of a sweep (it synchronises)."""

    def __init__(self, model, score_func, h: torch.Tensor, example_batch: torch.Tensor, test_set: bool = False,
                 logits: bool = False, adopt_input: bool = False):
        """``adopt_input``: the example batch (an int64 [2, BS] device tensor) IS the static input -- no private copy;
        a later call with that very tensor replays without the device-to-device copy of the ids (the caller refills it
        in place, or, like the bench, keeps one resident batch per scorer)."""
        self.model, self.score_func, self.h = model, score_func, h
        self.test_set, self.logits = test_set, logits
        dev = model.device
        prepped = model._prep_batch(example_batch)
        self.batch = prepped if (adopt_input and prepped is example_batch) else prepped.clone()
        self.stream = torch.cuda.Stream(dev)   # private: the model's workspaces of this stream belong to the scorer
        self.captures = 0
        self._capture()

    def _param_key(self):
        # (the module trees are walked once per capture, not per replay: a module gaining or losing a parameter goes
        #  through __setattr__ / load_state_dict(assign=True), which the version / address pairs below also notice for
        #  every parameter that existed at the capture)
        return tuple((p.data_ptr(), p._version) for p in self._params) + (self.model.precision,
                                                                           self.model.tail_precision)

    def stale(self) -> bool:
        """True when a parameter (storage or version) or a precision switch changed since the capture."""
        return self._param_key() != self._key

    def _capture(self):
        model, dev = self.model, self.model.device
        self._params = list(model.parameters()) + list(self.score_func.parameters())
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(2):  # sizes the per-stream workspaces and fills every parameter-derived cache
                model.score_pairs(self.batch, self.h, self.score_func, test_set=self.test_set, logits=self.logits)
            if not model.check_selection(self.stream):   # (first call of a stream sizes exactly: cannot overflow)
                model.score_pairs(self.batch, self.h, self.score_func, test_set=self.test_set, logits=self.logits)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = model.score_pairs(self.batch, self.h, self.score_func, test_set=self.test_set,
                                         logits=self.logits)
        # strong references to everything the captured launches read or write outside the graph's own pool
        raw = self.stream.cuda_stream
        self._keep = (model._z_cache, getattr(model, "_y_cache", None), getattr(model, "_zb_cache", None), model._folded,
                      getattr(model, "_tail_cache", None), getattr(model, "_score_fold_cache", None),
                      [(k, w, getattr(w, "entries", None), getattr(w, "item_pair", None), getattr(w, "run_lb", None))
                       for k, w in model._ws.items() if isinstance(k, tuple) and raw in k])
        self._key = self._param_key()
        self.captures += 1

    def __call__(self, batch: torch.Tensor, validate: bool = True) -> torch.Tensor:
        """Scores of ``batch`` ([2, BS] node ids, same BS as the example); the result tensor is reused by the next
        call.  Re-captures first when a parameter changed since the capture.  ``validate=False`` skips that check
        (one pass over the parameters' version counters on the host): for sweeps in which the caller knows that no
        parameter changes -- it then owes one ``stale()`` per sweep."""
        if batch.shape != self.batch.shape:
            raise ValueError(f"this graph was captured for batches of shape {tuple(self.batch.shape)}")
        if validate and self.stale():
            self._capture()
        if batch is not self.batch:
            self.batch.copy_(batch, non_blocking=True)
        self.graph.replay()
        return self.out

    def check(self) -> bool:
        """Synchronising status check: True when every batch replayed since the last check fitted the captured selection
        workspace.  Otherwise the scores since then are NaN; the graph is captured again with a workspace sized for
        the batch currently held (the last one replayed) and False is returned -- the caller replays those batches."""
        if self.model.check_selection(self.stream):
            return True
        self._capture()
        return False
