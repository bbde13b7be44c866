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

from ._lib import check


class GraphedScorer:
    """``scorer(batch) -> scores`` for candidate batches of ONE fixed size against a fixed encoder output ``h``.

    The selection workspace is sized from ``example_batch`` (twice its entry counts); a later batch that does not fit
    comes back as NaN: call ``scorer.check()`` before using the scores of a sweep (it synchronises)."""

    def __init__(self, model, score_func, h: torch.Tensor, example_batch: torch.Tensor, test_set: bool = False,
                 logits: bool = False, adopt_input: bool = False, stream: "torch.cuda.Stream | None" = None):
        """``adopt_input``: the example batch (an int64 [2, BS] device tensor) IS the static input -- no private copy;
        a later call with that very tensor replays without the device-to-device copy of the ids (the caller refills it
        in place, or, like the bench, keeps one resident batch per scorer).  ``stream``: the stream the step lives on
        (default: a new one).  The model's per-stream workspaces behind it are the scorer's: give it a stream nothing
        else scores on while the scorer is in use (a caller that rebuilds scorers often hands the same few streams back
        instead of leaving a set of workspaces behind for every new one)."""
        if getattr(model, "_multi_head", False):
            raise NotImplementedError("recorded plans and captured graphs replay the single-head inference launches; a "
                                      "model with num_heads > 1 scores through model.score_pairs(...)")
        self.model, self.score_func, self.h = model, score_func, h
        self.test_set, self.logits = test_set, logits
        dev = model.device
        prepped = model._prep_batch(example_batch)
        self.batch = prepped if (adopt_input and prepped is example_batch) else prepped.clone()
        # private: the model's workspaces of this stream belong to the scorer
        self.stream = stream if stream is not None else torch.cuda.Stream(dev)
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


class _StepRecorder:
    """What ``_lib.recording`` reports while one ``score_pairs`` call runs: the C-ABI launches (entry point, arguments)
    and stream hand-overs in issue order, and every tensor whose address went into an argument."""

    def __init__(self, skip):
        self.skip, self.calls, self.kept = skip, [], []

    def launch(self, name, fn):
        if name in self.skip:   # (pure queries: nothing to replay)
            return fn

        def call(*args):
            self.calls.append((name, fn, args))
            return fn(*args)
        return call

    def keep(self, t):
        self.kept.append(t)

    def wait(self, waiter, waited):
        self.calls.append((None, None, (waiter, waited)))


class PlannedScorer(GraphedScorer):
    """The same fixed-size step as a RECORDED LIST OF C-ABI LAUNCHES, replayed one ``hipLaunchKernel`` after the other.

    Why beside the HIP graph: ``score_pairs`` spends 0.13-0.15 ms of host time per step between its five launches (workspace
    look-ups, cache keys, pointer conversions), more than the device needs at D <= 128 once batches are pipelined over
    several streams -- the eager loop is bound by the host; and replayed graphs, which cost the host 0.04 ms per step,
    run their nodes with less overlap between streams than plain launches do (measured: tools/host_floor.py,
    tools/marginal_cost.py).  A plan keeps the launches plain and takes the host out: one recorded ``score_pairs`` call
    gives the entry points, their argument tuples (workspaces, tables and temporaries of that call, all kept alive by
    the plan) and the two stream hand-overs of the side stream; a replay calls them again, with the ids pointer
    replaced by the new batch's -- ~0.03 ms of host time per step.

    Same contract as ``GraphedScorer`` (fixed batch size and encoder output, ``check()`` before the scores of a sweep are
    used, re-recorded when a parameter changed), except that the launches always go to ``scorer.stream`` (and the model's
    side stream of it); ``__call__`` orders them against the caller's current stream unless told not to.  Only the C-ABI launches of the step
    are replayed -- which is all a step consists of on the paths ``score_pairs`` takes for the two-layer score head;
    the recording is checked against the eager result, bit for bit, before it is used."""

    _SKIP = frozenset(("lpf_strerror", "lpf_last_hip_error", "lpf_pair_rows_piece_floats", "lpf_abi_version",
                       "lpf_device_info", "lpf_select_plan_blocks", "lpf_train_partial_blocks"))

    def _capture(self):
        from . import _lib
        model, dev = self.model, self.model.device
        self._params = list(model.parameters()) + list(self.score_func.parameters())
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        kw = dict(test_set=self.test_set, logits=self.logits)
        with torch.cuda.stream(self.stream):
            for _ in range(2):  # sizes the per-stream workspaces and fills every parameter-derived cache
                model.score_pairs(self.batch, self.h, self.score_func, **kw)
            if not model.check_selection(self.stream):   # (first call of a stream sizes exactly: cannot overflow)
                model.score_pairs(self.batch, self.h, self.score_func, **kw)
            with _lib.recording(_StepRecorder(self._SKIP)) as rec:
                self.out = model.score_pairs(self.batch, self.h, self.score_func, **kw)
            calls, keep = rec.calls, rec.kept
            want = self.out.clone()
        b = self.batch   # (its extent in memory: a [2, BS] window of a longer id list has its rows far apart)
        base = b.data_ptr()
        nbytes = (sum((n - 1) * st for n, st in zip(b.shape, b.stride())) + 1) * b.element_size()
        # arguments that point INTO the ids (the tensor itself, its second row, ...) follow the batch of a replay; such a
        # value can only sit in a pointer slot: ints of other meaning are far below any device address
        self._plan = [(name, fn, args, tuple(i for i, a in enumerate(args)
                                             if fn is not None and isinstance(a, int) and base <= a < base + nbytes))
                      for name, fn, args in calls]
        raw = self.stream.cuda_stream
        self._keep = (keep, model._z_cache, getattr(model, "_y_cache", None), getattr(model, "_zb_cache", None),
                      model._folded, getattr(model, "_tail_cache", None), getattr(model, "_score_fold_cache", None),
                      [(k, w, getattr(w, "entries", None), getattr(w, "item_pair", None), getattr(w, "run_lb", None))
                       for k, w in model._ws.items() if isinstance(k, tuple) and raw in k])
        self._key = self._param_key()
        self.captures += 1
        self._replay(base)
        self.stream.synchronize()
        if not torch.equal(self.out, want):
            raise RuntimeError("PlannedScorer: the recorded launches do not reproduce the eager step (the step of this "
                               "configuration does work outside the C-ABI launches)")

    def _replay(self, ids_ptr: int):
        base = self.batch.data_ptr()
        for name, fn, args, slots in self._plan:
            if fn is None:
                args[0].wait_stream(args[1])
                continue
            if slots and ids_ptr != base:
                args = list(args)
                for i in slots:
                    args[i] += ids_ptr - base
            rc = fn(*args)
            if rc:
                check(rc, name)

    def __call__(self, batch: torch.Tensor, validate: bool = True, ordered: bool = True) -> torch.Tensor:
        """Scores of ``batch`` ([2, BS] int64 node ids on the device, same BS as the example), queued on
        ``scorer.stream``; the result tensor is reused by the next call.  ``ordered`` (default): the plan's stream first
        waits for the caller's current stream (whatever produced ``batch`` there is complete before the ids are read) and
        the caller's stream then waits for the plan's (the scores are complete for whatever the caller queues next) --
        the call behaves like ``score_pairs`` on the current stream.  ``ordered=False`` drops both hand-overs: for
        pipelines over several scorers whose ids have long been resident and whose consumer synchronises by itself."""
        if batch.shape != self.batch.shape:
            raise ValueError(f"this plan was recorded for batches of shape {tuple(self.batch.shape)}")
        if validate and self.stale():
            self._capture()
        cur = torch.cuda.current_stream(self.model.device) if ordered else None
        if ordered:
            self.stream.wait_stream(cur)
        self._last = batch
        if batch is self.batch:
            self._replay(self.batch.data_ptr())
        elif (batch.dtype == self.batch.dtype and batch.device == self.batch.device and
              batch.stride() == self.batch.stride()):
            # the ids are read where they are: the plan's stream must see them complete
            self._replay(batch.data_ptr())
        else:
            with torch.cuda.stream(self.stream):
                self.batch.copy_(batch, non_blocking=True)
            self._replay(self.batch.data_ptr())
        if ordered:
            cur.wait_stream(self.stream)
        return self.out

    def check(self) -> bool:
        """As ``GraphedScorer.check``; after an overflow the step is recorded again with a workspace sized for the LAST
        batch replayed (the plan takes a copy of it as its own ids; an adopted input tensor is left alone)."""
        if self.model.check_selection(self.stream):
            return True
        last = getattr(self, "_last", None)
        if last is not None and last is not self.batch:
            with torch.cuda.stream(self.stream):
                self.batch = self.model._prep_batch(last).clone()
        self._capture()
        return False
