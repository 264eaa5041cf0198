"""HIP-graph replay of the KFAC step for launch-bound models (BASELINE config 2: LeNet-5 is ~20 launches of a
few microseconds each per ``update + invert + sample_and_replace``; scripts/test.py:29-53 is that loop).

Every launch of the step takes its operands from kernel arguments (descriptor tables travel by value), runs on the
caller's stream (side streams fork from and join back into it through events) and needs no host synchronisation once
``invert(check=False)`` leaves the status words on the device - so the whole step can be captured once and replayed
while the tensors involved stay where they are:

    step = KFACStepGraph(kfac, add=0.5, multiply=1.0)        # warms up, captures
    for images in data:
        forward / backward                                    # records new inputs / gradients IN PLACE? no: see below
        step.replay()
    step.check()                                              # status words of the last replay

The captured ``update()`` reads the activation / gradient tensors that were recorded at capture time, BY ADDRESS: the
caller must produce the new batch in the same buffers (e.g. a captured forward/backward with static inputs, the usual
CUDA-graph training pattern) - `record_is_static()` tells whether the hooks saw the same addresses again.  The noise
stream position lives in a device word (`Curvature.use_device_noise_counter`), so every replay draws fresh noise and
the draws are the ones the eager path would have made.
"""
from typing import Optional

import torch

from . import ops
from .curvatures import KFAC


class KFACStepGraph:
    """Captures ``update + invert(check=False) + sample_and_replace`` of `kfac` once; `replay()` runs it again.

    Construction runs the step ``max(warmup, 2)`` times for real (launch plans, workspaces, side streams and the
    factors' `first` flags settle there) and once more under capture.  None of that is allowed to leak into the
    estimator: the Kronecker factors (a running SUM that `invert()` uses directly) and the position of the noise stream
    are snapshotted before the warm-up and restored after the capture, so that ``KFACStepGraph(...)`` followed by N
    replays leaves exactly N accumulated batches and N draws.  An estimator without factors yet gets zero factors (the
    captured `update()` accumulates; the first replay then equals the eager first update).  The parameters hold
    ``mean + sample`` of the last warm-up draw afterwards, as after any `sample_and_replace()`.

    The graph holds raw addresses: the sample plans, their scratch, the per-stream workspaces and the status words it
    was captured with are kept alive by this object (``eval_bnn`` on the same estimator evicts plans from the estimator's
    cache, `ops.release_workspaces()` empties the workspace cache: neither frees what the graph addresses), and
    `replay()` refuses to run if the estimator's factors, inverse factors or parameters are no longer the tensors it was
    captured with."""

    def __init__(self, kfac: KFAC, add=0.5, multiply=1.0, batch_size: Optional[int] = None, warmup: int = 3):
        self.kfac, self.add, self.multiply, self.batch_size = kfac, add, multiply, batch_size
        dev = next(kfac.model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("KFACStepGraph needs the model on an MI355X")
        had_state = bool(kfac.state)
        snapshot = {layer: [t.clone() for t in pair] for layer, pair in kfac.state.items()}
        kfac.use_device_noise_counter(True)
        counter0 = kfac._noise_counter.clone()
        self._addresses = self._record_addresses()
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(warmup, 2)):          # plans, workspaces, side streams and `first` flags settle here
                self._step()
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
            self._step()
        # undo what the warm-up steps did to the estimator (in place: the graph reads these addresses)
        with torch.cuda.stream(self.stream):
            for layer, pair in kfac.state.items():
                for i, t in enumerate(pair):
                    if had_state and layer in snapshot:
                        t.copy_(snapshot[layer][i])
                    else:
                        t.zero_()
            kfac._noise_counter.copy_(counter0)
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        # strong references to everything the captured launches address
        self._info = kfac._invert_info
        self._plans = dict(kfac._sample_plans())
        with ops._workspace_lock:
            self._workspaces = dict(ops._workspaces)
        self._tensors = [t for pair in kfac.state.values() for t in pair] + [t for pair in kfac.inv_state.values() for t in pair]
        self._state_addresses = self._state_ptrs()

    def _step(self):
        self.kfac.update(self.batch_size)
        self.kfac.invert(self.add, self.multiply, check=False)
        self.kfac.sample_and_replace()

    def _record_addresses(self):
        return tuple(t.data_ptr() for pair in self.kfac.record.values() for t in pair if t is not None)

    def _state_ptrs(self):
        k = self.kfac
        return (tuple(t.data_ptr() for pair in k.state.values() for t in pair),
                tuple(t.data_ptr() for pair in k.inv_state.values() for t in pair),
                tuple(p.data_ptr() for p in k.model.parameters()))

    def record_is_static(self) -> bool:
        """True if the hooks' recorded tensors sit where they sat at capture time (the replay reads those addresses)."""
        return self._record_addresses() == self._addresses

    def replay(self):
        if self._state_ptrs() != self._state_addresses:
            raise RuntimeError("KFACStepGraph.replay: factors, inverse factors or parameters of the estimator were "
                               "re-allocated since the capture (the graph holds their old addresses)")
        # (scratch workspaces, sample plans and status words cannot go away: this object holds them - a regrown or
        # released workspace of the cache only means that eager calls use another buffer than the graph)
        self.graph.replay()
        self.kfac._invert_info = self._info          # check() / check_invert() read the captured status words

    def check(self):
        """Raise ``RuntimeError`` if the LAST REPLAY met a factor that is not positive definite (the status words the
        graph was captured with; an eager `invert()` in between has its own)."""
        ops.check_chol_info(self._info)
