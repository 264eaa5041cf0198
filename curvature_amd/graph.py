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

from .curvatures import KFAC


class KFACStepGraph:
    def __init__(self, kfac: KFAC, add=0.5, multiply=1.0, batch_size: Optional[int] = None, warmup: int = 3):
        self.kfac, self.add, self.multiply, self.batch_size = kfac, add, multiply, batch_size
        dev = next(kfac.model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("KFACStepGraph needs the model on an MI355X")
        kfac.use_device_noise_counter(True)
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

    def _step(self):
        self.kfac.update(self.batch_size)
        self.kfac.invert(self.add, self.multiply, check=False)
        self.kfac.sample_and_replace()

    def _record_addresses(self):
        return tuple(t.data_ptr() for pair in self.kfac.record.values() for t in pair if t is not None)

    def record_is_static(self) -> bool:
        """True if the hooks' recorded tensors sit where they sat at capture time (the replay reads those addresses)."""
        return self._record_addresses() == self._addresses

    def replay(self):
        self.graph.replay()

    def check(self):
        self.kfac.check_invert()
