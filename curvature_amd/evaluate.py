"""Bayesian-network inference loop: ``eval_nn`` / ``eval_bnn`` of the reference's ``scripts/evaluate.py``
(:88-152) on the MI355X estimators.

``eval_bnn`` is the deployed hot loop of the path: per Monte-Carlo sample one ``sample_and_replace()`` (two
batched GEMM launches + one batched copy here) and a forward pass over the data; the predictive
distribution is the mean of the per-sample softmax outputs.  Unlike the reference, probabilities are
accumulated on the device and cross to the host once at the end (the reference concatenates logits batch
by batch and converts every sample to numpy).  SURVEY.md section 8(f), rank 3.

``overlap=True`` software-pipelines the loop: the model's state tensors get a second buffer set, and weight sample
k + 1 (and, with a layer shard, its all-gather) is produced on a second HIP stream into the set the forward sweep of
sample k is NOT reading.  Two events per set order the streams (sample written -> forward may read; forward done ->
next sample may overwrite).  The noise stream and every launch are the serial loop's, so the predictions are the
same bit for bit (tests/test_estimator_chain_gpu.py).  Measured on one MI355X (tools/bench_bnn_loop.py, KFAC, ms per
Monte-Carlo sample, serial -> overlapped): ResNet-50 batch 32: 6.36 -> 6.28; batch 32 x 4 sweeps: 21.6 -> 21.8; batch
256: 37.9 -> 37.9; LeNet-5 batch 100: 0.30 -> 0.64.  On one GPU there is nothing to win: a throughput-bound forward
sweep leaves no idle CUs for the 1.5 ms of sampling work, a launch-bound one (batch 32) is bound by the same Python
thread that enqueues the sample, and a tiny model pays for re-pointing its state tensors.  The default is therefore
the serial loop, and the pipelined one is switched on only for a layer-sharded estimator, where it takes the
all-gather of the sampled parameters (a wait on the other ranks, not local work) off the forward stream.
"""
from typing import Iterable, List, Tuple

import torch

from . import ops


def eval_nn(model: torch.nn.Module, dataset: Iterable, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Softmax predictions (N, classes) and labels (N,) over `dataset` (scripts/evaluate.py:88-119), as
    tensors on `device` / CPU respectively."""
    if device is None:
        device = next(model.parameters()).device
    model.eval()
    probs, labels_all = [], []
    with torch.no_grad():
        for images, labels in dataset:
            logits = model(images.to(device, non_blocking=True))
            probs.append(torch.softmax(logits, dim=1))
            if labels is not None:
                labels_all.append(labels.cpu() if isinstance(labels, torch.Tensor) else torch.as_tensor(labels))
    predictions = torch.cat(probs) if probs else torch.empty(0, device=device)
    labels = torch.cat(labels_all) if labels_all else torch.empty(0, dtype=torch.long)
    return predictions, labels


class _StateBufferSets:
    """Two storage sets for every state tensor of a model (parameters and buffers); `activate(i)` points the
    model at set i.  Set 0 is the storage the model came with."""

    def __init__(self, model: torch.nn.Module):
        self.live: List[torch.Tensor] = list(model.state_dict(keep_vars=True).values())
        first = [t.data for t in self.live]
        self.sets = [first, [t.clone() for t in first]]

    def activate(self, index: int) -> None:
        for t, buf in zip(self.live, self.sets[index]):
            t.data = buf

    def copy(self, dst: int, src: int) -> None:
        pairs = [(d, s_) for d, s_ in zip(self.sets[dst], self.sets[src]) if d.numel()]
        batched = [(d, s_) for d, s_ in pairs if d.is_contiguous() and s_.is_contiguous()]
        ops.CopyPlan([d for d, _ in batched], [s_ for _, s_ in batched]).run()      # one launch group, any dtype
        for d, s_ in pairs:
            if not (d.is_contiguous() and s_.is_contiguous()):
                d.copy_(s_)


_SIDE_STREAMS = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = _SIDE_STREAMS.get(index)
    if stream is None:
        stream = _SIDE_STREAMS[index] = torch.cuda.Stream(device)
    return stream


def _drop_plans_for(estimator, tensors) -> None:
    """Forget the estimator's (and its shard's) cached launch plans that mention any of `tensors` by address."""
    ptrs = {t.data_ptr() for t in tensors if t.numel()}

    def mentions(key) -> bool:
        if isinstance(key, (tuple, list)):
            return any(mentions(k) for k in key)
        return isinstance(key, int) and key in ptrs
    for owner, name in ((estimator, "_sample_plan_cache"), (estimator, "_reload_plans"),
                        (getattr(estimator, "shard", None), "_plans")):
        cache = getattr(owner, name, None) if owner is not None else None
        if isinstance(cache, dict):
            for key in [k for k in cache if mentions(k)]:
                del cache[key]


def eval_bnn(model: torch.nn.Module, dataset: Iterable, estimator, samples: int = 30, device=None,
             overlap: bool = None, samples_per_launch: int = 1):
    """Mean predictive distribution over `samples` posterior weight samples (scripts/evaluate.py:121-152,
    ``stats=False`` path).  Returns ``(mean_predictions, labels)`` as numpy arrays like the reference.
    The model is left at the last sampled weights, as in the reference, in its original storage.

    `overlap`: produce sample k + 1 on a second stream while the forward sweep of sample k runs (see the module
    docstring for what that does and does not buy); default: only for a layer-sharded estimator on the GPU.
    ``overlap=False`` is the reference's serial loop.

    `samples_per_launch` = S > 1: the weight samples are produced S at a time (`sample_many`: for KFAC two GEMM launches
    for S parameter sets, the triangular factors streamed once per S samples; the other estimators file S ordinary samples
    away) and loaded into the model one by one (`replace_from`); serial loop only."""
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    if overlap is None:
        shard = getattr(estimator, "shard", None)
        overlap = device.type == "cuda" and shard is not None and shard.world > 1
    model.eval()
    mean_predictions = None
    labels = None
    with torch.no_grad():
        if samples_per_launch > 1:
            done = 0
            while done < samples:
                group = min(int(samples_per_launch), samples - done)
                bank = estimator.sample_many(group)
                for k in range(group):
                    estimator.replace_from(bank, k)
                    predictions, labels = eval_nn(model, dataset, device)
                    mean_predictions = predictions if mean_predictions is None else mean_predictions + predictions
                done += group
        elif not overlap or samples < 2:
            for _ in range(samples):
                estimator.sample_and_replace()
                predictions, labels = eval_nn(model, dataset, device)
                mean_predictions = predictions if mean_predictions is None else mean_predictions + predictions
        else:
            main = torch.cuda.current_stream(device)
            side = _side_stream(device)         # one per device for the life of the process: scratch workspaces are
            sets = _StateBufferSets(model)      # cached per stream, a fresh stream per call would pin a new set each time
            written = [torch.cuda.Event(), torch.cuda.Event()]     # sample is complete in set i
            consumed = [torch.cuda.Event(), torch.cuda.Event()]    # the forward sweep has finished reading set i
            side.wait_stream(main)                                 # invert() etc. enqueued by the caller
            with torch.cuda.stream(side):
                estimator.sample_and_replace()                     # sample 0 -> set 0
                written[0].record(side)
            cur = 0
            for k in range(samples):
                cur, nxt = k % 2, 1 - k % 2
                if k + 1 < samples:
                    sets.activate(nxt)
                    with torch.cuda.stream(side):
                        if k >= 1:
                            side.wait_event(consumed[nxt])         # sweep k - 1 read set nxt
                        estimator.sample_and_replace()             # sample k + 1 -> set nxt, concurrent with sweep k
                        written[nxt].record(side)
                sets.activate(cur)
                main.wait_event(written[cur])
                predictions, labels = eval_nn(model, dataset, device)
                consumed[cur].record(main)
                mean_predictions = predictions if mean_predictions is None else mean_predictions + predictions
            if cur != 0:                                           # leave the last sample in the original storage
                sets.copy(0, 1)
                sets.activate(0)
            main.wait_stream(side)
            # the launch plans described for the temporary second buffer set keep a whole model copy alive: drop them
            _drop_plans_for(estimator, sets.sets[1])
        mean_predictions = mean_predictions / samples
    return mean_predictions.cpu().numpy(), labels.numpy()
