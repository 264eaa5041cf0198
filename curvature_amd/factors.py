"""Monte-Carlo Fisher driver: the outer loop of the reference's ``scripts/factors.py`` (``compute_factors``,
:33-62) on the MI355X estimators.

Per batch: ONE forward pass, then ``samples`` times {labels ~ Categorical(logits), ``loss.backward(
retain_graph=True)``, ``est.update(batch_size)``}.  The reference rebuilds both Kronecker factors for every
label draw although the A side depends only on the layer inputs, which do not change between the draws;
here (``share_inputs=True``, KFAC only) A is built once per forward pass with weight ``samples`` and the
other draws update G only - the same accumulated factors (to fp32 rounding) for about half the factor-build
work.  SURVEY.md section 8(f), rank 1.
"""
from types import SimpleNamespace
from typing import Any, Callable, Iterable, Optional, Union

import torch

from . import curvatures as curv


def _as_args(args: Any, **kw) -> SimpleNamespace:
    """The reference passes an argparse namespace (scripts/factors.py:33: args.estimator, .samples, .epochs,
    .device, .verbose); keyword arguments are accepted as well and win."""
    base = dict(estimator="kfac", samples=1, epochs=1, device=None, verbose=False)
    if args is not None:
        for k in base:
            if hasattr(args, k):
                base[k] = getattr(args, k)
    base.update({k: v for k, v in kw.items() if v is not None})
    return SimpleNamespace(**base)


def compute_factors(args: Any,
                    model: Union[torch.nn.Module, torch.nn.Sequential],
                    data: Iterable,
                    factors=None,
                    *,
                    estimator: Optional[str] = None,
                    samples: Optional[int] = None,
                    epochs: Optional[int] = None,
                    device=None,
                    share_inputs: bool = True,
                    label_sampler: Optional[Callable] = None):
    """``compute_factors(args, model, data, factors=None)`` of the reference (scripts/factors.py:33-62).

    `args` may be the reference's argparse namespace or ``None`` with the keyword arguments.  `data` yields
    ``(images, labels)`` (the dataset labels are ignored, as in the reference).  `label_sampler(logits,
    batch_index, sample_index)` replaces ``Categorical(logits).sample()`` (parity tests)."""
    a = _as_args(args, estimator=estimator, samples=samples, epochs=epochs, device=device)
    dev = a.device if a.device is not None else next(model.parameters()).device
    model.train()
    criterion = torch.nn.CrossEntropyLoss().to(dev)
    est_base = getattr(curv, a.estimator.upper())
    if a.estimator == 'efb':
        est = est_base(model, factors)
    else:
        est = est_base(model)
    shared = share_inputs and isinstance(est, curv.KFAC) and a.samples > 1

    for _ in range(a.epochs):
        for batch, (images, _labels) in enumerate(data):
            logits = model(images.to(dev, non_blocking=True))
            dist = torch.distributions.Categorical(logits=logits)
            for sample in range(a.samples):
                labels = dist.sample() if label_sampler is None else label_sampler(logits, batch, sample)
                loss = criterion(logits, labels)
                model.zero_grad()
                loss.backward(retain_graph=True)
                if not shared:
                    est.update(images.size(0))
                elif sample == 0:
                    est.update(images.size(0), input_weight=float(a.samples))   # A once, weighted; G of this draw
                else:
                    est.update(images.size(0), inputs=False)                    # G only
    return est
