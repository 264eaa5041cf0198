"""Bayesian-network inference loop: ``eval_nn`` / ``eval_bnn`` of the reference's ``scripts/evaluate.py``
(:88-152) on the MI355X estimators.

``eval_bnn`` is the deployed hot loop of the path: per Monte-Carlo sample one ``sample_and_replace()`` (two
batched GEMM launches + one batched copy here) and a forward pass over the data; the predictive
distribution is the mean of the per-sample softmax outputs.  Unlike the reference, probabilities are
accumulated on the device and cross to the host once at the end (the reference concatenates logits batch
by batch and converts every sample to numpy).  SURVEY.md section 8(f), rank 3.
"""
from typing import Iterable, Tuple

import torch


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


def eval_bnn(model: torch.nn.Module, dataset: Iterable, estimator, samples: int = 30, device=None):
    """Mean predictive distribution over `samples` posterior weight samples (scripts/evaluate.py:121-152,
    ``stats=False`` path).  Returns ``(mean_predictions, labels)`` as numpy arrays like the reference.
    The model is left at the last sampled weights, as in the reference."""
    if device is None:
        device = next(model.parameters()).device
    model.eval()
    mean_predictions = None
    labels = None
    with torch.no_grad():
        for _ in range(samples):
            estimator.sample_and_replace()
            predictions, labels = eval_nn(model, dataset, device)
            mean_predictions = predictions if mean_predictions is None else mean_predictions + predictions
        mean_predictions = mean_predictions / samples
    return mean_predictions.cpu().numpy(), labels.numpy()
