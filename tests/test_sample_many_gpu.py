"""Batched multi-sample generation (scripts/evaluate.py:134-139 draws one weight sample per forward sweep):
KFAC.sample_many(S) produces S parameter sets in two GEMM launches; parity with the oracle's per-sample restatement of
curvatures.py:387-392 + :77-82 for caller-supplied noise, and the BNN loop built on it."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_fro
from test_estimator_chain_gpu import backward, lenet, load

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _kfac(gpu):
    from curvature_amd.curvatures import KFAC
    g1 = load("g1_kfac_lenet.npz")
    model, layers = lenet(gpu, g1)
    kfac = KFAC(model)
    for b in range(2):
        kfac.update(backward(model, g1, b, gpu))
    kfac.invert(add=0.5, multiply=2.0)
    return model, layers, kfac


def test_sample_many_matches_the_oracle_sample_by_sample(gpu):
    import oracle.curvature_oracle as o
    model, layers, kfac = _kfac(gpu)
    S = 5
    torch.manual_seed(11)
    noise = {l: torch.randn(S, kfac.inv_state[l][0].shape[0], kfac.inv_state[l][1].shape[0], device=gpu) for l in layers}
    bank = kfac.sample_many(S, noise=noise)
    torch.cuda.synchronize()
    assert bank.count == S
    for layer in layers:
        L_A, L_G = (t.double().cpu() for t in kfac.inv_state[layer])
        w_mean = kfac.model_state_of(layer, "weight").double().cpu()
        b_mean = kfac.model_state_of(layer, "bias").double().cpu()
        for s in range(S):
            sample = o.kfac_sample(L_A, L_G, noise[layer][s].double().cpu())            # (m, n), bias = last column
            w_ref, b_ref = o.replace(sample, w_mean, b_mean)
            assert rel_fro(bank.weights[layer][s].view(layer.weight.shape), w_ref) < TOL
            assert rel_fro(bank.biases[layer][s], b_ref) < TOL
    # loading a set: exactly the bank's values in the parameters, everything else at its mean
    kfac.replace_from(bank, 3)
    for layer in layers:
        assert torch.equal(layer.weight.data, bank.weights[layer][3].view(layer.weight.shape))
        assert torch.equal(layer.bias.data, bank.biases[layer][3])
    # and it agrees with the one-sample path fed the same noise (different association of the two products: ~1e-7)
    kfac.sample_and_replace(noise={l: noise[l][3] for l in layers})
    for layer in layers:
        assert rel_fro(layer.weight.data, bank.weights[layer][3].view(layer.weight.shape)) < 1e-5


def test_sample_many_device_noise_is_standard_normal_and_fresh(gpu):
    model, layers, kfac = _kfac(gpu)
    kfac.noise_seed = 77
    bank = kfac.sample_many(64)
    first = {l: bank.weights[l].clone() for l in layers}
    layer = layers[2]                                           # Linear(400, 120): 64 x 120 x 400 draws
    dev = (first[layer] - kfac.model_state_of(layer, "weight").view(1, *first[layer].shape[1:]))
    assert abs(float(dev.mean())) < 5e-3 * float(dev.std())     # zero-mean fluctuations around the MAP weights
    bank2 = kfac.sample_many(64)                                # same buffers, new draws
    assert not torch.equal(bank2.weights[layer], first[layer])
    cov_ratio = float(bank2.weights[layer].var(0).mean() / first[layer].var(0).mean())
    assert 0.9 < cov_ratio < 1.1


def test_eval_bnn_with_samples_per_launch(gpu):
    from curvature_amd.evaluate import eval_bnn, eval_nn
    model, layers, kfac = _kfac(gpu)
    torch.manual_seed(3)
    data = [(torch.rand(16, 1, 28, 28), torch.arange(16) % 10) for _ in range(2)]
    ptrs = [p.data_ptr() for p in model.parameters()]
    kfac.noise_seed, kfac.noise_offset = 4321, 0
    pred, labels = eval_bnn(model, data, kfac, samples=7, device=gpu, samples_per_launch=4)      # groups of 4 + 3
    assert [p.data_ptr() for p in model.parameters()] == ptrs
    assert pred.shape == (32, 10) and abs(float(pred.sum(1).mean()) - 1.0) < 1e-5
    # the same draws by hand: two banks (4 and 3 sets) from the same noise stream position
    kfac.noise_seed, kfac.noise_offset = 4321, 0
    total = None
    with torch.no_grad():
        for group in (4, 3):
            bank = kfac.sample_many(group)
            for k in range(group):
                kfac.replace_from(bank, k)
                p, _ = eval_nn(model, data, gpu)
                total = p if total is None else total + p
    assert np.array_equal(pred, (total / 7).cpu().numpy())


@pytest.mark.parametrize("kind", ["diag", "efb", "inf"])
def test_generic_sample_many_files_ordinary_samples_away(gpu, kind):
    """The estimators without a fused form: `sample_many(S)` is S `sample_and_replace()` draws filed into a bank (same
    noise stream: set k of the bank equals the k-th ordinary draw bit for bit), so `eval_bnn(samples_per_launch=S)` is
    the serial loop."""
    from curvature_amd.curvatures import KFAC, EFB, INF, Diagonal
    from curvature_amd.evaluate import eval_bnn
    g1 = load("g1_kfac_lenet.npz")
    model, layers = lenet(gpu, g1)
    kfac, diag = KFAC(model), Diagonal(model)
    for b in range(2):
        n = backward(model, g1, b, gpu)
        kfac.update(n)
        diag.update(n)
    if kind == "diag":
        est = diag
    else:
        efb = EFB(model, kfac.state)
        backward(model, g1, 0, gpu)
        efb.update(8)
        est = efb if kind == "efb" else INF(model, diag.state, kfac.state, efb.state, eigvecs=efb.eigvecs)
        if kind == "inf":
            est.update(rank=10)
    est.invert(add=0.5, multiply=2.0)
    est.noise_seed, est.noise_offset = 555, 0
    want = []
    for _ in range(3):
        est.sample_and_replace()
        want.append([l.weight.detach().clone() for l in layers])
    est.noise_seed, est.noise_offset = 555, 0
    bank = est.sample_many(3)
    for k in range(3):
        for l, w in zip(layers, want[k]):
            assert torch.equal(bank.weights[l][k], w)
    torch.manual_seed(3)
    data = [(torch.rand(16, 1, 28, 28), torch.arange(16) % 10) for _ in range(2)]
    est.noise_seed, est.noise_offset = 9, 0
    a, _ = eval_bnn(model, data, est, samples=5, device=gpu, overlap=False)
    est.noise_seed, est.noise_offset = 9, 0
    b, _ = eval_bnn(model, data, est, samples=5, device=gpu, samples_per_launch=2)
    assert np.array_equal(a, b)
