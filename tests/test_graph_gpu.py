"""HIP-graph replay of the KFAC step (curvature_amd.graph): the captured update + invert(check=False) +
sample_and_replace must be bit-identical to the eager calls (scripts/test.py:29-53 is the loop), draw fresh noise at
every replay, and still report a non-positive-definite factor."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(gpu):
    from curvature_amd import models
    from curvature_amd.curvatures import KFAC
    g1 = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "g1_kfac_lenet.npz")).items()}
    model = models.lenet5()
    layers = [m for m in model.modules() if m.__class__.__name__ in ("Conv2d", "Linear")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    model = model.to(gpu).eval()
    kfac = KFAC(model)
    kfac.noise_seed = 4242
    for li, layer in enumerate(layers):                     # recorded inputs / gradients of golden g1, batch 0
        kfac.record[layer] = [g1[f"b0_l{li}_x"].to(gpu), g1[f"b0_l{li}_g"].to(gpu)]
    return model, layers, kfac


def test_replay_is_bit_identical_to_eager(gpu):
    from curvature_amd.graph import KFACStepGraph
    model_e, layers_e, eager = _setup(gpu)
    weights_e = []
    for step in range(6):
        eager.update(8)
        eager.invert(0.5, 1.0)
        eager.sample_and_replace()
        weights_e.append([l.weight.detach().clone() for l in layers_e] + [l.bias.detach().clone() for l in layers_e])
    model_g, layers_g, kfac = _setup(gpu)
    # construction warms up (three real steps) and captures, then puts factors and noise position back: the replays are
    # steps 0, 1, 2, ... of the eager loop (round-3 advisor: the warm-up used to stay accumulated in the factors)
    graph = KFACStepGraph(kfac, add=0.5, multiply=1.0, batch_size=8, warmup=3)
    assert graph.record_is_static()
    for layer in layers_g:
        assert all(float(t.abs().max()) == 0.0 for t in kfac.state[layer])
    for step in range(6):
        graph.replay()
        torch.cuda.synchronize()
        got = [l.weight.detach() for l in layers_g] + [l.bias.detach() for l in layers_g]
        for a, b in zip(got, weights_e[step]):
            assert torch.equal(a, b), step
    graph.check()
    for le, lg in zip(layers_e, layers_g):                  # accumulated factors and inverse factors too
        for a, b in zip(eager.state[le] + list(eager.inv_state[le]), kfac.state[lg] + list(kfac.inv_state[lg])):
            assert torch.equal(a, b)
    # successive replays drew different noise (the stream position lives on the device)
    assert not torch.equal(weights_e[4][0], weights_e[5][0])
    kfac.use_device_noise_counter(False)
    assert kfac.noise_offset == eager.noise_offset


def test_replay_reports_a_failed_factorisation(gpu):
    from curvature_amd.graph import KFACStepGraph
    model, layers, kfac = _setup(gpu)
    graph = KFACStepGraph(kfac, add=0.5, multiply=1.0, batch_size=8)
    graph.replay()
    graph.check()                                           # fine
    kfac.state[layers[2]][0].fill_(float("nan"))            # poison a factor in place: the captured sweep reads it
    graph.replay()
    with pytest.raises(RuntimeError, match="positive-definite"):
        graph.check()


def test_capture_of_an_inversion_with_far_updates(gpu):
    """Factors wider than 512 have far updates on a second stream.  Captured, the sweep keeps its chain on the capturing
    stream (two forked streams that depend on each other crashed hipStreamEndCapture): the replay must reproduce the
    eager result bit for bit - every tile is written by one workgroup in a fixed order - for one factor group and for
    two, on the chain-bound kernels and on the per-step launches (77 factors)."""
    from curvature_amd import ops
    for sizes in ([1100], [2304, 700, 64], [1100, 300] + [8] * 75):
        Fs = []
        for i, n in enumerate(sizes):
            torch.manual_seed(100 + i)
            X = torch.randn(n, n + 8, device=gpu)
            Fs.append((X @ X.t() / (n + 8)).contiguous())
        adds, muls = [0.7] * len(Fs), [20.0] * len(Fs)
        eager = [t.clone() for t in ops.chol_inv_lower(Fs, adds, muls)]
        outs = ops.chol_inv_lower(Fs, adds, muls, check=False)
        torch.cuda.synchronize()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            ops.chol_inv_lower(Fs, adds, muls, check=False, outs=outs)
        for t in outs:
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ops.check_chol_info(ops.chol_inv_lower.last_info)
        for a, b in zip(outs, eager):
            assert torch.equal(a, b), sizes


def test_construction_leaves_accumulated_factors_alone_and_replay_guards_its_addresses(gpu):
    """An estimator that already holds factors: building the graph must not add the warm-up batches to them, and a replay
    after the factors were re-allocated (the graph holds the old addresses) is refused."""
    from curvature_amd.graph import KFACStepGraph
    model, layers, kfac = _setup(gpu)
    kfac.update(8)
    kfac.update(8)
    before = {l: [t.clone() for t in kfac.state[l]] for l in layers}
    graph = KFACStepGraph(kfac, add=0.5, multiply=1.0, batch_size=8)
    for l in layers:
        for a, b in zip(kfac.state[l], before[l]):
            assert torch.equal(a, b)
    graph.replay()
    torch.cuda.synchronize()
    for l in layers:                                        # exactly one more batch
        assert torch.allclose(kfac.state[l][0], before[l][0] * 1.5, rtol=1e-5, atol=0)
    from curvature_amd import ops
    ops.release_workspaces()                                # the graph keeps what it addresses alive
    graph.replay()
    graph.check()
    kfac.state[layers[0]] = [t.clone() for t in kfac.state[layers[0]]]
    with pytest.raises(RuntimeError, match="re-allocated"):
        graph.replay()
