"""BlockDiagonal (reference curvatures.py:196-261) on the HIP path against golden g13
(the reference's own state / inverse factors / Linear samples on a small conv + linear net) and the CPU oracle
(Conv2d samples, which the reference itself cannot produce; the fused sample_and_replace)."""
import os

import numpy as np
import pytest
import torch

import oracle.curvature_oracle as o
from conftest import rel_fro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4            # north_star: relative Frobenius error vs the reference CPU path


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def small_net(gpu, g):
    model = torch.nn.Sequential(torch.nn.Conv2d(1, 2, 3), torch.nn.ReLU(), torch.nn.Flatten(),
                                torch.nn.Linear(2 * 4 * 4, 6), torch.nn.ReLU(), torch.nn.Linear(6, 4))
    layers = [m for m in model.modules() if m.__class__.__name__ in ("Conv2d", "Linear")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g[f"w_l{li}"])
            layer.bias.copy_(g[f"bias_l{li}"])
    return model.to(gpu).eval(), layers


def replay(layers, g, b, gpu):
    """The reference's own gradients of batch b (MIOpen's backward is not what is under test here)."""
    for li, layer in enumerate(layers):
        layer.weight.grad = g[f"b{b}_l{li}_gw"].to(gpu)
        layer.bias.grad = g[f"b{b}_l{li}_gb"].to(gpu)


def test_block_diagonal_against_the_reference(gpu):
    from curvature_amd.curvatures import BlockDiagonal
    g = load("g13_block_diagonal.npz")
    model, layers = small_net(gpu, g)
    est = BlockDiagonal(model)
    for b in range(2):
        replay(layers, g, b, gpu)
        est.update(batch_size=4)
        for li, layer in enumerate(layers):
            got = est.state[layer].cpu()
            assert rel_fro(got, g[f"state_after{b + 1}_l{li}"]) < 1e-6
            assert torch.equal(got, got.t())
    # scalar and per-layer hyper-parameters
    est.invert(add=0.5, multiply=2.0)
    for li, layer in enumerate(layers):
        assert rel_fro(est.inv_state[layer].cpu(), g[f"a_inv_l{li}"]) < TOL
    est.invert(add=[float(v) for v in g["b_add"]], multiply=[float(v) for v in g["b_mul"]])
    for li, layer in enumerate(layers):
        assert rel_fro(est.inv_state[layer].cpu(), g[f"b_inv_l{li}"]) < TOL
    # samples with the reference's noise: Linear layers against the reference, Conv2d against the oracle's (out, -1) form
    est.invert(add=0.5, multiply=2.0)
    for li, layer in enumerate(layers):
        smp = est.sample(layer, z=g[f"z_l{li}"].to(gpu)).cpu()
        if li > 0:
            assert rel_fro(smp, g[f"sample_l{li}"]) < TOL
        want = o.block_sample(g[f"a_inv_l{li}"], g[f"z_l{li}"], g[f"w_l{li}"].shape)
        assert smp.shape == want.shape and rel_fro(smp, want) < TOL


def test_block_diagonal_sample_and_replace(gpu):
    """The fused replace writes mean + z @ L onto the parameters in g's order; z is the library's Philox stream,
    reproduced here from the estimator's seed, so the expected parameters are exact up to GEMM rounding.  A second
    call starts from the mean again."""
    from curvature_amd import ops
    from curvature_amd.curvatures import BlockDiagonal
    g = load("g13_block_diagonal.npz")
    model, layers = small_net(gpu, g)
    est = BlockDiagonal(model)
    for b in range(2):
        replay(layers, g, b, gpu)
        est.update(batch_size=4)
    est.invert(add=0.5, multiply=2.0)
    est.noise_seed = 20260102
    for call in range(2):
        offset = est.noise_offset
        total = sum(est.inv_state[l].shape[0] for l in layers)
        z = ops.randn((total,), gpu, est.noise_seed, offset).cpu()
        est.sample_and_replace()
        pos = 0
        for li, layer in enumerate(layers):
            P = est.inv_state[layer].shape[0]
            x = z[pos:pos + P].double() @ est.inv_state[layer].cpu().double()
            pos += P
            n_w = layer.weight.numel()
            want_w = g[f"w_l{li}"].double() + x[:n_w].view(g[f"w_l{li}"].shape)
            want_b = g[f"bias_l{li}"].double() + x[n_w:]
            assert rel_fro(layer.weight.data.cpu().double(), want_w) < 1e-6, (call, li)
            assert rel_fro(layer.bias.data.cpu().double(), want_b) < 1e-6, (call, li)
        # every other state tensor is back at its mean
        for k, v in model.state_dict().items():
            if not any(v.data_ptr() == p.data_ptr() for l in layers for p in (l.weight, l.bias)):
                assert torch.equal(v, est.model_state[k])


def test_block_diagonal_sharded_covers_the_unsharded_result(gpu):
    """Two emulated ranks own disjoint layers; their states / inverse factors are the unsharded ones bit for bit,
    with per-layer hyper-parameters indexed by the GLOBAL layer position."""
    from curvature_amd import sharding
    from curvature_amd.curvatures import BlockDiagonal
    g = load("g13_block_diagonal.npz")
    model, layers = small_net(gpu, g)
    adds, muls = [float(v) for v in g["b_add"]], [float(v) for v in g["b_mul"]]

    def run(shard):
        est = BlockDiagonal(model, shard=shard)
        for b in range(2):
            replay(layers, g, b, gpu)
            est.update(batch_size=4)
        est.invert(add=adds, multiply=muls)
        return est
    full = run(None)
    owner = [0, 1, 0]
    seen = set()
    for rank in range(2):
        part = run(sharding.Shard(owner, rank, 2))
        assert set(part.state.keys()) == {l for l, o_ in zip(layers, owner) if o_ == rank}
        for layer in part.state:
            seen.add(layer)
            assert torch.equal(part.state[layer], full.state[layer])
            assert torch.equal(part.inv_state[layer], full.inv_state[layer])
    assert seen == set(layers)


def test_block_diagonal_refuses_cpu_tensors():
    from curvature_amd.curvatures import BlockDiagonal
    g = load("g13_block_diagonal.npz")
    model, layers = small_net(torch.device("cpu"), g)
    est = BlockDiagonal(model)
    replay(layers, g, 0, torch.device("cpu"))
    with pytest.raises(RuntimeError):
        est.update(batch_size=4)
