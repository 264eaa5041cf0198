"""End-to-end parity of the drop-in KFAC / Diagonal classes with the reference on LeNet-5:
same inputs, labels and noise as tools/make_golden.py fed to the reference (golden g1, g3, g4)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_fro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4            # north_star: relative Frobenius error vs the reference CPU path


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def lenet_with_golden_weights(gpu, g1):
    from curvature_amd import models
    model = models.lenet5()
    layers = [m for m in model.modules() if m.__class__.__name__ in ("Conv2d", "Linear")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    return model.to(gpu).eval(), layers


def run_batches(model, est_list, g1, gpu, nb=3):
    for b in range(nb):
        x = g1[f"b{b}_x"].to(gpu)
        labels = g1[f"b{b}_labels"].to(gpu)
        loss = torch.nn.functional.cross_entropy(model(x), labels)
        model.zero_grad()
        loss.backward()
        for est in est_list:
            est.update(batch_size=x.size(0))
        yield b


def test_kfac_update_invert_sample(gpu):
    from curvature_amd.curvatures import KFAC, Diagonal
    g1, g3, g4 = load("g1_kfac_lenet.npz"), load("g3_kfac_invert.npz"), load("g4_kfac_sample.npz")
    model, layers = lenet_with_golden_weights(gpu, g1)
    kfac, diag = KFAC(model), Diagonal(model)
    for b in run_batches(model, [kfac, diag], g1, gpu):
        if b in (0, 2):
            for li, layer in enumerate(layers):
                A, G = kfac.state[layer]
                assert rel_fro(A, g1[f"A_after{b + 1}_l{li}"]) < TOL
                assert rel_fro(G, g1[f"G_after{b + 1}_l{li}"]) < TOL
                assert rel_fro(diag.state[layer], g1[f"diag_after{b + 1}_l{li}"]) < TOL
    assert list(kfac.state.keys()) == layers                     # layer indexing: modules() order, bit-exact

    # invert: scalar form (README call) and per-layer list form
    kfac.invert(add=0.5, multiply=1)
    for li, layer in enumerate(layers):
        LA, LG = kfac.inv_state[layer]
        assert rel_fro(LA, g3[f"a_LA_l{li}"]) < TOL, (li, rel_fro(LA, g3[f"a_LA_l{li}"]))
        assert rel_fro(LG, g3[f"a_LG_l{li}"]) < TOL
    kfac.invert(add=g3["c_add"].tolist(), multiply=g3["c_mul"].tolist())
    for li, layer in enumerate(layers):
        LA, LG = kfac.inv_state[layer]
        # judged against the reference's fp64 twin ("c64_*": its own code on float64 factors): the reference's fp32
        # LAPACK chain is up to 4e-5 away from it on these factors, the north star's bar is 1e-4
        assert rel_fro(LA, g3[f"c64_LA_l{li}"]) < TOL, (li, rel_fro(LA, g3[f"c64_LA_l{li}"]))
        assert rel_fro(LG, g3[f"c64_LG_l{li}"]) < TOL, (li, rel_fro(LG, g3[f"c64_LG_l{li}"]))
        assert rel_fro(LA, g3[f"c_LA_l{li}"]) < 5e-4 and rel_fro(LG, g3[f"c_LG_l{li}"]) < 5e-4

    # sample with the reference's noise: use the reference's inverse factors to isolate the sampler
    kfac.invert(add=0.5, multiply=1)
    for li, layer in enumerate(layers):
        kfac.inv_state[layer] = (g3[f"a_LA_l{li}"].to(gpu), g3[f"a_LG_l{li}"].to(gpu))
    noise = {layer: g4[f"z_l{li}"].to(gpu) for li, layer in enumerate(layers)}
    for li, layer in enumerate(layers):
        s = kfac.sample(layer, z=noise[layer])
        assert rel_fro(s, g4[f"sample_l{li}"]) < TOL
    kfac.sample_and_replace(noise=noise)
    for li, layer in enumerate(layers):
        assert rel_fro(layer.weight, g4[f"w_new_l{li}"]) < TOL
        assert rel_fro(layer.bias, g4[f"b_new_l{li}"]) < TOL
        # and the perturbation itself, not just mean + perturbation
        dw = layer.weight.detach().cpu() - g1[f"w_l{li}"]
        assert rel_fro(dw, g4[f"w_new_l{li}"] - g1[f"w_l{li}"]) < 1e-3

    # full chain with our own factors: still within tolerance of the reference's sampled weights
    kfac.invert(add=0.5, multiply=1)
    kfac.sample_and_replace(noise=noise)
    for li, layer in enumerate(layers):
        assert rel_fro(layer.weight, g4[f"w_new_l{li}"]) < TOL

    # generic base-class path (sample + _replace) agrees with the fused one
    fused = [l.weight.detach().clone() for l in layers]
    kfac.model.load_state_dict(kfac.model_state)
    for layer in layers:
        kfac._replace(kfac.sample(layer, z=noise[layer]), layer.weight, layer.bias)
    for w, layer in zip(fused, layers):
        assert rel_fro(layer.weight, w) < 1e-6


def test_sampler_statistics(gpu):
    """Default (device Philox) noise: sample covariance of vec(W) matches (L_G L_G^T) kron (L_A L_A^T)."""
    from curvature_amd import ops
    z = ops.randn((4, 250000), gpu, seed=123)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05            # kurtosis of a normal
    z2 = ops.randn((4, 250000), gpu, seed=123)
    assert torch.equal(z, z2)                                    # counter based: reproducible
    z3 = ops.randn((4, 250000), gpu, seed=123, offset=250000)
    assert not torch.equal(z, z3)
    c = torch.corrcoef(z)
    assert float((c - torch.eye(4, device=gpu)).abs().max()) < 1e-2


def test_diagonal_invert_sample(gpu):
    from curvature_amd.curvatures import Diagonal
    import oracle.curvature_oracle as o
    g1 = load("g1_kfac_lenet.npz")
    model, layers = lenet_with_golden_weights(gpu, g1)
    diag = Diagonal(model)
    list(run_batches(model, [diag], g1, gpu))
    diag.invert(add=0.25, multiply=3.0)
    for li, layer in enumerate(layers):
        ref = o.rsqrt_affine(g1[f"diag_after3_l{li}"], 0.25, 3.0)
        assert rel_fro(diag.inv_state[layer], ref) < 1e-6
        z = torch.randn(ref.shape)
        assert rel_fro(diag.sample(layer, z=z.to(gpu)), o.diag_sample(ref, z)) < 1e-6
    diag.sample_and_replace()
    assert all(torch.isfinite(l.weight).all() for l in layers)


def test_cpu_model_is_rejected():
    """No CPU fallback: the product path fails loudly."""
    from curvature_amd.curvatures import KFAC
    from curvature_amd import models
    model = models.lenet5()
    kfac = KFAC(model)
    x = torch.rand(2, 1, 28, 28)
    loss = model(x).sum()
    loss.backward()
    with pytest.raises(RuntimeError):
        kfac.update(batch_size=2)


def test_sharded_ranks_cover_the_unsharded_result(gpu):
    """Layer sharding without the collective: every rank's update/invert/sample touches only its
    layers and the union over ranks equals the single-process result (the all-gather itself is covered by
    tests/test_sharding_gloo.py)."""
    from curvature_amd import sharding
    from curvature_amd.curvatures import KFAC
    g1 = load("g1_kfac_lenet.npz")
    results = {}
    for world, rank in [(1, 0), (2, 0), (2, 1)]:
        model, layers = lenet_with_golden_weights(gpu, g1)
        kfac = KFAC(model)
        costs = [sharding.layer_cost(n, m, K) for n, m, K in [(26, 6, 6272), (151, 16, 800), (401, 120, 8), (121, 84, 8), (85, 10, 8)]]
        if world > 1:
            kfac.shard = sharding.Shard(sharding.lpt_partition(costs, world), rank, 1)   # world=1 inside: no collective
            kfac.shard.world = 1
        list(run_batches(model, [kfac], g1, gpu, nb=1))
        kfac.invert(0.5, 1)
        results[(world, rank)] = {li: kfac.inv_state[l] for li, l in enumerate(layers) if l in kfac.inv_state}
    full = results[(1, 0)]
    assert len(full) == 5
    owned0, owned1 = set(results[(2, 0)]), set(results[(2, 1)])
    assert owned0 and owned1 and owned0.isdisjoint(owned1) and owned0 | owned1 == set(range(5))
    for part in (results[(2, 0)], results[(2, 1)]):
        for li, (LA, LG) in part.items():
            assert torch.equal(LA, full[li][0]) and torch.equal(LG, full[li][1])     # bitwise: same kernels, same inputs


def replay_batch(model_layers, ests, g1, b, gpu):
    """Feed batch `b` of golden g1 to the estimators WITHOUT running backward: the recorded layer inputs / raw
    grad_outputs go into KFAC.record, the recorded parameter gradients into .grad.  (MIOpen's weight-gradient
    kernels accumulate with atomics, so two backward passes of the same batch differ in the last bits; bitwise
    comparisons between separately built estimators need bit-identical inputs.)"""
    from curvature_amd.curvatures import KFAC
    for li, layer in enumerate(model_layers):
        layer.weight.grad = g1[f"b{b}_l{li}_gw"].to(gpu)
        layer.bias.grad = g1[f"b{b}_l{li}_gb"].to(gpu)
    for est in ests:
        if isinstance(est, KFAC):
            for li, layer in enumerate(model_layers):
                est.record[layer] = [g1[f"b{b}_l{li}_x"].to(gpu), g1[f"b{b}_l{li}_g"].to(gpu)]
        est.update(batch_size=8)


def test_sharded_efb_inf_diagonal_cover_the_unsharded_result(gpu):
    """The same for Diagonal / EFB / INF: with a Shard every rank decomposes, updates,
    inverts and samples only its own layers, per-layer hyper-parameter LISTS are still indexed by the global
    layer position, and the union over the ranks is bit-identical to the unsharded run."""
    from curvature_amd import sharding
    from curvature_amd.curvatures import KFAC, Diagonal, EFB, INF
    g1 = load("g1_kfac_lenet.npz")
    adds, muls = [0.5, 1.0, 2.0, 0.25, 3.0], [1.0, 10.0, 100.0, 5.0, 50.0]
    costs = [sharding.layer_cost(n, m, K) for n, m, K in [(26, 6, 6272), (151, 16, 800), (401, 120, 8), (121, 84, 8), (85, 10, 8)]]
    results = {}
    for world, rank in [(1, 0), (2, 0), (2, 1)]:
        model, layers = lenet_with_golden_weights(gpu, g1)
        shard = None
        if world > 1:
            shard = sharding.Shard(sharding.lpt_partition(costs, world), rank, 1)     # world=1 inside: no collective
        kfac, diag = KFAC(model, shard=shard), Diagonal(model, shard=shard)
        for b in range(2):
            replay_batch(layers, [kfac, diag], g1, b, gpu)
        efb = EFB(model, kfac.state, shard=shard)
        replay_batch(layers, [efb], g1, 0, gpu)
        inf = INF(model, diag.state, kfac.state, efb.state, shard=shard, eigvecs=efb.eigvecs)
        inf.update(rank=10)
        owned = [li for li, l in enumerate(layers) if shard is None or shard.owns(li)]
        for est in (diag, efb, inf):
            assert [layers.index(l) for l in est.state.keys()] == owned
            est.invert(add=adds, multiply=muls)
        noise = {l: torch.randn(kfac.state[l][0].shape[0] * kfac.state[l][1].shape[0],
                                generator=torch.Generator().manual_seed(li)).to(gpu)
                 for li, l in enumerate(layers) if li in owned}
        out = {}
        inf.sample_and_replace(noise=noise)
        for li in owned:
            out[li] = dict(diag=diag.inv_state[layers[li]].clone(), efb=efb.inv_state[layers[li]].clone(),
                           lam=efb.state[layers[li]].clone(), r=inf.inv_state[layers[li]][2].clone(),
                           Pc=inf.inv_state[layers[li]][3].clone(), w=layers[li].weight.detach().clone(),
                           b=layers[li].bias.detach().clone())
        # layers of other ranks stay at the mean (their values arrive through the all-gather in a real run)
        for li, l in enumerate(layers):
            if li not in owned:
                assert torch.equal(l.weight.data, inf.model_state_of(l, 'weight'))
        results[(world, rank)] = out
    full = results[(1, 0)]
    o0, o1 = set(results[(2, 0)]), set(results[(2, 1)])
    assert o0 and o1 and o0.isdisjoint(o1) and o0 | o1 == set(range(5))
    for part in (results[(2, 0)], results[(2, 1)]):
        for li, vals in part.items():
            for k, v in vals.items():
                assert torch.equal(v, full[li][k]), (li, k)


@pytest.mark.gpu
def test_copy_batched_matches_torch(gpu):
    """curv_copy_batched: any size / alignment / dtype, several launches' worth of buffers."""
    from curvature_amd import ops
    torch.manual_seed(3)
    sizes = [1, 3, 4, 17, 256, 1000, 65536 // 4, 65536 // 4 + 5, 300001] + [7 + i for i in range(120)]
    srcs, dsts = [], []
    base = torch.randn(sum(sizes) + len(sizes) + 8, device=gpu)
    off = 0
    for i, n in enumerate(sizes):
        off += (i % 3 == 0)                       # odd element offsets: 4-byte aligned only
        srcs.append(base[off:off + n])
        off += n
        dsts.append(torch.full((n,), float("nan"), device=gpu))
    srcs.append(torch.arange(5, device=gpu, dtype=torch.int64))          # 8-byte elements
    dsts.append(torch.zeros(5, device=gpu, dtype=torch.int64))
    srcs.append(torch.arange(11, device=gpu, dtype=torch.uint8)[1:])     # byte-aligned only
    dsts.append(torch.zeros(10, device=gpu, dtype=torch.uint8))
    ops.CopyPlan(dsts, srcs).run()
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)


@pytest.mark.gpu
def test_sample_and_replace_restores_every_state_tensor(gpu):
    """The batched reload must behave like load_state_dict(model_state) (curvatures.py:119): buffers and
    unselected parameters are reset too; selected layers end up at mean + sample."""
    from curvature_amd.curvatures import KFAC
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Flatten(), torch.nn.Linear(4 * 6 * 6, 5)).to(gpu)
    kfac = KFAC(model)
    x = torch.randn(8, 3, 6, 6, device=gpu)
    out = model(x)
    loss = torch.nn.functional.cross_entropy(out, torch.randint(0, 5, (8,), device=gpu))
    loss.backward()
    kfac.update(batch_size=8)
    kfac.invert(add=1.0, multiply=10.0)
    mean = {k: v.clone() for k, v in kfac.model_state.items()}
    with torch.no_grad():                         # perturb everything, as training / BN statistics would
        for v in model.state_dict().values():
            v.add_(1)
    kfac.sample_and_replace()
    torch.cuda.synchronize()
    state = model.state_dict()
    for k in ("1.weight", "1.bias", "1.running_mean", "1.running_var", "1.num_batches_tracked"):
        assert torch.equal(state[k], mean[k]), k
    for k in ("0.weight", "4.weight", "4.bias"):
        assert not torch.equal(state[k], mean[k]) and torch.isfinite(state[k]).all()
        assert float((state[k] - mean[k]).abs().max()) < 10.0
    kfac.sample_and_replace()                     # second call reuses the cached plan
    torch.cuda.synchronize()
    assert torch.equal(model.state_dict()["1.running_mean"], mean["1.running_mean"])


@pytest.mark.gpu
@pytest.mark.parametrize("share_inputs", [True, False])
def test_mc_fisher_driver_matches_reference_loop(gpu, share_inputs):
    """compute_factors (scripts/factors.py:33-62 of the reference) on LeNet-5: the accumulated KFAC factors
    after 2 batches x 3 label draws equal the reference's (golden g12), with and without building the A
    side once per forward pass."""
    from curvature_amd import models
    from curvature_amd.factors import compute_factors
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "g12_mc_fisher_lenet.npz")).items()}
    g1 = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "g1_kfac_lenet.npz")).items()}
    model = models.lenet5()
    layers = [l for l in model.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]
    with torch.no_grad():
        for li, layer in enumerate(layers):                   # the bundled MNIST weights the golden run used
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    model = model.to(gpu)
    samples = int(g["samples"])
    data = [(g[f"b{b}_x"], None) for b in range(2)]
    est = compute_factors(None, model, data, estimator="kfac", samples=samples, epochs=1, device=gpu,
                          share_inputs=share_inputs,
                          label_sampler=lambda logits, b, s: g[f"b{b}_s{s}_labels"].to(gpu))
    torch.cuda.synchronize()
    for li, layer in enumerate(layers):
        A, G = est.state[layer]
        assert rel_fro(A, g[f"A_l{li}"]) < 1e-5, (li, rel_fro(A, g[f"A_l{li}"]))
        assert rel_fro(G, g[f"G_l{li}"]) < 1e-5, (li, rel_fro(G, g[f"G_l{li}"]))


@pytest.mark.gpu
def test_eval_bnn_matches_oracle(gpu):
    """eval_bnn (scripts/evaluate.py:121-152): mean softmax over posterior samples, against the oracle's
    sampler + a CPU forward with the same factors, hyper-parameters and noise."""
    import copy
    import oracle.curvature_oracle as o
    from curvature_amd import models
    from curvature_amd.curvatures import KFAC
    from curvature_amd.evaluate import eval_bnn
    g1 = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "g1_kfac_lenet.npz")).items()}
    model = models.lenet5()
    layers = [l for l in model.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    cpu_model = copy.deepcopy(model).eval()
    cpu_layers = [l for l in cpu_model.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]
    model = model.to(gpu)
    kfac = KFAC(model)
    kfac.state = {l: [g1[f"A_after3_l{li}"].to(gpu), g1[f"G_after3_l{li}"].to(gpu)] for li, l in enumerate(layers)}
    kfac.invert(add=0.5, multiply=1)
    torch.manual_seed(7)
    data = [(torch.rand(6, 1, 28, 28), torch.arange(6)) for _ in range(2)]
    n_samples = 3
    noises = [{l: torch.randn(g1[f"A_after3_l{li}"].shape[0], g1[f"G_after3_l{li}"].shape[0])
               for li, l in enumerate(layers)} for _ in range(n_samples)]
    calls = {"i": 0}
    plain = kfac.sample_and_replace

    def with_noise():
        plain(noise={l: z.to(gpu) for l, z in noises[calls["i"]].items()})
        calls["i"] += 1
    kfac.sample_and_replace = with_noise
    mean_pred, labels = eval_bnn(model, data, kfac, samples=n_samples, device=gpu)
    assert labels.tolist() == list(range(6)) * 2 and mean_pred.shape == (12, 10)
    # oracle: same factors / hyper-parameters / noise on the CPU
    state = {cl: [g1[f"A_after3_l{li}"], g1[f"G_after3_l{li}"]] for li, cl in enumerate(cpu_layers)}
    inv = o.model_kfac_invert(state, 0.5, 1)
    mean = [(cl.weight.detach().clone(), cl.bias.detach().clone()) for cl in cpu_layers]
    ref = torch.zeros(12, 10, dtype=torch.float64)
    for s in range(n_samples):
        smp = o.model_kfac_sample(inv, cpu_model, {cl: noises[s][l] for cl, l in zip(cpu_layers, layers)})
        with torch.no_grad():
            for cl, (w, b) in zip(cpu_layers, mean):
                w_new, b_new = o.replace(smp[cl], w, b)
                cl.weight.copy_(w_new)
                cl.bias.copy_(b_new)
            ref += torch.cat([torch.softmax(cpu_model(x), dim=1) for x, _ in data]).double()
    ref /= n_samples
    assert float(np.abs(mean_pred - ref.numpy()).max()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(130, 200, 40), (128, 128, 1536), (200, 300, 2000), (64, 520, 3100), (512, 1100, 2304)])
def test_gemm_nt_kernel(gpu, shape):
    """curv_gemm_batched on products with K-contiguous operands (the LDS-DMA kernel): plain, with every epilogue, with
    the triangular cuts of the samplers, ragged M / N / K; against fp64 torch, and bit-reproducible."""
    from curvature_amd import ops
    M, N, K = shape
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=gpu)
    Bt = torch.randn(N, K, device=gpu)                     # B = Bt.t(): K-contiguous columns
    E = torch.randn(M, N, device=gpu)
    F = torch.randn(M, N, device=gpu)
    ref = A.double() @ Bt.double().t()
    scale = float(ref.abs().max())
    cases = [(ops.EPI_NONE, None, None, 1.0, 0.0, ref),
             (ops.EPI_SQUARE, None, None, 0.5, 0.0, 0.5 * ref * ref),
             (ops.EPI_MUL_E, E, None, 2.0, 0.0, 2.0 * ref * E.double()),
             (ops.EPI_ADD_E, E, None, -1.0, 0.0, -ref + E.double()),
             (ops.EPI_MUL_E_ADD_F, E, F, 1.5, 0.0, 1.5 * ref * E.double() + F.double())]
    jobs, wants = [], []
    for ep, e, f, alpha, beta, want in cases:
        C = torch.full((M, N), float("nan"), device=gpu)
        jobs.append(ops.Gemm(A, Bt.t(), C, alpha=alpha, beta=beta, epilogue=ep, E=e, F=f))
        wants.append(want)
    # accumulate onto an existing C (beta = 1)
    C0 = torch.randn(M, N, device=gpu)
    Cb = C0.clone()
    jobs.append(ops.Gemm(A, Bt.t(), Cb, alpha=1.0, beta=1.0))
    wants.append(ref + C0.double())
    ops.gemm_batched(jobs)
    for j, want in zip(jobs, wants):
        assert float((j.C.double() - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), scale), j.epilogue
    first = [j.C.clone() for j in jobs[:5]]
    ops.gemm_batched(jobs[:5])
    for a, j in zip(first, jobs[:5]):
        assert torch.equal(a, j.C)
    # triangular operands as in KFAC.sample: lower-triangular A (square, K = M), upper-triangular B (K = N)
    if K >= 1536:
        L = torch.tril(torch.randn(K, K, device=gpu))
        Z = torch.randn(N, K, device=gpu)
        out = torch.empty(K, N, device=gpu)
        ops.gemm_batched([ops.Gemm(L, Z.t(), out, tri=ops.TRI_A_LOWER)])
        want = L.double() @ Z.double().t()
        assert float((out.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        out3 = torch.empty(M, K, device=gpu)
        Lt = torch.tril(torch.randn(K, K, device=gpu))      # B = Lt.t() is upper triangular with K-contiguous columns
        ops.gemm_batched([ops.Gemm(A, Lt.t(), out3, tri=ops.TRI_B_UPPER)])
        want3 = A.double() @ Lt.double().t()
        assert float((out3.double() - want3).abs().max()) <= 2e-5 * float(want3.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(130, 200, 401), (256, 300, 2049), (129, 64, 7), (512, 1000, 4609)])
def test_gemm_nt_kernel_unaligned_operands(gpu, shape):
    """K % 4 != 0 (the last LDS-DMA step masks stale pixel groups) and operand rows that are only 4-byte aligned: the
    operands are column slices of wider buffers whose row pitch is not a multiple of 4 floats - L_A of a biased layer
    (2049 / 4609 wide) and EFB's `ua_t[:, :n0]` look like this."""
    from curvature_amd import ops
    M, N, K = shape
    torch.manual_seed(K)
    Abuf = torch.randn(M, K + 3, device=gpu)
    Bbuf = torch.randn(N, K + 5, device=gpu)
    A, Bt = Abuf[:, 1:K + 1], Bbuf[:, 2:K + 2]               # row pitches K + 3 / K + 5, first element at +1 / +2
    assert A.stride() == (K + 3, 1) and Bt.stride() == (K + 5, 1)
    E = torch.randn(M, N, device=gpu)
    C = torch.full((M, N), float("nan"), device=gpu)
    ops.gemm_batched([ops.Gemm(A, Bt.t(), C, alpha=0.5, epilogue=ops.EPI_ADD_E, E=E)])
    want = 0.5 * (A.double() @ Bt.double().t()) + E.double()
    assert float((C.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.gpu
def test_gemm_nt_split_k_with_triangular_cut_and_beta(gpu):
    """A launch with few tiles is K-sliced (partial slabs + a reduce launch that applies the epilogue): with a
    triangular cut only the slices that intersect the tile's K range exist, and beta = 1 must add the old C once."""
    from curvature_amd import ops
    torch.manual_seed(11)
    K = 2304
    L = torch.tril(torch.randn(K, K, device=gpu))
    Z = torch.randn(96, K, device=gpu)
    C0 = torch.randn(K, 96, device=gpu)
    out = C0.clone()
    ops.gemm_batched([ops.Gemm(L, Z.t(), out, beta=1.0, tri=ops.TRI_A_LOWER)])          # 18 x 1 tiles: sliced
    want = L.double() @ Z.double().t() + C0.double()
    assert float((out.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    Lt = torch.tril(torch.randn(K, K, device=gpu))
    A = torch.randn(100, K, device=gpu)
    C1 = torch.randn(100, K, device=gpu)
    out2 = C1.clone()
    ops.gemm_batched([ops.Gemm(A, Lt.t(), out2, alpha=2.0, beta=1.0, tri=ops.TRI_B_UPPER)])
    want2 = 2.0 * (A.double() @ Lt.double().t()) + C1.double()
    assert float((out2.double() - want2).abs().max()) <= 2e-5 * float(want2.abs().max())
    first = out2.clone()
    out3 = C1.clone()
    ops.gemm_batched([ops.Gemm(A, Lt.t(), out3, alpha=2.0, beta=1.0, tri=ops.TRI_B_UPPER)])
    assert torch.equal(first, out3)                           # fixed-order slice sums: bit-reproducible


@pytest.mark.gpu
def test_gemv_path_for_one_column_products(gpu):
    """N == 1 products with K-contiguous operands (the bias column of a sampled layer, INF's P_c x) run as row-wise
    dot products; with every epilogue, against fp64."""
    from curvature_amd import ops
    torch.manual_seed(3)
    M, K = 1000, 2049
    A = torch.randn(M, K, device=gpu)
    LA = torch.randn(K, K, device=gpu)
    col = LA.t()[:, K - 1:]                                   # (K, 1) view with unit row stride: row K - 1 of LA
    assert col.stride(0) == 1
    E = torch.randn(M, 1, device=gpu)
    F = torch.randn(M, 1, device=gpu)
    ref = A.double() @ col.double()
    for ep, e, f, alpha, want in ((ops.EPI_NONE, None, None, 1.0, ref), (ops.EPI_ADD_E, E, None, -1.0, -ref + E.double()),
                                  (ops.EPI_MUL_E_ADD_F, E, F, 2.0, 2.0 * ref * E.double() + F.double())):
        C = torch.full((M, 1), float("nan"), device=gpu)
        ops.gemm_batched([ops.Gemm(A, col, C, alpha=alpha, epilogue=ep, E=e, F=f)])
        assert float((C.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()), ep


@pytest.mark.gpu
def test_gemm_small_launch_kernel(gpu):
    """A launch whose products are all short and in NT layout (a LeNet-scale sampler) runs as 32 x 32 blocks with the K
    range split over the four waves of a workgroup (gemm_nt_small_kernel): ragged M / N / K down to 1, K beyond one
    pass (512), one-column products, triangular cuts, every epilogue, several products per launch; against fp64, and
    bit-reproducible."""
    from curvature_amd import ops
    torch.manual_seed(17)
    jobs, wants = [], []
    shapes = [(120, 401, 401), (84, 121, 121), (16, 151, 151), (6, 26, 26), (10, 85, 85), (33, 1, 401), (1, 1, 1),
              (5, 70, 3), (40, 40, 777), (31, 65, 1024), (64, 64, 513)]
    eps = [ops.EPI_NONE, ops.EPI_SQUARE, ops.EPI_MUL_E, ops.EPI_ADD_E, ops.EPI_MUL_E_ADD_F]
    for idx, (M, N, K) in enumerate(shapes):
        Abuf = torch.randn(M, K + 3, device=gpu)
        Bbuf = torch.randn(N, K + 1, device=gpu)
        A, Bt = Abuf[:, 2:K + 2], Bbuf[:, 1:K + 1]           # rows only 4-byte aligned
        E, F = torch.randn(M, N, device=gpu), torch.randn(M, N, device=gpu)
        ep = eps[idx % len(eps)]
        C0 = torch.randn(M, N, device=gpu)
        C = C0.clone()
        beta = 1.0 if idx % 3 == 0 else 0.0
        if beta == 0.0:
            C.fill_(float("nan"))
        jobs.append(ops.Gemm(A, Bt.t(), C, alpha=0.75, beta=beta, epilogue=ep,
                             E=E if ep >= ops.EPI_MUL_E else None, F=F if ep == ops.EPI_MUL_E_ADD_F else None))
        r = A.double() @ Bt.double().t()
        want = {ops.EPI_NONE: 0.75 * r, ops.EPI_SQUARE: 0.75 * r * r, ops.EPI_MUL_E: 0.75 * r * E.double(),
                ops.EPI_ADD_E: 0.75 * r + E.double(), ops.EPI_MUL_E_ADD_F: 0.75 * r * E.double() + F.double()}[ep]
        wants.append(want + beta * C0.double())
    # the sampler's triangular products: L_G z^T (A lower, K = M) and (.) L_A^T (B upper, K = N)
    for n in (120, 401, 37):
        L = torch.tril(torch.randn(n, n, device=gpu))
        Z = torch.randn(90, n, device=gpu)
        out = torch.full((n, 90), float("nan"), device=gpu)
        jobs.append(ops.Gemm(L, Z.t(), out, tri=ops.TRI_A_LOWER))
        wants.append(L.double() @ Z.double().t())
        out2 = torch.full((90, n), float("nan"), device=gpu)
        jobs.append(ops.Gemm(Z, L.t(), out2, tri=ops.TRI_B_UPPER))
        wants.append(Z.double() @ L.double().t())
    starts = [j.C.clone() for j in jobs]
    ops.gemm_batched(jobs)
    for j, want in zip(jobs, wants):
        assert float((j.C.double() - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1.0), (j.A.shape, j.B.shape)
    first = [j.C.clone() for j in jobs]
    for j, s in zip(jobs, starts):
        j.C.copy_(s)
    ops.gemm_batched(jobs)
    for a, j in zip(first, jobs):
        assert torch.equal(a, j.C)
