"""EFB, INF and the eigensolver against the reference's golden vectors (g5-g9).

Eigenvectors are unique only up to sign / rotation inside degenerate clusters, so the
eigensolver is judged by residual and orthogonality, and EFB / INF parity feeds the REFERENCE's
eigenvectors in, exactly as the reference's own EFB -> INF chain does."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_fro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def lenet(gpu, g1):
    from curvature_amd import models
    model = models.lenet5()
    layers = [m for m in model.modules() if m.__class__.__name__ in ("Conv2d", "Linear")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    return model.to(gpu).eval(), layers


def backward(model, g1, b, gpu):
    x, labels = g1[f"b{b}_x"].to(gpu), g1[f"b{b}_labels"].to(gpu)
    loss = torch.nn.functional.cross_entropy(model(x), labels)
    model.zero_grad()
    loss.backward()


@pytest.mark.parametrize("n", [1, 2, 31, 64, 65, 150, 401])
def test_eigh_residual_and_orthogonality(gpu, n):
    from curvature_amd import ops
    torch.manual_seed(n)
    X = torch.randn(n, max(n // 2, 1))                 # rank-deficient PSD, like a KFAC factor
    F = (X @ X.t() / X.shape[1]).float()
    F = ((F + F.t()) / 2).contiguous()
    (U,), (w,) = ops.eigh([F.to(gpu)], with_values=True)
    U, w, Fd = U.double().cpu(), w.double().cpu(), F.double()
    assert torch.all(w[1:] >= w[:-1] - 1e-9)                                  # ascending
    assert torch.linalg.norm(U.t() @ U - torch.eye(n, dtype=torch.float64)) < 1e-5 * n ** 0.5
    assert torch.linalg.norm(Fd @ U - U * w) < 1e-5 * max(float(torch.linalg.norm(Fd)), 1e-30)
    ref = torch.linalg.eigvalsh(Fd)
    assert torch.linalg.norm(w - ref) < 1e-5 * max(float(torch.linalg.norm(ref)), 1e-30)


def test_eigh_lenet_factors_batched(gpu):
    """All ten LeNet factors in one batched call; invariants of the golden eigenvectors hold for ours."""
    from curvature_amd import ops
    g1 = load("g1_kfac_lenet.npz")
    mats = [g1[f"{s}_after3_l{li}"].to(gpu) for li in range(5) for s in ("A", "G")]
    vecs, vals = ops.eigh(mats, with_values=True)
    for F, U, w in zip(mats, vecs, vals):
        F, U, w = F.double().cpu(), U.double().cpu(), w.double().cpu()
        n = F.shape[0]
        assert torch.linalg.norm(U.t() @ U - torch.eye(n, dtype=torch.float64)) < 1e-5 * n ** 0.5
        assert torch.linalg.norm(F @ U - U * w) < 1e-5 * torch.linalg.norm(F)
        assert rel_fro(w, torch.linalg.eigvalsh(F)) < 1e-5


def test_efb_chain(gpu):
    from curvature_amd.curvatures import EFB
    g1, g5, g6 = load("g1_kfac_lenet.npz"), load("g5_eigvecs_lenet.npz"), load("g6_efb_lenet.npz")
    model, layers = lenet(gpu, g1)
    factors = {l: [g1[f"A_after3_l{li}"].to(gpu), g1[f"G_after3_l{li}"].to(gpu)] for li, l in enumerate(layers)}
    efb = EFB(model, factors)                                         # runs our eigensolver
    for li, layer in enumerate(layers):                               # own eigvecs: valid decomposition
        UA, UG = efb.eigvecs[layer]
        assert UA.shape == g5[f"UA_l{li}"].shape and UG.shape == g5[f"UG_l{li}"].shape
    # parity: the reference's eigenvectors in (as its own chain has them)
    efb.eigvecs = {l: (g5[f"UA_l{li}"].to(gpu), g5[f"UG_l{li}"].to(gpu)) for li, l in enumerate(layers)}
    for b in range(2):
        backward(model, g1, b, gpu)
        efb.update(batch_size=8)
    for li, layer in enumerate(layers):
        assert rel_fro(efb.state[layer], g6[f"lambda_l{li}"]) < TOL
        assert rel_fro(efb.diags[layer], g6[f"diags_l{li}"]) < TOL
    efb.invert(add=0.5, multiply=2.0)
    for li, layer in enumerate(layers):
        assert rel_fro(efb.inv_state[layer], g6[f"inv_l{li}"]) < TOL
        s = efb.sample(layer, z=g6[f"z_l{li}"].to(gpu))
        assert rel_fro(s, g6[f"sample_l{li}"]) < TOL
    # fused whole-model path with the golden noise: parameters = mean + the reference's sample
    efb.sample_and_replace(noise={l: g6[f"z_l{li}"].to(gpu) for li, l in enumerate(layers)})
    for li, layer in enumerate(layers):
        smp = g6[f"sample_l{li}"].to(gpu)
        w_ref = g1[f"w_l{li}"].to(gpu) + smp[:, :-1].reshape(layer.weight.shape)
        b_ref = g1[f"bias_l{li}"].to(gpu) + smp[:, -1]
        assert rel_fro(layer.weight.data, w_ref) < TOL and rel_fro(layer.bias.data, b_ref) < TOL
    efb.sample_and_replace()
    assert all(torch.isfinite(l.weight).all() for l in layers)


def inf_from_golden(gpu, rank):
    from curvature_amd.curvatures import INF
    g1, g5, g6 = load("g1_kfac_lenet.npz"), load("g5_eigvecs_lenet.npz"), load("g6_efb_lenet.npz")
    model, layers = lenet(gpu, g1)
    factors = {l: [g1[f"A_after3_l{li}"].to(gpu), g1[f"G_after3_l{li}"].to(gpu)] for li, l in enumerate(layers)}
    lambdas = {l: g6[f"lambda_l{li}"].to(gpu) for li, l in enumerate(layers)}
    diags = {l: g6[f"diags_l{li}"].to(gpu) for li, l in enumerate(layers)}
    inf = INF(model, diags, factors, lambdas)
    inf.eigvecs = {l: (g5[f"UA_l{li}"].to(gpu), g5[f"UG_l{li}"].to(gpu)) for li, l in enumerate(layers)}
    inf.update(rank=rank)
    return inf, layers, g5, g6


@pytest.mark.parametrize("tag,rank", [("r10", 10), ("r100", 100), ("rall", 10 ** 9)])
def test_inf_update(gpu, tag, rank):
    g7 = load("g7_inf_update.npz")
    inf, layers, g5, g6 = inf_from_golden(gpu, rank)
    for li, layer in enumerate(layers):
        ua, ug, lam, D = inf.state[layer]
        I, J = g7[f"{tag}_I_l{li}"], g7[f"{tag}_J_l{li}"]
        # bit-exact index sets: the selected columns are exactly the reference's
        assert torch.equal(ua.cpu(), g5[f"UA_l{li}"][:, I]) and torch.equal(ug.cpu(), g5[f"UG_l{li}"][:, J])
        if f"{tag}_lam_l{li}" in g7:
            assert torch.equal(lam.cpu(), g7[f"{tag}_lam_l{li}"])
        if f"{tag}_D_l{li}" in g7:
            scale = torch.linalg.norm(g6[f"diags_l{li}"].double())
            assert float(torch.linalg.norm(D.double().cpu() - g7[f"{tag}_D_l{li}"].double()) / scale) < 1e-5


def test_inf_select_exact_and_ties(gpu):
    import oracle.curvature_oracle as o
    from curvature_amd import ops
    torch.manual_seed(0)
    for n, m, rank in [(7, 5, 3), (401, 120, 100), (26, 6, 10), (300, 200, 1), (64, 64, 4095)]:
        lam = torch.randn(n * m) ** 3
        I, J = ops.inf_select(lam.to(gpu), n, m, rank)
        Ir, Jr = o.inf_select(lam, m, rank)
        assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(J.cpu().numpy(), Jr)
    lam = torch.zeros(12 * 8)
    lam[[5, 17, 40]] = torch.tensor([3.0, -2.0, 1.0])
    I, J = ops.inf_select(lam.to(gpu), 12, 8, 5)          # ties at |0|: exactly 5 elements selected, any 2 zeros
    assert set([0, 2, 5]).issubset(set(I.tolist())) and len(I) <= 5 and len(J) <= 5


def test_inf_invert_per_layer_hyperparameters_match_the_one_pair_path(gpu):
    """`INF.invert` with ONE pair of hyper-parameters clamps / scales / inverts the arenas of `update()` in three launches;
    with per-layer lists it walks the layers.  Layer k of a list call must equal layer k of the one-pair call made with
    layer k's pair, bit for bit (r, P_c and the clamped D); a list of equal pairs takes the one-pair path again."""
    pairs = [(0.5, 2.0), (1.0, 10.0), (0.25, 1.0), (3.0, 0.5), (1.0, 1.0)]
    inf, layers, _, _ = inf_from_golden(gpu, 10)
    assert len(layers) == len(pairs)
    want = []
    for k, (a, m) in enumerate(pairs):
        inf.invert(add=a, multiply=m)
        want.append((inf.inv_state[layers[k]][2].clone(), inf.inv_state[layers[k]][3].clone(), inf.state[layers[k]][3].clone()))
    inf.invert(add=[a for a, _ in pairs], multiply=[m for _, m in pairs])
    for k, layer in enumerate(layers):
        assert torch.equal(inf.inv_state[layer][2], want[k][0])
        assert torch.equal(inf.inv_state[layer][3], want[k][1])
        assert torch.equal(inf.state[layer][3], want[k][2])
    inf.invert(add=[pairs[0][0]] * len(pairs), multiply=[pairs[0][1]] * len(pairs))
    assert torch.equal(inf.inv_state[layers[0]][2], want[0][0]) and torch.equal(inf.inv_state[layers[0]][3], want[0][1])


def test_inf_invert_and_sample(gpu):
    g8, g9 = load("g8_inf_invert.npz"), load("g9_inf_sample.npz")
    inf, layers, g5, g6 = inf_from_golden(gpu, 10)
    from curvature_amd.curvatures import INF
    add, mul = float(g8["add"]), float(g8["mul"])
    inf.invert(add=add, multiply=mul)
    for li, layer in enumerate(layers):
        ua, ug, r, Pc = inf.inv_state[layer]
        assert torch.equal(inf.state[layer][3].cpu(), g8[f"Dclamped_l{li}"]) or \
            rel_fro(inf.state[layer][3], g8[f"Dclamped_l{li}"]) < 1e-5          # clamped in place on `state`
        assert rel_fro(r, g8[f"r_l{li}"]) < TOL
        vtv = INF.vtv(ua, ug, g8[f"sigma_l{li}"].to(gpu), g8[f"r_l{li}"].to(gpu))
        assert rel_fro(vtv, g8[f"vtv_l{li}"]) < TOL
        # P_c: the reference's fp32 chain is itself noisy (2 Cholesky + 3 inverses in fp32); the bar is the
        # reference code run in fp64 (golden Pc64), and the fp32 reference within its own noise
        # (measured on these fixtures: 6e-8 .. 1e-7 against the fp64 twin; the reference's own fp32 result sits 2e-7 .. 9e-7 from it)
        assert rel_fro(Pc, g8[f"Pc64_l{li}"]) < 1e-5, rel_fro(Pc, g8[f"Pc64_l{li}"])
        noise = rel_fro(g8[f"Pc_l{li}"], g8[f"Pc64_l{li}"])
        assert rel_fro(Pc, g8[f"Pc_l{li}"]) < max(1e-5, 3 * noise)
        assert not torch.allclose(Pc, Pc.t())                                   # non-symmetric, as in the reference
    # sampler with the reference's inverse state and noise
    for li, layer in enumerate(layers):
        ua, ug = inf.inv_state[layer][0], inf.inv_state[layer][1]
        inf.inv_state[layer] = (ua, ug, g8[f"r_l{li}"].to(gpu), g8[f"Pc_l{li}"].to(gpu))
        s = inf.sample(layer, X=g9[f"X_l{li}"].to(gpu))
        assert s.shape == g9[f"sample_l{li}"].shape
        assert rel_fro(s, g9[f"sample_l{li}"]) < TOL, (li, rel_fro(s, g9[f"sample_l{li}"]))
    # the batched sample_and_replace with the reference's noise: mean + reference sample, through _replace
    inf.sample_and_replace(noise={l: g9[f"X_l{li}"].to(gpu) for li, l in enumerate(layers)})
    for li, layer in enumerate(layers):
        ref = g9[f"sample_l{li}"].to(gpu)                                      # (m, n), bias = last column
        w_mean = inf.model_state_of(layer, 'weight')
        want_w = w_mean + ref[:, :-1].reshape(w_mean.shape) if layer.bias is not None else w_mean + ref.reshape(w_mean.shape)
        assert rel_fro(layer.weight.data - w_mean, want_w - w_mean) < TOL
        if layer.bias is not None:
            b_mean = inf.model_state_of(layer, 'bias')
            assert rel_fro(layer.bias.data - b_mean, ref[:, -1]) < TOL
    inf.sample_and_replace()
    assert all(torch.isfinite(l.weight).all() for l in layers)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 130, 333, 1024, 1100])      # >= 1024: the 128 x 128 macro-tile kernel
def test_gemm_f64_triangular_operands(gpu, n):
    """curv_gemm_f64_batched with CURV_TRI64_* flags (the two products of INF.pre_sampler's triangular inverses):
    same result as the dense product when the operands really are triangular, all flag combinations, with the
    accumulating form (beta = 1) leaving the tiles above the diagonal untouched."""
    from curvature_amd import ops
    torch.manual_seed(n)
    full = [torch.randn(n, n, dtype=torch.float64, device=gpu) for _ in range(2)]
    lower = [torch.tril(f) for f in full]
    upper = [torch.triu(f) for f in full]
    cases = [(lower[0], lower[1], ops.TRI64_A_LOWER | ops.TRI64_B_LOWER),
             (upper[0], lower[1], ops.TRI64_A_UPPER | ops.TRI64_B_LOWER),
             (lower[0], upper[1], ops.TRI64_A_LOWER | ops.TRI64_B_UPPER),
             (upper[0], upper[1], ops.TRI64_A_UPPER | ops.TRI64_B_UPPER),
             (lower[0], full[1], ops.TRI64_A_LOWER), (full[0], upper[1], ops.TRI64_B_UPPER),
             (lower[0].t(), lower[1], ops.TRI64_A_UPPER | ops.TRI64_B_LOWER)]       # transposed view, as in pre_sampler
    outs = ops.gemm_f64_batched([ops.Gemm64(a, b, tri=t) for a, b, t in cases])
    for (a, b, _), c in zip(cases, outs):
        want = a @ b
        assert float((c - want).abs().max()) <= 1e-12 * float(want.abs().max())
    # T = A^-1 - B^-1 A^-1 exactly as pre_sampler_many does it
    T = lower[1].clone()
    ops.gemm_f64_batched([ops.Gemm64(lower[0], lower[1], T, alpha=-1.0, beta=1.0,
                                     tri=ops.TRI64_A_LOWER | ops.TRI64_B_LOWER)])
    want = lower[1] - lower[0] @ lower[1]
    assert float((T - want).abs().max()) <= 1e-12 * float(want.abs().max())
    assert float(torch.triu(T, 1).abs().max()) == 0.0


@pytest.mark.gpu
def test_gemm_f64_macro_tiles_all_layouts(gpu):
    """Products with both output edges >= 1024 run on 128 x 128 macro tiles (gemm_f64_macro_kernel): every operand
    layout (row-major, transposed views, column slices of wider buffers), ragged M / N / K, alpha / beta, and a call
    that mixes macro-tile and 64-tile products; against torch's fp64 product."""
    from curvature_amd import ops
    torch.manual_seed(5)
    M, N, K = 1150, 1030, 333
    A = torch.randn(M, K + 3, dtype=torch.float64, device=gpu)[:, 1:K + 1]
    At = torch.randn(K, M, dtype=torch.float64, device=gpu).t()
    B = torch.randn(K, N + 2, dtype=torch.float64, device=gpu)[:, 2:]
    Bt = torch.randn(N, K, dtype=torch.float64, device=gpu).t()
    small_a = torch.randn(70, 90, dtype=torch.float64, device=gpu)
    small_b = torch.randn(90, 600, dtype=torch.float64, device=gpu)
    jobs, wants = [], []
    for a in (A, At):
        for b in (B, Bt):
            C0 = torch.randn(M, N, dtype=torch.float64, device=gpu)
            jobs.append(ops.Gemm64(a, b, C0.clone(), alpha=0.5, beta=-2.0))
            wants.append(0.5 * (a @ b) - 2.0 * C0)
            jobs.append(ops.Gemm64(a, b))
            wants.append(a @ b)
    jobs.insert(3, ops.Gemm64(small_a, small_b))
    wants.insert(3, small_a @ small_b)
    outs = ops.gemm_f64_batched(jobs)
    for c, want in zip(outs, wants):
        assert float((c - want).abs().max()) <= 1e-12 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [200, 1100])
def test_gemm_f64_symmetric_product_lower_tiles_only(gpu, n):
    """TRI64_C_LOWER (the eigensolver's V^T V and V^T F V at its precision switch): every element on or below the diagonal
    equals the full product's BIT FOR BIT (same tiles, same K order), tiles strictly above the diagonal keep what C held;
    64-wide tiles (n = 200) and 128-wide macro tiles (n = 1100).  A non-square product with the flag is refused."""
    from curvature_amd import ops
    torch.manual_seed(n)
    V = torch.randn(n, n, dtype=torch.float64, device=gpu)
    full = ops.gemm_f64_batched([ops.Gemm64(V.t(), V)])[0]
    C = torch.full((n, n), 7.0, dtype=torch.float64, device=gpu)
    ops.gemm_f64_batched([ops.Gemm64(V.t(), V, C, tri=ops.TRI64_C_LOWER)])
    i, j = torch.tril_indices(n, n, device=gpu)
    assert torch.equal(C[i, j], full[i, j])
    T = 64 if n < 1024 else 128
    bi, bj = torch.arange(n, device=gpu)[:, None] // T, torch.arange(n, device=gpu)[None, :] // T
    assert bool((C[bj > bi] == 7.0).all())
    with pytest.raises(RuntimeError, match="square"):
        ops.gemm_f64_batched([ops.Gemm64(V[:50], V, tri=ops.TRI64_C_LOWER)])


@pytest.mark.gpu
def test_gemm_f64_many_products_per_call(gpu):
    """More products than one launch carries (24 descriptors travel as kernel arguments): 70 small ones and three
    macro-tile ones in one call, with triangular flags on some - every product against torch's fp64 result."""
    from curvature_amd import ops
    torch.manual_seed(9)
    jobs, wants = [], []
    for k in range(70):
        m, n, kk = 5 + 3 * k, 200 - 2 * k, 17 + k
        a = torch.randn(m, kk, dtype=torch.float64, device=gpu)
        b = torch.randn(kk, n, dtype=torch.float64, device=gpu)
        jobs.append(ops.Gemm64(a, b))
        wants.append(a @ b)
    for n in (1024, 1100, 1300):
        lo = torch.tril(torch.randn(n, n, dtype=torch.float64, device=gpu))
        up = torch.triu(torch.randn(n, n, dtype=torch.float64, device=gpu))
        jobs.insert(7 * (n % 9), ops.Gemm64(up, lo, tri=ops.TRI64_A_UPPER | ops.TRI64_B_LOWER))
        wants.insert(7 * (n % 9), up @ lo)
    outs = ops.gemm_f64_batched(jobs)
    for c, want in zip(outs, wants):
        assert c.shape == want.shape
        assert float((c - want).abs().max()) <= 1e-12 * max(float(want.abs().max()), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [200, 1100])
def test_gemm_f64_fused_epilogues(gpu, n):
    """The two epilogues INF.pre_sampler uses: C = alpha A B + beta E with E not C (also where a triangular cut leaves a
    tile's K range empty: the tile must still be written), and the float32 output alpha * rs[i] * cs[j] * (A B)[i, j];
    on 64-wide and on macro tiles."""
    from curvature_amd import ops
    torch.manual_seed(n)
    lo = [torch.tril(torch.randn(n, n, dtype=torch.float64, device=gpu)) for _ in range(2)]
    T = torch.full((n, n), float("nan"), dtype=torch.float64, device=gpu)
    ops.gemm_f64_batched([ops.Gemm64(lo[1], lo[0], T, alpha=-1.0, beta=1.0, E=lo[0], tri=ops.TRI64_A_LOWER | ops.TRI64_B_LOWER)])
    want = lo[0] - lo[1] @ lo[0]
    assert float((T - want).abs().max()) <= 1e-12 * float(want.abs().max())
    assert float(torch.triu(T, 1).abs().max()) == 0.0          # tiles above the diagonal: beta * E = 0, written
    rs = torch.rand(n, device=gpu) + 0.5
    cs = torch.rand(n, device=gpu) + 0.5
    out = torch.full((n, n), float("nan"), dtype=torch.float32, device=gpu)
    ops.gemm_f64_batched([ops.Gemm64(lo[0].t(), T, tri=ops.TRI64_A_UPPER | ops.TRI64_B_LOWER, out32=out, row_scale=rs, col_scale=cs,
                                     alpha=0.5)])
    ref = (0.5 * rs.double()[:, None] * (lo[0].t() @ T) * cs.double()[None, :]).float()
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    out2 = torch.empty(n, n, dtype=torch.float32, device=gpu)
    ops.gemm_f64_batched([ops.Gemm64(lo[0].t(), T, out32=out2)])          # no scale vectors, dense
    assert float((out2 - (lo[0].t() @ T).float()).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.gpu
def test_eigh_varied_spectra_batched(gpu):
    """One batched call over matrices that stress different parts of the block-Jacobi iteration (its sub-problem visits
    rotate the cross pairs of two 32-blocks, each block's inner pairs once per sweep): sizes around the 32 / 64 block edges,
    low rank, identity and diagonal (nothing to rotate), indefinite, rank one, two eigenvalues of high multiplicity.
    Residual, orthogonality and ascending order for every one of them."""
    from curvature_amd import ops
    torch.manual_seed(1)
    mats, names = [], []
    for n in (1, 2, 33, 64, 65, 129, 257, 600):
        X = torch.randn(n, max(1, n // 3), device=gpu)
        mats.append((X @ X.t() / X.shape[1]).contiguous()); names.append(f"lowrank{n}")
        mats.append(torch.eye(n, device=gpu)); names.append(f"eye{n}")
        mats.append(torch.diag(torch.rand(n, device=gpu) + 0.1)); names.append(f"diag{n}")
        Y = torch.randn(n, n, device=gpu)
        mats.append(((Y + Y.t()) / 2).contiguous()); names.append(f"indefinite{n}")
        v = torch.randn(n, 1, device=gpu)
        mats.append((v @ v.t()).contiguous()); names.append(f"rank1_{n}")
        Q, _ = torch.linalg.qr(torch.randn(n, n, device=gpu, dtype=torch.float64))
        lam = torch.cat([torch.full((n // 2,), 1.0), torch.full((n - n // 2,), 3.0)]).to(gpu, torch.float64)
        mats.append(((Q * lam) @ Q.t()).float().contiguous()); names.append(f"cluster{n}")
    vecs, vals = ops.eigh(mats, with_values=True)
    for M, U, w, name in zip(mats, vecs, vals, names):
        Md, Ud, wd = M.double(), U.double(), w.double()
        scale = max(float(Md.abs().max()), 1e-30)
        assert float((Md @ Ud - Ud * wd).abs().max()) <= 5e-5 * scale, name
        assert float((Ud.t() @ Ud - torch.eye(M.shape[0], device=gpu, dtype=torch.float64)).abs().max()) <= 1e-5, name
        if M.shape[0] > 1:
            assert bool((wd[1:] >= wd[:-1] - 1e-6 * scale).all()), name


@pytest.mark.gpu
def test_eigh_coordinate_order_cases(gpu):
    """The iteration runs on P sym(F) P^T with F's diagonal sorted descending and starts V at P^T: inputs where that
    permutation is the whole story or is degenerate.  A diagonal matrix in any order (its vectors are exactly the unit
    vectors, values exact), ties on the diagonal (equal and zero entries), the zero matrix, negative diagonals, and a
    graded matrix given in ascending, descending and shuffled order - all three orders give the same eigenvalues."""
    from curvature_amd import ops
    torch.manual_seed(3)
    n = 200
    dvals = torch.cat([torch.linspace(-2.0, 5.0, 150), torch.full((30,), 1.25), torch.zeros(20)])[torch.randperm(n)].to(gpu)
    (U,), (w,) = ops.eigh([torch.diag(dvals)], with_values=True)
    assert torch.equal(w, torch.sort(dvals).values)
    assert torch.equal(U.abs(), torch.eye(n, device=gpu)[:, torch.sort(dvals, stable=True).indices].abs()) or \
        float((torch.diag(dvals) @ U - U * w).abs().max()) == 0.0        # (inside the tie groups any unit vectors do)
    assert torch.equal((U != 0).sum(0), torch.ones(n, dtype=torch.long, device=gpu))
    (U0,), (w0,) = ops.eigh([torch.zeros(70, 70, device=gpu)], with_values=True)
    assert torch.equal(w0, torch.zeros(70, device=gpu)) and torch.equal(U0.abs().sum(0), torch.ones(70, device=gpu))
    m = 300
    Q, _ = torch.linalg.qr(torch.randn(m, m, dtype=torch.float64, device=gpu))
    G = torch.diag(torch.logspace(0, -4, m, dtype=torch.float64, device=gpu))
    G = G + 1e-3 * (Q * torch.logspace(0, -4, m, dtype=torch.float64, device=gpu)) @ Q.t()      # graded, PSD
    G = ((G + G.t()) / 2).float()
    perm = torch.randperm(m, device=gpu)
    flip = torch.arange(m - 1, -1, -1, device=gpu)
    variants = [G.contiguous(), G[flip][:, flip].contiguous(), G[perm][:, perm].contiguous()]
    vecs, vals = ops.eigh(variants, with_values=True)
    ref = torch.linalg.eigvalsh(G.double())
    for M, Uv, wv in zip(variants, vecs, vals):
        Md, Ud, wd = M.double(), Uv.double(), wv.double()
        assert float(torch.linalg.norm(Md @ Ud - Ud * wd) / torch.linalg.norm(Md)) < 1e-6
        assert float((Ud.t() @ Ud - torch.eye(m, dtype=torch.float64, device=gpu)).abs().max()) < 1e-6
        assert float((wd - ref).abs().max()) < 1e-6 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("count", [3, 70])
def test_chol_factor_inverse_with_right_hand_side(gpu, count):
    """`chol_factor_inverse(rhs=...)`: chol(M + d I)^-1 R by forward substitution inside the sweep (INF.pre_sampler's
    B_c^-1 A_c^-1 without B_c^-1), for sizes around the 64 / 256 block edges, with ordinary (inverting) matrices in the
    same call; against torch.linalg.solve_triangular in fp64."""
    from curvature_amd import ops
    torch.manual_seed(count)
    base = [100, 256, 300, 513, 700, 1100]
    sizes = base[:3] if count == 3 else (base[:4] * 18)[:count]
    mats, rhs, want = [], [], []
    for k, n in enumerate(sizes):
        X = torch.randn(n, n + 5, dtype=torch.float64, device=gpu)
        V = X @ X.t() / n
        C = torch.linalg.cholesky(V + torch.eye(n, dtype=torch.float64, device=gpu))
        mats.append(V)
        if k % 3 == 2:
            rhs.append(None)
            want.append(torch.linalg.inv(C))
        else:
            R = torch.tril(torch.randn(n, n, dtype=torch.float64, device=gpu))
            rhs.append(R)
            want.append(torch.linalg.solve_triangular(C, R, upper=False))
    got = ops.chol_factor_inverse(mats, [1.0] * len(mats), rhs=rhs)
    for g, w in zip(got, want):
        assert float((g - w).abs().max()) <= 1e-9 * float(w.abs().max())
        assert float(torch.triu(g, 1).abs().max()) == 0.0


@pytest.mark.gpu
def test_eigh_low_rank_path_and_its_fallback(gpu):
    """Wide matrices whose numerical rank is below half their width are decomposed through their range (csrc/eigh_lowrank.hip behind curv_syevd:
    Gaussian range finder, rank from a thresholded Cholesky of the Gram matrix, three Cholesky-QR passes, the iteration on
    the projected k x k matrix, an orthonormal complement for the null space); a wide matrix of full rank must fall back to
    the iteration on the whole matrix.  Same bars as the iteration: residual, orthogonality, ascending order, eigenvalues
    against torch's fp64 solver; the path taken is checked through `ops.eigh.last_lowrank`."""
    from curvature_amd import ops
    torch.manual_seed(5)
    n = 2304
    X = torch.relu(torch.randn(n, 600, device=gpu) + 0.5)               # rank 600: a KFAC factor with N L = 600 samples
    low = (X @ X.t() / 600).contiguous()
    Y = torch.randn(n, 2 * n, device=gpu)
    full = (Y @ Y.t() / (2 * n)).contiguous()
    small = (X[:500] @ X[:500].t() / 600).contiguous()                  # below the width the path looks at
    vecs, vals = ops.eigh([low, full, small], with_values=True)
    assert ops.eigh.last_lowrank == 1
    for F, U, w in zip((low, full, small), vecs, vals):
        Fd, Ud, wd = F.double(), U.double(), w.double()
        m = F.shape[0]
        assert U.dtype == torch.float32 and U.shape == F.shape and w.shape == (m,)
        assert bool((wd[1:] >= wd[:-1]).all())
        assert float(torch.linalg.norm(Ud.t() @ Ud - torch.eye(m, device=gpu, dtype=torch.float64))) < 1e-5 * m ** 0.5
        assert float(torch.linalg.norm(Fd @ Ud - Ud * wd)) < 1e-5 * float(torch.linalg.norm(Fd))
        ref = torch.linalg.eigvalsh(Fd)
        assert float(torch.linalg.norm(wd - ref)) < 1e-5 * float(torch.linalg.norm(ref))
    # deterministic, and independent of what else is in the batch (what a layer-sharded rank relies on)
    (U2,), (w2,) = ops.eigh([low], with_values=True)
    assert torch.equal(U2, vecs[0]) and torch.equal(w2, vals[0])


@pytest.mark.gpu
def test_eigh_low_rank_path_sharp_rank(gpu):
    """A classifier's input factor: 2049 wide, 32 samples - rank exactly 32, no oversampling for the range finder (its basis
    then depends on one step of subspace iteration; without it the residual was 2e-5)."""
    from curvature_amd import ops
    torch.manual_seed(0)
    X = torch.cat([torch.relu(torch.randn(2048, 32, device=gpu)), torch.ones(1, 32, device=gpu)])
    F = (X @ X.t() / 32).contiguous()
    (U,), (w,) = ops.eigh([F], with_values=True)
    assert ops.eigh.last_lowrank == 1 and ops.eigh.last_ranks == {0: 32}
    Fd, Ud, wd = F.double(), U.double(), w.double()
    assert float(torch.linalg.norm(Fd @ Ud - Ud * wd)) < 5e-6 * float(torch.linalg.norm(Fd))
    assert float(torch.linalg.norm(Ud.t() @ Ud - torch.eye(2049, device=gpu, dtype=torch.float64))) < 1e-5 * 2049 ** 0.5
    assert bool((wd[1:] >= wd[:-1]).all()) and int((wd > 1e-3).sum()) == 32


@pytest.mark.gpu
def test_eigh_low_rank_path_does_not_depend_on_the_batch(gpu):
    """Layer-sharded ranks decompose different subsets of a model's factors and must end with the same eigenvectors bit for
    bit: the projection path's products, factorisations and Gaussian matrices depend on the matrix alone."""
    from curvature_amd import ops
    torch.manual_seed(3)
    X = torch.randn(2304, 200, device=gpu)
    big = (X @ X.t() / 200).contiguous()
    Y = torch.randn(2048, 64, device=gpu)
    other = (Y @ Y.t() / 64).contiguous()
    small = [torch.randn(96, 96, device=gpu) for _ in range(3)]
    small = [(s + s.t()).contiguous() for s in small]
    (U_alone,), (w_alone,) = ops.eigh([big], with_values=True)
    assert ops.eigh.last_lowrank == 1
    U_many, w_many = ops.eigh([small[0], other, big, small[1], small[2]], with_values=True)
    assert ops.eigh.last_lowrank == 2 and sorted(ops.eigh.last_ranks) == [1, 2]
    assert torch.equal(U_alone, U_many[2]) and torch.equal(w_alone, w_many[2])


@pytest.mark.gpu
def test_lowrank_path_of_the_library_matches_the_python_glue(gpu):
    """The projection path lives behind `curv_syevd` since round 6 (csrc/eigh_lowrank.hip); round 5's Python glue over the same
    entry points is kept as its checker (tools/eigh_lowrank_reference.py): same Gaussian matrices, same products, same
    factorisations, same order - the eigenvectors and eigenvalues must agree bit for bit, on a rank-deficient matrix, on one
    whose rank is hit exactly, and on a full-rank one that both leave to the iteration on the whole matrix."""
    import os
    import sys
    from curvature_amd import ops
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import eigh_lowrank_reference as ref
    torch.manual_seed(11)
    X = torch.relu(torch.randn(2304, 500, device=gpu) + 0.3)
    low = (X @ X.t() / 500).contiguous()
    Z = torch.cat([torch.relu(torch.randn(2048, 32, device=gpu)), torch.ones(1, 32, device=gpu)])
    sharp = (Z @ Z.t() / 32).contiguous()
    Y = torch.randn(2048, 4096, device=gpu)
    full = (Y @ Y.t() / 4096).contiguous()
    mats = [low, sharp, full]
    want = ref.eigh_lowrank(mats, [0, 1, 2])
    assert sorted(want) == [0, 1]                                     # the full-rank matrix is left to the iteration
    vecs, vals = ops.eigh(mats, with_values=True)
    assert ops.eigh.last_ranks == {i: k for i, (_, _, k) in want.items()}
    for i, (U, w, _) in want.items():
        assert torch.equal(U, vecs[i]) and torch.equal(w, vals[i]), i

