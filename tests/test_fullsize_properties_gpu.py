"""Size-independent properties at the full benchmark configuration (ResNet-50, N = 32, 3x224x224): the oracle
would take minutes there, so the HIP path is checked through identities that hold at any size.

  factor build   exact symmetry; tr(A) = ||unfold(x)||^2 / (N L) (+1 for the ones row); accumulating the same
                 batch twice doubles the factors; the grouped launch equals per-layer launches bit for bit
  invert         L lower triangular and (L L^T)(sqrt(s) F + sqrt(n) I) = I for the largest factors
  sample         sample(layer, z) = (L_A z L_G^T)^T against a torch fp64 product, layer order = modules() order
"""
import pytest
import torch

from conftest import identity_residual_bound, rel_fro


@pytest.fixture(scope="module")
def resnet50_kfac(gpu):
    from curvature_amd import models
    from curvature_amd.curvatures import KFAC
    torch.manual_seed(0)
    model = models.resnet50().to(gpu).train()
    kfac = KFAC(model)
    x = torch.randn(32, 3, 224, 224, device=gpu)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    kfac.update(batch_size=32)
    torch.cuda.synchronize()
    return model, kfac


@pytest.mark.gpu
def test_factor_build_properties(gpu, resnet50_kfac):
    from curvature_amd import ops
    model, kfac = resnet50_kfac
    layers = kfac._layers()
    assert len(layers) == 54 and list(kfac.state.keys()) == layers          # modules() order, bit-exact indexing
    for layer in layers:
        A, G = kfac.state[layer]
        assert torch.equal(A, A.t()) and torch.equal(G, G.t())              # exactly symmetric
        assert torch.isfinite(A).all() and torch.isfinite(G).all()
        x, g = kfac.record[layer]
        N = x.shape[0]
        L = g.shape[2] * g.shape[3] if g.dim() == 4 else 1
        # tr(G) = (N / L) ||g||^2   (the hook's factor N of the reference folded into the scale)
        tr_g = float(g.double().pow(2).sum()) * N / L
        assert abs(float(torch.trace(G.double())) - tr_g) <= 1e-5 * abs(tr_g)
        if layer.__class__.__name__ == "Conv2d" and layer.kernel_size == (1, 1) and layer.stride == (1, 1):
            tr_a = float(x.double().pow(2).sum()) / (N * L)                  # unfold of a 1x1 conv is x itself
            assert abs(float(torch.trace(A.double())) - tr_a) <= 1e-5 * abs(tr_a)
        elif layer.__class__.__name__ == "Conv2d":
            unf = torch.nn.functional.unfold(x[:4].double(), layer.kernel_size, padding=layer.padding, stride=layer.stride)
            # diagonal of A restricted to 4 samples is a lower bound of the full diagonal (sum of squares)
            d4 = unf.pow(2).sum(dim=(0, 2)) / (N * L)
            assert (torch.diagonal(A.double())[:d4.numel()] + 1e-9 >= d4 * (1 - 1e-6)).all()
    # linearity: the same batch once more doubles every factor
    before = {l: [kfac.state[l][0].clone(), kfac.state[l][1].clone()] for l in layers}
    kfac.update(batch_size=32)
    torch.cuda.synchronize()
    for l in layers:
        assert rel_fro(kfac.state[l][0], 2 * before[l][0]) < 1e-6 and rel_fro(kfac.state[l][1], 2 * before[l][1]) < 1e-6
    # the grouped launch is deterministic and independent of what else is in the launch: three layers alone
    for l in (layers[0], layers[20], layers[53]):
        x, g = kfac.record[l]
        N = x.shape[0]
        if l.__class__.__name__ == "Conv2d":
            Lp = g.shape[2] * g.shape[3]
            A1 = torch.empty_like(before[l][0])
            ops.kfac_accumulate([ops.FactorJob(x.detach().contiguous(), A1, l.kernel_size, l.stride, l.padding,
                                               l.bias is not None, 1.0 / (N * Lp), True)])
        else:
            A1 = torch.empty_like(before[l][0])
            ops.kfac_accumulate([ops.FactorJob(x.detach().contiguous(), A1, (1, 1), (1, 1), (0, 0), l.bias is not None,
                                               1.0 / N, True)])
        torch.cuda.synchronize()
        assert rel_fro(A1, before[l][0]) < 1e-6
    for l in layers:                                   # restore single-batch factors for the tests below
        kfac.state[l][0].copy_(before[l][0])
        kfac.state[l][1].copy_(before[l][1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", [
    "conv1",                # 7x7 stride 2 pad 3 stem, n = 147, L = 12544           (patch kernel)
    "layer2.0.conv2",       # 3x3 stride 2, C = 128, 56x56 -> 28x28                 (patch kernel)
    "layer2.1.conv2",       # 3x3 stride 1, C = 128, 28x28: shifted correlations    (syrk_corr + flat kernel)
    "layer3.1.conv2",       # 3x3 stride 1, C = 256, 14x14: shifted correlations
    "layer4.1.conv2",       # 3x3 stride 1, C = 512, 7x7:   shifted correlations
    "layer1.0.conv2",       # 3x3 stride 1, C = 64, 56x56                           (patch kernel)
    "layer3.0.downsample.0",  # 1x1 stride 2, 512 -> 1024
])
def test_factor_build_matches_the_oracle_at_resnet50_geometry(gpu, resnet50_kfac, name):
    """The factors of the full-size N = 32 batch against the fp64 oracle (curvatures.py:329-350 restated in
    oracle.kfac_factors) at ResNet-50's real geometries — in particular the three shifted-correlation classes,
    whose k-slicing, Xp row pitch and stage counts differ from every small test geometry.  1e-4 relative Frobenius
    (north_star); all blocks of A are compared, not only its diagonal."""
    import oracle.curvature_oracle as o
    model, kfac = resnet50_kfac
    layer = dict(model.named_modules())[name]
    x, g = kfac.record[layer]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))          # the box's 256 logical CPUs make torch's CPU GEMM crawl
    try:
        A, G = o.kfac_factors(x.detach().double().cpu(), g.detach().double().cpu(),
                              has_bias=layer.bias is not None, **o.layer_geometry(layer))
    finally:
        torch.set_num_threads(threads)
    ea, eg = rel_fro(kfac.state[layer][0], A), rel_fro(kfac.state[layer][1], G)
    assert ea < 1e-4 and eg < 1e-4, (name, ea, eg)
    # off-diagonal (kh, kw) blocks on their own: a wrong shift or border strip would hide behind the diagonal's mass
    n0 = A.shape[0]
    off = ~torch.eye(n0, dtype=torch.bool)
    d = (kfac.state[layer][0].double().cpu() - A)[off]
    assert float(d.norm() / A[off].norm()) < 1e-4


@pytest.mark.gpu
def test_invert_and_sample_properties(gpu, resnet50_kfac):
    model, kfac = resnet50_kfac
    kfac.invert(add=1.0, multiply=1000.0)
    layers = kfac._layers()
    big = sorted(layers, key=lambda l: -kfac.state[l][0].shape[0])[:2] + [layers[0], layers[53]]
    for layer in big:
        for F, Lf in zip(kfac.state[layer], kfac.inv_state[layer]):
            n = F.shape[0]
            assert torch.equal(Lf, torch.tril(Lf))
            M = (1000.0 ** 0.5) * F.double() + torch.eye(n, device=gpu, dtype=torch.float64)
            M = (M + M.t()) / 2
            R = (Lf.double() @ Lf.double().t()) @ M - torch.eye(n, device=gpu, dtype=torch.float64)
            assert float(torch.linalg.norm(R)) / n ** 0.5 < identity_residual_bound(M), (n, float(torch.linalg.norm(R)) / n ** 0.5)
    layer = big[0]
    LA, LG = kfac.inv_state[layer]
    torch.manual_seed(3)
    z = torch.randn(LA.shape[0], LG.shape[0], device=gpu)
    s = kfac.sample(layer, z=z)
    ref = (LA.double() @ z.double() @ LG.double().t()).t()
    assert rel_fro(s, ref) < 1e-5
    mean = {k: v.clone() for k, v in kfac.model_state.items()}
    kfac.sample_and_replace()
    torch.cuda.synchronize()
    state = model.state_dict()
    changed = sum(int(not torch.equal(state[k], mean[k])) for k in state)
    assert all(torch.isfinite(v).all() for v in state.values())
    assert changed == 54 + 1                      # 54 weights and the one bias (fc); BatchNorm tensors restored


def _fp64_oracle_L(F, add, mul):
    """oracle.chol_of_inverse on the matrix the reference hands to LAPACK: the damping formed in fp32 exactly as
    curvatures.py:368-375 forms it, everything behind it in fp64 on the host cores."""
    import oracle.curvature_oracle as o
    Fc = F.detach().cpu()
    reg = torch.tensor(mul ** 0.5, dtype=torch.float32) * Fc + torch.diag(Fc.new_full((Fc.shape[0],), add ** 0.5))
    M = ((reg + reg.t()) / 2.0).double()
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        return o.chol_of_inverse(M)
    finally:
        torch.set_num_threads(threads)


@pytest.mark.gpu
@pytest.mark.parametrize("add,mul", [(1.0, 1000.0), (69.0, 25771.0)])
def test_forward_error_of_L_on_the_wide_resnet50_factors(gpu, resnet50_kfac, add, mul):
    """Entry-wise parity of ``L = chol_lower((sqrt(s) F + sqrt(n) I)^-1)`` (curvatures.py:378-379) for the factors that
    carry the headline workload: the three 4608-wide A factors, one 2304-wide and the classifier's 2049-wide one, at
    the bench's (1, 1000) and at the README's ResNet-50 pair (69, 25771; README.rst:262), against the fp64 oracle
    on the CPU, through BOTH launch forms of the sweep: the whole model in one call (per-step launches, outer panels
    of 6 block columns) and a call with at most 64 factors (block squares swept by one launch).  The triangular
    inverse outside the block squares is accumulated in fp32 over up to 72 block steps (supd32_kernel); the bar is
    1e-6 relative Frobenius, two orders below the north star's 1e-4."""
    from curvature_amd import ops
    model, kfac = resnet50_kfac
    layers = kfac._layers()
    wide = [l for l in layers if kfac.state[l][0].shape[0] == 4608]
    mid = next(l for l in layers if kfac.state[l][0].shape[0] == 2304)
    fc = layers[53]
    assert len(wide) == 3 and kfac.state[fc][0].shape[0] == 2049
    picks = wide + [mid, fc]
    kfac.invert(add=add, multiply=mul)                                        # whole model: 108 factors in one sweep
    factors = [kfac.state[l][0] for l in picks]
    chain = ops.chol_inv_lower(factors, [add] * len(factors), [mul] * len(factors))   # <= 64 factors: chain form
    torch.cuda.synchronize()
    worst = 0.0
    for layer, F, L_chain in zip(picks, factors, chain):
        exact = _fp64_oracle_L(F, add, mul)
        L_model = kfac.inv_state[layer][0]
        for form, L in (("model", L_model), ("chain", L_chain)):
            assert torch.equal(L, torch.tril(L))
            err = rel_fro(L, exact)
            worst = max(worst, err)
            print(f"n = {F.shape[0]} ({add}, {mul}) {form} sweep: forward error of L {err:.2e}")
            assert err < 1e-6, (F.shape[0], form, err)
    kfac.invert(add=1.0, multiply=1000.0)                                     # what the tests below expect


@pytest.mark.gpu
def test_a_rank_share_inverts_like_the_whole_model(gpu, resnet50_kfac):
    """Layer sharding at full size: the factors of one rank's layers (here the largest layer alone, and a nine-layer
    share) go through the chain-bound kernels of the inversion (at most 64 factors in a call: block squares swept by
    one launch, quarter-form panel products), the whole model through the per-step launches.  Both must satisfy the
    defining identity, and agree with each other to output rounding."""
    from curvature_amd import ops
    model, kfac = resnet50_kfac
    kfac.invert(add=1.0, multiply=1000.0)
    layers = kfac._layers()
    largest = max(layers, key=lambda l: kfac.state[l][0].shape[0])
    share = [layers[i] for i in (2, 5, 15, 23, 25, 27, 33, 40, 49)]
    for group in ([largest], share):
        factors = [F for layer in group for F in kfac.state[layer]]
        outs = ops.chol_inv_lower(factors, [1.0] * len(factors), [1000.0] * len(factors))
        whole = [L for layer in group for L in kfac.inv_state[layer]]
        for F, L_rank, L_whole in zip(factors, outs, whole):
            n = F.shape[0]
            assert torch.equal(L_rank, torch.tril(L_rank))
            assert rel_fro(L_rank, L_whole.cpu()) < 1e-6
            if n >= 1024:
                M = (1000.0 ** 0.5) * F.double() + torch.eye(n, device=gpu, dtype=torch.float64)
                M = (M + M.t()) / 2
                R = (L_rank.double() @ L_rank.double().t()) @ M - torch.eye(n, device=gpu, dtype=torch.float64)
                assert float(torch.linalg.norm(R)) / n ** 0.5 < identity_residual_bound(M)


@pytest.mark.gpu
def test_resnet18_full_estimator_chain(gpu):
    """SURVEY 8(d) config 3 at full size: Diagonal -> KFAC -> EFB -> INF(rank = 100) -> invert -> sample on an
    ImageNet ResNet-18 (random init, N = 32 as SURVEY 8(d) prescribes).  No oracle at this size; checked through
    what must hold anyway."""
    from curvature_amd import models
    from curvature_amd.curvatures import Diagonal, KFAC, EFB, INF
    torch.manual_seed(0)
    N, rank = 32, 100
    model = models.resnet18().to(gpu).train()
    diag, kfac = Diagonal(model), KFAC(model)
    x = torch.randn(N, 3, 224, 224, device=gpu)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    diag.update(N)
    kfac.update(N)
    layers = kfac._layers()
    assert len(layers) == 21 and list(diag.state.keys()) == layers
    efb = EFB(model, kfac.state)
    efb.update(N)
    inf = INF(model, diag.state, kfac.state, efb.state, eigvecs=efb.eigvecs)   # reuse, do not decompose again
    # eigenvectors of the largest factor: orthonormal, and they diagonalise it
    big = max(layers, key=lambda l: kfac.state[l][0].shape[0])
    A = kfac.state[big][0].double()
    U = efb.eigvecs[big][0].double()
    n = A.shape[0]
    assert n == 4608
    assert (U.t() @ U - torch.eye(n, dtype=torch.float64, device=gpu)).abs().max().item() < 1e-5
    D = U.t() @ A @ U
    off = D - torch.diag(torch.diagonal(D))
    assert off.norm().item() <= 1e-5 * A.norm().item()
    w = torch.diagonal(D)
    assert (w[1:] >= w[:-1] - 1e-6 * w.abs().max()).all()     # ascending, like symeig
    for layer in layers:
        assert (efb.state[layer] >= 0).all()                  # Lambda accumulates squares
    inf.update(rank=rank)
    for layer in layers:
        ua, ug, lam, corr = inf.state[layer]
        n_l, m_l = kfac.state[layer][0].shape[0], kfac.state[layer][1].shape[0]
        a, b = ua.shape[1], ug.shape[1]
        assert ua.shape[0] == n_l and ug.shape[0] == m_l and 1 <= a <= min(rank, n_l) and 1 <= b <= min(rank, m_l)
        assert a * b >= min(rank, n_l * m_l) and lam.numel() == a * b and corr.numel() == n_l * m_l
        # the selected columns are columns of the full eigenvector matrices, in ascending index order
        UA = inf.eigvecs[layer][0]
        idx = [int((UA == ua[:, k:k + 1]).all(dim=0).nonzero()[0]) for k in (0, a - 1)]
        assert idx[0] <= idx[1]
    for est in (diag, kfac, efb, inf):
        est.invert(1.0, 1000.0)
        est.sample_and_replace()
        changed = 0
        for k, v in model.state_dict().items():             # mean = the state captured at construction (:41)
            assert torch.isfinite(v).all(), k
            changed += int(not torch.equal(v, est.model_state[k]))
        assert changed == 22                                   # 21 weights + the fc bias; BatchNorm restored
    L_A = kfac.inv_state[big][0]
    assert torch.equal(L_A, torch.tril(L_A))


@pytest.mark.gpu
def test_config5_resnet50_inf_chain(gpu, resnet50_kfac):
    """SURVEY 8(d) config 5 on one GPU: ResNet-50 (N = 32) through Diagonal -> KFAC -> EFB -> INF(rank = 100) ->
    invert at (1, 1000) and at the README's INF values (145307, 60; README.rst:263) -> sample_and_replace, with
    the size-independent properties of the ResNet-18 chain plus a per-layer comparison with the oracle
    (oracle.inf_invert / inf_sampler, fp64) on the layers small enough for its explicit Kronecker matrix."""
    import oracle.curvature_oracle as o
    from curvature_amd.curvatures import Diagonal, EFB, INF
    model, kfac = resnet50_kfac
    N, rank = 32, 100
    layers = kfac._layers()
    diag = Diagonal(model)
    diag.update(N)                                             # gradients of the fixture's backward pass
    efb = EFB(model, kfac.state)
    efb.update(N)
    assert list(diag.state.keys()) == layers and list(efb.state.keys()) == layers
    # eigenvectors of a large and a small factor: orthonormal, diagonalising, ascending
    for layer in (max(layers, key=lambda l: kfac.state[l][0].shape[0]), layers[1]):
        for F, U in zip(kfac.state[layer], efb.eigvecs[layer]):
            Fd, Ud = F.double(), U.double()
            n = Fd.shape[0]
            assert (Ud.t() @ Ud - torch.eye(n, dtype=torch.float64, device=gpu)).abs().max().item() < 1e-5
            D = Ud.t() @ Fd @ Ud
            assert (D - torch.diag(torch.diagonal(D))).norm().item() <= 1e-5 * Fd.norm().item()
            w = torch.diagonal(D)
            assert (w[1:] >= w[:-1] - 1e-6 * w.abs().max()).all()
    for layer in layers:
        assert (efb.state[layer] >= 0).all() and torch.isfinite(efb.state[layer]).all()
    inf = INF(model, diag.state, kfac.state, efb.state, eigvecs=efb.eigvecs)
    inf.update(rank=rank)
    small = []
    for layer in layers:
        ua, ug, lam, corr = inf.state[layer]
        n_l, m_l = kfac.state[layer][0].shape[0], kfac.state[layer][1].shape[0]
        a, b = ua.shape[1], ug.shape[1]
        assert ua.shape[0] == n_l and ug.shape[0] == m_l and 1 <= a <= min(rank, n_l) and 1 <= b <= min(rank, m_l)
        assert a * b >= min(rank, n_l * m_l) and lam.numel() == a * b and corr.numel() == n_l * m_l
        small.append((n_l * m_l * a * b, layer))
    # bit-exact index sets against the oracle's selection on the same Lambda, for three layers
    for _, layer in sorted(small, key=lambda t: t[0])[:3]:
        U_A, U_G = efb.eigvecs[layer]
        ref = o.inf_update(U_A.cpu(), U_G.cpu(), efb.state[layer].cpu(), diag.state[layer].cpu(), rank)
        ua, ug, lam, corr = inf.state[layer]
        assert torch.equal(ua.cpu(), ref[0]) and torch.equal(ug.cpu(), ref[1]) and torch.equal(lam.cpu(), ref[2])
        scale = float(torch.linalg.norm(diag.state[layer].double()))
        assert float(torch.linalg.norm(corr.double().cpu() - ref[3].double())) / scale < 1e-5
    for add, mul in ((1.0, 1000.0), (145307.0, 60.0)):
        pre = {l: tuple(t.clone() for t in inf.state[l]) for _, l in sorted(small, key=lambda t: t[0])[:3]}
        inf.invert(add, mul)
        for layer in layers:
            ua, ug, r, Pc = inf.inv_state[layer]
            assert torch.isfinite(r).all() and torch.isfinite(Pc).all() and Pc.shape == (ua.shape[1] * ug.shape[1],) * 2
        # oracle in fp64 on the three smallest layers: r, P_c and a sample for the same noise.  The reference keeps
        # sigma and r as fp32 tensors (curvatures.py:525-526) and P_c is a sensitive function of them when V_s^T V_s
        # is ill-conditioned (the line printed below says how far the exact P_c moves when r is NOT rounded to fp32),
        # so the chain is judged on the SAME fp32 sigma and r (what pre_sampler receives, :528), promoted to fp64
        # inside the oracle
        for layer, st in pre.items():
            ua, ug, lam, corr = (t.double().cpu() for t in st)
            _, sigma64, r64, _, Pc_exact_r = o.inf_invert(ua, ug, lam, corr, add, mul)
            _, _, r, Pc = inf.inv_state[layer]
            assert rel_fro(r, r64) < 1e-6
            sigma32 = (mul * st[2]).sqrt().double().cpu()             # fp32 arithmetic, as the reference / the kernel
            assert rel_fro(sigma32, sigma64) < 1e-6
            r32 = r.double().cpu()
            Pc64 = o.inf_pre_sampler_from_vtv(o.inf_vtv(ua, ug, sigma32, r32), sigma32)
            e_pc = rel_fro(Pc, Pc64)
            X = torch.randn(r.numel(), generator=torch.Generator().manual_seed(5))
            s = inf.sample(layer, X=X.to(gpu))
            e_s = rel_fro(s, o.inf_sampler(ua, ug, r32, Pc64, X.double()))
            moves = rel_fro(Pc64, Pc_exact_r)
            print(f"config 5, ({add}, {mul}), n*m={r.numel()}, ab={Pc.shape[0]}: P_c err {e_pc:.2e}, sample err {e_s:.2e}; "
                  f"P_c moves by {moves:.1e} when r is not rounded to fp32")
            # 1e-4 (north_star), except where the layer's own conditioning is worse than that: `moves` is how far the
            # EXACT P_c travels when its input r is rounded to the fp32 the reference stores it in (one rounding of
            # one input), and no fp32-storing chain can be held to less than that.  Which layers are that
            # ill-conditioned varies from run to run (MIOpen's backward pass is not deterministic)
            tol = max(1e-4, 2.0 * moves)
            assert e_s < tol, (add, mul, e_s, moves)
            assert e_pc < tol, (add, mul, e_pc, moves)
        inf.sample_and_replace()
        changed = 0
        for k, v in model.state_dict().items():
            assert torch.isfinite(v).all(), k
            changed += int(not torch.equal(v, inf.model_state[k]))
        assert changed == 54 + 1
    for est in (diag, efb):
        est.invert(1.0, 1000.0)
        est.sample_and_replace()
        changed = sum(int(not torch.equal(v, est.model_state[k])) for k, v in model.state_dict().items())
        assert changed == 54 + 1
    kfac.model.load_state_dict(kfac.model_state)
