"""Pin the CPU oracle (oracle/curvature_oracle.py) against outputs of the reference itself.

The golden vectors were produced by tools/make_golden.py importing /root/reference; see that script
for what each file holds.  These tests run on CPU (`-m "not gpu"`)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle.curvature_oracle as o
from conftest import rel_fro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LENET_GEOM = [dict(kernel_size=(5, 5), stride=(1, 1), padding=(2, 2)),
              dict(kernel_size=(5, 5), stride=(1, 1), padding=(0, 0)),
              dict(kernel_size=None, stride=None, padding=None),
              dict(kernel_size=None, stride=None, padding=None),
              dict(kernel_size=None, stride=None, padding=None)]


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def test_kron_doctest():
    """The reference's only known-answer test (curvature/utils.py:301-309)."""
    g = load("g10_kron.npz")
    assert torch.equal(o.kron(g["a"], g["b"]), g["ab"])
    assert torch.equal(g["ab"], torch.tensor([[0, 5, 0, 10], [6, 7, 12, 14], [0, 15, 0, 20], [18, 21, 24, 28]]))
    assert torch.allclose(o.kron(g["c"], g["d"]), g["cd"], rtol=0, atol=0)


def test_kfac_update_lenet():
    g = load("g1_kfac_lenet.npz")
    for li in range(5):
        A = G = None
        for b in range(3):
            a, gg = o.kfac_factors(g[f"b{b}_l{li}_x"], g[f"b{b}_l{li}_g"], has_bias=True, **LENET_GEOM[li])
            A, G = (a, gg) if A is None else (A + a, G + gg)
            if b in (0, 2):
                assert rel_fro(A, g[f"A_after{b + 1}_l{li}"]) < 1e-6
                assert rel_fro(G, g[f"G_after{b + 1}_l{li}"]) < 1e-6


def test_kfac_update_conv_shapes():
    g = load("g2_kfac_convshapes.npz")
    for li in range(5):
        geom = dict(kernel_size=None, stride=None, padding=None)
        if f"l{li}_geom" in g:
            k = g[f"l{li}_geom"].tolist()
            geom = dict(kernel_size=(k[0], k[1]), stride=(k[2], k[3]), padding=(k[4], k[5]))
        A, G = o.kfac_factors(g[f"l{li}_x"], g[f"l{li}_g"], has_bias=bool(g[f"l{li}_bias"]), **geom)
        assert rel_fro(A, g[f"l{li}_A"]) < 1e-6
        assert rel_fro(G, g[f"l{li}_G"]) < 1e-6


def test_diag_update_lenet():
    g = load("g1_kfac_lenet.npz")
    for li in range(5):
        d = sum(o.diag_update(g[f"b{b}_l{li}_gw"], g[f"b{b}_l{li}_gb"], 8) for b in range(3))
        assert rel_fro(d, g[f"diag_after3_l{li}"]) < 1e-6


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_kfac_invert(tag):
    g1, g3 = load("g1_kfac_lenet.npz"), load("g3_kfac_invert.npz")
    hyper = {"a": (0.5, 1), "b": (1.0, 1000.0), "c": (g3["c_add"].tolist(), g3["c_mul"].tolist())}[tag]
    for li in range(5):
        n, s = o.layer_hyper(hyper[0], hyper[1], li, 5)
        A, G = g1[f"A_after3_l{li}"], g1[f"G_after3_l{li}"]
        LA, LG = o.kfac_invert(A, G, n, s)
        # same algorithm as the reference, but fp32 LAPACK round-off depends on the host CPU / MKL code
        # path (up to ~1e-4 here between machines): the fp64 twin below is what pins the algorithm
        assert rel_fro(LA, g3[f"{tag}_LA_l{li}"]) < 1e-3
        assert rel_fro(LG, g3[f"{tag}_LG_l{li}"]) < 1e-3
        if True:                                     # every set has an fp64 twin since round 6 ("c64_*")
            LA64, LG64 = o.kfac_invert(A.double(), G.double(), n, s)
            assert rel_fro(LA64, g3[f"{tag}64_LA_l{li}"]) < 1e-6
            assert rel_fro(LG64, g3[f"{tag}64_LG_l{li}"]) < 1e-6


def test_kfac_sample_and_replace():
    g1, g3, g4 = load("g1_kfac_lenet.npz"), load("g3_kfac_invert.npz"), load("g4_kfac_sample.npz")
    for li in range(5):
        s = o.kfac_sample(g3[f"a_LA_l{li}"], g3[f"a_LG_l{li}"], g4[f"z_l{li}"])
        assert rel_fro(s, g4[f"sample_l{li}"]) < 1e-6
        w, b = o.replace(g4[f"sample_l{li}"], g1[f"w_l{li}"], g1[f"bias_l{li}"])
        assert torch.equal(w, g4[f"w_new_l{li}"]) and torch.equal(b, g4[f"b_new_l{li}"])


def test_eigenvectors_span():
    """Eigenvectors are unique only up to sign / rotation inside degenerate clusters:
    compare through residual and orthogonality, and through the invariant F U = U diag(w)."""
    g1, g5 = load("g1_kfac_lenet.npz"), load("g5_eigvecs_lenet.npz")
    for li in range(5):
        for side, key in (("A", "UA"), ("G", "UG")):
            F = g1[f"{side}_after3_l{li}"].double()
            U = o.eigenvectors(F)
            Ug = g5[f"{key}_l{li}"].double()
            n = F.shape[0]
            for V in (U, Ug):
                assert torch.linalg.norm(V.t() @ V - torch.eye(n, dtype=V.dtype)) < 1e-4 * n ** 0.5
                w = torch.diagonal(V.t() @ F @ V)
                assert torch.linalg.norm(F @ V - V * w) < 1e-4 * torch.linalg.norm(F)


def test_efb():
    g1, g5, g6 = load("g1_kfac_lenet.npz"), load("g5_eigvecs_lenet.npz"), load("g6_efb_lenet.npz")
    for li in range(5):
        UA, UG = g5[f"UA_l{li}"], g5[f"UG_l{li}"]
        lam = sum(o.efb_update(UA, UG, g1[f"b{b}_l{li}_gw"], g1[f"b{b}_l{li}_gb"]) for b in range(2))
        dia = sum(o.diag_update(g1[f"b{b}_l{li}_gw"], g1[f"b{b}_l{li}_gb"], 8) for b in range(2))
        assert rel_fro(lam, g6[f"lambda_l{li}"]) < 1e-5
        assert rel_fro(dia, g6[f"diags_l{li}"]) < 1e-6
        inv = o.rsqrt_affine(g6[f"lambda_l{li}"], 0.5, 2.0)
        assert rel_fro(inv, g6[f"inv_l{li}"]) < 1e-6
        s = o.efb_sample(UA, UG, g6[f"inv_l{li}"], g6[f"z_l{li}"])
        assert rel_fro(s, g6[f"sample_l{li}"]) < 1e-5


@pytest.mark.parametrize("tag,rank", [("r10", 10), ("r100", 100), ("rall", 10 ** 9)])
def test_inf_update(tag, rank):
    g5, g6, g7 = load("g5_eigvecs_lenet.npz"), load("g6_efb_lenet.npz"), load("g7_inf_update.npz")
    for li in range(5):
        ua, ug, lam, D, I, J = o.inf_update(g5[f"UA_l{li}"], g5[f"UG_l{li}"], g6[f"lambda_l{li}"],
                                            g6[f"diags_l{li}"], rank)
        if I is None:
            I, J = np.arange(ua.shape[1]), np.arange(ug.shape[1])
        assert np.array_equal(I, g7[f"{tag}_I_l{li}"].numpy())       # bit-exact index sets
        assert np.array_equal(J, g7[f"{tag}_J_l{li}"].numpy())
        if f"{tag}_lam_l{li}" in g7:
            assert torch.equal(lam, g7[f"{tag}_lam_l{li}"])
        if f"{tag}_D_l{li}" in g7:
            ref = g7[f"{tag}_D_l{li}"]
            # D = d - sif_diag cancels heavily; compare against the scale of the minuend
            scale = torch.linalg.norm(g6[f"diags_l{li}"].double())
            assert float(torch.linalg.norm(D.double() - ref.double()) / scale) < 1e-5


def test_inf_invert_and_sample():
    g5, g7, g8, g9 = (load("g5_eigvecs_lenet.npz"), load("g7_inf_update.npz"), load("g8_inf_invert.npz"),
                      load("g9_inf_sample.npz"))
    add, mul = float(g8["add"]), float(g8["mul"])
    for li in range(5):
        I, J = g7[f"r10_I_l{li}"], g7[f"r10_J_l{li}"]
        ua, ug = g5[f"UA_l{li}"][:, I], g5[f"UG_l{li}"][:, J]
        Dc, sigma, r, vtv, Pc = o.inf_invert(ua, ug, g7[f"r10_lam_l{li}"], g7[f"r10_D_l{li}"], add, mul)
        assert torch.equal(Dc, g8[f"Dclamped_l{li}"])
        assert rel_fro(sigma, g8[f"sigma_l{li}"]) < 1e-6
        assert rel_fro(r, g8[f"r_l{li}"]) < 1e-6
        assert rel_fro(vtv, g8[f"vtv_l{li}"]) < 1e-5
        # the fp32 chain (2 Cholesky + 3 inverses) is noisy in the reference itself: judge the fp32
        # restatement loosely and the fp64 restatement against the reference's fp64 twin tightly
        assert rel_fro(Pc, g8[f"Pc_l{li}"]) < 5e-2
        out64 = o.inf_invert(ua.double(), ug.double(), g7[f"r10_lam_l{li}"].double(),
                             g7[f"r10_D_l{li}"].double(), add, mul)
        assert rel_fro(out64[4], g8[f"Pc64_l{li}"]) < 1e-5
        s = o.inf_sampler(ua, ug, g8[f"r_l{li}"], g8[f"Pc_l{li}"], g9[f"X_l{li}"])
        assert rel_fro(s, g9[f"sample_l{li}"]) < 1e-5


def test_layer_tables_match_models():
    """Layer order / (n, m, L) of the build's own model definitions == the reference's (bit-exact)."""
    from curvature_amd import models
    with open(os.path.join(GOLD, "g11_layer_tables.json")) as fh:
        tables = json.load(fh)
    specs = {"lenet5": (models.lenet5(), (1, 28, 28)), "resnet18": (models.resnet18(), (3, 224, 224)),
             "resnet50": (models.resnet50(), (3, 224, 224))}
    for name, (model, shape) in specs.items():
        rows = models.layer_table(model, shape)
        ref = tables[name]
        assert len(rows) == len(ref)
        for a, b in zip(rows, ref):
            assert (a["index"], a["name"], a["kind"], a["n"], a["m"], a["L"], a["has_bias"]) == \
                   (b["index"], b["name"], b["kind"], b["n"], b["m"], b["L"], b["has_bias"]), (name, a, b)


def test_mc_fisher_loop():
    """The oracle's restatement of the reference driver loop (scripts/factors.py:47-61) against the factors
    the reference's own KFAC accumulated in that loop (golden g12)."""
    from curvature_amd import models
    g, g1 = load("g12_mc_fisher_lenet.npz"), load("g1_kfac_lenet.npz")
    model = models.lenet5()
    layers = [l for l in model.modules() if l.__class__.__name__ in ("Linear", "Conv2d")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    model.train()
    state = o.mc_fisher_kfac(model, [g[f"b{b}_x"] for b in range(2)], int(g["samples"]),
                             lambda logits, b, s: g[f"b{b}_s{s}_labels"])
    for li, layer in enumerate(layers):
        assert rel_fro(state[layer][0], g[f"A_l{li}"]) < 1e-6
        assert rel_fro(state[layer][1], g[f"G_l{li}"]) < 1e-6


def _rel(a, b):
    return float(torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()))


def test_inf_sampler_gauge_dependence_of_the_reference():
    """The measurement behind the gauge remark in tests/test_estimator_chain_gpu.py::test_inf_end_to_end_own_chain, on the CPU oracle (a restatement of the reference's sampler,
    pinned against g9): same state, eigenvector columns negated -> a different sample."""
    import oracle.curvature_oracle as o
    g1, g5, g6, g8, g9 = (load(n) for n in ("g1_kfac_lenet.npz", "g5_eigvecs_lenet.npz", "g6_efb_lenet.npz",
                                            "g8_inf_invert.npz", "g9_inf_sample.npz"))
    add, mul = float(g8["add"]), float(g8["mul"])
    moved = []
    for li in range(2):
        out = []
        for flip in (False, True):
            UA, UG = g5[f"UA_l{li}"].double(), g5[f"UG_l{li}"].double()
            if flip:
                UA, UG = UA.clone(), UG.clone()
                UA[:, ::2] *= -1
                UG[:, 1::3] *= -1
            lam = sum(o.efb_update(UA, UG, g1[f"b{b}_l{li}_gw"].double(), g1[f"b{b}_l{li}_gb"].double()) for b in range(2))
            ua, ug, l, D, _, _ = o.inf_update(UA, UG, lam, g6[f"diags_l{li}"].double(), 10)
            _, _, r, _, Pc = o.inf_invert(ua, ug, l, D, add, mul)
            out.append(o.inf_sampler(ua, ug, r, Pc, g9[f"X_l{li}"].double()))
        assert _rel(out[0], g9[f"sample_l{li}"]) < 1e-6          # unflipped: the reference's sample
        moved.append(_rel(out[1], out[0]))
    assert moved[0] > 5e-3 and moved[1] > 2e-2, moved                # 1.7e-2 and 8.2e-2 measured


def test_block_diagonal():
    """BlockDiagonal (curvatures.py:196-261) against golden g13: state after one and two batches, both inverse sets,
    the Linear layers' samples."""
    g = load("g13_block_diagonal.npz")
    for li in range(3):
        state = None
        for b in range(2):
            upd = o.block_update(g[f"b{b}_l{li}_gw"], g[f"b{b}_l{li}_gb"], 4)
            state = upd if state is None else state + upd
            assert torch.equal(state, g[f"state_after{b + 1}_l{li}"])
        assert rel_fro(o.block_invert(state, 0.5, 2.0), g[f"a_inv_l{li}"]) < 1e-5
        n, s = float(g["b_add"][li]), float(g["b_mul"][li])
        assert rel_fro(o.block_invert(state, n, s), g[f"b_inv_l{li}"]) < 1e-5
        if li > 0:        # the reference's sample only works for Linear layers
            smp = o.block_sample(g[f"a_inv_l{li}"], g[f"z_l{li}"], g[f"w_l{li}"].shape)
            assert torch.allclose(smp, g[f"sample_l{li}"], rtol=0, atol=1e-6)
            assert smp.shape == (g[f"w_l{li}"].shape[0], g[f"w_l{li}"].shape[1] + 1)
    # Conv2d: weight part as (out, -1), bias as the last column - what `_replace` consumes
    smp = o.block_sample(g["a_inv_l0"], g["z_l0"], g["w_l0"].shape)
    assert smp.shape == (2, 10)
    x = g["z_l0"] @ g["a_inv_l0"]
    assert torch.equal(smp[:, :9].reshape(-1), x[:18]) and torch.equal(smp[:, 9], x[18:])


def test_shifted_correlation_decomposition_of_a_3x3_factor():
    """The identity behind csrc/syrk_corr.hip, in fp64 on the CPU: the A factor of a 3x3 / stride 1 / pad 1
    convolution (oracle.unfold_input, i.e. F.unfold as in curvatures.py:329-337) equals, block by block for
    p = (kh, kw) >= q = (kh', kw'), the whole-image correlation at the relative shift minus the border row / column the
    window of (kh, kw) leaves out, plus the corner counted twice: 13 + 12 + 4 component matrices instead of 45 blocks."""
    import itertools
    torch.manual_seed(0)
    N, C, H, W = 3, 4, 6, 5
    x = torch.randn(N, C, H, W, dtype=torch.float64)
    U = o.unfold_input(x, (3, 3), (1, 1), (1, 1), False)            # (9 C, N H W), rows (c, kh, kw)
    A = U @ U.t()

    def corr(us, vs, dh, dw):
        out = torch.zeros(C, C, dtype=torch.float64)
        for u, v in itertools.product(us, vs):
            if 0 <= u + dh < H and 0 <= v + dw < W:
                out += torch.einsum('nc,nd->cd', x[:, :, u, v], x[:, :, u + dh, v + dw])
        return out
    rows, cols = range(H), range(W)
    components = set()
    for kh, kw, kh2, kw2 in itertools.product(range(3), repeat=4):
        if 3 * kh + kw < 3 * kh2 + kw2:
            continue
        dh, dw = kh2 - kh, kw2 - kw
        assert dh < 0 or (dh == 0 and dw <= 0)
        blk = corr(rows, cols, dh, dw)
        components.add(("F", dh, dw))
        if kh == 0 and dh == 0:
            blk -= corr([H - 1], cols, 0, dw); components.add(("RB", dw))
        if kh == 2 and dh == 0:
            blk -= corr([0], cols, 0, dw); components.add(("RT", dw))
        if kw == 0 and dw == 0:
            blk -= corr(rows, [W - 1], dh, 0); components.add(("CR", dh))
        if kw == 2 and dw == 0:
            blk -= corr(rows, [0], dh, 0); components.add(("CL", dh))
        if dh == 0 and dw == 0 and kh != 1 and kw != 1:
            blk += corr([H - 1 if kh == 0 else 0], [W - 1 if kw == 0 else 0], 0, 0); components.add(("PT", kh, kw))
        assert torch.allclose(blk, A[3 * kh + kw::9][:, 3 * kh2 + kw2::9], rtol=0, atol=1e-12)
    assert len(components) == 29
