#!/usr/bin/env python
"""The projection path of the eigensolver as Python glue over the C ABI - round 5's `ops._eigh_lowrank`, kept as the CHECKER
of its C++ successor (csrc/eigh_lowrank.hip, behind curv_syevd since round 6): the same products, factorisations and
Gaussian matrices through the same entry points, so the two must agree bit for bit
(tests/test_efb_inf_gpu.py::test_lowrank_path_of_the_library_matches_the_python_glue).  Not on the product path."""
import os
import sys
from typing import Sequence

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd.ops import (Gemm64, TRI64_B_UPPER, chol_factor_inverse, eigh, gemm_f64_batched, randn)  # noqa: E402

LOWRANK_PROBE = 0.5           # columns of the range finder over the width: ranks up to this share take the path
LOWRANK_MIN_N = 2048          # narrower matrices converge in a few cheap sweeps anyway
LOWRANK_GRAM_PIVOT = 1e-13    # relative pivot of the Gram matrix below which a direction counts as noise (3e-7 of ||F||)
LOWRANK_RESIDUAL = 3e-6       # ||F - P F P|| / ||F|| the projection must reach (the iteration's own bar above 1024: 5e-6)


def eigh_lowrank(mats: Sequence[torch.Tensor], index: Sequence[int]) -> dict:
    """Eigendecomposition of wide symmetric matrices whose numerical rank is below half their width - the Kronecker factor of a
    layer with more rows than samples went into it (ResNet-50's 4608-wide factors at N = 32: rank 1568) - without iterating on
    the null space: F = sym(F) in fp64;
      Y = F Omega (n x n/2, Gaussian)            range of F, columns in general position
      G = Y^T Y, Cholesky with a pivot threshold   -> k = numerical rank (the first pivot that falls to the threshold)
      Q = orth(Y[:, :k]), Q = orth(F Q)            three Cholesky-QR passes each (G's condition is F's squared); one step
                                                   of subspace iteration
      B = Q^T F Q (k x k);  accept iff ||F||^2 - ||B||^2 <= (3e-6 ||F||)^2     (P = Q Q^T is an orthogonal projector)
      B = W diag(lam) W^T                          the block-Jacobi iteration, on a matrix (k/n)^3 the size
      U = [ Z | Q W ],  w = [ 0 | lam ], sorted    Z = an orthonormal basis of the complement of Q (Gaussian, projected,
                                                   two Cholesky-QR passes): F Z is below the residual bar by construction
    Every product is curv_gemm_f64_batched, every factorisation curv_chol_factor_inverse; a matrix that fails any test
    (rank >= n/2, a failed factorisation, residual above the bar) is left to the caller's iteration on the whole matrix.
    Returns {index: (U, w, k)} of the matrices it decomposed.  Deterministic: the Gaussian matrices depend on n only."""
    out = {}
    log = eigh.lowrank_log = []                          # (index, what happened): diagnostics, tools/eigh_lowrank_breakdown.py
    dev = mats[0].device
    S, nS2, Y, G, r_of = [], [], [], [], []
    jobs = []
    for F in mats:
        n = F.shape[0]
        Fd = F.double()
        Sd = (Fd + Fd.t()) * 0.5
        S.append(Sd)
        nS2.append((Sd * Sd).sum())
        r = int(n * float(os.environ.get("CURV_EIGH_PROBE", LOWRANK_PROBE))) // 64 * 64
        r_of.append(r)
        om = randn((n, r), dev, 0x5EED0000 + n, 0).double()
        jobs.append(Gemm64(Sd, om))
    Y = gemm_f64_batched(jobs)
    G = gemm_f64_batched([Gemm64(y.t(), y) for y in Y])
    thr = torch.stack([g.diagonal().max() for g in G]).mul(LOWRANK_GRAM_PIVOT).tolist()
    chol_factor_inverse(G, [0.0] * len(G), check=False, pivot_mins=thr)
    info = chol_factor_inverse.last_info.tolist()
    live = []                                           # (position in mats, k)
    for p, (inf, r) in enumerate(zip(info, r_of)):
        k = inf - 1 if inf > 0 else r
        if inf >= 0 and 16 <= k < r - 8:
            live.append((p, k))
        else:
            log.append((index[p], f"rank {k} of {r} probed (status {inf}): left to the iteration"))
    if not live:
        return out

    def cholqr(cols, grams=None):
        """orthonormalise the columns of each (n x k) matrix: G = C^T C, L L^T = G, C L^-T; None where G is not positive definite"""
        if grams is None:
            grams = gemm_f64_batched([Gemm64(c.t(), c) for c in cols])
        X = chol_factor_inverse(grams, [0.0] * len(grams), check=False)
        bad = chol_factor_inverse.last_info.tolist()
        if any(bad):
            log.append((-1, f"Cholesky-QR pass: status words {bad}"))
        q = gemm_f64_batched([Gemm64(c, x.t(), tri=TRI64_B_UPPER) for c, x in zip(cols, X)])
        return [None if b != 0 else t for t, b in zip(q, bad)]

    def keep(flags, *lists):
        idx = [i for i, f in enumerate(flags) if f]
        return [[lst[i] for i in idx] for lst in lists]

    pos = [p for p, _ in live]
    ks = [k for _, k in live]
    Q = cholqr([Y[p][:, :k] for p, k in live], [G[p][:k, :k].contiguous() for p, k in live])
    for _ in range(2):
        pos, ks, Q = keep([q is not None for q in Q], pos, ks, Q)
        if not pos:
            return out
        Q = cholqr(Q)
    pos, ks, Q = keep([q is not None for q in Q], pos, ks, Q)
    if not pos:
        return out
    # one step of subspace iteration, Q <- orth(F Q).  Without it the basis of a matrix whose rank was hit exactly (k = rank,
    # no oversampling: a sharp drop of the pivots) is only as good as the k x k Gaussian mixing matrix is conditioned - the
    # weak directions tilt by the noise of F over its smallest singular value (measured: rank 32 of 2049, residual 2e-5);
    # multiplied by F once more the noise directions fall by their eigenvalue ratio
    Q = gemm_f64_batched([Gemm64(S[p], q) for p, q in zip(pos, Q)])
    for _ in range(3):
        Q = cholqr(Q)
        pos, ks, Q = keep([q is not None for q in Q], pos, ks, Q)
        if not pos:
            return out
    SQ = gemm_f64_batched([Gemm64(S[p], q) for p, q in zip(pos, Q)])
    B = gemm_f64_batched([Gemm64(q.t(), sq) for q, sq in zip(Q, SQ)])
    B = [(b + b.t()) * 0.5 for b in B]
    res2 = torch.stack([nS2[p] - (b * b).sum() for p, b in zip(pos, B)])
    bar = torch.stack([nS2[p] for p in pos]).mul(LOWRANK_RESIDUAL ** 2)
    ok = (res2 <= bar).tolist()
    for p, k, r2, b2, good in zip(pos, ks, res2.tolist(), bar.tolist(), ok):
        log.append((index[p], f"rank {k}: projection residual {max(r2, 0.0) ** 0.5 / (b2 ** 0.5 / LOWRANK_RESIDUAL):.2e} of ||F||" +
                    ("" if good else ": above the bar, left to the iteration")))
    pos, ks, Q, B = keep(ok, pos, ks, Q, B)
    if not pos:
        return out
    # the small problems: the block-Jacobi iteration (they have full rank by construction: not projected again)
    W, lam = eigh([b.float().contiguous() for b in B], with_values=True, _project=False)
    # complement of Q: Gaussian, projected, orthonormalised twice
    Z = []
    for p, k, q in zip(pos, ks, Q):
        n = S[p].shape[0]
        Z.append(randn((n, n - k), dev, 0x5EED8000 + n, 0).double())
    for _ in range(2):
        T = gemm_f64_batched([Gemm64(q.t(), z) for q, z in zip(Q, Z)])
        Z = gemm_f64_batched([Gemm64(q, t, alpha=-1.0, beta=1.0, E=z) for q, t, z in zip(Q, T, Z)])
        Z = cholqr(Z)
        if any(z is None for z in Z):
            pos, ks, Q, W, lam, Z = keep([z is not None for z in Z], pos, ks, Q, W, lam, Z)
            if not pos:
                return out
    for p, k, q, wk, lk, z in zip(pos, ks, Q, W, lam, Z):
        n = S[p].shape[0]
        Ucat = torch.empty(n, n, dtype=torch.float32, device=dev)
        gemm_f64_batched([Gemm64(q, wk.double(), out32=Ucat[:, n - k:])])
        Ucat[:, :n - k].copy_(z)
        w = torch.cat([torch.zeros(n - k, dtype=torch.float32, device=dev), lk])
        w_sorted, order = torch.sort(w, stable=True)
        out[index[p]] = (Ucat.index_select(1, order).contiguous(), w_sorted.contiguous(), k)
    return out


