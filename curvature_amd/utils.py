"""`get_eigenvectors`, `get_eigenvalues` and `kron` of the reference (curvature/utils.py:21-60, 288-310)."""
from typing import Dict, List, Sequence, Union

import torch
from torch import Tensor
from torch.nn import Module

from . import ops


def get_eigenvectors(factors: Dict[Module, Tensor]) -> Dict[Module, Tensor]:
    """Eigenvectors (columns, eigenvalues ascending) of both Kronecker factors of every layer in `factors`.

    The reference decomposes F + F^T (utils.py:55-58); F is exactly symmetric here, so F itself has the
    same eigenvectors.  Computed by the library's batched block-Jacobi eigensolver (curv_syevd), all
    matrices of the dict advancing together.  Nothing is memoised: the factors are accumulated in place
    through raw pointers, so neither their address nor torch's version counter identifies their contents.
    The reference decomposes the same factors in the EFB and again in the INF constructor
    (curvatures.py:403, 473); pass ``INF(..., eigvecs=efb.eigvecs)`` to reuse the first result.

    With a layer-sharded KFAC the dict holds this rank's layers only, so the decomposition is sharded too."""
    layers = list(factors.keys())
    mats = []
    for layer in layers:
        xxt, ggt = factors[layer]
        mats.extend([xxt, ggt])
    vecs = ops.eigh(mats)
    return {layer: (vecs[2 * i], vecs[2 * i + 1]) for i, layer in enumerate(layers)}


def get_eigenvalues(factors: Union[Sequence, Dict], verbose: bool = False) -> Tensor:
    """Eigenvalues of KFAC, EFB or diagonal factors, concatenated over the layers (utils.py:21-42).

    An entry of length 2 is a pair of Kronecker factors: its eigenvalues are the outer product of the two
    factors' ascending eigenvalues, flattened row-major (i * m + j, A side first).  Any other entry (an EFB
    Lambda or a diagonal-Fisher matrix) contributes its entries as they are.  `factors` may be a list (as in
    the reference) or an estimator ``state`` dict.  All Kronecker factors are decomposed in ONE batched
    eigensolver call (`w` output of curv_syevd); the outer products are one batched GEMM launch (K = 1)."""
    items: List = list(factors.values()) if isinstance(factors, dict) else list(factors)
    mats = [f for item in items if len(item) == 2 for f in item]
    vals = ops.eigh(mats, with_values=True)[1] if mats else []
    out, jobs, pos = [], [], 0
    for item in items:
        if len(item) == 2:
            wa, wg = vals[pos], vals[pos + 1]
            pos += 2
            outer = torch.empty(wa.numel(), wg.numel(), dtype=torch.float32, device=wa.device)
            jobs.append(ops.Gemm(wa.view(-1, 1), wg.view(1, -1), outer))
            out.append(outer.view(-1))
        else:
            out.append(item.contiguous().view(-1))
    ops.gemm_batched(jobs)
    return ops.concat(out) if out else Tensor()


def kron(a: Tensor, b: Tensor) -> Tensor:
    """Kronecker product with the reference's index convention (utils.py:288-310): out[i*p + k, j*q + l] =
    a[i, j] b[k, l] for b of shape (p, q).  Only for callers that want the explicit matrix: the estimators
    never materialise it.  One elementwise kernel (curv_kron); GPU tensors only, like every
    other entry point (no CPU fallback)."""
    return ops.kron(a, b)
