#!/usr/bin/env python
"""Stage times of INF.invert at ResNet-50 size (config 5): the stages of INF.pre_sampler_many run one by one with a
synchronisation between them (the product path enqueues them back to back).  Diagnostics only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops  # noqa: E402
from curvature_amd.curvatures import Diagonal, KFAC, EFB, INF  # noqa: E402


def main():
    N = 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet50().to(dev).train()
    diag, kfac = Diagonal(model), KFAC(model)
    x = torch.randn(N, 3, 224, 224, device=dev)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    diag.update(N)
    kfac.update(N)
    efb = EFB(model, kfac.state)
    efb.update(N)
    inf = INF(model, diag.state, kfac.state, efb.state, eigvecs=efb.eigvecs)
    inf.update(rank=100)
    inf.invert(1.0, 1000.0)
    torch.cuda.synchronize()
    regs = [(ua, ug, torch.ones_like(lam), r) for (ua, ug, r, _), (_, _, lam, _) in
            ((inf.inv_state[l], inf.state[l]) for l in inf.state)]

    def stage(name, fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        print(f"{name:34s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms", flush=True)
        return out

    def vtv():
        first, parts = [], []
        for ua, ug, sigma, r in regs:
            (n, a), (m, b) = ua.shape, ug.shape
            PA, PG = ops.colpairs_sym(ua), ops.colpairs_sym(ug)
            r2 = ops.square_f64(r).view(n, m)
            first.append(ops.Gemm64(PA.t(), r2))
            parts.append((PG, sigma, a, b))
        Ms = ops.gemm_f64_batched(first)
        V4s = ops.gemm_f64_batched([ops.Gemm64(M, PG) for M, (PG, _, _, _) in zip(Ms, parts)])
        return [ops.inf_vtv_assemble_sym(V4.contiguous(), sigma, a, b) for V4, (_, sigma, a, b) in zip(V4s, parts)]

    vtvs = stage("V_s^T V_s (closed form)", vtv)
    invA = stage("sweep A: chol(vtv)^-1 (1 per layer)", lambda: ops.chol_factor_inverse(vtvs, [0.0] * len(vtvs)))
    Ts = stage("sweep B with right-hand side: T = A^-1 - chol(vtv + I)^-1 A^-1",
               lambda: ops.chol_factor_inverse(vtvs, [1.0] * len(vtvs), rhs=invA, rhs_minus=True))
    # the round-3 / early round-4 form of the same two stages, for comparison: both inverses explicitly, then a product
    mats, adds = [], []
    for v in vtvs:
        mats += [v, v]
        adds += [0.0, 1.0]
    inv = stage("  (explicit form: factor-and-invert sweep, 2 per layer)", lambda: ops.chol_factor_inverse(mats, adds))
    T2 = [torch.empty_like(t) for t in Ts]
    stage("  (explicit form: T = A^-1 - B^-1 A^-1, tri x tri product)", lambda: ops.gemm_f64_batched(
        [ops.Gemm64(inv[2 * i + 1], inv[2 * i], T, alpha=-1.0, beta=1.0, E=inv[2 * i], tri=ops.TRI64_A_LOWER | ops.TRI64_B_LOWER)
         for i, T in enumerate(T2)]))
    print("  explicit vs right-hand side form of T: max difference",
          max(float((a - b).abs().max()) for a, b in zip(Ts, T2)))
    del inv, T2
    outs = [torch.empty(t.shape, dtype=torch.float32, device=t.device) for t in Ts]
    stage("P_c = diag(s) A^-T T diag(s) (upper x lower, fp32 out)", lambda: ops.gemm_f64_batched(
        [ops.Gemm64(invA[i].t(), T, tri=ops.TRI64_A_UPPER | ops.TRI64_B_LOWER, out32=outs[i], row_scale=regs[i][2].contiguous(),
                    col_scale=regs[i][2].contiguous()) for i, T in enumerate(Ts)]))
    n3 = sum(float(v.shape[0]) ** 3 for v in vtvs)
    print(f"sum (ab)^3 = {n3:.3e}: each sweep {2 / 3 * n3 / 1e12:.2f} TFLOP, the tri x tri product it replaces {n3 / 3 / 1e12:.2f}, upper x lower {2 * n3 / 3 / 1e12:.2f}")
    stage("inf.invert(1, 1000) as a whole", lambda: inf.invert(1.0, 1000.0))


if __name__ == "__main__":
    main()
