#!/usr/bin/env python
"""Race detector for the inversion sweep: the same factors inverted many times (whole model, a 9-layer share, one large factor)
must give bit-identical results every time - a missing dependency between the sweep's streams shows up as a flipped bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    sizes = []
    for r in rows:
        sizes += [r["n"], r["m"]]
    for tag, sel in (("whole model", sizes), ("share", [s for i, s in enumerate(sizes) if i % 6 == 0]), ("one 4608", [4608]), ("three 4608", [4608] * 3)):
        Fs = []
        for i, n in enumerate(sel):
            torch.manual_seed(i)
            k = min(n + 8, 4096)
            X = torch.randn(n, k, device=dev)
            Fs.append((X @ X.t() / k).contiguous())
        add, mul = [1.0] * len(Fs), [1000.0] * len(Fs)
        ref = [o.clone() for o in ops.chol_inv_lower(Fs, add, mul)]
        bad = 0
        for it in range(reps):
            outs = ops.chol_inv_lower(Fs, add, mul, check=(it % 7 == 0))
            if it % 10 == 9 or it == reps - 1:
                torch.cuda.synchronize()
                bad += sum(int(not torch.equal(a, b)) for a, b in zip(outs, ref))
        print(f"{tag}: {len(Fs)} factors x {reps} calls: {bad} mismatching outputs")


if __name__ == "__main__":
    main()
