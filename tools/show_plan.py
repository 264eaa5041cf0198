#!/usr/bin/env python
"""Print the SYRK launch plan (host-only, no GPU needed) for a model's layer shapes."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import _lib, models  # noqa: E402

NAMES = "dim Ho Wo NS R Wc nchunks RS PS SS nch ntiles cpi nslices nitems base TM vec4 cshift nsub direct rshift pre dma flops".split()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    model, chw = {"lenet5": (models.lenet5, (1, 28, 28)), "resnet18": (models.resnet18, (3, 224, 224)),
                  "resnet50": (models.resnet50, (3, 224, 224))}[args.model]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_syrk import conv_layers
    layers = conv_layers(model(), chw)
    descs = []
    for m, ishape, oshape in layers:
        if m.__class__.__name__ == "Conv2d":
            k, s, p = m.kernel_size, m.stride, m.padding
            descs.append(dict(N=args.batch, C=ishape[1], H=ishape[2], W=ishape[3], kh=k[0], kw=k[1], sh=s[0],
                              sw=s[1], ph=p[0], pw=p[1], has_bias=int(m.bias is not None)))
            descs.append(dict(N=args.batch, C=oshape[1], H=oshape[2], W=oshape[3], kh=1, kw=1, sh=1, sw=1, ph=0, pw=0,
                              has_bias=0))
        else:
            descs.append(dict(N=args.batch, C=ishape[1], H=1, W=1, kh=1, kw=1, sh=1, sw=1, ph=0, pw=0,
                              has_bias=int(m.bias is not None)))
            descs.append(dict(N=args.batch, C=oshape[1], H=1, W=1, kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=0))
    n = len(descs)
    arr = (_lib.curv_factor_desc * n)()
    for d, a in zip(descs, arr):
        for k, v in d.items():
            setattr(a, k, v)
        a.scale = 1.0
    L = _lib.lib()
    nf = 25                  # CURV_PLAN_INFO_FIELDS
    out = (ctypes.c_longlong * (nf * n))()
    rc = L.curv_kfac_plan_info(arr, n, out)
    print("rc", rc, L.curv_last_error())
    seen, tot = set(), 0
    for i in range(n):
        o = out[nf * i:nf * i + nf]
        tot += o[14]
        key = tuple(o[:14]) + tuple(o[16:])
        if key in seen:
            continue
        seen.add(key)
        d = descs[i]
        print(f"C={d['C']:5d} H={d['H']:4d} k={d['kh']} s={d['sh']}", dict(zip(NAMES, o)))
    print("items", tot, "workspace MB", L.curv_kfac_workspace_bytes(arr, n) / 1e6)


if __name__ == "__main__":
    main()
