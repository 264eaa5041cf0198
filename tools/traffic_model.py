#!/usr/bin/env python
"""Operand bytes of the LDS-DMA factor-build kernel from the launch plan (host only, no GPU): what syrk_flat_kernel streams
from L2 (every work item its own panels) against what it would fetch from memory if every operand row were fetched once
per factor, per model:   python tools/traffic_model.py [resnet50|resnet18|densenet121] [N]
The gap between the measured FETCH_SIZE (x 2 on gfx950, tools/fetch_calib.sh) and these two figures says how much of the
re-streaming the L2s absorb."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvature_amd import _lib, models  # noqa: E402

NF = 25
NAMES = "dim Ho Wo NS R Wc nchunks RS PS SS nch ntiles cpi nslices nitems base TM vec4 cshift nsub direct rshift pre dma flops".split()


def geometries(model, N, chw=(3, 224, 224)):
    geoms = []
    hooks = []

    def hook(layer, inp, out):
        x = inp[0]
        if isinstance(layer, torch.nn.Conv2d):
            geoms.append(dict(N=N, C=x.shape[1], H=x.shape[2], W=x.shape[3], kh=layer.kernel_size[0], kw=layer.kernel_size[1],
                              sh=layer.stride[0], sw=layer.stride[1], ph=layer.padding[0], pw=layer.padding[1],
                              has_bias=int(layer.bias is not None), what="A %s" % (tuple(x.shape[1:]),)))
            geoms.append(dict(N=N, C=out.shape[1], H=out.shape[2], W=out.shape[3], kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=0,
                              what="G %s" % (tuple(out.shape[1:]),)))
        else:
            geoms.append(dict(N=N, C=x.shape[-1], H=1, W=1, kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=int(layer.bias is not None),
                              what="A linear %d" % x.shape[-1]))
            geoms.append(dict(N=N, C=out.shape[-1], H=1, W=1, kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=0, what="G linear %d" % out.shape[-1]))

    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        model.eval()(torch.zeros(1, *chw))
    for h in hooks:
        h.remove()
    return geoms


def plan(geoms):
    n = len(geoms)
    arr = (_lib.curv_factor_desc * n)()
    for d, a in zip(geoms, arr):
        for k, v in d.items():
            if k != "what":
                setattr(a, k, v)
        a.scale = 1.0
    out = (ctypes.c_longlong * (NF * n))()
    rc = _lib.lib().curv_kfac_plan_info(arr, n, out)
    assert rc == 0, _lib.lib().curv_last_error()
    return [dict(zip(NAMES, out[NF * i:NF * i + NF])) for i in range(n)]


def operand_bytes(geoms, plans, N):
    once = streamed = other = 0.0
    rows = []
    for g, p in zip(geoms, plans):
        Ho = (g["H"] + 2 * g["ph"] - g["kh"]) // g["sh"] + 1
        Wo = (g["W"] + 2 * g["pw"] - g["kw"]) // g["sw"] + 1
        if p["dma"] == 1:
            K = N * Ho * Wo
            P = -(-p["dim"] // 128)
            a = p["dim"] * K * 4.0
            s = a * P                       # every 128-row panel is an operand of P upper-triangular tiles (once as a diagonal tile)
            once += a
            streamed += s
            rows.append((s, g["what"], p["dim"], K, P, p["nslices"], a, s))
        elif p["dma"] == 2:
            # 29 shifted correlations of the zero-padded image copy: 13 symmetric, 16 full (syrk_corr.hip)
            C = g["C"]
            K = N * (g["H"] + 2) * (g["W"] + 2)
            a = C * K * 4.0
            if C >= 128:
                P = C // 128
                s = 13 * a * P + 16 * 2 * a * P
            else:
                s = 10 * 2 * 128 * K * 4.0     # ten packed pair tiles, two 128-row operands each
            once += a
            streamed += s
            rows.append((s, g["what"] + " (correlations)", p["dim"], K, -(-C // 128), 0, a, s))
        else:
            other += p["dim"] * N * Ho * Wo * 4.0
    return once, streamed, other, rows


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    geoms = geometries(getattr(models, name)(), N)
    plans = plan(geoms)
    once, streamed, other, rows = operand_bytes(geoms, plans, N)
    rows.sort(reverse=True)
    print("%s, N = %d: LDS-DMA kernel operands: %.2f GB if every row were fetched once, %.2f GB streamed by the work items"
          % (name, N, once / 1e9, streamed / 1e9))
    print("(factors of the register-staged kernels: %.2f GB of patch rows, not modelled)" % (other / 1e9))
    for s, what, dim, K, P, ns, a, st in rows[:25]:
        print("  %-44s dim %5d K %7d P %2d slices %3d: once %7.1f MB, streamed %7.1f MB" % (what, dim, K, P, ns, a / 1e6, st / 1e6))


if __name__ == "__main__":
    main()
