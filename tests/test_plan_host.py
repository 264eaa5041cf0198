"""Host-side launch planning of the SYRK kernel (curv_kfac_plan_info): every plan must respect the
kernel's LDS / staging-register budgets and cover the whole K range."""
import ctypes
import itertools

import pytest

from curvature_amd import _lib

NF = 25
NAMES = "dim Ho Wo NS R Wc nchunks RS PS SS nch ntiles cpi nslices nitems base TM vec4 cshift nsub direct rshift pre dma flops".split()
PANEL_WORDS, PRE_PANEL_WORDS, KTAB_MAX, SLOTS, THREADS = 8704, 6528, 1024, 32, 256


def plan(descs):
    n = len(descs)
    arr = (_lib.curv_factor_desc * n)()
    for d, a in zip(descs, arr):
        for k, v in d.items():
            setattr(a, k, v)
        a.scale = 1.0
    out = (ctypes.c_longlong * (NF * n))()
    rc = _lib.lib().curv_kfac_plan_info(arr, n, out)
    assert rc == 0, _lib.lib().curv_last_error()
    return [dict(zip(NAMES, out[NF * i:NF * i + NF])) for i in range(n)]


def geom(N, C, H, W, k, s, p, bias):   # (W: image width; CASES pass H twice)
    return dict(N=N, C=C, H=H, W=W, kh=k, kw=k, sh=s, sw=s, ph=p, pw=p, has_bias=bias)


CASES = [geom(N, C, H, H, k, s, p, b)
         for (N, C, H, k, s, p, b) in
         [(32, 3, 224, 7, 2, 3, 0), (32, 64, 56, 1, 1, 0, 0), (32, 64, 56, 3, 1, 1, 0), (32, 128, 56, 3, 2, 1, 0),
          (32, 256, 56, 1, 2, 0, 0), (32, 512, 7, 3, 1, 1, 0), (32, 2048, 7, 1, 1, 0, 0), (32, 2048, 1, 1, 1, 0, 1),
          (100, 1, 28, 5, 1, 2, 1), (100, 6, 14, 5, 1, 0, 1), (100, 400, 1, 1, 1, 0, 1), (1, 1, 1, 1, 1, 0, 0),
          (7, 33, 17, 3, 3, 0, 1), (2, 5, 300, 11, 4, 5, 0), (3, 700, 5, 1, 1, 0, 0), (3, 37, 28, 1, 1, 0, 0),
          (3, 37, 1, 1, 1, 0, 0), (2, 37, 9, 1, 2, 0, 0),
          # DenseNet widths (64 + 32 k channels): the LDS-DMA kernel with a ragged last tile row / column
          (8, 96, 56, 1, 1, 0, 0), (8, 224, 28, 1, 1, 0, 0), (3, 992, 7, 1, 1, 0, 0), (8, 160, 14, 1, 1, 0, 0),
          (8, 336, 28, 1, 1, 0, 0), (5, 144, 9, 1, 1, 0, 0)]]


@pytest.mark.parametrize("d", CASES)
def test_plan_respects_budgets(d):
    p = plan([d])[0]
    compact = d["kh"] == 1 and d["kw"] == 1
    Ho = (d["H"] + 2 * d["ph"] - d["kh"]) // d["sh"] + 1
    Wo = (d["W"] + 2 * d["pw"] - d["kw"]) // d["sw"] + 1
    flat = compact and d["sh"] == 1 and d["ph"] == 0
    # a strided 1x1 or a kh x kw > 1 convolution that the LDS-DMA kernel takes: flattened too - through its unfolded copy
    sub = not flat and p["dma"] == 1
    if sub:
        assert not d["has_bias"] and p["dim"] % 16 == 0 and p["dim"] >= 96
    flat = flat or sub
    assert p["dim"] == d["C"] * d["kh"] * d["kw"] + d["has_bias"]
    assert (p["Ho"], p["Wo"]) == ((1, Ho * Wo) if flat else (Ho, Wo))
    K = d["N"] * Ho * Wo
    if p["dma"] == 2:
        # assembled from shifted correlations (syrk_corr.hip): no items of its own; 13 whole-image correlations over
        # rows padded to W + 2 (one symmetric) + 12 border strips (four symmetric) + 4 corner products
        assert (d["kh"], d["kw"], d["sh"], d["ph"], d["has_bias"]) == (3, 3, 1, 1, 0) and d["N"] >= 8
        assert d["C"] % 128 == 0 or d["C"] == 64
        assert p["nitems"] == 0 and p["nsub"] == 0
        C, N, H, W = d["C"], d["N"], d["H"], d["W"]
        if C == 64:
            # packed pair tiles: four 128x128 non-symmetric tiles hold the 13 whole-image correlations (16 blocks),
            # one tile per border-strip array, two symmetric 128-row tiles the four corner products
            want = 4 * 2 * 128 * 128 * N * H * (W + 2) + 2 * 2 * 128 * 128 * N * (W + 2) + 2 * 2 * 128 * 128 * N * (H + 2)
            want += 2 * 128 * 129 * N
            assert p["flops"] == want
            assert p["flops"] < 0.45 * p["dim"] * (p["dim"] + 1) * K
            return
        sym, full = C * (C + 1), 2 * C * C
        plane = H * (W + 2)
        shifts = [dh * (W + 2) + dw for dh in (-1, -2) for dw in range(-2, 3)] + [-1, -2]
        want = sym * N * plane + sum(full * N * (plane + sft) for sft in shifts)
        want += 2 * (sym + 2 * full) * N * (W + 2) + 2 * (sym + 2 * full) * N * (H + 2) + 4 * sym * N
        assert p["flops"] == want
        assert p["flops"] < 0.4 * p["dim"] * (p["dim"] + 1) * K          # the point of it
        return
    assert p["flops"] == p["dim"] * (p["dim"] + 1) * K
    assert p["TM"] in (64, 128)
    # an unsliced 128x128-tile factor writes its tiles itself (direct epilogue): no sub-tiles for the reduce pass
    assert p["direct"] == int(p["TM"] == 128 and p["nslices"] == 1)
    assert p["nsub"] == (0 if p["direct"] else p["ntiles"] * (p["TM"] // 64) ** 2)
    if p["dma"]:
        # LDS-DMA kernel (syrk_flat.hip): flattened factor, 128-row tiles (the last one may be ragged: widths of 16 k
        # channels from 96 on), no bias row; K = the stream of the factor's 4-pixel groups (ceil(HW / 4) per sample, sample
        # after sample), in stages of four groups
        assert flat and not d["has_bias"] and p["dim"] % 16 == 0 and p["dim"] >= 96 and p["TM"] == 128
        groups = d["N"] * max(-(-(Ho * Wo) // 4), 4)
        assert p["nchunks"] == -(-groups // 4)
        P = -(-p["dim"] // 128)
        assert p["ntiles"] == P * (P + 1) // 2 and p["nitems"] == p["ntiles"] * p["nslices"]
        assert p["cpi"] * p["nslices"] >= p["nchunks"] and p["cpi"] * (p["nslices"] - 1) < p["nchunks"]
        return
    assert p["NS"] * p["SS"] + 16 <= (PRE_PANEL_WORDS if p["pre"] else PANEL_WORDS)      # LDS patch per panel
    rows_in = p["R"] if compact else (p["R"] - 1) * d["sh"] + d["kh"]
    cols_in = p["Wc"] if compact else (p["Wc"] - 1) * d["sw"] + d["kw"]
    assert p["RS"] >= cols_in and p["PS"] >= rows_in * p["RS"] and p["SS"] >= p["nch"] * p["PS"]
    prow = p["NS"] * p["nch"] * rows_in
    if p["vec4"]:
        assert p["Wc"] % 4 == 0 and (prow << p["cshift"]) * 4 <= SLOTS * THREADS
    elif p["pre"]:
        # full-width chunks of a kh x kw > 1 convolution: LDS-DMA from the pre-tiled copy (syrk_pre.hip); the image of
        # a (panel, sample) is one run of SS words moved in 16-byte lanes
        assert not compact and p["Wc"] == Wo and p["pre"] == 1 and p["SS"] % 4 == 0 and p["SS"] < p["nch"] * p["PS"] + 4
        n_sg, n_rg = -(-d["N"] // p["NS"]), -(-p["Ho"] // p["R"])
        assert ((n_sg * n_rg * p["NS"] * d["C"] + p["nch"]) * p["PS"] + 64) * 4 < 2 ** 31
    elif flat and p["nch"] & (p["nch"] - 1) == 0:
        assert (prow << p["cshift"]) <= SLOTS * THREADS and (1 << p["cshift"]) >= cols_in
    else:
        # general staging: 2^rshift row lanes walk the folded (sample, row) index, the others split channels
        row_lanes = THREADS >> p["cshift"]
        assert (1 << p["cshift"]) >= cols_in and (1 << p["rshift"]) <= row_lanes
        passes = -(-(p["NS"] * rows_in) // (1 << p["rshift"]))
        groups = -(-p["nch"] // (row_lanes >> p["rshift"]))
        assert passes * groups <= SLOTS and p["SS"] == groups * (row_lanes >> p["rshift"]) * p["PS"]
    assert p["NS"] * p["R"] * p["Wc"] <= 4096            # k values per chunk
    # chunk grid covers all of K = N * Ho * Wo
    n_rg, n_cg, n_sg = -(-p["Ho"] // p["R"]), -(-p["Wo"] // p["Wc"]), -(-d["N"] // p["NS"])
    assert p["nchunks"] == n_rg * n_cg * n_sg
    assert p["cpi"] * p["nslices"] >= p["nchunks"] and p["cpi"] * (p["nslices"] - 1) < p["nchunks"]
    P = -(-p["dim"] // p["TM"])
    assert p["ntiles"] == P * (P + 1) // 2 and p["nitems"] == p["ntiles"] * p["nslices"]


def test_densenet_widths_take_the_lds_dma_kernel():
    for C in (96, 144, 160, 224, 336, 480, 992):
        assert plan([geom(8, C, 14, 14, 1, 1, 0, 0)])[0]["dma"] == 1
    for C in (64, 37, 100):          # narrower than 96 / no multiple of 16: the register-staged kernel
        assert plan([geom(8, C, 14, 14, 1, 1, 0, 0)])[0]["dma"] == 0


def test_item_bases_tile_the_work_list():
    """Factors are laid out in the work list by descending work per item; together their [base, base + nitems) ranges
    cover the patch kernel's list exactly once (two lists: register-staged factors and the pre-tiled variant's).  In
    the LDS-DMA kernel's list the caller's factors share the list with
    the virtual factors of assembled 3x3 factors (not reported): ranges are disjoint and ascending, and without such
    a factor in the set they tile it exactly as well."""
    everything = plan(CASES)
    assert any(p["dma"] == 1 for p in everything) and any(p["dma"] == 0 for p in everything)
    assert any(p["dma"] == 2 for p in everything) and any(p["pre"] for p in everything)
    for kernel, pre, subset in ((0, 0, everything), (0, 1, everything), (1, 0, everything),
                                (1, 0, [p for p in plan([c for c in CASES if not (c["kh"] == 3 and c["sh"] == 1 and c["C"] % 128 == 0)])])):
        base = 0
        exact = kernel == 0 or not any(p["dma"] == 2 for p in subset)
        for p in sorted((p for p in subset if p["dma"] == kernel and p["pre"] == pre), key=lambda p: p["base"]):
            assert p["base"] == base if exact else p["base"] >= base
            base = p["base"] + p["nitems"]


def test_invalid_geometry_is_rejected():
    arr = (_lib.curv_factor_desc * 1)()
    for k, v in geom(2, 3, 4, 4, 7, 1, 0, 0).items():        # kernel larger than the unpadded input
        setattr(arr[0], k, v)
    out = (ctypes.c_longlong * NF)()
    assert _lib.lib().curv_kfac_plan_info(arr, 1, out) == 2   # CURV_ERR_INVALID


def test_launch_form_of_the_unsharded_model_is_decided_by_the_library():
    """curv_kfac_path_for: the form a launch of exactly these factors takes on its own, with EVERY gate of the small
    form (csrc/syrk_small.hip: small_plan), not only the flop bound a caller could restate: a deep, cheap model
    (more than 96 factors) runs grouped when unsharded, so its layer-sharded ranks must be told GROUPED although each
    share is tiny.  Host only."""
    from curvature_amd import ops
    tiny = (4, 16, 1, 1, (1, 1), (1, 1), (0, 0), True)              # Linear(15, .) with bias at N = 4
    assert ops.kfac_path_for([tiny] * 10) == _lib.PATH_SMALL
    assert ops.kfac_path_for([tiny] * 96) == _lib.PATH_SMALL
    assert ops.kfac_path_for([tiny] * 98) == _lib.PATH_GROUPED       # more than 4 argument blocks of 24 factors
    lenet = [(100, 1, 28, 28, (5, 5), (1, 1), (2, 2), True), (100, 6, 28, 28, (1, 1), (1, 1), (0, 0), False),
             (100, 6, 14, 14, (5, 5), (1, 1), (0, 0), True), (100, 16, 10, 10, (1, 1), (1, 1), (0, 0), False),
             (100, 400, 1, 1, (1, 1), (1, 1), (0, 0), True), (100, 120, 1, 1, (1, 1), (1, 1), (0, 0), False)]
    assert ops.kfac_path_for(lenet) == _lib.PATH_SMALL
    resnet_layer = [(32, 256, 56, 56, (1, 1), (1, 1), (0, 0), False)]   # 2 * 256^2 * 100 352: far over the flop bound
    assert ops.kfac_path_for(lenet + resnet_layer) == _lib.PATH_GROUPED
    # a hint in the descriptors does not leak into the decision
    arr = (_lib.curv_factor_desc * 98)()
    for d in arr:
        d.N, d.C, d.H, d.W, d.kh, d.kw, d.sh, d.sw, d.has_bias, d.path_hint = 4, 16, 1, 1, 1, 1, 1, 1, 1, _lib.PATH_SMALL
    assert _lib.lib().curv_kfac_path_for(arr, 98) == _lib.PATH_GROUPED
    assert ops.kfac_path_for([]) == _lib.PATH_GROUPED
