"""bench.py's arithmetic, on the CPU: the rooflines a reader recomputes from the line, the eigensolver's byte count with
projected factors, the thread count of the CPU leg, and the staleness guard of `roofline.traffic`."""
import importlib.util
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_arith", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_roofline_of_invert_and_sample(bench):
    dims = [(4608, 512), (256, 64)]
    out = bench.roofline_phases(dims, [0, 1], invert_ms=2.0, sample_ms=0.5)
    n3 = 4608.0 ** 3 + 512.0 ** 3 + 256.0 ** 3 + 64.0 ** 3
    inv = out["invert"]
    assert inv["gflop"] == pytest.approx(2.0 / 3.0 * n3 / 1e9)
    # half the flops on each MFMA type: time at both roofs over the measured time
    roof_s = n3 / 3.0 / 78.6e12 + n3 / 3.0 / 157.3e12
    assert inv["frac"] == pytest.approx(roof_s / 2.0e-3) and inv["roof_ms"] == pytest.approx(roof_s * 1e3)
    assert inv["frac_all_fp64_roof"] == pytest.approx(2.0 / 3.0 * n3 / 2.0e-3 / 78.6e12)
    smp = out["sample_and_replace"]
    flops = 4608.0 ** 2 * 512 + 4608.0 * 512 ** 2 + 256.0 ** 2 * 64 + 256.0 * 64 ** 2
    assert smp["gflop"] == pytest.approx(flops / 1e9) and smp["frac"] == pytest.approx(flops / 0.5e-3 / 157.3e12)
    # a layer-sharded rank prices its own layers only
    assert bench.roofline_phases(dims, [1], 1.0, 1.0)["invert"]["gflop"] == pytest.approx(2.0 / 3.0 * (256.0 ** 3 + 64.0 ** 3) / 1e9)


def test_eigensolver_bytes_count_projected_factors_at_their_rank(bench):
    state = {"a": (torch.empty(4608, 1), torch.empty(512, 1)), "b": (torch.empty(64, 1), torch.empty(64, 1))}
    state = {k: (torch.empty(A.shape[0], A.shape[0], device="meta"), torch.empty(G.shape[0], G.shape[0], device="meta"))
             for k, (A, G) in state.items()}

    def per_sweep(widths):
        return sum(max(n / 32.0 - 1.0, 1.0) * 14.0 * float(n) ** 2 for n in widths)

    plain = bench._efb_eig_fracs(state, 1.0, 100.0, 10, "")
    assert plain["eigensolver_hbm_gbytes_per_sweep"] == pytest.approx(per_sweep([4608, 512, 64, 64]) / 1e9)
    assert plain["eigensolver_frac"] == pytest.approx(per_sweep([4608, 512, 64, 64]) * 10 / 0.1 / 8.0e12)
    proj = bench._efb_eig_fracs(state, 1.0, 100.0, 10, "", ranks={0: 1568})
    assert proj["eigensolver_hbm_gbytes_per_sweep"] == pytest.approx(per_sweep([1568, 512, 64, 64]) / 1e9)
    assert proj["eigensolver_projected"] == {"0": 1568}
    flops = 2.0 * (512 ** 2 * 4608 + 512 * 4608 ** 2) + 2.0 * (64 ** 3 + 64 ** 3)
    assert plain["efb_update_frac"] == pytest.approx(flops / 1e-3 / 157.3e12)


def test_cpu_leg_takes_the_smallest_thread_count_near_the_fastest_probe(bench):
    assert bench.pick_threads({256: 3.3, 64: 0.017, 32: 0.019, 16: 0.031}) == 32
    assert bench.pick_threads({256: 3.3, 64: 0.016, 32: 0.017, 16: 0.021}) == 32
    assert bench.pick_threads({8: 0.5, 4: 0.9}) == 8
    assert bench.pick_threads({64: 1.0, 32: 1.0, 16: 1.0}) == 16


def test_traffic_file_is_tied_to_the_factor_build_sources(bench):
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_syrk_pmc.json")))
    assert len(bench.syrk_source_sha16()) == 16
    # the committed counter file belongs to the committed factor-build sources: otherwise `roofline.traffic` is withheld
    assert rec.get("source_sha16") == bench.syrk_source_sha16(), "re-run tools/collect_profiles.sh: the factor-build sources changed"
    assert 3.9e9 < rec["hbm_bytes_per_launch"] < 3.0e10


def test_roofline_of_the_inf_phases(bench):
    """vtv in closed form + 2 q^3 for the fp64 chain (q = a b), never the reference's 2 n m q^2."""
    shapes = [(147, 64, 30, 20), (4608, 512, 50, 40)]
    out = bench.inf_rooflines(shapes, update_ms=10.0, invert_ms=50.0, sample_ms=2.0)
    vtv = sum(a * (a + 1.0) * n * m + a * (a + 1.0) * m * b * (b + 1.0) / 2 for n, m, a, b in shapes)
    chain = sum(2.0 * (a * b) ** 3 for n, m, a, b in shapes)
    inv = out["inf_invert"]
    assert inv["gflop_vtv"] == pytest.approx(vtv / 1e9) and inv["gflop_chain"] == pytest.approx(chain / 1e9)
    assert inv["frac"] == pytest.approx((vtv + chain) / 50e-3 / 78.6e12)
    assert inv["gflop"] * 1e9 < sum(2.0 * n * m * (a * b) ** 2 for n, m, a, b in shapes)      # far below the explicit form
    assert out["inf_update"]["gbyte"] == pytest.approx(sum(12.0 * n * m for n, m, a, b in shapes) / 1e9)
    assert out["inf_update"]["frac"] == pytest.approx(out["inf_update"]["gbyte"] * 1e9 / 10e-3 / 8.0e12)
    assert out["inf_sample"]["frac"] == pytest.approx(out["inf_sample"]["gbyte"] * 1e9 / 2e-3 / 8.0e12)
