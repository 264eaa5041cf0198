"""Parity of curv_chol_inv_lower (KFAC.invert, curvature/curvatures.py:354-385) with the reference.

Two bars: (1) against the reference's own fp32 outputs (golden g3) within the north_star tolerance 1e-4
relative Frobenius where the reference's fp32 LAPACK noise permits; (2) against the oracle run in fp64 on
the SAME fp32-formed damped matrix, where only output rounding to fp32 remains (1e-6)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_fro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def damp32_then_64(F, add, mul):
    """The reference's fp32 damping (curvatures.py:368-375), result promoted to fp64."""
    reg = torch.tensor(mul ** 0.5, dtype=torch.float32) * F + torch.diag(F.new_full((F.shape[0],), add ** 0.5))
    return ((reg + reg.t()) / 2.0).double()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_lenet_golden(gpu, tag):
    import oracle.curvature_oracle as o
    from curvature_amd import ops
    g1, g3 = load("g1_kfac_lenet.npz"), load("g3_kfac_invert.npz")
    hyper = {"a": (0.5, 1), "b": (1.0, 1000.0), "c": (g3["c_add"].tolist(), g3["c_mul"].tolist())}[tag]
    factors, adds, muls, refs = [], [], [], []
    for li in range(5):
        n, s = o.layer_hyper(hyper[0], hyper[1], li, 5)
        for side in ("A", "G"):
            factors.append(g1[f"{side}_after3_l{li}"].to(gpu))
            adds.append(n)
            muls.append(s)
            refs.append(g3[f"{tag}_L{side}_l{li}"])
    outs = ops.chol_inv_lower(factors, adds, muls)
    torch.cuda.synchronize()
    for F, a, s, L, ref in zip(factors, adds, muls, outs, refs):
        assert torch.equal(L, torch.tril(L))                       # zeros above the diagonal
        exact = o.chol_of_inverse(damp32_then_64(F.cpu(), a, s))
        assert rel_fro(L, exact) < 1e-6, rel_fro(L, exact)
        # the reference's fp32 result: its own distance to the exact answer bounds what parity can mean
        ref_noise = rel_fro(ref, exact)
        assert rel_fro(L, ref) < max(1e-4, 2 * ref_noise), (rel_fro(L, ref), ref_noise)
        if tag == "a":
            assert rel_fro(L, ref) < 1e-4, rel_fro(L, ref)          # north_star bar at the README call (0.5, 1)


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 130, 200, 577])
def test_random_spd_sizes(gpu, n):
    import oracle.curvature_oracle as o
    from curvature_amd import ops
    torch.manual_seed(n)
    X = torch.randn(n, 2 * n + 3)
    F = (X @ X.t() / X.shape[1]).float()
    F = (F + F.t()) / 2
    L = ops.chol_inv_lower([F.to(gpu)], [0.3], [7.0])[0]
    exact = o.chol_of_inverse(damp32_then_64(F, 0.3, 7.0))
    assert rel_fro(L, exact) < 1e-6, rel_fro(L, exact)
    # L L^T (sqrt(s) F + sqrt(n) I) = I
    M = damp32_then_64(F, 0.3, 7.0)
    I = L.double().cpu() @ L.double().cpu().t() @ M
    assert torch.linalg.norm(I - torch.eye(n, dtype=torch.float64)) / n ** 0.5 < 1e-4


def test_batched_mixed_sizes(gpu):
    """Many factors of different sizes advance together through the blocked sweep."""
    import oracle.curvature_oracle as o
    from curvature_amd import ops
    torch.manual_seed(0)
    sizes = [5, 300, 64, 129, 1, 450, 70]
    Fs = []
    for n in sizes:
        X = torch.randn(n, n + 10)
        F = (X @ X.t() / X.shape[1]).float()
        Fs.append(((F + F.t()) / 2))
    adds = [0.1 * (i + 1) for i in range(len(sizes))]
    muls = [1.0 + i for i in range(len(sizes))]
    outs = ops.chol_inv_lower([F.to(gpu) for F in Fs], adds, muls)
    for F, a, s, L in zip(Fs, adds, muls, outs):
        assert rel_fro(L, o.chol_of_inverse(damp32_then_64(F, a, s))) < 1e-6


def test_not_positive_definite_raises(gpu):
    from curvature_amd import ops
    F = -torch.eye(70, device=gpu)
    with pytest.raises(RuntimeError):
        ops.chol_inv_lower([F], [0.0], [1.0])
    # rank-deficient factor without damping is singular as well (pivot exactly 0)
    v = torch.ones(40, 1, device=gpu)
    with pytest.raises(RuntimeError):
        ops.chol_inv_lower([(v @ v.t()).contiguous()], [0.0], [1.0])


@pytest.mark.parametrize("count", [3, 80])
def test_early_verdict_equals_the_status_words(gpu, count, monkeypatch):
    """`curv_chol_inv_lower_status`: the status words copied to pinned host memory before the finalize passes are the
    words the device holds when the call is complete, for a chain-bound call (3 factors: one group) and a whole-model call
    (80: two groups on two streams) with failing factors among them; the inverse factors are bit-identical to the plain
    call's, and the failure is reported for the same factors with the same pivots."""
    from curvature_amd import ops
    torch.manual_seed(count)
    sizes = ([70, 300, 1100] * 27)[:count]
    Fs = []
    for k, n in enumerate(sizes):
        X = torch.randn(n, n + 5, device=gpu)
        Fs.append((X @ X.t() / n).contiguous())
    adds, muls = [0.5] * count, [1.0] * count
    good = [t.clone() for t in ops.chol_inv_lower(Fs, adds, muls)]
    monkeypatch.setenv("CURV_EARLY_STATUS", "0")
    plain = ops.chol_inv_lower(Fs, adds, muls)
    assert all(torch.equal(a, b) for a, b in zip(good, plain))
    monkeypatch.delenv("CURV_EARLY_STATUS")
    bad = [1, count - 1]
    for b in bad:
        Fs[b] = Fs[b].clone()
        Fs[b][5, 5] = -3.0                         # pivot 6 fails (no damping can save it at add = 0.5)
    messages = []
    for switch in ("1", "0"):
        monkeypatch.setenv("CURV_EARLY_STATUS", switch)
        with pytest.raises(RuntimeError) as err:
            ops.chol_inv_lower(Fs, adds, muls)
        messages.append(str(err.value))
    assert messages[0] == messages[1] and str(bad) in messages[0].replace(" ", "").replace(",", ", ")
    monkeypatch.delenv("CURV_EARLY_STATUS")
    outs = ops.chol_inv_lower(Fs, adds, muls, check=False)
    words = outs.info.cpu()
    assert sorted(torch.nonzero(words).flatten().tolist()) == bad
    for k in range(count):
        if k not in bad:
            assert torch.equal(outs[k], good[k])


def test_many_large_factors_no_race(gpu):
    """More block-column workgroups than the chip holds at once: late workgroups must still see the
    un-factorised diagonal block (regression test for an in-place write-back race)."""
    import oracle.curvature_oracle as o
    from curvature_amd import ops
    torch.manual_seed(5)
    sizes = [1500] * 6 + [700] * 10 + [64] * 40
    Fs = []
    for n in sizes:
        X = torch.randn(n, n + 8, device=gpu)
        F = X @ X.t() / X.shape[1]
        Fs.append(((F + F.t()) / 2).contiguous())
    outs = ops.chol_inv_lower(Fs, [1.0] * len(Fs), [1000.0] * len(Fs))
    for idx in (0, 5, 6, 15, 16, 55):
        exact = o.chol_of_inverse(damp32_then_64(Fs[idx].cpu(), 1.0, 1000.0))
        assert rel_fro(outs[idx], exact) < 1e-6


def test_same_result_whatever_the_stream_layout(gpu, tmp_path):
    """The sweep's stream layout (CU-masked side streams, or plain low-priority ones when CURV_FREE_CUS=0 / the
    runtime has no CU masks) must not change a single bit: every tile is written by exactly one workgroup and
    the accumulation order inside a tile is fixed.  The setting is read once per process, hence a subprocess."""
    import subprocess
    import sys
    from curvature_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sizes = [1100, 300, 64, 2304]

    def factors():
        out = []
        for i, n in enumerate(sizes):
            g = torch.Generator().manual_seed(100 + i)
            X = torch.randn(n, n + 8, generator=g)
            out.append((X @ X.t() / (n + 8)).contiguous())
        return out

    here = [L.cpu() for L in ops.chol_inv_lower([F.to(gpu) for F in factors()], [1.0] * 4, [1000.0] * 4)]
    # the same factors inside a call with more than 64 of them: two factor groups on four streams
    filler = [torch.eye(8, device=gpu) * (i + 1.0) for i in range(70)]
    here += [L.cpu() for L in ops.chol_inv_lower([F.to(gpu) for F in factors()] + filler, [1.0] * 74, [1000.0] * 74)[:4]]
    script = tmp_path / "inv.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from curvature_amd import ops\n"
        f"sizes = {sizes!r}\n"
        "Fs = []\n"
        "for i, n in enumerate(sizes):\n"
        "    g = torch.Generator().manual_seed(100 + i)\n"
        "    X = torch.randn(n, n + 8, generator=g)\n"
        "    Fs.append((X @ X.t() / (n + 8)).contiguous().cuda())\n"
        "Ls = list(ops.chol_inv_lower(Fs, [1.0] * 4, [1000.0] * 4))\n"
        "filler = [torch.eye(8, device='cuda') * (i + 1.0) for i in range(70)]\n"
        "Ls += list(ops.chol_inv_lower(Fs + filler, [1.0] * 74, [1000.0] * 74))[:4]\n"
        f"torch.save([L.cpu() for L in Ls], {str(tmp_path / 'out.pt')!r})\n")
    # ... nor may any of the sweep's other orchestration switches: plain event records instead of events riding on the
    # launches, the caller's stream joining at once, the set's streams created in another order (the large chain's queue on
    # the caller's pipe), the small group on one stream
    for extra in ({"CURV_FREE_CUS": "0"}, {"CURV_EXT_EVENTS": "0", "CURV_LATE_JOIN": "0"}, {"CURV_STREAM_ORDER": "xm01a"},
                  {"CURV_SMALL_ONE_STREAM": "1", "CURV_FORK_AFTER_NEAR": "1"}):
        subprocess.run([sys.executable, str(script)], check=True, env=dict(os.environ, **extra), timeout=300)
        there = torch.load(tmp_path / "out.pt")
        for a, b in zip(here, there):
            assert torch.equal(a, b), extra


def test_chain_bound_and_throughput_forms_agree(gpu):
    """A call with few factors sweeps a panel's block square in one launch whose workgroups hand tiles to each other
    (chol_square_kernel, quarter-form panel products); the same factors inside a call with many factors take the
    per-step launches.  Both forms must reproduce the fp64 oracle, for block counts that are not multiples of the
    panel width (1, 2, 3 and 4 block rows in the last square) and for several factors advancing together."""
    import oracle.curvature_oracle as o
    from curvature_amd import ops
    torch.manual_seed(11)
    sizes = [64, 100, 190, 256, 321, 700, 1100]
    Fs = []
    for n in sizes:
        X = torch.randn(n, n + 12, device=gpu)
        F = X @ X.t() / X.shape[1]
        Fs.append(((F + F.t()) / 2).contiguous())
    adds, muls = [0.3] * len(Fs), [50.0] * len(Fs)
    few = ops.chol_inv_lower(Fs, adds, muls)
    filler = [torch.eye(8, device=gpu) * (i + 1.0) for i in range(70)]          # 77 factors: the throughput form
    many = ops.chol_inv_lower(Fs + filler, adds + [0.0] * 70, muls + [1.0] * 70)
    for F, L_few, L_many in zip(Fs, few, many):
        exact = o.chol_of_inverse(damp32_then_64(F.cpu(), 0.3, 50.0))
        assert rel_fro(L_few, exact) < 1e-6
        assert rel_fro(L_many, exact) < 1e-6
        assert rel_fro(L_few, L_many.cpu()) < 1e-6


def test_chain_bound_form_reports_failed_pivots(gpu):
    """A non-positive pivot in any block row of a square ends in the status word, not in a hang: the workgroups of the
    square kernel keep posting their flags when a factorisation fails."""
    from curvature_amd import ops
    torch.manual_seed(12)
    for n, bad_at in ((200, 130), (500, 20), (500, 470)):
        X = torch.randn(n, n + 8, device=gpu)
        F = (X @ X.t() / X.shape[1]).contiguous()
        F[bad_at, bad_at] = -5.0
        with pytest.raises(RuntimeError):
            ops.chol_inv_lower([F], [0.0], [1.0])
    # and the next call on the same workspace is clean again
    X = torch.randn(300, 320, device=gpu)
    F = (X @ X.t() / 320).contiguous()
    out = ops.chol_inv_lower([F], [1.0], [1.0])[0]
    assert torch.isfinite(out).all()


def test_chain_bound_form_is_reproducible_beside_other_work(gpu):
    """The square kernel's workgroups wait for each other through flags; whatever order the hardware runs them in -
    alone, or squeezed between the GEMMs of another stream - every tile is produced by one workgroup from the same
    operands in the same order: repeated calls must agree bit for bit, and no wait may run into its bound."""
    from curvature_amd import ops
    for sizes in ([2304, 700, 64], [1024] * 6 + [333] * 10):
        Fs = []
        for i, n in enumerate(sizes):
            torch.manual_seed(40 + i)
            X = torch.randn(n, n + 8, device=gpu)
            Fs.append((X @ X.t() / (n + 8)).contiguous())
        adds, muls = [0.5] * len(Fs), [30.0] * len(Fs)
        ref = [t.clone() for t in ops.chol_inv_lower(Fs, adds, muls)]
        side = torch.cuda.Stream()
        A = torch.randn(2048, 2048, device=gpu)
        for it in range(12):
            if it % 2:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        A @ A
            outs = ops.chol_inv_lower(Fs, adds, muls)
            for a, b in zip(outs, ref):
                assert torch.equal(a, b)
        torch.cuda.synchronize()


def _stream_probe(order: str):
    """Fresh process: three unrelated HIP streams created before / after the estimator; returns (sha of inv_state, ms)."""
    import subprocess
    import sys
    code = r'''
import ctypes, hashlib, sys, time, torch
from curvature_amd.curvatures import KFAC
order = sys.argv[1]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(1500, 700), torch.nn.ReLU(), torch.nn.Linear(700, 300), torch.nn.ReLU(),
                            torch.nn.Linear(300, 40)).to(dev)
hip = ctypes.CDLL("libamdhip64.so")
keep = []
def extra():
    for _ in range(3):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
        keep.append(h)
if order == "before": extra()
kfac = KFAC(model)
if order == "after": extra()
x = torch.randn(64, 1500, device=dev)
model(x).logsumexp(1).sum().backward()
kfac.update(batch_size=64)
kfac.invert(add=0.1, multiply=10.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
kfac.invert(add=0.1, multiply=10.0)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
h = hashlib.sha256()
for layer in kfac.state:
    for t in kfac.inv_state[layer]:
        h.update(t.cpu().numpy().tobytes())
print("RESULT", h.hexdigest(), ms)
'''
    proc = subprocess.run([sys.executable, "-c", code, order], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=300, cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)))
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, proc.stderr[-2000:]
    _, sha, ms = line[0].split()
    return sha, float(ms)


def test_inversion_is_bit_identical_whatever_streams_the_process_holds(gpu):
    """Round-3 review: the whole-model sweep runs on internal streams whose mapping onto hardware queues depends on the
    streams the process created before them.  That is a timing property only: three unrelated streams created before
    the estimator, after it (the estimator constructor creates the library's stream set: curv_init_streams), or not at
    all give bit-identical inverse factors (two factor groups on two internal streams + two far-update streams: the
    1501-wide factor is above the group split)."""
    shas = {order: _stream_probe(order)[0] for order in ("none", "before", "after")}
    assert shas["none"] == shas["before"] == shas["after"], shas


def test_repeated_sweeps_are_bit_identical(gpu):
    """The sweep runs on four streams (two chains, two streams for the far updates and the fp32 inverse) that hand over through
    events; a missing dependency shows up as a flipped bit sooner or later.  Two factor groups, fp32 inverse, per-step chain
    (more than 64 factors) and the chain-bound forms (a handful): the same inputs 25 times each, every output compared bit for
    bit with the first call's (tools/stress_invert_determinism.py is the long version on the ResNet-50 sizes)."""
    from curvature_amd import ops
    for sizes in ([1536, 1000, 640, 401, 130, 64] * 12, [1536, 640, 130]):
        Fs = []
        for i, n in enumerate(sizes):
            torch.manual_seed(i)
            X = torch.randn(n, n + 8, device=gpu)
            Fs.append((X @ X.t() / (n + 8)).contiguous())
        add, mul = [1.0] * len(Fs), [100.0] * len(Fs)
        ref = [o.clone() for o in ops.chol_inv_lower(Fs, add, mul)]
        for _ in range(25):
            outs = ops.chol_inv_lower(Fs, add, mul)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(outs, ref))


def test_checked_inversion_does_not_depend_on_streams_created_before_the_estimator(gpu):
    """Until round 5 three raw HIP streams created before the estimator put the large factor group's chain on the command-
    processor pipe of the caller's stream - which waited for the sweep - and invert() of the ResNet-50 factors took 11 ms
    instead of 6.9 (LAB_NOTEBOOK R5.6).  The checked call now joins the caller's stream after the host wait: the two
    process layouts must stay within 25 % of each other (they are within 2 %; the old ratio was 1.6)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    times = {}
    for mode in ("none", "before"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--stream-probe", mode, "--batch", "32"],
                             check=True, capture_output=True, text=True, timeout=600).stdout
        times[mode] = json.loads(out.strip().splitlines()[-1])["invert_ms"]
    assert times["before"] < 1.25 * times["none"], times
