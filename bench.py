#!/usr/bin/env python
"""Headline benchmark: KFAC factor-build + invert + sample throughput on ResNet-50 (BASELINE.json).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch whose activations / gradients are already resident
in HBM: ``KFAC.update()`` (factor build, all layers) + ``KFAC.invert(1.0, 1000.0)`` +
``KFAC.sample_and_replace()`` on a random-init ImageNet ResNet-50 (54 layers, N = 32, synthetic
3x224x224 inputs).  With N > 1 GPUs the layers are sharded across ranks (disjoint layer groups, replicated
forward/backward, one RCCL all-gather of the sampled weights per step): total work is fixed, so scaling is
"strong".  Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA = 157.3e12      # /opt/skills/guides/MI355X_MICROARCH.md: f32-input MFMA, dense


def layer_dims(layers, record):
    """(n, m, K, build flops) per layer from the recorded activations / gradients; the build flops come from the
    library's launch plan (curv_kfac_plan_info), not from a copy of its rules."""
    from curvature_amd import sharding
    return sharding.layer_dims(layers, {l: (tuple(record[l][0].shape), tuple(record[l][1].shape)) for l in layers})


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _probe_threads():
    """Thread count for the CPU leg.  SURVEY 8(d) asks for os.cpu_count() threads; on many-core hosts torch/MKL at
    hundreds of threads runs this workload (im2col copies, GEMMs and LAPACK calls of 64..4608-wide factors) several
    times SLOWER than at 16-64 threads, which would flatter the GPU.  A bounded probe (one 3x3-conv factor build +
    one 1152-wide invert) is timed at os.cpu_count() and at 64 / 32 / 16 threads; the smallest count within 25 % of the
    fastest is used and every probe time is reported."""
    import oracle.curvature_oracle as o
    total = os.cpu_count() or 1
    cands = sorted({c for c in (total, 64, 32, 16) if c <= total}, reverse=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 128, 28, 28, generator=g)
    gr = torch.randn(4, 128, 28, 28, generator=g)
    times = {}
    for c in cands:
        torch.set_num_threads(c)
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            A, G = o.kfac_factors(x, gr, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), has_bias=False)
            o.kfac_invert(A, G, 1.0, 1000.0)
            best = min(best, time.perf_counter() - t0)
        times[c] = best
    # the probe is 16-20 ms long and cannot tell 64 threads from 32; the whole workload can (same host: update 4.3 s at 64
    # threads, 2.0 s at 32; invert 6.6 s against 3.1): among the counts within 25 % of the fastest probe the SMALLEST one is used
    return pick_threads(times), total, times


def pick_threads(times):
    """The smallest thread count whose probe time is within 25 % of the fastest one."""
    best = min(times.values())
    return min(c for c, t in times.items() if t <= 1.25 * best)


def cpu_baseline(batch_full, seed, budget_s=75.0):
    """The oracle (torch-CPU restatement of the reference path, validated against the imported reference) timed
    on this box's host cores on the SAME workload: ResNet-50, N = `batch_full`, update + invert(1, 1000) +
    sample_and_replace of all 54 layers (CPU model stated, 2 warm-ups, median of 5; thread count: see
    _probe_threads).  The leg is bounded by `budget_s` seconds of CPU work: if the spec's 7 passes of a phase do
    not fit, fewer passes are run and the line says how many (nothing is scaled)."""
    import statistics
    import oracle.curvature_oracle as o
    from curvature_amd import models
    cores, total, probe = _probe_threads()
    torch.set_num_threads(cores)
    torch.manual_seed(seed)
    model = models.resnet50().train()
    x = torch.randn(batch_full, 3, 224, 224)
    rec, _, _ = o.capture(model, x, seed=seed)

    def timed(fn, share):
        """(median seconds, warm-ups, timed passes, last result) within `share` of the budget."""
        t0 = time.perf_counter()
        out = fn()
        first = time.perf_counter() - t0
        allowed = share * budget_s
        if first * 7 <= allowed:
            warm, reps = 2, 5
        elif first * 3 <= allowed:
            warm, reps = 1, 2
        else:
            return first, 0, 1, out                    # the single pass is the measurement
        for _ in range(warm - 1):
            out = fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), warm, reps, out

    t_update, wu, ru, state = timed(lambda: o.model_kfac_update({}, model, rec), 0.6)
    t_invert, wi, ri, inv = timed(lambda: o.model_kfac_invert(state, 1.0, 1000.0), 0.3)

    def sample_pass():
        samples = o.model_kfac_sample(inv, model)
        for layer, s in samples.items():
            o.replace(s, layer.weight.data, layer.bias.data if layer.bias is not None else None)
    t_sample, ws, rs_, _ = timed(sample_pass, 0.1)
    step = t_update + t_invert + t_sample
    n_layers = len(state)
    probe_txt = ", ".join(f"{c} threads {t * 1e3:.0f} ms" for c, t in sorted(probe.items(), reverse=True))
    return {"value": n_layers / step, "unit": "layers/s", "cores": cores, "host_cpu_count": total, "kind": "port",
            "cpu_model": _cpu_model(),
            "sample": f"oracle/curvature_oracle.py on the same workload (ResNet-50, N={batch_full}, all {n_layers} layers) on "
                      f"'{_cpu_model()}' (os.cpu_count() = {total}), torch {cores} threads = the smallest count within 25 % of the fastest of a probe "
                      f"({probe_txt}): update {t_update:.2f} s (median of {ru} after {wu} warm-ups), invert(1, 1000) "
                      f"{t_invert:.2f} s (median of {ri} after {wi}), sample_and_replace {t_sample:.2f} s (median of {rs_} "
                      f"after {ws}); KFAC leg only, no EFB/INF leg",
            "update_s": t_update, "invert_s": t_invert, "sample_s": t_sample,
            "invert_plus_sample_s": t_invert + t_sample}


def cpu_baseline_subprocess(batch_full, timeout_s=300):
    """Run the CPU leg in a child process with a hard time limit (a torch CPU op cannot be interrupted from inside),
    so that the default `python bench.py` always finishes within minutes.  The child never touches the GPU."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--batch", str(batch_full)]
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        for line in reversed(proc.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "layers/s", "cores": None, "kind": "port",
                "sample": f"CPU leg failed (rc {proc.returncode}): {proc.stderr.strip()[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "layers/s", "cores": None, "kind": "port",
                "sample": f"CPU leg did not finish within {timeout_s} s on this host and was stopped"}


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json's other configurations (2: LeNet-5 on one GPU, 3: ResNet-18 KFAC + EFB, 5: the ResNet-50 EFB / INF
# chain), measured AFTER the headline timed region and never inside it.  Each entry is a wall-clock time around the
# call with the GPU idle before and after (torch.cuda.synchronize on both sides), median of a few repeats.
# ------------------------------------------------------------------------------------------------------------------
PEAK_HBM = 8.0e12             # MI355X_MICROARCH.md: HBM3E spec (6.3e12 achievable)
PEAK_F64_MFMA = 78.6e12       # datasheet; reachable: profiles/r04_micro_mfma_f64.txt (rounds 2-3 priced against a mis-measured 48e12)


def inf_rooflines(shapes, update_ms, invert_ms, sample_ms):
    """Roofline of the three INF phases from closed-form counts (curvatures.py:487-507, 538-600), a reader can
    recompute every figure from `shapes` = [(n, m, a, b)] per layer:
      invert   V_s^T V_s in closed form on distinct column pairs, fp64: a(a+1) n m + a(a+1) m b(b+1)/2 flops; then with
               q = a b:  chol(vtv)^-1 (2/3) q^3  +  T = A^-1 - chol(vtv + I)^-1 A^-1 by forward substitution inside the
               second sweep (2/3) q^3  +  P_c = diag(s) A^-T T diag(s) (upper x lower) (2/3) q^3 = 2 q^3, all on the
               fp64 MFMA (78.6 TFLOP/s).  Never the reference's 2 n m q^2.
      update   index selection + gathers + sif_diag = (U_A^2) Lam_lr (U_G^2)^T: 2 n a b + 2 n b m flops, against
               HBM: lambda, diags read, the correction written, 4 B each (12 B per parameter)
      sample   five skinny products 2 b m n + 2 b n a + 2 q^2 + 2 m b a + 2 n a m flops; HBM: X, r, r^2, Y read / written
               and the weight read-modify-written (24 B per parameter) + P_c (4 q^2 B)."""
    vtv = sum(a * (a + 1.0) * n * m + a * (a + 1.0) * m * b * (b + 1.0) / 2.0 for n, m, a, b in shapes)
    chain = sum(2.0 * (float(a) * b) ** 3 for n, m, a, b in shapes)
    inv_flops = vtv + chain
    upd_flops = sum(2.0 * n * a * b + 2.0 * n * b * m for n, m, a, b in shapes)
    upd_bytes = sum(12.0 * n * m for n, m, a, b in shapes)
    smp_flops = sum(2.0 * b * m * n + 2.0 * b * n * a + 2.0 * (float(a) * b) ** 2 + 2.0 * m * b * a + 2.0 * n * a * m
                    for n, m, a, b in shapes)
    smp_bytes = sum(24.0 * n * m + 4.0 * (float(a) * b) ** 2 for n, m, a, b in shapes)
    return {
        "inf_invert": {"bound": "mfma (fp64)", "gflop": inv_flops / 1e9, "gflop_vtv": vtv / 1e9, "gflop_chain": chain / 1e9,
                       "sum_q3": sum((float(a) * b) ** 3 for n, m, a, b in shapes),
                       "peak": PEAK_F64_MFMA / 1e12, "unit": "TFLOP/s", "achieved": inv_flops / (invert_ms * 1e-3) / 1e12,
                       "roof_ms": inv_flops / PEAK_F64_MFMA * 1e3, "frac": inv_flops / (invert_ms * 1e-3) / PEAK_F64_MFMA},
        "inf_update": {"bound": "hbm", "gbyte": upd_bytes / 1e9, "gflop": upd_flops / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                       "achieved": upd_bytes / (update_ms * 1e-3) / 1e9, "roof_ms": upd_bytes / PEAK_HBM * 1e3,
                       "frac": upd_bytes / (update_ms * 1e-3) / PEAK_HBM,
                       "note": "54 layers x (selection read-back + gathers + two small products): launch- and host-bound, not a bandwidth figure"},
        "inf_sample": {"bound": "hbm", "gbyte": smp_bytes / 1e9, "gflop": smp_flops / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                       "achieved": smp_bytes / (sample_ms * 1e-3) / 1e9, "roof_ms": smp_bytes / PEAK_HBM * 1e3,
                       "frac": smp_bytes / (sample_ms * 1e-3) / PEAK_HBM,
                       "note": "five dependent stages of skinny products (K = a, b <= rank): stage latency, not bytes"},
    }


def _timed_gpu(fn, reps=3, warm=1):
    import statistics
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3


def _backward_once(model, x):
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits.detach()).sample()
    model.zero_grad()
    torch.nn.functional.cross_entropy(logits, labels).backward()


def _inf_anchor_layer(n=147, m=64, seed=0):
    """One synthetic layer of ResNet-50's stem size (n = 147, m = 64) for the INF.invert anchor: random orthonormal
    eigenvector matrices, a decaying positive Lambda, a positive diagonal (CPU tensors)."""
    g = torch.Generator().manual_seed(seed)
    U_A = torch.linalg.qr(torch.randn(n, n, generator=g))[0].contiguous()
    U_G = torch.linalg.qr(torch.randn(m, m, generator=g))[0].contiguous()
    lam = (torch.rand(m, n, generator=g) ** 8).contiguous()        # heavy-tailed, unstructured: rank 100 -> a b of a few thousand
    diag = (lam.mean() * (0.5 + torch.rand(m, n, generator=g))).contiguous()
    return U_A, U_G, lam, diag


def other_configs_gpu(dev, model50, kfac50, batch):
    """Configs 2, 3 and 5 on this GPU.  `kfac50` holds the headline workload's factors (ResNet-50, N = batch)."""
    from curvature_amd import models, ops
    from curvature_amd.curvatures import KFAC, EFB, INF
    out = {}

    # ---- config 2: LeNet-5, N = 100, synthetic 28x28 U[0,1) inputs: update + invert(0.5, 1) + sample_and_replace
    torch.manual_seed(0)
    lenet = models.lenet5().to(dev).eval()
    k2 = KFAC(lenet)
    _backward_once(lenet, torch.rand(100, 1, 28, 28, device=dev))

    def lenet_step():
        k2.update(batch_size=100)
        k2.invert(add=0.5, multiply=1)
        k2.sample_and_replace()
    for _ in range(5):
        lenet_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        lenet_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    out["config2_lenet5_n100"] = {"workload": "LeNet-5 N=100: KFAC.update + invert(0.5, 1) + sample_and_replace, 5 layers",
                                  "ms_per_step": ms, "layers_per_s": 5 / ms * 1e3,
                                  "bound": "launch latency (0.6 GFLOP and 0.04 MB per sample: ~16 launches per step)"}
    try:        # the same step captured once and replayed as a HIP graph (curvature_amd.graph; bit-identical to eager)
        from curvature_amd.graph import KFACStepGraph
        k2.restart_accumulation()
        k2.update(batch_size=100)
        step_graph = KFACStepGraph(k2, add=0.5, multiply=1, batch_size=100)
        for _ in range(5):
            step_graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step_graph.replay()
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t0) / 100 * 1e3
        step_graph.check()
        out["config2_lenet5_n100"].update({"graph_replay_ms_per_step": gms, "graph_replay_layers_per_s": 5 / gms * 1e3})
        del step_graph
    except Exception as exc:
        out["config2_lenet5_n100"]["graph_replay_error"] = f"{type(exc).__name__}: {exc}"
    del k2, lenet

    # ---- config 3: ImageNet ResNet-18, N = 32: KFAC step, then EFB (eigenvectors, update, invert, sample)
    torch.manual_seed(0)
    r18 = models.resnet18().to(dev).train()
    k3 = KFAC(r18)
    _backward_once(r18, torch.randn(32, 3, 224, 224, device=dev))
    c3 = {"workload": "ResNet-18 N=32, 21 layers: KFAC update / invert(1, 1000) / sample_and_replace; EFB constructor "
                      "(eigenvectors of the 42 factors), update, invert(1, 1000), sample_and_replace"}
    k3.update(batch_size=32)
    c3["kfac_update_ms"] = _timed_gpu(lambda: k3.update(batch_size=32))
    k3.restart_accumulation()                                                         # back to one batch
    k3.update(batch_size=32)
    c3["kfac_invert_ms"] = _timed_gpu(lambda: k3.invert(1.0, 1000.0))
    c3["kfac_sample_and_replace_ms"] = _timed_gpu(k3.sample_and_replace)
    c3["kfac_step_ms"] = c3["kfac_update_ms"] + c3["kfac_invert_ms"] + c3["kfac_sample_and_replace_ms"]
    for key in ("efb_eigenvectors_first_call_ms", "efb_eigenvectors_ms"):   # the first call allocates the workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e3 = EFB(r18, k3.state)
        torch.cuda.synchronize()
        c3[key] = (time.perf_counter() - t0) * 1e3
    c3["efb_eigensolver_sweeps"] = int(getattr(ops.eigh, "last_sweeps", 0))
    ranks3 = dict(getattr(ops.eigh, "last_ranks", {}))
    r18.load_state_dict(k3.model_state)
    _backward_once(r18, torch.randn(32, 3, 224, 224, device=dev))
    c3["efb_update_ms"] = _timed_gpu(lambda: e3.update(32))
    c3.update(_efb_eig_fracs(k3.state, c3["efb_update_ms"], c3["efb_eigenvectors_ms"], c3["efb_eigensolver_sweeps"], "efb_", ranks3))
    c3["efb_invert_ms"] = _timed_gpu(lambda: e3.invert(1.0, 1000.0))
    c3["efb_sample_and_replace_ms"] = _timed_gpu(e3.sample_and_replace)
    out["config3_resnet18_kfac_efb"] = c3
    del e3, k3, r18
    torch.cuda.empty_cache()

    # ---- the README's other model family (README.rst:259-267): DenseNet-121, N = batch, KFAC step
    try:
        torch.manual_seed(0)
        dn = models.densenet121().to(dev).train()
        kd = KFAC(dn)
        _backward_once(dn, torch.randn(batch, 3, 224, 224, device=dev))
        kd._count_flops = True
        kd.update(batch_size=batch)
        kd._count_flops = False
        plan = float(getattr(kd, "_last_flops", 0.0))
        cd = {"workload": f"DenseNet-121 N={batch}, 121 layers (1x1 inputs 64 + 32 k wide, 58 3x3 with C = 128): KFAC update / "
                          "invert(1, 1000) / sample_and_replace"}
        cd["kfac_update_ms"] = _timed_gpu(lambda: kd.update(batch_size=batch))
        cd["kfac_update_plan_gflop"] = plan / 1e9
        cd["kfac_update_frac"] = plan / (cd["kfac_update_ms"] * 1e-3) / PEAK_F32_MFMA
        kd.restart_accumulation()
        kd.update(batch_size=batch)
        cd["kfac_invert_ms"] = _timed_gpu(lambda: kd.invert(1.0, 1000.0))
        cd["kfac_sample_and_replace_ms"] = _timed_gpu(kd.sample_and_replace)
        cd["kfac_step_ms"] = cd["kfac_update_ms"] + cd["kfac_invert_ms"] + cd["kfac_sample_and_replace_ms"]
        out["densenet121_kfac"] = cd
        for h in kd.hooks:
            h.remove()
        del kd, dn
        torch.cuda.empty_cache()
    except Exception as exc:                                                           # a leg of its own: never the headline
        out["densenet121_kfac"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- config 5: ResNet-50 chain on the headline's factors: eigenvectors, EFB update, INF update / invert / sample
    c5 = {"workload": f"ResNet-50 N={batch}, 54 layers, on the headline run's KFAC factors: EFB constructor (eigenvectors of "
                      "the 108 factors), efb.update, INF(..., eigvecs=efb.eigvecs).update(rank=100), inf.invert at "
                      "(1, 1000) and at the README's (145307, 60), inf.sample_and_replace"}
    for key in ("eigenvectors_first_call_ms", "eigenvectors_ms"):     # the first call allocates the 5.6 GB workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e5 = EFB(model50, kfac50.state)
        torch.cuda.synchronize()
        c5[key] = (time.perf_counter() - t0) * 1e3
    c5["eigensolver_sweeps"] = int(getattr(ops.eigh, "last_sweeps", 0))
    ranks5 = dict(getattr(ops.eigh, "last_ranks", {}))
    model50.load_state_dict(kfac50.model_state)
    _backward_once(model50, torch.randn(batch, 3, 224, 224, device=dev))
    c5["efb_update_ms"] = _timed_gpu(lambda: e5.update(batch))
    c5.update(_efb_eig_fracs(kfac50.state, c5["efb_update_ms"], c5["eigenvectors_ms"], c5["eigensolver_sweeps"], "", ranks5))
    inf = INF(model50, e5.diags, kfac50.state, e5.state, eigvecs=e5.eigvecs)
    c5["inf_update_rank100_ms"] = _timed_gpu(lambda: inf.update(rank=100), reps=2)
    c5["inf_invert_1_1000_ms"] = _timed_gpu(lambda: inf.invert(1.0, 1000.0), reps=2)
    c5["inf_invert_145307_60_ms"] = _timed_gpu(lambda: inf.invert(145307.0, 60.0), reps=2)
    c5["inf_sample_and_replace_ms"] = _timed_gpu(inf.sample_and_replace)
    c5["efb_invert_ms"] = _timed_gpu(lambda: e5.invert(1.0, 1000.0))
    c5["efb_sample_and_replace_ms"] = _timed_gpu(e5.sample_and_replace)
    shapes5 = [(int(ua.shape[0]), int(ug.shape[0]), int(ua.shape[1]), int(ug.shape[1])) for ua, ug, _, _ in inf.state.values()]
    c5["inf_shapes_n_m_a_b"] = shapes5
    c5["roofline_phases"] = inf_rooflines(shapes5, c5["inf_update_rank100_ms"], c5["inf_invert_1_1000_ms"],
                                          c5["inf_sample_and_replace_ms"])
    del inf, e5
    torch.cuda.empty_cache()
    # the eigensolver on FULL-RANK factors: everything above decomposes factors of ONE batch (K = N L = 1568 < 4608 for
    # the three widest: numerical rank 1568, taken through their range), the favourable case.  Factors accumulated over
    # a dataset (scripts/factors.py:46-61) are full rank: four DIFFERENT batches here (K = 6272 > 4608)
    kfac50.restart_accumulation()
    for b in range(4):
        model50.load_state_dict(kfac50.model_state)
        torch.manual_seed(1000 + b)
        _backward_once(model50, torch.randn(batch, 3, 224, 224, device=dev))
        kfac50.update(batch_size=batch)
    for key in ("eigenvectors_full_rank_first_call_ms", "eigenvectors_full_rank_ms"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e5f = EFB(model50, kfac50.state)
        torch.cuda.synchronize()
        c5[key] = (time.perf_counter() - t0) * 1e3
    full = _efb_eig_fracs(kfac50.state, 1.0, c5["eigenvectors_full_rank_ms"], int(getattr(ops.eigh, "last_sweeps", 0)), "",
                          dict(getattr(ops.eigh, "last_ranks", {})))
    c5["eigensolver_full_rank"] = {"batches_accumulated": 4, "sweeps": int(getattr(ops.eigh, "last_sweeps", 0)),
                                   **{k: v for k, v in full.items() if k.startswith("eigensolver")}}
    del e5f
    out["config5_resnet50_efb_inf_chain"] = c5
    # the same INF.invert on ONE synthetic layer of the stem's size, the only size at which the reference's explicit
    # (n m) x (a b) Kronecker chain can be timed on the host (cpu_oracle.inf_invert_anchor_ms)
    # built through the estimator itself (INF.update on this GPU selects the low-rank part): nothing of oracle/ runs in this leg
    U_A, U_G, lam, diag = _inf_anchor_layer()
    stem = torch.nn.Sequential(torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)).to(dev)
    conv = stem[0]
    anchor = INF(stem, {conv: diag.to(dev)}, {conv: [U_A.to(dev), U_G.to(dev)]}, {conv: lam.to(dev)},
                 eigvecs={conv: (U_A.to(dev), U_G.to(dev))})
    anchor.update(rank=100)
    ua_d, ug_d = anchor.state[conv][0], anchor.state[conv][1]
    out["inf_invert_anchor"] = {"layer": "synthetic n=147, m=64 (ResNet-50 stem size), rank 100: a x b = %d x %d" % (ua_d.shape[1], ug_d.shape[1]),
                                "gpu_ms": _timed_gpu(lambda: anchor.invert(1.0, 1000.0), reps=5)}
    return out


def other_configs_cpu(budget_s=100.0):
    """The oracle on the host cores for the other configurations, bounded: a leg whose predicted time exceeds 60 s
    (eigendecompositions and the explicit Kronecker chain of INF at ResNet size) is skipped and says so."""
    import oracle.curvature_oracle as o
    from curvature_amd import models
    cores, total, _ = _probe_threads()
    torch.set_num_threads(cores)
    res = {"cores": cores, "host_cpu_count": total, "kind": "port"}
    t_start = time.perf_counter()

    def left():
        return budget_s - (time.perf_counter() - t_start)

    # config 2 (and BASELINE.json's config 1: the reference's own CPU-runnable case): LeNet-5, N = 100
    torch.manual_seed(0)
    lenet = models.lenet5().eval()
    rec, _, _ = o.capture(lenet, torch.rand(100, 1, 28, 28), seed=0)

    def lenet_step():
        st = o.model_kfac_update({}, lenet, rec)
        inv = o.model_kfac_invert(st, 0.5, 1.0)
        for layer, smp in o.model_kfac_sample(inv, lenet).items():
            o.replace(smp, layer.weight.data, layer.bias.data)
    for _ in range(3):
        lenet_step()
    t0 = time.perf_counter()
    for _ in range(20):
        lenet_step()
    res["config2_lenet5_n100_ms_per_step"] = (time.perf_counter() - t0) / 20 * 1e3

    # how fast is a symmetric eigendecomposition here?  (n = 768, scaled with n^3 for the predictions below)
    M = torch.randn(768, 768)
    M = M @ M.t()
    torch.linalg.eigh(M)
    t0 = time.perf_counter()
    torch.linalg.eigh(M)
    eigh_unit = (time.perf_counter() - t0) / 768.0 ** 3

    # config 3: ResNet-18, N = 32
    if left() > 40:
        torch.manual_seed(0)
        r18 = models.resnet18().train()
        rec, _, _ = o.capture(r18, torch.randn(32, 3, 224, 224), seed=0)
        t0 = time.perf_counter()
        st = o.model_kfac_update({}, r18, rec)
        res["config3_resnet18_kfac_update_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        inv = o.model_kfac_invert(st, 1.0, 1000.0)
        res["config3_resnet18_kfac_invert_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for layer, smp in o.model_kfac_sample(inv, r18).items():
            o.replace(smp, layer.weight.data, layer.bias.data if layer.bias is not None else None)
        res["config3_resnet18_kfac_sample_and_replace_ms"] = (time.perf_counter() - t0) * 1e3
        est = eigh_unit * sum(f.shape[0] ** 3 for pair in st.values() for f in pair)
        if est < min(60.0, left()):
            t0 = time.perf_counter()
            for pair in st.values():
                for f in pair:
                    o.eigenvectors(f)
            res["config3_resnet18_eigenvectors_ms"] = (time.perf_counter() - t0) * 1e3
        else:
            res["config3_resnet18_eigenvectors_ms"] = None
            res["config3_resnet18_eigenvectors_note"] = (f"skipped: torch.linalg.eigh of the 42 factors predicted at {est:.0f} s "
                                                         "on this host (n^3 scaling of a measured 768-wide decomposition)")
        del st, inv, rec, r18
    else:
        res["config3_note"] = "skipped: CPU budget of the other-configs leg exhausted"

    # config 5: eigenvectors of the 108 ResNet-50 factors; the reference's explicit-Kronecker pre_sampler is out of reach
    n3 = 4.528e11                                          # sum of n^3 + m^3 over ResNet-50's 54 layers
    est = eigh_unit * n3
    res["config5_resnet50_eigenvectors_ms"] = None
    if est < 60.0 and left() > est + 25.0:
        torch.manual_seed(0)
        r50 = models.resnet50().train()
        rec, _, _ = o.capture(r50, torch.randn(32, 3, 224, 224), seed=0)
        st = o.model_kfac_update({}, r50, rec)
        t0 = time.perf_counter()
        for pair in st.values():
            for f in pair:
                o.eigenvectors(f)
        res["config5_resnet50_eigenvectors_ms"] = (time.perf_counter() - t0) * 1e3
        del st, rec, r50
    else:
        res["config5_resnet50_eigenvectors_note"] = (f"skipped: torch.linalg.eigh of the 108 factors predicted at {est:.0f} s on this "
                                                     f"host, {left():.0f} s of the leg's budget left")
    res["config5_inf_note"] = ("INF.invert / sample at ResNet-50 size: no CPU figure - the reference's pre_sampler materialises a "
                               "(n m) x (a b) matrix per layer (SURVEY 3.3: 18-46 s and 1.2-4 GB per layer on 8 cores; fp32 Cholesky "
                               "fails on some layers), far beyond the 60 s bound of this leg; inf_invert_anchor_ms is the one layer "
                               "size at which it can be timed")
    if left() > 20:
        # one synthetic layer of the stem's size through the oracle's literal INF.invert (explicit Kronecker matrix);
        # other_configs.inf_invert_anchor.gpu_ms is the HIP path on the same layer
        U_A, U_G, lam, diag = _inf_anchor_layer()
        ua, ug, lam_lr, corr, _, _ = o.inf_update(U_A, U_G, lam, diag, 100)
        o.inf_invert(ua, ug, lam_lr, corr, 1.0, 1000.0)
        t0 = time.perf_counter()
        o.inf_invert(ua, ug, lam_lr, corr, 1.0, 1000.0)
        res["inf_invert_anchor_ms"] = (time.perf_counter() - t0) * 1e3
    res["seconds_used"] = time.perf_counter() - t_start
    return res


def other_configs_cpu_subprocess(timeout_s=240):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-other-only"]
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        for line in reversed(proc.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"note": f"CPU leg failed (rc {proc.returncode}): {proc.stderr.strip()[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"note": f"CPU leg did not finish within {timeout_s} s on this host and was stopped"}


def stream_probe(mode: str, batch: int):
    """Child process of `stream_probe_subprocess`: invert(1, 1000) of the headline workload's factors in a process that
    holds three unrelated HIP streams (hipStreamCreateWithFlags, what RCCL or a data loader would add), created
    "before" or "after" the estimator - whose constructor creates the library's own stream set (curv_init_streams) -
    or not at all ("none").  The whole-model sweep is sensitive to streams created BEFORE its own."""
    import ctypes
    import statistics
    from curvature_amd import models
    from curvature_amd.curvatures import KFAC
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = models.resnet50().to(dev).train()
    hip = ctypes.CDLL("libamdhip64.so")
    keep = []

    def extra():
        for _ in range(3):
            h = ctypes.c_void_p()
            if hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) != 0:
                raise RuntimeError("hipStreamCreateWithFlags failed")
            keep.append(h)
    if mode == "before":
        extra()
    kfac = KFAC(model)
    if mode == "after":
        extra()
    _backward_once(model, torch.randn(batch, 3, 224, 224, device=dev))
    kfac.update(batch_size=batch)
    check = os.environ.get("BENCH_PROBE_CHECK", "1") != "0"      # 0: the unchecked entry point (no host wait inside the call)
    for _ in range(3):
        kfac.invert(add=1.0, multiply=1000.0, check=check)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kfac.invert(add=1.0, multiply=1000.0, check=check)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"invert_ms": statistics.median(ts)}))


def _efb_eig_fracs(state, efb_update_ms, eig_ms, sweeps, prefix, ranks=None):
    """Roofline fractions a reader can recompute from the line: EFB.update = U_G^T grad U_A per layer (curvatures.py:424-427),
    2 (m^2 n + m n^2) flops on the fp32 MFMA path; the block-Jacobi eigensolver is HBM-bound: a sweep of an n-wide matrix is
    n / 32 - 1 rounds, and a round of the fp32 phase moves 14 n^2 bytes (the symmetric two-sided pass over A32: 6 n^2, the
    column pass over V32: 8 n^2; DESIGN K4).  Every matrix is priced at the sweep count of the slowest one and at the fp32
    phase's bytes (the two or three fp64 sweeps move twice as much): the fraction is an estimate, good to ~20 %.  A wide
    rank-deficient factor that went through its range (csrc/eigh_lowrank.hip; `ranks` = position -> k) is priced at the k x k
    problem the iteration ran on - the range finder's products are not counted."""
    flops = sum(2.0 * (G.shape[0] ** 2 * A.shape[0] + G.shape[0] * A.shape[0] ** 2) for A, G in state.values())
    widths = [n for A, G in state.values() for n in (A.shape[0], G.shape[0])]
    widths = [(ranks or {}).get(i, n) for i, n in enumerate(widths)]
    per_sweep = sum(max(n / 32.0 - 1.0, 1.0) * 14.0 * float(n) ** 2 for n in widths)
    out = {"efb_update_gflop": flops / 1e9,
           "efb_update_frac": flops / (efb_update_ms * 1e-3) / PEAK_F32_MFMA,
           "efb_update_frac_of": "2 (m^2 n + m n^2) flops per layer / time / 157.3 TFLOP/s (fp32 MFMA)"}
    if sweeps:
        out["eigensolver_projected"] = {str(i): k for i, k in (ranks or {}).items()}
        out["eigensolver_hbm_gbytes_per_sweep"] = per_sweep / 1e9
        out["eigensolver_frac"] = per_sweep * sweeps / (eig_ms * 1e-3) / PEAK_HBM
        out["eigensolver_frac_of"] = ("sweeps x sum over factors of (n / 32 - 1) rounds x 14 n^2 bytes / time / 8 TB/s (HBM spec); "
                                      "every factor priced at the slowest one's sweep count, a projected one (eigensolver_projected: position -> rank) at its "
                                      "k x k problem")
    return out


def stream_probe_subprocess(mode: str, batch: int, timeout_s=180):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--stream-probe", mode, "--batch", str(batch)]
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        for line in reversed(proc.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)["invert_ms"]
        return f"failed (rc {proc.returncode}): {proc.stderr.strip()[-200:]}"
    except subprocess.TimeoutExpired:
        return f"did not finish within {timeout_s} s"


def roofline_phases(dims, owned, invert_ms, sample_ms):
    """Roofline of the two phases that are not the factor build, from quantities in the line itself.
    invert(): (2/3)(n^3 + m^3) flops per layer (one Cholesky factorisation + one triangular inverse, curvatures.py:368-385);
    since round 5 the factorisation half runs on fp64 MFMA (78.6 TFLOP/s) and the inverse half on fp32 MFMA (157.3):
    `frac` = time at those two roofs / measured time; `frac_all_fp64_roof` prices all flops at 78.6 (rounds 1-4's figure).
    sample_and_replace(): (L_A z L_G^T)^T per layer (:387-392) after the triangular cut: n^2 m + n m^2 flops, fp32 MFMA."""
    n3 = sum(float(dims[i][0]) ** 3 + float(dims[i][1]) ** 3 for i in owned)
    inv_flops = (2.0 / 3.0) * n3
    roof_s = (n3 / 3.0) / PEAK_F64_MFMA + (n3 / 3.0) / PEAK_F32_MFMA
    smp_flops = sum(float(dims[i][0]) ** 2 * dims[i][1] + float(dims[i][0]) * float(dims[i][1]) ** 2 for i in owned)
    return {
        "invert": {"bound": "mfma (factorisation fp64, triangular inverse fp32)", "gflop": inv_flops / 1e9,
                   "gflop_fp64": n3 / 3.0 / 1e9, "gflop_fp32": n3 / 3.0 / 1e9,
                   "peak_fp64": PEAK_F64_MFMA / 1e12, "peak_fp32": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                   "achieved": inv_flops / (invert_ms * 1e-3) / 1e12,
                   "roof_ms": roof_s * 1e3, "frac": roof_s / (invert_ms * 1e-3),
                   "frac_all_fp64_roof": inv_flops / (invert_ms * 1e-3) / PEAK_F64_MFMA},
        "sample_and_replace": {"bound": "mfma", "gflop": smp_flops / 1e9, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                               "achieved": smp_flops / (sample_ms * 1e-3) / 1e12,
                               "frac": smp_flops / (sample_ms * 1e-3) / PEAK_F32_MFMA,
                               "flops_counted": "n^2 m + n m^2 per layer: both products after the triangular cut"},
    }


def syrk_source_sha16() -> str:
    """sha256 (first 16 hex digits) over the factor-build sources: what `roofline.traffic`'s PMC file must match."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "curvature_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "syrk*.hip")) + [os.path.join(csrc, "syrk_plan.h")]):
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes (one per GPU) with
    torch.distributed.run and return its exit code.  Called before this process has made any HIP call (torch is
    imported, the GPU is not initialised), and it starts CHILD processes - it never replaces this one."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # does not initialise the GPU on this image
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_other_configs:
        cmd.append("--no-other-configs")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the untimed-region legs for BASELINE.json's configs 2, 3 and 5 (`other_configs`)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-other-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stream-probe", default=None, choices=["none", "before", "after"], help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.stream_probe is not None:               # child of stream_probe_subprocess
        stream_probe(args.stream_probe, args.batch)
        return
    if args.cpu_baseline_only:                      # child of cpu_baseline_subprocess: CPU only, no GPU call
        print(json.dumps(cpu_baseline(args.batch, 0)))
        return
    if args.cpu_other_only:                         # child of other_configs_cpu_subprocess: CPU only, no GPU call
        print(json.dumps(other_configs_cpu()))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("BENCH_SINGLE_DEVICE"):       # test hook: all ranks on GPU 0 (with BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the library's internal streams before anything else creates streams on this device (RCCL's communicator streams
    # below; DESIGN 3 K2 / INTEGRATION.md): the estimator constructor would do it, but only after init_process_group
    from curvature_amd import _lib as _curv_lib
    _curv_lib.init_streams(dev)
    if world > 1:
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from curvature_amd import _lib, models, sharding
    from curvature_amd.curvatures import KFAC

    seed = 0
    torch.manual_seed(seed)
    # BatchNorm in batch-statistics mode: a random-init network in eval mode (running stats 0/1) lets the
    # activations explode with depth, which no trained network does; train mode keeps them O(1)
    model = models.resnet50().to(dev).train()
    kfac = KFAC(model)
    layers = kfac._layers()
    x = torch.randn(args.batch, 3, 224, 224, device=dev)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits.detach()).sample()      # scripts/test.py:39-40
    loss = torch.nn.functional.cross_entropy(logits, labels)
    model.zero_grad()
    loss.backward()                       # activations / gradients of every layer now resident in HBM
    dims = layer_dims(layers, kfac.record)
    if world > 1:
        kfac.shard = sharding.make_layer_shard(dims, rank, world)
    owned = [i for i, _ in kfac._owned()]

    L = _lib.lib()
    # one set of events per timed step: they are read after the timed region, so that the steps run back to
    # back (the only host synchronisation inside a step is invert()'s read-back of its status words)
    hip_ev = [(L.curv_event_create(), L.curv_event_create()) for _ in range(max(args.steps, 1))]
    tev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(max(args.steps, 1))]

    # The bench repeats ONE batch.  update() accumulates (the reference's `+=`), so the rank-deficient factors of that
    # batch grow linearly with the step count while the damping of invert(1, 1000) stays: after ~200 accumulations
    # the fp32 rounding noise of the largest factor exceeds the damping and its Cholesky factorisation fails, as it
    # would in the reference (tools/accumulation_limit.py).  Every RESTART-th step therefore starts the accumulation
    # again (its update() overwrites instead of adding); 15 of 16 steps time the accumulating form.
    RESTART = 16
    counter = {"steps": 0}

    def step(e):
        if counter["steps"] % RESTART == 0 and counter["steps"] > 0:
            kfac.restart_accumulation()
        counter["steps"] += 1
        if e is not None:
            e[0].record()
        kfac.update(batch_size=args.batch)
        if e is not None:
            e[1].record()
        kfac.invert(add=1.0, multiply=1000.0)
        if e is not None:
            e[2].record()
        kfac.sample_and_replace()
        if e is not None:
            e[3].record()

    kfac._timing_events = hip_ev[0]
    kfac._count_flops = True              # first warm-up update: ask the launch plan what it executes
    for _ in range(max(args.warmup, 1)):
        step(None)
        kfac._count_flops = False
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        kfac._timing_events = hip_ev[k]
        step(tev[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    phase = [0.0, 0.0, 0.0]
    syrk_ms = 0.0
    for k in range(args.steps):
        for p in range(3):
            phase[p] += tev[k][p].elapsed_time(tev[k][p + 1])
        ms = __import__("ctypes").c_float(0.0)
        _lib.check(L.curv_event_elapsed_ms(hip_ev[k][0], hip_ev[k][1], ms), "curv_event_elapsed_ms")
        syrk_ms += ms.value
    rank_phases = None
    if world > 1:
        cdev = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        # per-rank phase times (max / min over ranks) and the number of ranks of the library's own RCCL communicator
        mine = torch.tensor([p / args.steps for p in phase], dtype=torch.float64, device=cdev)
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        owned_n = torch.tensor([float(len(owned))], dtype=torch.float64, device=cdev)
        owned_hi, owned_lo = owned_n.clone(), owned_n.clone()
        dist.all_reduce(owned_hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(owned_lo, op=dist.ReduceOp.MIN)
        # every rank must hold the same parameters after the all-gather of sample_and_replace
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1).to(cdev)
        d_hi, d_lo = digest.clone(), digest.clone()
        dist.all_reduce(d_hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(d_lo, op=dist.ReduceOp.MIN)
        rank_phases = {"update_ms": [float(lo[0]), float(hi[0])], "invert_ms": [float(lo[1]), float(hi[1])],
                       "sample_and_replace_ms": [float(lo[2]), float(hi[2])], "what": "[min, max] over ranks",
                       "layers_owned": [int(owned_lo), int(owned_hi)],
                       "parameters_identical_on_all_ranks": bool(float(d_hi) == float(d_lo)),
                       "collective_backend": dist.get_backend(),
                       "allgather": "curv_allgather_weights (library communicator)" if kfac.shard.rccl_ranks() else
                                    "torch.distributed all-gather on padded shards",
                       "rccl_ranks": kfac.shard.rccl_ranks() if kfac.shard.rccl_ranks() else (world if dist.get_backend() == "nccl" else 0)}

    # work of the dominant kernels (the factor build) on this rank.  `plan_flops`: the multiply-add flops the launch
    # plan really executes (curv_kfac_plan_info): n (n + 1) K per symmetric factor, and for the 3x3 / stride 1 / pad 1
    # A factors - assembled from 13 shifted correlations + border strips instead of 45 blocks - the flops of those
    # correlations (3.1x fewer).  `direct_flops`: SURVEY 8(d)'s figure, every factor as one symmetric product.
    direct_flops = sum((dims[i][0] * (dims[i][0] + 1.0) + dims[i][1] * (dims[i][1] + 1.0)) * dims[i][2] for i in owned)
    dense_flops = sum(2.0 * (dims[i][0] ** 2 + dims[i][1] ** 2) * dims[i][2] for i in owned)
    plan_flops = float(getattr(kfac, "_last_flops", direct_flops))
    syrk_s = syrk_ms / args.steps * 1e-3
    achieved = plan_flops / syrk_s / 1e12

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; they are
    # collected by tools/collect_profiles.sh (rocprofv3, separate FETCH_SIZE / WRITE_SIZE passes, gfx950
    # correction of MI355X_MICROARCH.md) on this same workload and committed under profiles/
    # The figure is only reported while the factor-build sources are the ones it was measured on (`source_sha16` of the
    # file = sha256 over csrc/syrk*.hip + syrk_plan.h, written by tools/collect_profiles.sh); otherwise null.
    traffic, traffic_source = None, None
    for name in ("r06_syrk_pmc.json",):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if world == 1 and args.batch == 32 and os.path.exists(pmc_path):
            try:
                rec = json.load(open(pmc_path))
                if rec.get("source_sha16") == syrk_source_sha16():
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/{name}: rocprofv3 PMC passes of this workload on these sources, NOT measured in this run"
                else:
                    traffic_source = (f"profiles/{name} was measured on other factor-build sources (sha {rec.get('source_sha16')} "
                                      f"vs {syrk_source_sha16()}): not reported; re-run tools/collect_profiles.sh")
                break
            except Exception:
                traffic = None

    if rank == 0:
        n_layers = len(layers)
        out = {
            "metric": "KFAC factor-build + invert + sample throughput",
            "value": n_layers * args.steps / elapsed,
            "unit": "layers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ResNet-50 random-init (seed 0), synthetic 3x224x224 N(0,1) inputs, N=32: "
                                   "KFAC.update + invert(1.0, 1000.0) + sample_and_replace, 54 layers",
                       "batch": args.batch, "layers": n_layers, "accumulation_restarts_every": 16,
                       "parallelism": f"layer-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": "the whole factor build of update(): curv::syrk_flat_kernel + curv::syrk_pre_kernel + "
                                                       "curv::syrk_patch_kernel (LDS-DMA kernel for flattened factors and the shifted "
                                                       "correlations of 3x3 factors; implicit-im2col kernel with LDS-DMA staging from "
                                                       "pre-tiled copies; its register-staged variant on a side stream) WITH the padding / "
                                                       "pre-tiling passes in front of them and the k-slice reduction and 3x3 assembly "
                                                       "passes behind them: HIP events on the launch stream around everything "
                                                       "curv_kfac_accumulate_ex enqueues", "achieved": achieved,
                         "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s", "frac": achieved / (PEAK_F32_MFMA / 1e12),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "flops_counted": "what the launch plan executes (curv_kfac_plan_info): n(n+1)K per symmetric "
                                          "factor; 3x3/s1/p1 A factors as 13 shifted correlations + border strips",
                         "plan_gflop": plan_flops / 1e9,
                         "direct_symmetric_gflop": direct_flops / 1e9,
                         "direct_symmetric_tflops": direct_flops / syrk_s / 1e12,
                         "dense_equivalent_tflops": dense_flops / syrk_s / 1e12,
                         "kernel_ms": syrk_s * 1e3,
                         "update_call_ms": phase[0] / args.steps,
                         "frac_of_update_call": plan_flops / (phase[0] / args.steps * 1e-3) / PEAK_F32_MFMA},
            "phases_ms": {"update": phase[0] / args.steps, "invert": phase[1] / args.steps,
                          "sample_and_replace": phase[2] / args.steps},
            "roofline_phases": roofline_phases(dims, owned, phase[1] / args.steps, phase[2] / args.steps),
        }
        if rank_phases is not None:
            out["ranks"] = rank_phases
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args.batch)
        if world == 1 and not args.no_other_configs:
            # BASELINE.json's configs 2, 3 and 5: after the timed region, never inside it
            try:
                other = other_configs_gpu(dev, model, kfac, args.batch)
            except Exception as exc:                         # the headline line must survive a failure here
                other = {"error": f"{type(exc).__name__}: {exc}"}
            # invert() of this workload in fresh processes that hold three unrelated streams
            other["stream_population"] = {
                "what": "KFAC.invert(1, 1000) of the headline workload, median of 10, in a fresh process with three "
                        "unrelated HIP streams created before / after the estimator (whose constructor creates the "
                        "library's stream set), and without them",
                "invert_ms_no_extra_streams": stream_probe_subprocess("none", args.batch),
                "invert_ms_with_3_extra_streams": stream_probe_subprocess("after", args.batch),
                "invert_ms_with_3_extra_streams_created_before_the_estimator": stream_probe_subprocess("before", args.batch)}
            if not args.no_cpu_baseline:
                other["cpu_oracle"] = other_configs_cpu_subprocess()
            out["other_configs"] = other
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
