"""The N-rank body of bench.py, run for real before the driver does: two rank processes started by torch.distributed.run
on 127.0.0.1 (gloo rendezvous and collectives, both ranks on GPU 0 - the box has one GPU and RCCL refuses two ranks on one
device), the ResNet-50 workload layer-sharded over them.  Checks the JSON line rank 0 prints: n_gpus, the sharded
parallelism, finite phase times, and that both ranks end the run with identical parameters (the one all-gather of
sample_and_replace; reference basis for the sharding: per-layer independence, curvature/curvatures.py:20-21, 117-129)."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_end_to_end():
    # (child processes: nothing in THIS process has to touch the GPU first, and nothing is exec'ed)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.update(BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--no-cpu-baseline", "--no-other-configs"]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, proc.stdout[-2000:]                      # rank 0 prints, rank 1 does not
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "layer-sharded x2"
    assert out["scaling"] == "strong" and out["unit"] == "layers/s"
    for key in ("update", "invert", "sample_and_replace"):
        assert math.isfinite(out["phases_ms"][key]) and out["phases_ms"][key] > 0.0
    assert math.isfinite(out["value"]) and out["value"] > 0.0
    ranks = out["ranks"]
    assert ranks["parameters_identical_on_all_ranks"] is True
    assert ranks["collective_backend"] == "gloo" and ranks["rccl_ranks"] == 0
    lo, hi = ranks["layers_owned"]
    assert 0 < lo <= hi < 54 and lo + hi >= 54 - hi                  # both ranks own layers; together all 54
    for key in ("update_ms", "invert_ms", "sample_and_replace_ms"):
        assert 0.0 < ranks[key][0] <= ranks[key][1]
    frac = out["roofline_phases"]["invert"]["frac"]
    assert 0.0 < frac < 1.0
