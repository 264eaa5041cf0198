"""bench.py's launcher logic on the CPU: `--gpus N` without a torchrun environment starts N CHILD rank processes
through torch.distributed.run on 127.0.0.1 (and never touches the GPU or replaces the process first); inside a
torchrun environment it must not spawn again; WORLD_SIZE and --gpus must agree."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_spawns_child_ranks(monkeypatch):
    bench = load_bench()
    import subprocess
    seen = {}

    def fake_call(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before spawning")))
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3", "--no-cpu-baseline"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    tail = cmd[script + 1:]
    assert tail[tail.index("--gpus") + 1] == "4" and tail[tail.index("--steps") + 1] == "7"
    assert tail[tail.index("--warmup") + 1] == "3" and "--no-cpu-baseline" in tail
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_more_gpus_than_visible_is_refused(monkeypatch):
    bench = load_bench()
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "only 1 GPU" in str(exc.value.code)


def test_world_size_must_match_gpus(monkeypatch):
    bench = load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "WORLD_SIZE=2" in str(exc.value.code)
