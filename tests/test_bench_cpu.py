"""bench.py's launcher logic, no GPU: `python bench.py --gpus N` (WORLD_SIZE unset) must start
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>` as a CHILD process -- never exec, never a
GPU call first -- and relay rank 0's JSON line with the child's exit code."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_builds_the_torchrun_child_and_relays_the_line(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return subprocess.CompletedProcess(cmd, 0, stdout="noise\n" + json.dumps({"n_gpus": 4, "ranks_seen": 4}) + "\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(argparse.Namespace(gpus=4))
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in cmd
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert json.loads(out.out.strip()) == {"n_gpus": 4, "ranks_seen": 4} and "noise" in out.err


def test_self_launch_propagates_a_failing_child(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 3, stdout="boom\n"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(argparse.Namespace(gpus=2))
    assert e.value.code == 3


def test_main_routes_to_the_launcher_only_without_world_size():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index('"WORLD_SIZE" not in os.environ and args.gpus > 1') < body.index("torch.cuda.set_device")
    assert "os.exec" not in src
