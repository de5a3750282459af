"""bench.py's launch path without a GPU: `--gpus N` with no launcher around it must start ONE child -- torch.distributed.run
with N ranks on 127.0.0.1 -- before anything touches the GPU, hand the child's JSON line through as the only line on stdout,
and return the child's exit code (a hang that the ranks' watchdog turned into exit 3 stays a failure)."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load_bench():
    import importlib
    return importlib.import_module("bench")


def test_defaults_follow_the_contract(monkeypatch):
    bench = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.workload, a.mode) == (1, "gw_200k", "ensemble")
    assert a.steps > 0 and a.warmup > 0 and a.cutoff == 0.6


def test_gpus_n_spawns_one_torchrun_child_and_passes_its_line_through(monkeypatch, capsys):
    bench = _load_bench()
    seen = {}
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 4})

    def fake_run(cmd, env=None, stdout=None):
        seen["cmd"], seen["env"] = cmd, env
        out = ("RCCL version : 2.26.6\n" + line + "\n[Gloo] Rank 0 is connected\n").encode()
        return types.SimpleNamespace(returncode=0, stdout=out)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.spawn_ranks(bench.parse_args())
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # no GPU in this container: fewer devices than ranks -> the rehearsal note and a gloo rendezvous for torch
    assert "MMX_BENCH_REHEARSAL" in seen["env"] and seen["env"]["MMX_DIST_BACKEND"] == "gloo"
    out = capsys.readouterr()
    assert out.out.strip() == line                      # stdout: the JSON line alone
    assert "RCCL version" in out.err and "[Gloo]" in out.err


def test_a_failed_child_stays_a_failure(monkeypatch, capsys):
    bench = _load_bench()
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, stdout=None: types.SimpleNamespace(
        returncode=3, stdout=(json.dumps({"metric": "m", "dd": {"error": "no result after 240 s"}}) + "\n").encode()))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    assert bench.spawn_ranks(bench.parse_args()) == 3
    assert capsys.readouterr().out.startswith('{"metric"')
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, stdout=None: types.SimpleNamespace(returncode=0, stdout=b"nothing\n"))
    assert bench.spawn_ranks(bench.parse_args()) == 5   # a child that printed no line is not a success either
