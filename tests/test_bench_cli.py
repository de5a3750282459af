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


def test_strong_scaling_keys_are_top_level_and_a_hung_leg_exits_3():
    """north_star's 1 -> 8 GPU curve must be readable from the lines' top-level keys, and a decomposed leg that never comes
    back must cost the `dd` object only -- the line is printed -- while the job ends with exit status 3 (a rank hung in a
    collective holds its GPU: the launcher and the driver have to see a failure).  The stall is injected; no GPU involved."""
    import threading
    import time
    bench = _load_bench()
    out = {"metric": "m", "value": 1.0}
    bench.hoist_strong(out, {"value": 2400.0, "single_gpu_iters_per_s": 780.0, "speedup": 3.08, "parallel_efficiency": 0.385,
                             "ranks_rccl_saw": 8, "critical_path_us": 250.0}, 8)
    for k in bench.STRONG_KEYS:
        assert k in out
    assert out["strong_1m_iters_per_s"] == 2400.0 and out["strong_1m_ranks_rccl_saw"] == 8 and out["strong_1m_speedup"] == 3.08
    # a leg that hangs: the watchdog prints the line (rank 0) and leaves with status 3
    emitted, exits, release = [], [], threading.Event()

    def hung(progress):
        progress["stage"] = "injected stall"
        release.wait(10.0)
        return {"value": None}

    line = {"metric": "m", "value": 3300.0}
    t = threading.Thread(target=lambda: bench.guarded_leg(hung, 0.2, 0, 2, line, exit_fn=lambda c: (exits.append(c), release.set()),
                                                          emit_fn=emitted.append), daemon=True)
    t.start()
    t.join(10.0)
    assert exits == [3]
    assert len(emitted) == 1 and "injected stall" in emitted[0]["dd"]["error"] and emitted[0]["value"] == 3300.0
    assert emitted[0]["strong_1m_iters_per_s"] is None and "exit status 3" in emitted[0]["strong_1m_note"]
    # a rank that is not rank 0 prints nothing and leaves with the same status (after giving rank 0 time to print)
    exits2, rel2 = [], threading.Event()
    t0 = time.perf_counter()
    t = threading.Thread(target=lambda: bench.guarded_leg(lambda p: rel2.wait(10.0), 0.1, 1, 2, {}, exit_fn=lambda c: (exits2.append(c), rel2.set()),
                                                          emit_fn=emitted.append), daemon=True)
    t.start()
    t.join(10.0)
    assert exits2 == [3] and len(emitted) == 1 and time.perf_counter() - t0 >= 3.0
    # a leg that comes back in time: nothing fires
    exits3 = []
    leg = bench.guarded_leg(lambda p: {"value": 5.0}, 5.0, 0, 2, {}, exit_fn=exits3.append, emit_fn=emitted.append)
    assert leg == {"value": 5.0} and exits3 == [] and len(emitted) == 1
