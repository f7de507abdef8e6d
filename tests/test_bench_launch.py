"""`python bench.py --gpus N` starts its own rank processes (bench.launch_ranks) — driven here on the CPU with STUB children
(TRS_BENCH_CHILD), because the real ranks need GPUs: the parent must reach the rank processes with the environment
torch.distributed.run would give them, relay exactly rank 0's JSON line on stdout, and exit with the children's code.  The parent
itself must import neither torch nor the HIP library.  The N > 1 timing protocol the ranks run (shard.timed_steps, max_over_ranks,
allgather_timed) is covered by tests/test_sharding_gloo.py on two gloo ranks; one GPU run through the same launch path is
tests/test_bench_contract.py::test_bench_self_launch_on_the_gpu."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(tmp_path, stub_body, n=2, extra=()):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(textwrap.dedent(stub_body))
    env = dict(os.environ, TRS_BENCH_CHILD=str(stub))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "20", "--warmup", "5", *extra], capture_output=True, text=True,
                          timeout=120, cwd=ROOT, env=env)


def test_parent_reaches_the_ranks_and_relays_rank0s_line(tmp_path):
    out = _run(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        print(f"rank {rank} chatter that is not the line")
        print("stderr chatter", file=sys.stderr)
        if rank == 0:
            print(json.dumps({"metric": "stub", "n_gpus": world, "argv": sys.argv[1:], "launcher": os.environ["TRS_BENCH_LAUNCHER"],
                              "torch_in_parent": False}))
    """, n=3)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                       # ONE JSON line on stdout; the chatter went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["launcher"] == "self"
    assert d["argv"] == ["--gpus", "3", "--steps", "20", "--warmup", "5"]       # the ranks see the parent's own command line
    for r in range(3):
        assert f"rank {r} chatter" in out.stderr


def test_parent_exits_with_a_failing_ranks_code_and_stops_the_others(tmp_path):
    out = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)            # a healthy rank that would wait at a barrier for ever: the parent must end it
    """)
    assert out.returncode == 7
    assert "rank 1 exited with code 7" in out.stderr


def test_parent_does_not_import_torch_or_the_hip_library(tmp_path):
    """The parent of the ranks must not initialise the GPU: checked by making torch and ctypes.CDLL unusable in the parent process only."""
    poison = tmp_path / "sitecustomize.py"
    poison.write_text(textwrap.dedent("""
        import os, sys
        if "RANK" not in os.environ:                     # the parent (the ranks get RANK from it)
            class _Refuse:
                def find_spec(self, name, path=None, target=None):
                    if name == "torch" or name.startswith("torch."):
                        raise ImportError("the parent of the rank processes imported torch")
                    return None
            sys.meta_path.insert(0, _Refuse())
            import ctypes
            def _no_cdll(*a, **k):
                raise OSError("the parent of the rank processes loaded a shared library")
            ctypes.CDLL = _no_cdll
    """))
    stub = tmp_path / "stub_rank.py"
    stub.write_text("import os, json\nif os.environ['RANK'] == '0':\n    print(json.dumps({'ok': True}))\n")
    env = dict(os.environ, TRS_BENCH_CHILD=str(stub), PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip()) == {"ok": True}


def test_spawn_flag_takes_the_launch_path_at_one_gpu(tmp_path):
    out = _run(tmp_path, """
        import json, os
        print(json.dumps({"world": os.environ["WORLD_SIZE"], "rank": os.environ["RANK"]}))
    """, n=1, extra=("--spawn",))
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip()) == {"world": "1", "rank": "0"}


def test_a_rank_that_stalls_in_a_phase_says_which_and_exits_3(tmp_path):
    """bench.phase (round 5): a rendezvous / barrier / collective that can only stall on ANOTHER rank is bracketed by a watchdog — the rank
    names the phase it is stuck in and exits with code 3 instead of waiting for torch.distributed's own timeout; the parent then stops the other ranks.  Driven with
    stub ranks that use the real `phase` class of bench.py: rank 1 never reaches the barrier rank 0 waits at."""
    out = _run(tmp_path, f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        if os.environ["RANK"] == "0":
            with bench.phase("host barrier (gloo)", 0.5):
                time.sleep(30)        # the peer never arrives
        else:
            time.sleep(30)
    """)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert "rank 0 stalled in phase 'host barrier (gloo)'" in out.stderr
    assert "rank 0 exited with code 3" in out.stderr
