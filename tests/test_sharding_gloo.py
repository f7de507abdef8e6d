"""N > 1 path on CPU: two gloo ranks, each stepping its own shard (oracle backend injected — tests only), one
all-gather of episode returns; the gathered vector must equal a single-process run over all envs."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

N_TOTAL, STEPS = 48, 80


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from triton_racer_sim_amd import _ffi
    from triton_racer_sim_amd.shard import ShardedEnvs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = _ffi.Api(ctypes.CDLL(os.path.join(ROOT, "oracle", "libtrsim_oracle.so")), "trso_")
    sh = ShardedEnvs(N_TOTAL, rank, world, device=0, _api=api, render=False, auto_reset=True)
    # the protocol bench.py's ranks run at N > 1: K steps between two HOST barriers (timed per rank, MAX over ranks), then the one
    # all-gather timed on its own
    from triton_racer_sim_amd.shard import max_over_ranks, timed_steps
    calls = []
    barrier = lambda: (calls.append("barrier"), dist.barrier())
    wall = timed_steps(sh.env, lambda k: (calls.append(f"run{k}"), sh.step_synthetic(k, 1)), STEPS, barrier, lambda: calls.append("sync"))
    assert calls == ["sync", "barrier", f"run{STEPS}", "sync", "barrier"], calls
    wall_max = max_over_ranks(wall + rank)                  # rank 1's value is larger by construction: both ranks must get it
    full, gather_s = sh.allgather_timed("ep_return", dist.barrier, None)
    idx = sh.allgather("cte")
    np.save(os.path.join(out_dir, f"wall_{rank}.npy"), np.array([wall, wall_max, gather_s]))
    np.save(os.path.join(out_dir, f"ret_{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"cte_{rank}.npy"), idx.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_equals_single_process(tmp_path, make_env):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ref = make_env("oracle", n_envs=N_TOTAL, render=False, auto_reset=True)
    ref.step_synthetic(STEPS, 1)
    w0, w1 = np.load(tmp_path / "wall_0.npy"), np.load(tmp_path / "wall_1.npy")
    assert w0[0] > 0 and w1[0] > 0 and w0[2] > 0 and w1[2] > 0
    assert w0[1] == w1[1] == w1[0] + 1                     # MAX over ranks, the same on every rank
    for rank in (0, 1):                                    # every rank holds the whole vector, in global-id order
        assert np.array_equal(np.load(tmp_path / f"ret_{rank}.npy"), ref.fetch("ep_return"))
        assert np.array_equal(np.load(tmp_path / f"cte_{rank}.npy"), ref.fetch("cte"))


def test_shard_range():
    from triton_racer_sim_amd.shard import shard_range
    assert [shard_range(4096, r, 8) for r in (0, 7)] == [(0, 512), (3584, 512)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
