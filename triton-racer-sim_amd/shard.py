"""Env shards over the GPUs of one node: one process per GPU, rank r owns the contiguous global env ids
``[r * n_local, (r + 1) * n_local)``.  Shards never exchange state (the track is replicated, RNG streams and
start poses are keyed by GLOBAL env id), so the data path has no collective; the only exchange is ONE
all-gather of the per-env episode returns (RCCL over xGMI with the ``nccl`` backend, 4 B per env — latency
bound, issued per reporting interval, never per step).  The reference has no distributed layer at all
(SURVEY.md §5); this is the MI355X-native fan-out named by BASELINE.json's north_star.
"""
import time

import numpy as np

from .env import BatchedEnv


def shard_range(n_total, rank, world):
    if n_total % world:
        raise ValueError(f"n_total={n_total} is not divisible by world_size={world}")
    n_local = n_total // world
    return rank * n_local, n_local


def timed_steps(env, run, steps, host_barrier, stream_sync=None):
    """The timed region of a sharded run (bench.py, N >= 1): [barrier + stream synchronisation] t0 — ``run(steps)`` — ``env.sync()`` t1
    [barrier].  Returns this rank's ``t1 - t0`` in seconds; the caller takes the MAX over ranks (``max_over_ranks``).

    ``host_barrier`` must not need the GPU (a gloo barrier, or a no-op at world size 1): a resident worker holds a workgroup slot
    and most of the LDS of every CU, so a device-side barrier (an RCCL kernel) would have to wait for the worker to leave and the
    region would time the worker's restart instead of K steps.  ``env.sync()`` returns when every posted / launched step is complete
    in memory (resident mode: the completion flags the worker writes; launch mode: the stream has drained) — the same condition a
    device synchronisation gives, without ending the worker.  ``stream_sync``: the caller's own stream (torch's), synchronised on
    both sides outside [t0, t1]."""
    env.sync()
    if stream_sync is not None:
        stream_sync()
    host_barrier()
    t0 = time.perf_counter()
    run(steps)
    env.sync()
    t1 = time.perf_counter()
    if stream_sync is not None:
        stream_sync()
    host_barrier()
    return t1 - t0


def max_over_ranks(value, group=None):
    """MAX of a host float over the ranks of ``group`` (a gloo group: no GPU work); the value itself without a process group."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class ShardedEnvs:
    def __init__(self, n_total, rank=0, world=1, device=None, _api=None, **env_kw):
        self.n_total, self.rank, self.world = int(n_total), int(rank), int(world)
        self.base, self.n_local = shard_range(self.n_total, self.rank, self.world)
        self.env = BatchedEnv(n_envs=self.n_local, env_id_base=self.base, device=self.rank if device is None else device,
                              _api=_api, **env_kw)

    def step_synthetic(self, n_steps=1, steps_per_launch=1):
        self.env.step_synthetic(n_steps, steps_per_launch)

    def comm_init(self, unique_id=None, exchange=None):
        """Build the C-ABI communicator (``trs_comm_init``: RCCL without torch).  ``unique_id``: rank 0's 128 bytes, already
        distributed by the caller; or ``exchange``: a callable ``bytes_or_None -> bytes`` that broadcasts rank 0's id (rank 0
        passes the id, the others ``None``) — e.g. a file, an environment variable or a ``torch.distributed`` store."""
        if self.world > 1 and unique_id is None:
            if exchange is None:
                raise ValueError("world > 1: pass rank 0's unique id or an exchange callable")
            unique_id = exchange(self.env.comm_unique_id() if self.rank == 0 else None)
        elif self.world == 1 and unique_id is None and exchange is not None:
            unique_id = exchange(self.env.comm_unique_id())
        self.env.comm_init(self.rank, self.world, unique_id)
        self._c_comm = True

    def allgather_timed(self, name="ep_return", host_barrier=None, device_sync=None):
        """The job's ONE exchange, timed on its own (BASELINE configs[3]: one all-gather per reporting interval, never per step):
        the worker is asked to leave first (outside the timing), then ``[barrier] t0 — all-gather — device synchronisation t1``.
        Returns ``(gathered, seconds)``."""
        self.env.quiesce()
        if device_sync is not None:
            device_sync()
        if host_barrier is not None:
            host_barrier()
        t0 = time.perf_counter()
        out = self.allgather(name)
        if device_sync is not None:
            device_sync()
        return out, time.perf_counter() - t0

    def allgather(self, name="ep_return"):
        """All ranks receive the full ``[n_total]`` vector of a per-env float32 field, ordered by global env id."""
        import torch
        import torch.distributed as dist
        if getattr(self, "_c_comm", False) and name == "ep_return":           # RCCL behind the C ABI: no torch process group needed
            return torch.from_numpy(self.env.allgather_returns())
        if not dist.is_initialized():
            return torch.from_numpy(self.env.fetch(name))
        if dist.get_backend() == "nccl":                       # device-resident, zero copy: RCCL all-gather over xGMI
            # a resident worker holds one workgroup slot on EVERY CU and most of their LDS; the collective's kernel of another
            # stream must not have to squeeze in beside it (and with N > 1 must not wait on a peer whose worker never yields):
            # ask the worker to leave first, as the C-ABI path does (trs_allgather_returns -> quiesce); the next step restarts it
            self.env.quiesce()
            local = torch.as_tensor(self.env.device_array(name), device="cuda")      # (device_array waits for the env's stream)
            out = torch.empty(self.n_total, dtype=local.dtype, device="cuda")
        else:                                                  # gloo (CPU tests)
            local = torch.from_numpy(np.ascontiguousarray(self.env.fetch(name)))
            out = torch.empty(self.n_total, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local)
        return out

    def close(self):
        self.env.close()
