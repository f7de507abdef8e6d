#!/usr/bin/env python3
"""round 5: what slows the resident worker's steps once bench.py has its process groups (104 -> 78-82 M env-steps/s at --steps 20, 108 -> 95-97 M at
--steps 2000 with --force-dist at world size 1)?  One mode per process: none | gloo | nccl | nccl_lazy | nccl+gather | nccl+gloo | all."""
import datetime, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
import torch
import torch.distributed as dist
from triton_racer_sim_amd.env import BatchedEnv
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29600 + (os.getpid() % 300)))
to = datetime.timedelta(seconds=60)
host_group = None
if mode == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1, timeout=to)
elif mode.startswith("nccl") or mode == "all":
    torch.cuda.set_device(0)
    if mode == "nccl_lazy":
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=to)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=to)
    if mode in ("nccl+gloo", "all"):
        host_group = dist.new_group(backend="gloo", timeout=to)
    if mode in ("nccl+gather", "all"):
        warm = torch.zeros(1024, device="cuda")
        dist.all_gather_into_tensor(warm, torch.ones(1024, device="cuda"))
        torch.cuda.synchronize()
if mode == "lazy+gather":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=to)
    warm = torch.zeros(1024, device="cuda")
    dist.all_gather_into_tensor(warm, torch.ones(1024, device="cuda"))
    torch.cuda.synchronize()
if mode == "sleep3":
    torch.zeros(1, device="cuda"); torch.cuda.synchronize(); time.sleep(3.0)
if mode == "nccl_destroyed":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=to)
    dist.destroy_process_group()
n = 1024
env = BatchedEnv(n_envs=n, auto_reset=True)
env.set_step_mode(True, 100000)
t0 = time.perf_counter()
while time.perf_counter() - t0 < float(os.environ.get("PW", "0.08")):
    env.step_synthetic(400, 1); env.sync()
res = []
for k in (20, 20, 2000, 2000, 2000, 2000, 20):
    env.sync(); torch.cuda.current_stream().synchronize()
    a = time.perf_counter(); env.step_synthetic(k, 1); env.sync(); b = time.perf_counter()
    res.append(f"{k}: {(b - a) / k * 1e6:.2f} us/step")
print(f"{mode:12s}", " | ".join(res), "| threads", len(os.listdir("/proc/self/task")), flush=True)
if dist.is_initialized():
    env.quiesce(); dist.destroy_process_group()
