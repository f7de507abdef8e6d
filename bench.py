#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused physics + camera step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu E | --total-envs T] [--steps-per-launch 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes ITSELF (launch_ranks: plain child
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, created before this process imports torch or
touches a GPU), relays rank 0's JSON line and exits with the children's code.  Under torch.distributed.run the same file is the rank.

A "step" = one pass of the hot path over every env of every shard: bicycle-model integration, binary64
nearest-point index, cte/done/return, and one 120x160 RGB frame per env written to HBM (inputs and
outputs device-resident; controls from the counter-based generator of include/trsim_spec.h).
Steps reach the GPU the consumer-paced way by default (--step-mode resident): every step is posted on its own to the resident
worker kernel (trs_set_step_mode), the way Car.start calls GymInterface.step once per tick; --step-mode launch is one kernel
launch per step.
Workload at N = 1: BASELINE.json configs[2] (1024 envs, physics + 120x160 RGB pinhole rasteriser, one MI355X).
Workload at N > 1: BASELINE.json configs[3] (4096 envs in total, sharded over the N GPUs = 512 per GPU at N = 8,
one RCCL all-gather of the episode returns closing the job).  --envs-per-gpu / --total-envs override either.
Rank 0 prints ONE JSON line.

Timing (round 3).  The metric is a steady-state rate.  An idle MI355X needs ~25-40 ms of work to raise its clocks, so untimed steps
of the same workload run for PREWARM_S first, then the W warm-up steps.  In resident mode the worker stays on the GPU across the warm-up -> timed
boundary: `value` is the host wall clock from the post of the first timed step to the completion flag of the last (env.sync() waits on the flags
the worker writes once a step's frames are in memory); a device-wide synchronisation there would wait for the worker to leave, so torch's stream
is synchronised instead.  `roofline.achieved` comes from a second pass of the same K steps bracketed by HIP events on the worker's own stream
(= one whole worker launch: start-up, K steps, exit).  --profile-mode (scripts/profile.sh) runs the pre-warm by launches (a resident worker's
pre-warm dispatch would be of no fixed length) and makes every worker dispatch of a trace serve exactly K steps.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREWARM_S = 0.08               # untimed steps of the same workload before the W warm-up steps: GPU clocks at their loaded level


# What v_mfma_f32_32x32x16_f16 delivers on this chip when all 1,024 SIMDs stream it from registers (scripts/mfma_issue.hip, profiles/r04_mfma_issue.txt:
# 32 clocks per MFMA at the ~1.65 GHz the chip sustains under that load).  The roofline's `peak` stays the data sheet's 2.5 PFLOP/s; this is reported beside it.
MEASURED_MFMA_TFLOPS = 1729.0

def algorithmic_bytes(h, w, render, depth=False):
    # H*W*3 image bytes written once (+ H*W*4 of fp32 depth) + 88 B of state / control / telemetry per env-step
    return (h * w * 3 if render else 0) + (h * w * 4 if render and depth else 0) + 88


def usable_cores():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU box grants a
    16-CPU share of a 256-thread host; oversubscribing OpenMP there runs slower than one thread)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    cores = min(cores, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    cores = min(cores, max(1, quota // period))
            break
        except Exception:
            continue
    return min(cores, int(os.environ.get("TRS_CPU_BASELINE_THREADS", "16")))


def cpu_baseline(n_envs, h, w, budget_s=12.0):
    """Times the CPU oracle (kind "port": the reference has no dynamics/camera to run, and its Python cannot
    travel) on this host, on a bounded sample of the same workload.  Checker code is only TIMED here."""
    import numpy as np
    from triton_racer_sim_amd import _ffi
    from triton_racer_sim_amd.env import BatchedEnv
    lib = os.path.join(ROOT, "oracle", "libtrsim_oracle.so")
    if not os.path.exists(lib):
        return None
    api = _ffi.Api(ctypes.CDLL(lib), "trso_")
    cores = usable_cores()
    env = BatchedEnv(n_envs=n_envs, img_h=h, img_w=w, auto_reset=True, _api=api)
    set_threads = api.cdll.trso_set_threads
    set_threads.restype, set_threads.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    out = {}
    for label, threads, budget in (("all", cores, budget_s * 0.6), ("one", 1, budget_s * 0.25)):
        used = set_threads(env._h, threads)
        env.step_synthetic(1, 1)
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            env.step_synthetic(2, 1)
            steps += 2
        dt = time.perf_counter() - t0
        out[label] = (n_envs * steps / dt, used, steps)
    env.close()
    # what the reference itself executes per tick, minus the external simulator:
    # Car-style tick + dict pool + pure-Python L1 nearest-point (oracle/pyref.py restates track_data_process.py:89-104)
    py_rate = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyref
        py_rate = pyref.time_reference_loop(budget_s * 0.15)
    except Exception:
        py_rate = None
    v, used, steps = out["all"]
    return {
        "value": round(v, 1), "unit": "env-steps/s", "cores": used, "kind": "port",
        "sample": f"oracle/libtrsim_oracle.so (scalar C + OpenMP over envs), {n_envs} envs x {steps} steps of the same workload",
        "single_thread": round(out["one"][0], 1),
        "python_reference_loop_1env": None if py_rate is None else round(py_rate, 1),
        "reference_design_ceiling": "20 env-steps/s per car (car_templates/manage.py:38)",
    }


def pilot_weights(h, w):
    """Random-init Keras_2D_CNN weights (Glorot-uniform kernels, zero biases; keras_train.py:127-174) and the MACs of one forward pass."""
    import numpy as np
    rng = np.random.default_rng(0)
    spec = [(5, 2, 3, 24), (5, 2, 24, 32), (5, 2, 32, 64), (3, 1, 64, 64), (3, 1, 64, 64), (3, 1, 64, 128), (3, 1, 128, 128)]
    ws, ih, iw, macs = [], h, w, 0
    for k, s_, cin, cout in spec:
        ih, iw = (ih - k) // s_ + 1, (iw - k) // s_ + 1
        lim = (6.0 / (k * k * (cin + cout))) ** 0.5
        ws += [rng.uniform(-lim, lim, (k, k, cin, cout)).astype("float32"), np.zeros(cout, "float32")]
        macs += ih * iw * cout * k * k * cin
    dims = [ih * iw * 128, 100, 50, 25, 2]
    for a_, b_ in zip(dims[:-1], dims[1:]):
        lim = (6.0 / (a_ + b_)) ** 0.5
        ws += [rng.uniform(-lim, lim, (a_, b_)).astype("float32"), np.zeros(b_, "float32")]
        macs += a_ * b_
    return ws, macs


def config1_car_loop(device, ticks=1500):
    """BASELINE configs[0] on the GPU box: 1 env through `Car.tick` (the reference's drive loop, core/car.py:45-53) with a constant-control part
    and HipGymInterface (frame uint8[120,160,3] + 5 Python floats copied to the host every tick, one synchronisation).  ticks/s with the sleep
    disabled, one launch per tick and with the resident worker."""
    from triton_racer_sim_amd.components import HipGymInterface
    from triton_racer_sim_amd.core import Car, Component

    class Const(Component):
        def __init__(self):
            super().__init__(outputs=["mux/steering", "mux/throttle", "mux/breaking", "usr/reset"])

        def step(self, *a):
            return 0.05, 0.5, None, False

    out = {}
    for label, res in (("launch", False), ("resident", True)):
        car = Car(loop_hz=1e9, verbose=False)
        gym = HipGymInterface(gym_config={"scene_name": "generated_track", "hip_device": device, "hip_resident": res, "hip_resident_idle_us": 100000})
        car.addComponent(Const())
        car.addComponent(gym)
        for _ in range(200):
            car.tick()
        t0 = time.perf_counter()
        for _ in range(ticks):
            car.tick()
        dt = time.perf_counter() - t0
        img, x = car.pool.get_value("cam/img"), car.pool.get_value("gym/x")
        assert img.shape == (120, 160, 3) and img.dtype.name == "uint8" and type(x) is float
        out[label] = ticks / dt
        car.stop()
    return {"ticks_per_s_launch": round(out["launch"], 1), "ticks_per_s_resident": round(out["resident"], 1), "ticks": ticks,
            "us_per_tick_launch": round(1e6 / out["launch"], 2), "us_per_tick_resident": round(1e6 / out["resident"], 2),
            "pcie_inclusive_MBps_resident": round(out["resident"] * 57600 / 1e6, 1),
            "note": "BASELINE configs[0]: 1 env through Car.tick (core/car.py:45-53) = constant-control part + HipGymInterface.step (trs_step_host + trs_fetch_outputs: the frame "
                    "and 5 Python floats on the host every tick), sleep disabled; host wall clock; latency-bound (FFI + launch or post + one synchronisation), never part of `value`; "
                    "the reference paces this loop at 20 ticks/s (car_templates/manage.py:38)"}


def physics_256(device, steps=4000):
    """BASELINE configs[1]: 256 envs, physics only (trs_physics_kernel, 88 algorithmic bytes per env-step: latency-bound, the HBM roofline means nothing
    here — reported as us per step).  One step per launch, 16 steps per launch, and the consumer-paced tick (trs_step once per tick with device controls)
    in launch mode and posted to the resident physics worker."""
    import torch
    from triton_racer_sim_amd.env import BatchedEnv
    n = 256
    env = BatchedEnv(n_envs=n, render=False, auto_reset=True, device=device)
    B = algorithmic_bytes(0, 0, False)
    out = {"envs": n, "steps": steps, "bytes_per_env_step": B}
    env.step_synthetic(4000, 16)
    env.sync()
    for label, spl in (("1", 1), ("16", 16)):
        env.event_record(0)
        env.step_synthetic(steps, spl)
        env.event_record(1)
        ms = env.event_elapsed_ms(0, 1)
        out[f"us_per_step_spl{label}"] = round(ms * 1e3 / steps, 3)
        out[f"us_per_launch_spl{label}"] = round(ms * 1e3 / (steps / spl), 3)
        out[f"env_steps_per_s_spl{label}"] = round(n * steps / (ms * 1e-3), 1)
        out[f"GBps_spl{label}"] = round(B * n * steps / (ms * 1e-3) / 1e9, 2)
    st = (torch.rand(n, device="cuda") * 2 - 1) * 0.3
    th = torch.rand(n, device="cuda") * 0.6 + 0.2
    torch.cuda.synchronize()
    for label, res in (("launch", False), ("resident", True)):
        try:
            env.set_step_mode(res, 100000)
        except Exception as exc:
            out[f"tick_{label}_error"] = str(exc)
            continue
        for _ in range(200):
            env.step_device(st.data_ptr(), th.data_ptr())
        env.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            env.step_device(st.data_ptr(), th.data_ptr())
        env.sync()
        dt = time.perf_counter() - t0
        out[f"tick_{label}_us"] = round(dt * 1e6 / steps, 3)
        out[f"tick_{label}_env_steps_per_s"] = round(n * steps / dt, 1)
        lock = 500
        t0 = time.perf_counter()
        for _ in range(lock):
            env.step_device_wait(st.data_ptr(), th.data_ptr())
        out[f"tick_{label}_lock_step_us"] = round((time.perf_counter() - t0) * 1e6 / lock, 3)
    env.close()
    out["note"] = ("BASELINE configs[1]: 256 envs, bicycle physics + binary64 nearest point, no camera; spl = steps per launch of trs_physics_kernel with the in-kernel control "
                   "generator (device time by HIP events); tick_* = trs_step once per tick with device-resident controls (core/car.py:45-53), host wall clock: one launch per tick, "
                   "or posted to the resident physics worker (trs_set_step_mode); lock_step = trs_step_wait (post, wait for the telemetry)")
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def image_path_leg(device):
    """ImgPreprocessing on the device (components/img_preprocessing.py:37-102) on the env's own frames: per 1024 frames of
    120x160 and per 256 of 240x320 — trim, trim + dynamic brightness, trim + HSV masks, + Canny: us per batch, GB/s of algorithmic traffic (2 x H x W x 3 B
    per frame: read once + written once) and the fraction of the 8 TB/s HBM peak.  Fixed lengths, clocks pre-warmed (PREWARM_S), HIP events on the env's stream."""
    from triton_racer_sim_amd.env import BatchedEnv
    cfgs = (("trim", {"preprocessing_contrast_enhancement_ratio": 1.2}),
            ("trim_dynamic_brightness", {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True}),
            ("trim_hsv_masks", {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_color_filter_enabled": True}),
            ("trim_dynamic_masks_canny", {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True,
                                          "preprocessing_color_filter_enabled": True, "preprocessing_edge_detection_enabled": True}))
    out = {}
    for n, h, w in ((1024, 120, 160), (256, 240, 320)):
        env = BatchedEnv(n_envs=n, auto_reset=True, img_h=h, img_w=w, device=device)
        env.step_synthetic(20, 1)
        leg = {}
        for name, cfg in cfgs:
            pc = env.pre_config(cfg)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < PREWARM_S:
                for _ in range(20):
                    env.preprocess_latest(pc)
                env.sync()
            reps = 100
            env.event_record(0)
            for _ in range(reps):
                env.preprocess_latest(pc)
            env.event_record(1)
            env.sync()
            us = env.event_elapsed_ms(0, 1) * 1e3 / reps
            gbs = 2.0 * h * w * 3 * n / us / 1e3
            leg[name] = {"us_per_batch": round(us, 2), "GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
        out[f"{n}x{h}x{w}"] = leg
        env.close()
    out["note"] = ("trs_preprocess on device-resident frames (components/img_preprocessing.py:37-102), one batch per call; algorithmic traffic = one read + one write of every "
                   "frame byte; device time by HIP events, 100 calls after an 80 ms pre-warm per configuration")
    return out


def filtered_step_leg(device):
    """cam/processed_img straight from the rasteriser (trs_set_frame_filter): the resident step at 1024 envs x 120x160 with raw frames, with the static
    filter (trim + HSV masks: the palette is filtered on the host, zero extra traffic) and with dynamic brightness (per-env palettes inside the worker kernel):
    us per step, every step posted on its own; host wall clock between completion flags, worker resident (the definition of the main line's `value`)."""
    from triton_racer_sim_amd.env import BatchedEnv
    static = {"preprocessing_color_filter_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2}
    dyn = dict(static, preprocessing_dynamic_brightness_enabled=True)
    n, steps = 1024, 2000
    env = BatchedEnv(n_envs=n, auto_reset=True, device=device)
    env.set_step_mode(True, 100000)
    out = {}
    for name, flt in (("raw_frames", None), ("static_palette_filter", static), ("dynamic_brightness", dyn)):
        if flt is None:
            env.set_frame_filter(enabled=False)
        else:
            env.set_frame_filter(flt)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < PREWARM_S:
            env.step_synthetic(400, 1)
            env.sync()
        t1 = time.perf_counter()
        env.step_synthetic(steps, 1)
        env.sync()
        wall = time.perf_counter() - t1
        us = wall * 1e6 / steps
        out[name] = {"us_per_step": round(us, 3), "env_steps_per_s": round(n * steps / wall, 1),
                     "frac_of_hbm_peak": round(algorithmic_bytes(120, 160, True) * n * steps / wall / 1e9 / HBM_PEAK_GBS, 5)}
    env.close()
    out["note"] = ("1024 envs x 120x160, resident worker, every step posted on its own, 2000 steps after an 80 ms pre-warm each; the filter of "
                   "components/img_preprocessing.py:81-102 (+ HSV masks :65-71) applied by the rasteriser itself: no second pass over the frames")
    return out


def hilly_track_leg(device):
    """A track with elevation (include/trsim_spec.h, round 5): the reference's mountain track (car_templates/track_data/mountain_track.json), 1024 envs x 120x160 —
    a frame's view pitch follows the slope ahead, the row tables and the depth frame are evaluated per env inside the kernel.  Resident worker, every step posted on
    its own; host wall clock between completion flags, 2000 steps after an 80 ms pre-warm; RGB and RGB + depth."""
    from triton_racer_sim_amd.env import BatchedEnv
    n, steps = 1024, 2000
    out = {}
    for name, depth in (("rgb", False), ("rgb_depth", True)):
        env = BatchedEnv(n_envs=n, auto_reset=True, track="mountain_track", depth=depth, device=device)
        env.set_step_mode(True, 100000)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < PREWARM_S:
            env.step_synthetic(400, 1)
            env.sync()
        t1 = time.perf_counter()
        env.step_synthetic(steps, 1)
        env.sync()
        wall = time.perf_counter() - t1
        B = algorithmic_bytes(120, 160, True, depth)
        out[name] = {"us_per_step": round(wall * 1e6 / steps, 3), "env_steps_per_s": round(n * steps / wall, 1), "frac_of_hbm_peak": round(B * n * steps / wall / 1e9 / HBM_PEAK_GBS, 5)}
        env.close()
    out["note"] = ("mountain_track (2,664 points, 4.2 units of height): per-env view pitch from the slope ahead (up to 4.1 deg), row tables / fogged palette / row depth "
                   "evaluated per env and frame by the raster team (HILLS instantiation of trs_worker_kernel); the flat generated track is the main line")
    return out


class phase:
    """``with phase("name", seconds):`` — a phase of a multi-rank run that can only stall on ANOTHER rank (rendezvous, barrier, collective).  When it
    takes longer than ``seconds`` the rank says which phase it is stuck in and exits with code 3 (the launcher then stops the other ranks): no silent
    wait for torch.distributed's own timeout.  At world size 1 nothing can stall on a peer; the guard is the same code all the same."""
    def __init__(self, name, seconds, rank=None):
        self.name, self.seconds = name, float(seconds)
        self.rank = os.environ.get("RANK", "0") if rank is None else rank

    def _expired(self):
        sys.stderr.write(f"bench.py: rank {self.rank} stalled in phase '{self.name}' for more than {self.seconds:.0f} s; giving up (exit 3)\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        import threading
        self.t = threading.Timer(self.seconds, self._expired)
        self.t.daemon = True
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


PHASE_S = float(os.environ.get("TRS_BENCH_PHASE_S", "120"))     # how long a barrier / collective may take before the rank gives up
RENDEZVOUS_S = 2.0 * PHASE_S                                     # ... and the rendezvous of the process groups: the ranks of a fresh box finish `import torch` up to a minute or two apart


def launch_ranks(n_ranks, argv):
    """Parent of a self-launched run: one child process per rank (= per GPU), each running THIS file with the environment
    torch.distributed.run would give it.  The parent imports neither torch nor the HIP library and never touches a GPU (a process that
    has initialised the GPU must not spawn / exec its replacement on this pool); it relays the children's stdout lines that are
    JSON objects (rank 0's line) to its own stdout, everything else to stderr, and returns the first non-zero exit code (the other ranks
    are then terminated by PID) or 0.  TRS_BENCH_CHILD (tests: a stub) replaces the script the ranks run."""
    import signal
    import subprocess
    import threading
    child = os.environ.get("TRS_BENCH_CHILD", os.path.abspath(__file__))
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TRS_BENCH_LAUNCHER="self")
        # dmabuf IPC.  Source: this pool's environment notes (the image exports it; "the host driver only supports dmabuf IPC, and without it RCCL /
        # device-tensor sharing across processes fails with hipIpcGetMemHandle: invalid argument") — NOT a run of ours: no N > 1 RCCL run has been
        # possible on the one-GPU boxes this repo is built on.  setdefault: an environment that sets it otherwise is left alone.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, child] + list(argv), env=env, stdout=subprocess.PIPE, text=True, bufsize=1))

    def relay(proc):
        for line in proc.stdout:
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()

    threads = [threading.Thread(target=relay, args=(p_,), daemon=True) for p_ in procs]
    for t in threads:
        t.start()
    rc = 0
    try:
        alive = set(range(n_ranks))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    sys.stderr.write(f"bench.py: rank {r} exited with code {code}; stopping the other ranks\n")
                    for o in alive:
                        procs[o].terminate()
            time.sleep(0.02)
    except KeyboardInterrupt:
        for p_ in procs:
            if p_.poll() is None:
                p_.send_signal(signal.SIGINT)
        rc = 130
    for p_ in procs:
        try:
            p_.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p_.kill()
    for t in threads:
        t.join(timeout=5)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="envs per shard (default: 1024 at N = 1 = configs[2]; 4096 / N at N > 1 = configs[3])")
    ap.add_argument("--total-envs", type=int, default=None, help="envs over all shards (alternative to --envs-per-gpu)")
    ap.add_argument("--img-h", type=int, default=120)
    ap.add_argument("--img-w", type=int, default=160)
    ap.add_argument("--steps-per-launch", type=int, default=1)
    ap.add_argument("--step-mode", choices=("resident", "launch"), default="resident",
                    help="resident (default): trs_set_step_mode(TRS_STEP_RESIDENT) - a worker kernel stays on the GPU and every step is POSTED on its own "
                         "(consumer-paced, no launch per step); launch: one kernel launch per step (--steps-per-launch K: K steps per launch)")
    ap.add_argument("--resident", action="store_true", help="same as --step-mode resident")
    ap.add_argument("--no-render", action="store_true", help="physics only (BASELINE configs[1] shape)")
    ap.add_argument("--pilot", action="store_true", help="closed loop with cnn_2d_speed_control inference on the device frame each step (BASELINE configs[4] shape)")
    ap.add_argument("--depth", action="store_true", help="also write the binary32 depth frame (BASELINE configs[4] frame format)")
    ap.add_argument("--pilot-tuning", default="", help="with --pilot: kernel choices for measurements, e.g. dense=2,ksplit=32 (fields of trs_pilot_tuning, include/trsim.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the informational 8-steps-per-launch leg (profiling runs: only the timed kernel in the trace)")
    ap.add_argument("--profile-mode", action="store_true", help="for rocprofv3 runs (scripts/profile.sh): the clock pre-warm goes by launches (trs_step_kernel), and a resident worker is asked to leave after the "
                                                                  "warm-up, so that every trs_worker_kernel dispatch of the trace serves exactly --steps (or --warmup) steps")
    ap.add_argument("--force-dist", action="store_true", help="initialise the nccl process group even at world size 1 (path rehearsal)")
    ap.add_argument("--spawn", action="store_true", help="go through the self-launch path (launch_ranks) even at --gpus 1: the rank runs as a child process of a parent that never touches the GPU")
    ap.add_argument("--share-gpu", action="store_true", help="REHEARSAL of the N > 1 code path on a box with fewer GPUs than ranks: every rank uses GPU 0 and the process group is gloo "
                                                             "(RCCL refuses two ranks on one device); the line says so and its value means nothing")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # self-launch: this process becomes the parent of the rank processes; it has imported neither torch nor the HIP library
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launcher = os.environ.get("TRS_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else ("self" if "WORLD_SIZE" not in os.environ else "external"))
    if world != args.gpus:
        args.gpus = world

    import torch
    from triton_racer_sim_amd.shard import ShardedEnvs, max_over_ranks, timed_steps

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    if args.share_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} but this node shows {torch.cuda.device_count()} GPU(s); --gpus N needs N GPUs (--share-gpu rehearses the code path on fewer)")
    torch.cuda.set_device(local_rank)
    dist = None
    host_group = None
    if world > 1 or args.force_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_gpu:
            with phase("init_process_group(gloo)", RENDEZVOUS_S):
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=RENDEZVOUS_S + 30))
            host_group = dist.group.WORLD
        else:
            # device collectives (the one all-gather): nccl = RCCL over xGMI.  Barriers and the MAX over ranks: a gloo group, i.e. host
            # sockets — a barrier must not need CU resources while every CU holds a resident worker
            # No device_id: the communicator is then created by the warm-up all-gather below (torch.cuda.set_device above names the GPU).  Measured at
            # world size 1 (scripts/r05_dist_probe.py, profiles/r05_dist_probe.txt): with the EAGER communicator of device_id= the first posted steps
            # after every stream synchronisation ran at 12.0-13.0 us instead of 10.0 and the next 2000 at 10.7 instead of 9.6 (-20 % on the driver's
            # --steps 20 line, -10 % at --steps 2000); created lazily by its first collective the same communicator costs nothing measurable.
            with phase("init_process_group(nccl)", RENDEZVOUS_S):
                dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=RENDEZVOUS_S + 30))
            with phase("new_group(gloo)", RENDEZVOUS_S):
                host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=RENDEZVOUS_S + 30))
        if dist.get_world_size() != args.gpus:
            sys.exit(f"rank {rank}: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    if args.envs_per_gpu is not None and args.total_envs is not None:
        sys.exit("give --envs-per-gpu or --total-envs, not both")
    if args.envs_per_gpu is not None:
        n, picked = args.envs_per_gpu, "override (--envs-per-gpu)"
    elif args.total_envs is not None:
        if args.total_envs % world:
            sys.exit("--total-envs must be divisible by the number of GPUs")
        n, picked = args.total_envs // world, "override (--total-envs)"
    elif world == 1:
        n, picked = 1024, "BASELINE configs[2]"
    else:
        if 4096 % world:
            sys.exit("BASELINE configs[3] (4096 envs) needs a GPU count that divides 4096; use --envs-per-gpu")
        n, picked = 4096 // world, "BASELINE configs[3]"
    fixed_total = args.envs_per_gpu is None and world > 1          # total fixed as N grows -> strong scaling
    render = not args.no_render
    shard = ShardedEnvs(n * world, rank, world, device=local_rank, img_h=args.img_h, img_w=args.img_w, render=render, auto_reset=True,
                        depth=args.depth)
    env = shard.env
    spl = max(1, args.steps_per_launch)
    if args.pilot:
        ws, macs = pilot_weights(args.img_h, args.img_w)
        if args.pilot_tuning:
            env.pilot_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.pilot_tuning.split(","))})
        env.pilot_load(ws)
        pilot_flops = 2.0 * macs
        run = lambda k_: env.step_pilot(k_)
    else:
        run = lambda k_: env.step_synthetic(k_, spl)
    # the pilot loop goes through launches (the pilot's kernels need the CUs' LDS); physics-only envs have their own resident worker since round 4
    resident = bool((args.resident or args.step_mode == "resident") and not args.pilot and spl == 1)
    # (--share-gpu: several ranks on ONE GPU.  Their resident workers cannot be on the GPU together; since round 5 the library arbitrates — a launch that does
    # not get the whole GPU is called off and that rank steps by launches until its next try, workers leave every 50 ms — so the rehearsal runs the
    # resident path like the real run; round 4 had to rehearse with launches: "resident worker gave up", gpurun_out/r04_full_2.log.)
    if resident:
        # idle_us: the worker leaves by itself after this long without a post.  The library's default (2 ms) suits an interactive loop; a
        # benchmark whose host thread can be descheduled for milliseconds (a tracer attached, the GIL) would see its worker leave and be
        # relaunched mid-run — every explicit boundary below asks it to leave (quiesce) instead.
        env.set_step_mode(True, 100000)

    # The timed region is triton_racer_sim_amd.shard.timed_steps: [env.sync + stream synchronisation + barrier] t0 - K steps - env.sync t1
    # [stream synchronisation + barrier], MAX of (t1 - t0) over ranks.  env.sync() = every step handed to the env so far is complete in memory
    # (resident mode: the completion flags the worker writes; launch mode: the env's stream has drained).  While a resident worker is on the GPU
    # a DEVICE-wide synchronisation would wait for that kernel to end (it leaves idle_us after the last post), so resident mode synchronises
    # torch's stream instead, and the barrier between ranks is a HOST barrier (gloo): the worker stays across the warm-up -> timed boundary on
    # every rank (steady state: what a consumer that keeps posting sees).
    def host_barrier():
        if dist is not None:
            with phase("host barrier (gloo)", PHASE_S):
                dist.barrier(group=host_group)

    stream_sync = (lambda: torch.cuda.current_stream().synchronize()) if resident else torch.cuda.synchronize

    if dist is not None and not args.share_gpu:   # warm the communicator (RCCL: connection set-up over xGMI) before any worker is resident
        with phase("warm-up all-gather (RCCL connection set-up)", PHASE_S):
            warm = torch.zeros(n * world, device="cuda")
            dist.all_gather_into_tensor(warm, torch.full((n,), float(rank + 1), device="cuda"))
            torch.cuda.synchronize()
        seen = sorted(set(warm.cpu().tolist()))
        if seen != [float(r + 1) for r in range(world)]:
            sys.exit(f"rank {rank}: the warm-up all-gather returned contributions {seen[:16]}, expected one per rank 1..{world}")
    # An idle MI355X takes ~25-40 ms of work to raise its clocks: the first 2,000 steps of a fresh process run 12-14 % slower than the
    # steady state (profiles/r03_steady_state.txt: 10.85 us per step, then 9.46-9.58).  The metric is a steady-state rate, and
    # the driver's --warmup 5 is 50 us, so the clocks are brought up first — with untimed steps of the same workload, for PREWARM_S
    # seconds — and the W warm-up steps follow.  `config.prewarm_s` says so in the line.
    t_pw = time.perf_counter()
    if args.profile_mode and resident:
        # under the profiler the pre-warm steps go by LAUNCHES (trs_step_kernel: its own row of the stats table), so that every trs_worker_kernel
        # dispatch in the trace is one of the three equal ones below, and all three run at loaded clocks like the line's own (until round 5 the
        # profiled command had no pre-warm at all: its first dispatch ran 15-20 % slow and pulled the table's average off the line's figure)
        env.set_step_mode(False)
        while time.perf_counter() - t_pw < PREWARM_S:
            run(400)
            env.sync()
        env.set_step_mode(True)
    while not args.profile_mode and time.perf_counter() - t_pw < PREWARM_S:
        run(50 if args.pilot else 400)
        env.sync()
    run(max(args.warmup, 1))
    if args.profile_mode and resident:
        env.quiesce()

    def timed_run(k_):
        if not resident:
            env.event_record(0)
        run(k_)
        if not resident:
            env.event_record(1)

    wall = timed_steps(env, timed_run, args.steps, host_barrier, stream_sync)
    # the job's ONE exchange: an all-gather of the episode returns (4 B per env) per reporting interval — configs[3]: one per 1000 steps —
    # so it closes the job here, timed on its own, not inside the K timed steps (a resident worker is asked to leave first, outside the timing)
    gathered, allgather_s = None, None
    if dist is not None:
        with phase("all-gather of ep_return", PHASE_S):
            gathered, allgather_s = shard.allgather_timed("ep_return", host_barrier, torch.cuda.synchronize)
        with phase("MAX over ranks (gloo)", PHASE_S):
            allgather_s = max_over_ranks(allgather_s, host_group)
        if gathered.numel() != n * world:
            sys.exit(f"rank {rank}: the all-gather returned {gathered.numel()} values, expected {n} x {world}")
        mine = torch.as_tensor(env.fetch("ep_return"))
        if not torch.equal(gathered.cpu()[shard.base:shard.base + n], mine):
            sys.exit(f"rank {rank}: the gathered returns do not hold this shard's values at [{shard.base}, {shard.base + n})")
    if resident:
        # the dominant kernel's launch duration by HIP events on its own stream: an event on that stream makes the worker leave, so this
        # is a SECOND pass over the same K steps, bracketed by events = exactly one trs_worker_kernel launch (start-up, K steps, exit)
        env.quiesce()
        torch.cuda.synchronize()
        env.event_record(0)
        run(args.steps)
        env.event_record(1)
        env.sync()
    kernel_ms = env.event_elapsed_ms(0, 1)
    if dist is not None:
        with phase("MAX of the event time over ranks (gloo)", PHASE_S):
            kernel_ms = max_over_ranks(kernel_ms, host_group)        # (rank 0's own event time x world was not the job's)
    mode_after = env.step_mode()

    # informational only (never part of `value`): the same workload with 8 steps per launch, timed separately
    also = None
    if render and not args.pilot and spl == 1 and world == 1 and not args.no_also:
        env.set_step_mode(False)
        env.sync()
        # ... one kernel launch per step, controls from the in-kernel generator (round 1's headline path)
        env.event_record(2)
        env.step_synthetic(args.steps, 1)
        env.event_record(3)
        ms1 = env.event_elapsed_ms(2, 3)
        env.event_record(2)
        env.step_synthetic(args.steps, 8)
        env.event_record(3)
        ms8 = env.event_elapsed_ms(2, 3)
        # ... and with externally produced controls: a device-resident action sequence, one control set per step (trs_step_sequence)
        seq_steer = (torch.rand((args.steps, n), device="cuda") * 2 - 1) * 0.3
        seq_thr = torch.rand((args.steps, n), device="cuda") * 0.6 + 0.2
        torch.cuda.synchronize()
        env.event_record(4)
        env.step_sequence_device(seq_steer.data_ptr(), seq_thr.data_ptr(), n_steps=args.steps, steps_per_launch=8)
        env.event_record(5)
        ms_seq = env.event_elapsed_ms(4, 5)
        # ... and the consumer-paced call: ONE trs_step per tick with that tick's (device-resident) controls, n_steps = 1 per call,
        # the way Car.start drives GymInterface.step (core/car.py:45-53)
        one_steer = (torch.rand(n, device="cuda") * 2 - 1) * 0.3
        one_thr = torch.rand(n, device="cuda") * 0.6 + 0.2
        p_steer, p_thr = one_steer.data_ptr(), one_thr.data_ptr()          # the consumer holds its control tensors: their addresses are taken once
        torch.cuda.synchronize()
        for _ in range(min(50, args.steps)):
            env.step_device(p_steer, p_thr)
        env.sync()
        t1 = time.perf_counter()
        env.event_record(6)
        for _ in range(args.steps):
            env.step_device(p_steer, p_thr)
        env.event_record(7)
        ms_one = env.event_elapsed_ms(6, 7)
        wall_one = time.perf_counter() - t1
        lock_steps = min(args.steps, 500)
        env.sync()
        t4 = time.perf_counter()
        for _ in range(lock_steps):                               # lock-step with launches: step, wait for the frame (stream synchronisation), step
            env.step_device(p_steer, p_thr)
            env.sync()
        wall_lock_launch = time.perf_counter() - t4
        # ... the same loop in resident mode: every call only posts its control pointers to the worker kernel
        env.set_step_mode(True)
        for _ in range(min(50, args.steps)):
            env.step_device(p_steer, p_thr)
        env.sync()
        t2 = time.perf_counter()
        env.event_record(6)                                       # (asks the worker to leave: the timed span holds a whole worker launch)
        for _ in range(args.steps):
            env.step_device(p_steer, p_thr)
        env.event_record(7)
        ms_res = env.event_elapsed_ms(6, 7)
        wall_res = time.perf_counter() - t2
        # ... and lock-step: the consumer waits for every frame before it posts the next step (PCIe round trips included)
        env.step_device(p_steer, p_thr)
        env.sync()
        t3 = time.perf_counter()
        for _ in range(lock_steps):
            env.step_device(p_steer, p_thr)
            env.sync()
        wall_lock = time.perf_counter() - t3
        env.step_device_wait(p_steer, p_thr)
        t3 = time.perf_counter()
        for _ in range(lock_steps):                                       # ... the same tick through trs_step_wait: one FFI crossing
            env.step_device_wait(p_steer, p_thr)
        wall_lock1 = time.perf_counter() - t3
        # ... and SURVEY 8(f-1): cnn_2d_speed_control inference on the device frame every step, actions fed back (closed loop, launch mode: the
        # pilot's kernels need the CUs' LDS, so the resident worker is asked to leave first)
        pilot_leg = None
        try:
            env.set_step_mode(False)
            env.sync()
            pws, pmacs = pilot_weights(args.img_h, args.img_w)
            env.pilot_load(pws)
            psteps = 200                                              # a fixed 40 ms whatever --steps says: 20 steps (4 ms) would be timed at idle clocks
            t_pw = time.perf_counter()
            while time.perf_counter() - t_pw < PREWARM_S:              # the same pre-warm as the main line (100 steps were 20 ms of the 25-40 the clocks need)
                env.step_pilot(50)
                env.sync()
            env.event_record(2)
            env.step_pilot(psteps)
            env.event_record(3)
            ms_p = env.event_elapsed_ms(2, 3)
            ptf = 2.0 * pmacs * n * psteps / (ms_p * 1e-3) / 1e12
            pilot_leg = {"env_steps_per_s": round(n * psteps / (ms_p * 1e-3), 1), "us_per_step": round(ms_p * 1e3 / psteps, 3), "tflops_fp16": round(ptf, 1),
                         "frac_of_mfma_peak": round(ptf / 2500.0, 5), "frac_of_measured_mfma_ceiling": round(ptf / MEASURED_MFMA_TFLOPS, 5),
                         "note": "trs_step_pilot: env step + cnn_2d_speed_control forward (fp16 MFMA convolutions, fp32 accumulate, fp32 tail) + KerasPilot.step per step, random-init weights; "
                                 "device time by HIP events; `python bench.py --pilot` reports this loop as the main line"}
        except Exception as exc:                                    # the leg is informational: never lose the main line to it
            pilot_leg = {"error": str(exc)}
        # ... and BASELINE configs[4]'s per-GPU share: 512 envs x 240x320 RGB + fp32 depth with the same pilot in the loop (its own handle)
        pilot5_leg = None
        if (args.img_h, args.img_w) == (120, 160) and not args.depth:
            try:
                from triton_racer_sim_amd.env import BatchedEnv
                env5 = BatchedEnv(n_envs=512, img_h=240, img_w=320, depth=True, auto_reset=True, device=local_rank)
                w5, macs5 = pilot_weights(240, 320)
                env5.pilot_load(w5)
                env5.step_synthetic(2, 1)
                t_pw = time.perf_counter()
                while time.perf_counter() - t_pw < PREWARM_S:               # clocks up again after the host-side weight generation
                    env5.step_pilot(20)
                    env5.sync()
                env5.event_record(0)
                env5.step_pilot(60)
                env5.event_record(1)
                ms5 = env5.event_elapsed_ms(0, 1)
                tf5 = 2.0 * macs5 * 512 * 60 / (ms5 * 1e-3) / 1e12
                pilot5_leg = {"env_steps_per_s": round(512 * 60 / (ms5 * 1e-3), 1), "us_per_step": round(ms5 * 1e3 / 60, 3), "tflops_fp16": round(tf5, 1),
                              "frac_of_mfma_peak": round(tf5 / 2500.0, 5), "frac_of_measured_mfma_ceiling": round(tf5 / MEASURED_MFMA_TFLOPS, 5),
                              "note": "512 envs x 240x320 RGB + fp32 depth + cnn_2d_speed_control in the loop = one GPU's share of BASELINE configs[4]; device time by HIP events"}
                env5.close()
            except Exception as exc:
                pilot5_leg = {"error": str(exc)}
        # ... and BASELINE configs[3]'s per-GPU share: the LAST shard of the 4096-env job (512 envs, global ids 3584..4095), every step
        # posted on its own to its resident worker — the only driver-side evidence for the 8-GPU target while no 8-GPU node runs this
        shard_leg = None
        if (args.img_h, args.img_w) == (120, 160) and not args.depth:
            try:
                from triton_racer_sim_amd.env import BatchedEnv
                env3 = BatchedEnv(n_envs=512, env_id_base=3584, img_h=120, img_w=160, auto_reset=True, device=local_rank)
                env3.set_step_mode(True, 100000)
                s3 = max(args.steps, 2000)
                env3.step_synthetic(8000, 1)                                # 40 ms: clocks up
                env3.sync()
                t5 = time.perf_counter()
                env3.step_synthetic(s3, 1)
                env3.sync()                                             # completion flags: the worker stays resident (steady state)
                wall3 = time.perf_counter() - t5
                env3.event_record(0)                                    # (the worker leaves) ... and one worker launch bracketed by HIP events
                env3.step_synthetic(s3, 1)
                env3.event_record(1)
                ms3 = env3.event_elapsed_ms(0, 1)
                B3 = algorithmic_bytes(120, 160, True)
                shard_leg = {"env_steps_per_s": round(512 * s3 / wall3, 1), "us_per_step": round(wall3 * 1e6 / s3, 3),
                             "frac_of_hbm_peak": round(B3 * 512 * s3 / wall3 / 1e9 / HBM_PEAK_GBS, 5),
                             "frac_of_hbm_peak_by_hip_events": round(B3 * 512 * s3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "x8_gpus_env_steps_per_s": round(8 * 512 * s3 / wall3, 1),
                             "note": "512 envs at env_id_base 3584 = one GPU's share of BASELINE configs[3] (4096 envs over 8 GPUs), resident worker, every step posted on its own; "
                                     "host wall clock between completion flags (steady state), and one worker launch by HIP events; x8 = what eight independent shards add up to "
                                     "(no data-path collective; the one all-gather of 4 B per env is outside the step loop)"}
                env3.close()
            except Exception as exc:
                shard_leg = {"error": str(exc)}
        # ... BASELINE configs[0]: ONE env through the Component / DataPool / Car loop (core/car.py:45-53), frame + 5 Python floats to the host every
        # tick (HipGymInterface.step = trs_step_host + trs_fetch_outputs) — the reference's own shape, sleep disabled; fixed length
        car_leg = None
        try:
            car_leg = config1_car_loop(local_rank)
        except Exception as exc:
            car_leg = {"error": str(exc)}
        # ... BASELINE configs[1]: 256 envs, bicycle physics only (no camera), one GPU; fixed lengths (a consumer-paced tick and multi-step launches)
        phys_leg = None
        try:
            phys_leg = physics_256(local_rank)
        except Exception as exc:
            phys_leg = {"error": str(exc)}
        # ... the image path (ImgPreprocessing on the device) and the step with the filter behind the rasteriser: fixed shapes and lengths
        image_leg = filtered_leg = None
        if (args.img_h, args.img_w) == (120, 160) and not args.depth:
            try:
                image_leg = image_path_leg(local_rank)
            except Exception as exc:
                image_leg = {"error": str(exc)}
            try:
                filtered_leg = filtered_step_leg(local_rank)
            except Exception as exc:
                filtered_leg = {"error": str(exc)}
        hilly_leg = None
        if (args.img_h, args.img_w) == (120, 160) and not args.depth:
            try:
                hilly_leg = hilly_track_leg(local_rank)
            except Exception as exc:
                hilly_leg = {"error": str(exc)}
        env.set_step_mode(resident, 100000)
        Bx = algorithmic_bytes(args.img_h, args.img_w, render, args.depth)
        rate = lambda ms: round(n * args.steps / (ms * 1e-3), 1)
        frac = lambda ms: round(Bx * n * args.steps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        also = {"launch_per_step": {"env_steps_per_s": rate(ms1), "frac_of_hbm_peak": frac(ms1), "us_per_step": round(ms1 * 1e3 / args.steps, 3),
                                    "note": "trs_step_synthetic(steps, 1) in launch mode: one trs_step_kernel launch per step, physics of step t overlapping the raster of t - 1 across launches (open loop); device time by HIP events"},
                "single_step_call": {"env_steps_per_s": rate(ms_one), "frac_of_hbm_peak": frac(ms_one), "us_per_call": round(ms_one * 1e3 / args.steps, 3),
                                     "host_wall_us_per_call": round(wall_one * 1e6 / args.steps, 3),
                                     "lock_step_us_per_call": round(wall_lock_launch * 1e6 / lock_steps, 3),
                                     "note": "trs_step(device controls, n_steps = 1) called once per step: the consumer-paced path (one launch per call, raster waits for that step's physics); device time by HIP events"},
                "resident_single_step_call": {"env_steps_per_s": rate(ms_res), "frac_of_hbm_peak": frac(ms_res), "us_per_call": round(ms_res * 1e3 / args.steps, 3),
                                              "host_wall_us_per_call": round(wall_res * 1e6 / args.steps, 3),
                                              "lock_step_us_per_call": round(wall_lock1 * 1e6 / lock_steps, 3), "lock_step_two_calls_us": round(wall_lock * 1e6 / lock_steps, 3),
                                              "note": "trs_set_step_mode(TRS_STEP_RESIDENT): the same calls only POST to the resident worker kernel (no launch, no kernel boundary, tables staged once); device time = the worker launch that served all the calls, by HIP events; lock_step = one trs_step_wait per tick: post, wait for the frame (host wall clock; lock_step_two_calls_us: trs_step + trs_sync)"},
                "sequence_steps_per_launch_8": {"env_steps_per_s": rate(ms_seq), "frac_of_hbm_peak": frac(ms_seq),
                                                "note": "trs_step_sequence: a different device-resident control set per step (open-loop action sequences), 8 steps per launch"},
                "steps_per_launch_8": {"env_steps_per_s": rate(ms8), "frac_of_hbm_peak": frac(ms8),
                                       "note": "synthetic controls; physics team runs 8 steps ahead inside one launch (LDS hand-off); device time by HIP events"}}
        if pilot_leg:
            also["pilot_closed_loop"] = pilot_leg
        if pilot5_leg:
            also["pilot_closed_loop_512x240x320_depth"] = pilot5_leg
        if shard_leg:
            also["shard_512_of_4096"] = shard_leg
        if car_leg:
            also["config1_car_loop"] = car_leg
        if phys_leg:
            also["physics_256"] = phys_leg
        if image_leg:
            also["image_path"] = image_leg
        if filtered_leg:
            also["filtered_step"] = filtered_leg
        if hilly_leg:
            also["hilly_track"] = hilly_leg
    if dist is not None:
        wall = max_over_ranks(wall, host_group)

    if rank == 0:
        total_env_steps = n * world * args.steps
        value = total_env_steps / wall
        B = algorithmic_bytes(args.img_h, args.img_w, render, args.depth)
        # camera on: the library pipelines a call over ceil(steps/spl)+1 launches of trs_step_kernel (physics runs ahead of the
        # raster inside and across launches); physics only: steps/spl launches of trs_physics_kernel
        launches = ((args.steps + spl - 1) // spl + 1) if render else (args.steps + spl - 1) // spl
        if resident:
            launches = 1                                                # one trs_worker_kernel launch serves every posted step of the timed region
        avg_launch_s = kernel_ms * 1e-3 / launches
        per_launch = n * args.steps / launches                         # env-steps one launch completes
        achieved = B * per_launch / avg_launch_s / 1e9                 # GB/s of algorithmic bytes, dominant (only) kernel
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        traffic_key = f"{n}x{args.img_h}x{args.img_w}" + ("+depth" if args.depth else "") + (":resident" if resident else f":spl{spl}")
        if os.path.exists(pmc):
            try:
                # scripts/summarize_profile.py writes {"per_env_step": {"<envs>x<H>x<W>[+depth]:<resident|splK>": HBM bytes per env-step}} from the
                # WRITE_SIZE / FETCH_SIZE passes of scripts/profile.sh (unit + gfx950 corrections of MI355X_MICROARCH.md applied); per launch = x the
                # env-steps one launch completes, like `achieved`
                with open(pmc) as f:
                    per_step = json.load(f).get("per_env_step", {}).get(traffic_key)
                traffic = None if per_step is None else per_step * per_launch
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/s (whole node) at N envs x 120x160 RGB",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "strong" if fixed_total else "weak",
            "vs_baseline": None, "dtype": "f32 state / f64 nearest-point / u8 image", "data": "synthetic",
            "config": {
                "workload": f"{n} envs/GPU x {world} GPU, bicycle physics + L1 nearest point"
                            + (f" + {args.img_h}x{args.img_w} RGB pinhole rasteriser" if render else " (no camera)")
                            + (" + fp32 depth" if render and args.depth else "")
                            + f" = {picked}" + (f": {n * world} envs in total over {world} GPUs, one RCCL all-gather of ep_return" if world > 1 else "")
                            + ", generated_track 1185 pts, synthetic controls seed 0x5EED, auto-reset",
                "envs_total": n * world, "envs_per_gpu": n, "img_h": args.img_h, "img_w": args.img_w, "depth": bool(args.depth), "steps_per_launch": spl, "step_mode": "resident worker (posted steps)" if resident else "one launch per call", "prewarm_s": PREWARM_S if (resident or not args.profile_mode) else 0.0,
                "timing": ("host wall clock from the post of the first timed step to the completion flag of the last (worker resident across the warm-up -> timed boundary)"
                           + ("; per rank between two host (gloo) barriers, MAX over ranks" if world > 1 else "") + "; "
                           "roofline.achieved from a second pass of the same steps bracketed by HIP events = one whole worker launch") if resident else "host wall clock around the timed steps; HIP events on the env's stream for roofline.achieved", "sharding": f"{world} shard(s), no data-path collective; one all-gather of ep_return closes the job, timed separately (`allgather`)" if world > 1 else "1 shard",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_key": traffic_key,
                "frac_by_wall_clock": round(B * n * args.steps / wall / 1e9 / HBM_PEAK_GBS, 5),
                "kernel": ("trs_worker_kernel" if render else "trs_physics_worker_kernel") if resident else ("trs_step_kernel" if render else "trs_physics_kernel"), "avg_launch_us": round(avg_launch_s * 1e6, 3),
                "bytes_per_env_step": B, "env_steps_per_launch": round(per_launch, 2), "launches": launches,
            },
        }
        if also:
            line["also"] = also
        if args.pilot:
            # closed loop with the CNN: the convolutions dominate (MFMA-bound class); the device time of a step spans
            # the env launch, the conv / dense launches and the tail, so the figure below is a whole-step rate, a lower bound
            # on the conv kernels' own rate
            tf = pilot_flops * n * args.steps / (kernel_ms * 1e-3) / 1e12
            line["config"]["workload"] += " + cnn_2d_speed_control inference in the loop (random-init weights, closed loop)"
            line["dtype"] += " / fp16 MFMA convolutions, f32 accumulate"
            line["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 5),
                                "measured_ceiling": MEASURED_MFMA_TFLOPS, "frac_of_measured_ceiling": round(tf / MEASURED_MFMA_TFLOPS, 5),
                                "traffic": None, "kernel": "per step: trs_step_kernel + trs_conv12_band_kernel (conv1 + conv2 fused) + trs_conv_frame5_kernel (conv3) + trs_conv_chain_kernel (conv4..7 in one launch) + trs_pilot_dense_kernel (dense1) + trs_pilot_tail_kernel at 120x160; frames too large for LDS: trs_conv_frame5_kernel on row bands (conv3) and one trs_conv_frame_kernel launch per 3x3 layer",
                                "flops_per_frame": pilot_flops, "avg_step_us": round(kernel_ms * 1e3 / args.steps, 3),
                                "note": "whole closed-loop step by HIP events; per-layer times in profiles/r04_pilot_layers.txt"}
        line["config"]["launcher"] = launcher          # "self": python bench.py --gpus N started the ranks (launch_ranks); "torch.distributed.run"; "external"
        if resident:
            # the same K steps with the worker's start-up and exit inside the clock (round 2's definition of `value`): one whole
            # trs_worker_kernel launch by HIP events — what a consumer that posts K steps and then stops sees.  `value` is the steady-state rate.
            line["value_incl_worker_launch"] = round(n * world * args.steps / avg_launch_s, 1)
            line["roofline"]["covers"] = "one whole worker launch (start-up + K steps + exit) by HIP events; `value` and frac_by_wall_clock cover the K steps alone, worker resident"
        if gathered is not None:
            line["allgather"] = {"us": round(allgather_s * 1e6, 1), "ranks": dist.get_world_size(), "values": int(gathered.numel()), "backend": dist.get_backend(),
                                 "floats_per_rank": n, "returns_mean": round(float(gathered.float().mean().item()), 4),
                                 "note": "ONE all-gather of ep_return closes the job (configs[3]: one per 1000 steps), timed on its own after the K steps: "
                                         "[worker asked to leave, host barrier] t0 - all_gather_into_tensor - device synchronisation t1, MAX over ranks; never inside `value`"}
            line["config"]["allgather_returns_mean"] = line["allgather"]["returns_mean"]
        if resident:
            line["config"]["step_mode_at_end"] = {"mode": mode_after[0], "fell_back_to_launches": mode_after[1]}   # (trs_get_step_mode: a rank that had to share its GPU says so)
        if world > 1:
            line["config"]["n_gt_1_status"] = ("REHEARSAL on one GPU" if args.share_gpu else "nccl + gloo groups, resident workers, one RCCL all-gather") + \
                "; the N > 1 resident path had never run on N GPUs when this file was written (one-GPU boxes only): unmeasured on hardware until the driver's run"
        if args.share_gpu:
            line["config"]["rehearsal"] = (f"{world} ranks SHARING GPU 0 over gloo: a rehearsal of the N > 1 code path, not a measurement (resident mode selected on every rank; "
                                           "the library lets one worker at a time have the GPU: the ranks take turns of <= 50 ms or step by launches)")
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(min(n, 1024), args.img_h, args.img_w) if render else None
            if cb:
                line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)

    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
