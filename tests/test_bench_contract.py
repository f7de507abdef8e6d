"""bench.py prints ONE JSON line with the driver's contract keys (plus the roofline / cpu_baseline objects)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--envs-per-gpu", "256"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and ROOFLINE <= set(d["roofline"])
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["value"] > 1e6 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0 and 0 < d["roofline"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # every BASELINE config has a leg in the driver's line: configs[0] 1 env through the Car loop, configs[1] 256 envs physics only,
    # configs[2] the line itself, configs[3] / [4] one GPU's share
    also = d["also"]
    assert {"config1_car_loop", "physics_256", "shard_512_of_4096", "pilot_closed_loop", "pilot_closed_loop_512x240x320_depth"} <= set(also)
    c1 = also["config1_car_loop"]
    assert "error" not in c1 and c1["ticks_per_s_launch"] > 100 and c1["ticks_per_s_resident"] > 100
    p2 = also["physics_256"]
    assert "error" not in p2 and p2["envs"] == 256 and p2["us_per_step_spl1"] > 0 and p2["us_per_step_spl16"] > 0 and p2["env_steps_per_s_spl16"] > 1e6
    assert p2["tick_launch_us"] > 0 and p2["tick_resident_us"] > 0            # the consumer-paced tick, launched and posted to the resident physics worker
    assert d["value_incl_worker_launch"] > 0 and d["config"]["launcher"] in ("self", "external", "torch.distributed.run")
    # the image path and the filtered step are in the driver's line
    ip = also["image_path"]
    assert "error" not in ip
    for shape in ("1024x120x160", "256x240x320"):
        assert {"trim", "trim_dynamic_brightness", "trim_hsv_masks", "trim_dynamic_masks_canny"} <= set(ip[shape])
        for leg in ip[shape].values():
            assert leg["us_per_batch"] > 0 and leg["GBps"] > 0 and 0 < leg["frac_of_hbm_peak"] < 1
    fs = also["filtered_step"]
    assert "error" not in fs and {"raw_frames", "static_palette_filter", "dynamic_brightness"} <= set(fs)
    assert all(fs[k]["us_per_step"] > 0 for k in ("raw_frames", "static_palette_filter", "dynamic_brightness"))
    ht = also["hilly_track"]                                                 # a track with elevation (round 5): the mountain track's step in the driver's line
    assert "error" not in ht and ht["rgb"]["us_per_step"] > 0 and ht["rgb_depth"]["us_per_step"] > 0 and 0 < ht["rgb"]["frac_of_hbm_peak"] < 1


def _json_lines(out):
    return [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


@pytest.mark.gpu
def test_bench_self_launch_on_the_gpu():
    """`python bench.py --gpus 1 --spawn`: the rank runs as a child of a parent that never touches the GPU — the path `--gpus N` takes for N > 1."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "40", "--warmup", "5", "--envs-per-gpu", "256",
                          "--no-also", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _json_lines(out)
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["launcher"] == "self" and d["value"] > 1e6 and "allgather" not in d


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_sharing_the_gpu():
    """The N > 1 branch of bench.py end to end on the one GPU of this box: two self-launched ranks, each with its own shard, K steps between host
    barriers, MAX over ranks, one all-gather timed on its own.  Both ranks share GPU 0, so the group is gloo (RCCL refuses two ranks on one device)
    and the value is not a measurement — the line says so.  Both ranks select RESIDENT mode, as the real run does: the library lets one worker at a
    time have the GPU (round 5: a launch that does not get the whole GPU is called off and that rank steps by launches until its next try; round 4
    failed here with "resident worker gave up", gpurun_out/r04_full_2.log, and had to rehearse with launches)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "20", "--warmup", "5", "--total-envs", "512"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = _json_lines(out)
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["config"]["envs_total"] == 512 and d["config"]["envs_per_gpu"] == 256
    assert d["config"]["launcher"] == "self" and "rehearsal" in d["config"]
    ag = d["allgather"]
    assert ag["ranks"] == 2 and ag["floats_per_rank"] == 256 and ag["us"] > 0 and ag["backend"] == "gloo"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "cpu_baseline" not in d
    assert ag["values"] == 512 and "unmeasured on hardware" in d["config"]["n_gt_1_status"]
    assert d["config"]["step_mode"].startswith("resident") and d["config"]["step_mode_at_end"]["mode"] in ("resident", "launch")


@pytest.mark.gpu
def test_bench_nccl_branch_at_world_size_one():
    """`--force-dist`: the process groups of the N > 1 branch at world size 1 on this box's GPU — nccl (= RCCL) for the one all-gather, a gloo group
    for the host barriers and the MAX over ranks — and the exchange through nccl: every call of the N > 1 branch that the two-rank rehearsal (gloo
    only: RCCL refuses two ranks on one device) cannot reach."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "512",
                          "--no-also", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = _json_lines(out)
    assert len(lines) == 1
    d = json.loads(lines[0])
    ag = d["allgather"]
    assert ag["backend"] == "nccl" and ag["ranks"] == 1 and ag["floats_per_rank"] == 512 and ag["us"] > 0
    assert d["n_gpus"] == 1 and d["value"] > 1e6
