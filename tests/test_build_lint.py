"""Build-time screen of the generated gfx950 code (CPU: hipcc cross-compiles to assembly).

An in-place packed FP32 op whose overlapping source is read across halves (`v_pk_mul_f32 v[14:15], v[12:13], v[14:15]
op_sel:[0,1]`: lo result from the source's HIGH register, which the same instruction overwrites) gave +-0 in the last 16
lanes now and then while other kernels shared the CU (scripts/coresidency_probe.py; DESIGN.md "Packed FP32").  hipcc picks
that form on its own from float2 arithmetic, so the assembly of every kernel is screened for it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "triton-racer-sim_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
PACKED = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\s+(v\[\d+:\d+\]),\s*(.*)$")


def cross_half_in_place(line):
    m = PACKED.match(line)
    if not m:
        return False
    dst, rest = m.group(2), m.group(3)
    ops = [o.strip() for o in rest.split(" op_sel")[0].split(" neg_")[0].split(",")]
    sel = {"op_sel": None, "op_sel_hi": None}
    for key in sel:
        k = re.search(key + r":\[([01,]+)\]", rest)
        if k:
            sel[key] = [int(x) for x in k.group(1).split(",")]
    for i, op in enumerate(ops):
        if op != dst:
            continue
        lo = sel["op_sel"][i] if sel["op_sel"] and i < len(sel["op_sel"]) else 0          # default: lo result reads the low register
        hi = sel["op_sel_hi"][i] if sel["op_sel_hi"] and i < len(sel["op_sel_hi"]) else 1  # default: hi result reads the high register
        if lo != 0 or hi != 1:
            return True
    return False


def test_the_screen_recognises_the_faulty_form():
    assert cross_half_in_place("\tv_pk_mul_f32 v[14:15], v[12:13], v[14:15] op_sel:[0,1]")
    assert not cross_half_in_place("\tv_pk_mul_f32 v[14:15], v[12:13], v[14:15]")
    assert not cross_half_in_place("\tv_pk_fma_f32 v[54:55], v[54:55], v[50:51], s[26:27] op_sel_hi:[1,0,1]")
    assert not cross_half_in_place("\tv_pk_fma_f32 v[28:29], v[14:15], v[2:3], v[0:1] op_sel_hi:[0,1,1]")
    assert cross_half_in_place("\tv_pk_add_f32 v[32:33], v[32:33], v[30:31] op_sel_hi:[0,1]")


@pytest.mark.parametrize("src", ["trsim_hip.hip", "trsim_pilot.hip"])
def test_no_in_place_cross_half_packed_fp32(src, tmp_path):
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    import __graft_entry__ as g
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(CSRC, src)],
                          stderr=subprocess.DEVNULL)
    bad = [l.strip() for l in open(out) if cross_half_in_place(l)]
    assert not bad, f"{src}: in-place packed FP32 ops that read the overwritten register across halves: {bad[:5]}"
