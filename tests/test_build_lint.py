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
# EVERY packed op with a 64-bit destination (today the three TUs emit v_pk_{mul,add,fma}_f32 only; v_pk_mov_b32 and whatever a
# later compiler picks are covered by the same pattern).  16-bit packed forms have 32-bit operands (one register, no pair):
# the pattern does not match them and `test_packed_mnemonics_are_known` fails when one appears, so that it gets looked at.
PACKED = re.compile(r"^\s*(v_pk_\w+)\s+(v\[(\d+):(\d+)\]),\s*(.*)$")
KNOWN_PACKED = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mov_b32"}
# 16-bit packed forms keep both halves in ONE 32-bit register: in place is half-for-half unless op_sel / op_sel_hi swap the halves
KNOWN_PACKED_16 = {"v_pk_max_i16", "v_pk_min_i16", "v_pk_sub_i16",     # (v_pk_sub_i16: the Canny suppression compares two magnitudes per instruction; v_pk_min_i16: the fp16 activations saturate at 65504)
                   "v_pk_add_u16", "v_pk_add_i16", "v_pk_lshlrev_b16", "v_pk_ashrrev_i16", "v_pk_mad_i16", "v_pk_mad_u16", "v_pk_mul_lo_u16",    # the Canny Sobel on packed int16 (round 3)
                   "v_pk_mul_f16", "v_pk_add_f16", "v_pk_min_u16"}   # the fused head stages its pixels as x / 256: a byte permute into the mantissas of (4.0, 4.0) and one packed subtract per pair (round 4)


def swapped_halves_in_place_16(line):
    """True for a 16-bit packed op whose destination is also a source read with non-default half selection."""
    m = re.match(r"\s*(v_pk_\w+16)\s+v(\d+),\s*(.*)$", line)
    if not m:
        return False
    dst, rest = int(m.group(2)), m.group(3)
    ops = [o.strip() for o in rest.split(" op_sel")[0].split(",")]
    sel = re.search(r"op_sel:\[([01,]+)\]", rest)
    sel_hi = re.search(r"op_sel_hi:\[([01,]+)\]", rest)
    lo = [int(x) for x in sel.group(1).split(",")] if sel else [0] * len(ops)
    hi = [int(x) for x in sel_hi.group(1).split(",")] if sel_hi else [1] * len(ops)
    for i, op in enumerate(ops):
        if op == f"v{dst}" and ((i < len(lo) and lo[i] != 0) or (i < len(hi) and hi[i] != 1)):
            return True
    return False
PAIR = re.compile(r"^v\[(\d+):(\d+)\]$")


def cross_half_in_place(line):
    """True for a packed op whose 64-bit destination overlaps a source pair such that a half of the result is computed from a
    register the other half's result is written to: the same pair with a cross-half select (the form that misbehaved), or a
    pair shifted by one register whose overlapping register is read by the half that does not write it."""
    m = PACKED.match(line)
    if not m:
        return False
    d_lo, d_hi, rest = int(m.group(3)), int(m.group(4)), m.group(5)
    if d_hi != d_lo + 1:
        return False
    ops = [o.strip() for o in rest.split(" op_sel")[0].split(" neg_")[0].split(",")]
    sel = {"op_sel": None, "op_sel_hi": None}
    for key in sel:
        k = re.search(key + r":\[([01,]+)\]", rest)
        if k:
            sel[key] = [int(x) for x in k.group(1).split(",")]
    for i, op in enumerate(ops):
        pm = PAIR.match(op)
        if not pm:
            continue
        s_lo, s_hi = int(pm.group(1)), int(pm.group(2))
        if s_hi != s_lo + 1 or s_hi < d_lo or s_lo > d_hi:
            continue                                           # no shared register
        lo = sel["op_sel"][i] if sel["op_sel"] and i < len(sel["op_sel"]) else 0          # default: lo result reads the low register
        hi = sel["op_sel_hi"][i] if sel["op_sel_hi"] and i < len(sel["op_sel_hi"]) else 1  # default: hi result reads the high register
        reg_for_lo, reg_for_hi = (s_hi if lo else s_lo), (s_hi if hi else s_lo)
        # the lo result is written to d_lo, the hi result to d_hi: each half may only read "its own" destination register
        if reg_for_lo == d_hi or reg_for_hi == d_lo:
            return True
    return False


def test_the_screen_recognises_the_faulty_form():
    assert cross_half_in_place("\tv_pk_mul_f32 v[14:15], v[12:13], v[14:15] op_sel:[0,1]")
    assert not cross_half_in_place("\tv_pk_mul_f32 v[14:15], v[12:13], v[14:15]")
    assert not cross_half_in_place("\tv_pk_fma_f32 v[54:55], v[54:55], v[50:51], s[26:27] op_sel_hi:[1,0,1]")
    assert not cross_half_in_place("\tv_pk_fma_f32 v[28:29], v[14:15], v[2:3], v[0:1] op_sel_hi:[0,1,1]")
    assert cross_half_in_place("\tv_pk_add_f32 v[32:33], v[32:33], v[30:31] op_sel_hi:[0,1]")
    assert cross_half_in_place("\tv_pk_mov_b32 v[4:5], v[4:5], v[4:5] op_sel:[1,0]")          # any packed mnemonic, not only the arithmetic ones
    assert cross_half_in_place("\tv_pk_mul_f32 v[14:15], v[13:14], v[20:21]")                  # shifted pair: the hi result reads v14, which the lo result overwrites
    assert cross_half_in_place("\tv_pk_mul_f32 v[14:15], v[15:16], v[20:21] op_sel_hi:[1,1]")       # ... and the mirror image: the lo result reads v15, which the hi result overwrites
    assert not cross_half_in_place("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]")       # no shared register


_ASM = {}


def device_asm(src, tmp_path_factory):
    if src not in _ASM:
        if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        import __graft_entry__ as g
        flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
        out = tmp_path_factory.mktemp("asm") / (src + ".s")
        subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(CSRC, src)],
                              stderr=subprocess.DEVNULL)
        _ASM[src] = open(out).read()
    return _ASM[src]


SOURCES = ["trsim_hip.hip", "trsim_resident.hip", "trsim_pilot.hip"]


@pytest.mark.parametrize("src", SOURCES)
def test_no_in_place_cross_half_packed_op(src, tmp_path_factory):
    bad = [l.strip() for l in device_asm(src, tmp_path_factory).splitlines() if cross_half_in_place(l)]
    assert not bad, f"{src}: packed ops that read a register the other half's result overwrites: {bad[:5]}"


@pytest.mark.parametrize("src", SOURCES)
def test_packed_mnemonics_are_known(src, tmp_path_factory):
    text = device_asm(src, tmp_path_factory)
    seen = set(re.findall(r"^\s*(v_pk_\w+)", text, flags=re.M))
    assert seen <= KNOWN_PACKED | KNOWN_PACKED_16, f"{src}: new packed instruction forms {sorted(seen - KNOWN_PACKED - KNOWN_PACKED_16)}: extend the screen (16-bit forms have one-register operands)"
    bad = [l.strip() for l in text.splitlines() if swapped_halves_in_place_16(l)]
    assert not bad, f"{src}: 16-bit packed ops that swap halves in place: {bad[:5]}"


def test_the_16_bit_screen():
    assert not swapped_halves_in_place_16("\tv_pk_max_i16 v5, v5, 0")
    assert not swapped_halves_in_place_16("\tv_pk_max_i16 v5, v6, v5 op_sel_hi:[1,1]")
    assert swapped_halves_in_place_16("\tv_pk_max_i16 v5, v5, v6 op_sel:[1,0] op_sel_hi:[0,1]")
    assert not swapped_halves_in_place_16("\tv_pk_max_i16 v5, v7, v6 op_sel:[1,0] op_sel_hi:[0,1]")


@pytest.mark.parametrize("src,kernels", [("trsim_hip.hip", "trs_step_kernel"), ("trsim_resident.hip", "trs_worker_kernel")])
def test_map_kernels_have_no_static_lds_and_no_scratch(src, kernels, tmp_path_factory):
    """The rasteriser addresses the class map at LDS offset 0: the dynamic segment starts there exactly when the kernel has no
    static __shared__ (a compile-time property: .group_segment_fixed_size).  Also: no scratch, no VGPR spills."""
    text = device_asm(src, tmp_path_factory)
    blocks = re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", text, flags=re.S)
    mine = [b for b in blocks if kernels in b]
    assert mine, f"no metadata for {kernels}"
    for b in mine:
        assert re.search(r"\.group_segment_fixed_size:\s*0\b", b), b[:400]
        assert re.search(r"\.private_segment_fixed_size:\s*0\b", b), b[:400]
        assert re.search(r"\.vgpr_spill_count:\s*0\b", b), b[:400]
