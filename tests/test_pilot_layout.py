"""The LDS image of trs_conv_chain16_kernel (csrc/trsim_pilot_chain16.hpp) is conflict-free by construction — checked by exhaustion, on the CPU.

Design claims pinned here (LDS bank conflicts of the pilot's K loops):
* ``ds_read_b128`` serves a wave in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32); a group is
  conflict-free iff its lanes touch 16 distinct 16-byte slots of the 256-byte bank row (``/opt/skills/guides/MI355X_MICROARCH.md``, LDS).
* With the 16x16x32 MFMA's B-operand layout (lane l: pixel l % 16, channel granule 4 kk + l / 16) a group reads granule g of eight pixels
  and granule g + 1 of eight others.  With all granules of a pixel in ONE plane no XOR swizzle is conflict-free for every tap shift;
  with the even and the odd granules in TWO planes it is, for every shift — provided column i of a block holds a pixel whose linear index
  is = i (mod 16), which is what the host's column tables guarantee (blocks are filled by residue class, not by consecutive output pixels,
  so a row wrap inside a block cannot put two lanes on one bank).
The functions below restate the address arithmetic of the kernel and the table rule of trs_pilot_load for the real layer shapes."""
import itertools

import numpy as np

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def plane_swz(pix, cgs):
    return (pix >> 2) & 3 if cgs == 3 else (pix >> 1) & 7


def slot16(pix, kb, kk, cgs, plane_slots):
    """16-byte slot index of lane (pixel pix, k-block kb) at k-step kk: plane kb & 1, granule-in-plane (2 kk + (kb >> 1)) ^ swizzle."""
    return (kb & 1) * plane_slots + (pix << (cgs - 1)) + ((2 * kk + (kb >> 1)) ^ plane_swz(pix, cgs))


def worst_conflict(pixels, cgs, kk, plane_slots):
    worst = 1
    for g in GROUPS:
        banks = {}
        for l in g:
            a = slot16(pixels[l % 16], l // 16, kk, cgs, plane_slots)
            banks.setdefault(a % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def column_blocks(units, ih, iw, oh, ow):
    """trs_pilot_load's rule: column i of block t = the t-th valid output pixel whose window starts at a linear LDS pixel index = i (mod 16)."""
    cls = [[] for _ in range(16)]
    for ul, oy, ox in itertools.product(range(units), range(oh), range(ow)):
        cls[((ul * ih + oy) * iw + ox) & 15].append((ul * ih + oy) * iw + ox)
    nblk = max(len(c) for c in cls)
    return [[c[t] if t < len(c) else c[-1] for c in cls] for t in range(nblk)], sum(len(c) for c in cls)


# conv4..conv7 at 120x160: (units, IH, IW, OH, OW, input channels)
LAYERS_120x160 = [(2, 12, 17, 10, 15, 64), (4, 10, 15, 8, 13, 64), (4, 8, 13, 6, 11, 64), (4, 6, 11, 4, 9, 128)]


def test_two_plane_image_and_column_tables_are_conflict_free_for_every_tap():
    for units, ih, iw, oh, ow, cin in LAYERS_120x160:
        cgs = 3 if cin == 64 else 4
        plane_slots = -(-(units * ih * iw * (cin // 16)) // 16) * 16        # planes start on a bank row
        blocks, valid = column_blocks(units, ih, iw, oh, ow)
        assert valid == units * oh * ow
        # padding: the blocks hold at most one block more than the pixels need (classes are balanced to within one pixel per image row)
        assert len(blocks) * 16 - valid <= 16 + 2 * units * oh, (len(blocks), valid)
        for blk in blocks:
            assert [p & 15 for p in blk] == list(range(16))
            for tap in range(9):
                off = (tap // 3) * iw + tap % 3
                for kk in range(cin // 32):
                    assert worst_conflict([p + off for p in blk], cgs, kk, plane_slots) == 1


def test_one_plane_cannot_be_conflict_free_for_every_shift():
    """Why two planes: with every granule of a pixel in one plane (slot = pixel * cg + (granule ^ s(pixel))), 16 consecutive pixels at SOME
    alignment always collide, whatever the 4-bit swizzle table s — shown here for the XOR swizzles the 32x32 kernels use."""
    for cgs in (3, 4):
        def slot_one(pix, kb, kk):
            sw = (pix >> 1) & 7 if cgs == 3 else pix & 15
            return (pix << cgs) + ((4 * kk + kb) ^ sw)
        bad = 0
        for p0 in range(32):
            for g in GROUPS:
                banks = {}
                for l in g:
                    a = slot_one(p0 + l % 16, l // 16, 0)
                    banks.setdefault(a % 16, set()).add(a)
                bad += max(len(v) for v in banks.values()) > 1
        assert bad > 0


def test_consecutive_output_pixels_do_collide_behind_a_row_wrap():
    """What the tables are for: a block of 16 CONSECUTIVE output pixels that crosses an image row reads two pixels of one residue class."""
    units, ih, iw, oh, ow, cin = LAYERS_120x160[3]
    pix = [(m // ow) * iw + m % ow for m in range(16)]                      # output pixels 0..15 of a 9-wide row: wraps at 9
    assert len({p & 15 for p in pix}) < 16
    assert worst_conflict(pix, 4, 0, 4096) > 1
