"""LDS layouts of the f16 LDS-patch kernels (conv3x3_halo.hip), checked against the banking model of MI355X_MICROARCH.md (LDS section):
a wave64 `ds_read_b128` is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- one LDS
cycle per group when the group's sixteen 16-byte slots fall on sixteen different slots of the 256-byte bank row (bank = (addr / 4)
mod 64).  The grouping is built for 32 CONSECUTIVE rows; a 16-wide output tile puts two tile rows into one 32-pixel MFMA block, and
with lane = pixel the two halves of a hardware group read patch rows PW = 18 / 20 / 24 apart that collide mod 16 -- found with
SQ_LDS_BANK_CONFLICT (8.5 M cycles per 4K launch, profiles/r04_halo4_pmc.log) and removed by h_pix (lanes of one hardware group take
the sixteen pixels of ONE tile row).  This file restates h_pix / h_swz / the fragment addresses in Python and counts conflicts: none
with h_pix, some without (the negative control), for every tap, slice and dilation the kernels run; the weight images (128-byte and
64-byte rows) and the epilogue staging rows are checked the same way.  Host logic only: nothing here touches a GPU."""
import itertools

import pytest

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def extra_cycles_b128(addr_of_lane):
    """extra LDS cycles of one ds_read_b128 / 16-byte access pattern: per hardware group, (max number of lanes on one 16-byte slot
    of the 256-byte bank row) - 1; identical addresses broadcast"""
    extra = 0
    for g in GROUPS:
        slots = {}
        for l in g:
            a = addr_of_lane(l)
            assert a % 16 == 0
            slots.setdefault((a // 16) % 16, set()).add(a)
        extra += max(len(v) for v in slots.values()) - 1
    return extra


def h_swz(row):
    return (row >> 1) & 7


def h_pix(r, tw):
    return ((r & 15) | ((((((r >> 2) & 3) + 1) >> 1) & 1) ^ (r >> 4)) << 4) if tw == 16 else r


def patch_read_conflicts(tw, th, d, pix):
    PW = tw + 2 * d
    total = 0
    for blk, tap, sl in itertools.product(range(th * tw // 32), range(9), range(4)):
        ky, kx = divmod(tap, 3)

        def addr(lane):
            r, hh = lane & 31, lane >> 5
            R = blk * 32 + pix(r, tw)
            ty, tx = divmod(R, tw)
            p = (ty + ky * d) * PW + tx + kx * d
            return p * 128 + (((2 * sl + hh) ^ h_swz(p)) * 16)

        total += extra_cycles_b128(addr)
    return total


def test_h_pix_is_a_permutation_that_keeps_a_hardware_group_on_one_tile_row():
    assert sorted(h_pix(r, 16) for r in range(32)) == list(range(32))
    assert [h_pix(r, 32) for r in range(32)] == list(range(32))
    for g in GROUPS[:2]:
        rows = {h_pix(r, 16) >> 4 for r in g}
        cols = sorted(h_pix(r, 16) & 15 for r in g)
        assert len(rows) == 1 and cols == list(range(16))


@pytest.mark.parametrize("d", [1, 2, 4])
def test_patch_fragment_reads_are_conflict_free_with_h_pix(d):
    assert patch_read_conflicts(16, 16, d, h_pix) == 0
    assert patch_read_conflicts(32, 8, d, h_pix) == 0   # (one tile row per block: consecutive patch rows as the hardware expects)
    assert patch_read_conflicts(32, 7, d, h_pix) == 0   # the strip region of 135 x 240
    # the negative control: lane = pixel, what the kernels did before
    assert patch_read_conflicts(16, 16, d, lambda r, tw: r) > 0


@pytest.mark.parametrize("row_bytes", [128, 64])
def test_weight_fragment_reads_are_conflict_free(row_bytes):
    """a fragment's 32 rows are consecutive image rows; 128-byte rows: chunk ^ ((row >> 1) & 7), 64-byte rows (half-tap steps):
    chunk ^ ((row >> 2) & 3)"""
    slices = 4 if row_bytes == 128 else 2
    for base_row, kl in itertools.product((0, 32, 96, 224), range(slices)):
        def addr(lane):
            r, hh = lane & 31, lane >> 5
            row = base_row + r
            sw = h_swz(r) if row_bytes == 128 else (r >> 2) & 3
            return row * row_bytes + (((2 * kl + hh) ^ sw) * 16)

        assert extra_cycles_b128(addr) == 0


def staging_read_conflicts(rowb, lpr):
    total = 0
    for it, half in itertools.product(range(32 * lpr // 64), range(2)):
        def addr(lane):
            row, col = it * (64 // lpr) + lane // lpr, lane % lpr
            return row * rowb + col * 32 + half * 16

        total += extra_cycles_b128(addr)
    return total


def test_epilogue_staging_reads():
    """read-back of a wave's staged 32-pixel block: `lpr` lanes per pixel row, 32 bytes of f32 per lane (two b128 reads).  The 4-wave
    form (128 channels per wave: 16 lanes per row, 528-byte rows) is conflict-free.  The 64-channel forms (8 lanes per row, 272-byte
    rows: the 8-wave LDS-patch kernels, conv1x1_areg, the f16 -> f16 epilogue of conv_igemm_kernel.h) are NOT under the hardware's
    real lane groups -- half of their groups take one extra cycle, and no row padding repairs it (a lane -> row mapping that gives a
    hardware group two whole rows would): 32 extra LDS cycles per 32-pixel block against ~1,000 cycles of stores, recorded here so
    that the number is known, not so that it is fixed."""
    assert staging_read_conflicts(4 * 128 + 16, 16) == 0
    assert staging_read_conflicts(2 * 128 + 16, 8) == 32
    assert min(staging_read_conflicts(256 + pad, 8) for pad in range(0, 272, 16)) == 32


# ---- conv_hl.hip (INFUR_DTYPE_F16_HL): 64-byte hi rows and 32-byte lo rows of a 32-channel K step, DMA pieces lane-linear ----
def hl_swz64(row):
    return (row >> 2) & 3


def hl_swz32(row):
    return (row >> 3) & 1


@pytest.mark.parametrize("blocks", [1, 2, 4])
def test_conv_hl_fragment_reads_are_conflict_free(blocks):
    """lane (row r = lane & 31, half h = lane >> 5) reads hi chunk 2 h + kk (kk = 0, 1) and lo chunk h of its row; a wave's row blocks are
    32 rows apart, which leaves both swizzles unchanged"""
    for blk in range(blocks):
        for kk in range(2):
            assert extra_cycles_b128(lambda l: ((blk * 32 + (l & 31)) * 64 + (((2 * (l >> 5) + kk) ^ hl_swz64(l & 31)) * 16))) == 0
        assert extra_cycles_b128(lambda l: ((blk * 32 + (l & 31)) * 32 + (((l >> 5) ^ hl_swz32(l & 31)) * 16))) == 0
    # negative controls: the same reads without the swizzles collide (4-way on the 64-byte rows, 2-way on the 32-byte rows)
    assert extra_cycles_b128(lambda l: (l & 31) * 64 + (2 * (l >> 5)) * 16) > 0
    assert extra_cycles_b128(lambda l: (l & 31) * 32 + (l >> 5) * 16) > 0


def test_conv_hl_dma_pieces_cover_the_image_exactly_once():
    """a hi piece = rows 16 p .. 16 p + 15 (lane l: row l >> 2 at position l & 3 = data chunk (l & 3) ^ swz64(row)), a lo piece = rows 32 p ..
    32 p + 31 (lane l: row l >> 1 at position l & 1): every (row, data chunk) of the image is written once, at the address the fragment
    reads look for it"""
    for rows in (128, 256):
        where = {}
        for p in range(rows // 16):
            for l in range(64):
                row = 16 * p + (l >> 2)
                chunk = (l & 3) ^ hl_swz64(row)
                addr = p * 1024 + l * 16
                assert (row, chunk) not in where
                where[(row, chunk)] = addr
        for row in range(rows):
            for q in range(4):
                assert where[(row, q)] == row * 64 + ((q ^ hl_swz64(row)) * 16)
        where = {}
        for p in range(rows // 32):
            for l in range(64):
                row = 32 * p + (l >> 1)
                chunk = (l & 1) ^ hl_swz32(row)
                where[(row, chunk)] = p * 1024 + l * 16
        for row in range(rows):
            for q in range(2):
                assert where[(row, q)] == row * 32 + ((q ^ hl_swz32(row)) * 16)


# ---- conv_hl_areg.hip: a wave's residual / output slot (32 rows of 2 W bytes hi and W bytes lo, W = 64 or 128 channels per wave),
#      DMA pieces of 1 KB of whole rows, accumulator-layout accesses in place, and the row permutation of the weight tile ----
def ah_swz8(row):
    return (row >> 1) & 7


def ah_pi(r):
    return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)


@pytest.mark.parametrize("wch", [64, 128])
def test_conv_hl_areg_slot_pieces_and_accumulator_accesses_agree(wch):
    """piece p, lane l lands at p * 1024 + l * 16; the kernel sends it the data chunk (l % chunks_per_row) ^ swz(row) of row
    rows_per_piece * p + l // chunks_per_row.  The epilogue looks for channels 32 j + 16 s + 8 h .. + 7 of row r at hi chunk
    (4 j + 2 s + h) ^ swz(r) and at lo chunk (2 j + s) ^ swz(r), half h: both must be where the DMA put them, and every chunk of the
    slot must be written exactly once"""
    for plane, row_bytes in (("hi", 2 * wch), ("lo", wch)):
        chunks = row_bytes // 16
        rows_pp = 1024 // row_bytes
        swz = ah_swz8 if chunks >= 8 else hl_swz64
        where = {}
        for p in range(32 * row_bytes // 1024):
            for l in range(64):
                row = rows_pp * p + l // chunks
                d = (l % chunks) ^ swz(row)
                assert 0 <= d < chunks and (row, d) not in where
                where[(row, d)] = p * 1024 + l * 16
        assert len(where) == 32 * chunks
        for r in range(32):
            for j in range(wch // 32):
                for s_ in range(2):
                    for h in range(2):
                        d = 4 * j + 2 * s_ + h if plane == "hi" else 2 * j + s_
                        assert where[(r, d)] == r * row_bytes + ((d ^ swz(r)) * 16)


def test_conv_hl_areg_slot_hi_accesses_are_conflict_free():
    """the in-place b128 accesses of the 64-channel form: lane (r, h) at chunk (4 j + 2 s + h) ^ swz8(r) of its 128-byte row"""
    for j in range(2):
        for s_ in range(2):
            assert extra_cycles_b128(lambda l: (l & 31) * 128 + (((4 * j + 2 * s_ + (l >> 5)) ^ ah_swz8(l & 31)) * 16)) == 0


def test_conv_hl_areg_row_permutation_gives_eight_consecutive_channels():
    """weight-tile row R holds output channel pi(R); in the MFMA's C/D layout lane (pixel, h) register 4 g + e is row 8 g + 4 h + e, so
    registers 8 s .. 8 s + 7 must be the consecutive channels 16 s + 8 h .. + 7"""
    assert sorted(ah_pi(r) for r in range(32)) == list(range(32))
    for h in range(2):
        for s_ in range(2):
            chans = [ah_pi(8 * (reg // 4) + 4 * h + reg % 4) for reg in range(8 * s_, 8 * s_ + 8)]
            assert chans == [16 * s_ + 8 * h + t for t in range(8)]
