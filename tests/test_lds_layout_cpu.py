"""CPU model of the LDS images of bgemm8_kernel (scldm_amd/csrc/bgemm8.hpp): the LDS-DMA staging map must be a bijection onto a
unit's 16 KB, the swizzle of the staging side and of the fragment reads must agree, and every lane group of a fragment read
must fall on distinct banks (lane groups and bank formulas: /opt/skills/guides/MI355X_MICROARCH.md, LDS table - ds_read_b128 is
served in four groups of 16 lanes over 64 banks of 4 bytes, ds_read_b64_tr_b16 in two groups of 32).  Pure index arithmetic:
this file restates the address formulas of the header next to the property each one exists for."""
import itertools

UNIT = 128 * 128          # bytes of a staging unit
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def kc_stage(tid, q):
    """thread -> (LDS byte offset inside the unit, unit row, source k chunk) of staging instruction q (k-contiguous operand)."""
    wave, lane = tid >> 6, tid & 63
    dst = q * 8192 + wave * 1024 + lane * 16          # wave base + 16 lane: what the LDS-DMA writes
    row = q * 64 + (tid >> 3)
    c_src = (tid & 7) ^ ((tid >> 4) & 7)
    return dst, row, c_src


def kc_read(lane, ks, row0):
    """lane -> (byte offset, row, k chunk it must hold) of the ds_read_b128 of k step ks for the 32-row block at row0."""
    row = row0 + (lane & 31)
    chunk = 2 * ks + (lane >> 5)
    off = (lane & 31) * 128 + (((2 * ks + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4) + row0 * 128
    return off, row, chunk


def test_k_contiguous_unit_staging_is_a_bijection_and_matches_the_reads():
    where = {}
    for q, tid in itertools.product(range(2), range(512)):
        dst, row, c = kc_stage(tid, q)
        assert dst == row * 128 + ((c ^ ((row >> 1) & 7)) << 4)          # slot = chunk ^ ((row >> 1) & 7)
        assert dst not in where
        where[dst] = (row, c)
    assert sorted(where) == list(range(0, UNIT, 16))
    for row0, ks, lane in itertools.product(range(0, 128, 32), range(4), range(64)):
        off, row, chunk = kc_read(lane, ks, row0)
        assert where[off] == (row, chunk)


def test_k_contiguous_fragment_reads_are_bank_conflict_free():
    for row0, ks in itertools.product(range(0, 128, 32), range(4)):
        for group in B128_GROUPS:
            banks = set()
            for lane in group:
                off, _, _ = kc_read(lane, ks, row0)
                for b in range(4):
                    banks.add(((off >> 2) + b) % 64)
            assert len(banks) == 64                                       # 16 lanes x 16 bytes = every bank exactly once


def mc_stage(tid, q):
    """m-contiguous operand: thread -> (LDS byte offset, k row, source chunk of 8 consecutive m)."""
    wave, lane = tid >> 6, tid & 63
    dst = q * 8192 + wave * 1024 + lane * 16
    krow = q * 32 + (tid >> 4)
    c_src = (tid & 15) ^ (((tid >> 4) & 3) << 2)
    return dst, krow, c_src


def mc_read(lane, ks, seg, half):
    """lane -> (byte offset, k row, first of its 4 consecutive columns) of one ds_read_b64_tr_b16 (seg: the block's 64-byte segment)."""
    g, i = lane >> 4, lane & 15
    krow = 16 * ks + 8 * (g >> 1) + (i >> 2) + 4 * half
    off = krow * 256 + ((seg ^ (i >> 2)) << 6) + 32 * (g & 1) + 8 * (i & 3)
    col = seg * 32 + 16 * (g & 1) + 4 * (i & 3)
    return off, krow, col


def test_m_contiguous_unit_staging_is_a_bijection_and_matches_the_transposing_reads():
    where = {}
    for q, tid in itertools.product(range(2), range(512)):
        dst, krow, c = mc_stage(tid, q)
        assert dst == krow * 256 + ((c ^ ((krow & 3) << 2)) << 4)         # 64-byte segments XORed with k row & 3
        assert dst not in where
        where[dst] = (krow, c)
    assert sorted(where) == list(range(0, UNIT, 16))
    for ks, seg, half, lane in itertools.product(range(4), range(4), range(2), range(64)):
        off, krow, col = mc_read(lane, ks, seg, half)
        krow_w, c = where[off & ~15]
        assert krow_w == krow and c * 8 + (off & 15) // 2 == col          # the 8-byte piece holds columns col .. col + 3 of k row krow


def test_m_contiguous_transposing_reads_are_bank_conflict_free():
    for ks, seg, half in itertools.product(range(4), range(4), range(2)):
        for group in (range(0, 32), range(32, 64)):
            banks = set()
            for lane in group:
                off, _, _ = mc_read(lane, ks, seg, half)
                banks.update({(off >> 2) % 64, ((off >> 2) + 1) % 64})
            assert len(banks) == 64                                       # 32 lanes x 8 bytes = every bank exactly once


def test_bf16_epilogue_image_rows_come_back_whole():
    """EPI 8192: lane writes the 8-byte piece of (row r, columns nl .. nl + 3) at chunk (nl / 8) ^ (r % 32); the row-wise pass reads
    chunk j ^ (r % 32) for global chunk j: every row must come back in column order."""
    for r in (0, 1, 31, 32, 77, 255):
        img = {}
        for nl in range(0, 256, 4):
            img[r * 512 + (((nl >> 3) ^ (r & 31)) << 4) + (nl & 4) * 2] = nl
        for j in range(32):
            base = r * 512 + ((j ^ (r & 31)) << 4)
            assert img[base] == 8 * j and img[base + 8] == 8 * j + 4
