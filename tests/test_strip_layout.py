"""The index arithmetic the frame-strip kernels rest on (flucoma-core_amd/csrc/kernels_nmf_strip.hip), restated and checked on
the CPU: the XOR-swizzled LDS image of W is a bijection and both MFMA operand read patterns are conflict-free at 16-byte
bank-group granularity; the element order of a numerator partial covers every (bin, column) of a bin pair exactly once; the
deal of frame quads over workgroups is contiguous, complete and as even as the launcher assumes."""
import itertools


def wl_off(f, c):
    """byte offset of row f (0..31 inside a bin pair), 16-byte chunk c = columns (2c, 2c+1) -- kernels_nmf_strip.hip wl_off"""
    g = ((((c >> 1) ^ ((f >> 1) & 3)) << 2) | ((((f & 1) << 1) | (c & 1)) ^ ((f >> 3) & 3)))
    return ((f >> 1) & 15) * 256 + g * 16


def lane_xyb(lane):
    return lane & 3, (lane >> 2) & 3, lane >> 4   # x, blk, y of the 4x4x4 four-block MFMA


def test_lds_image_is_a_bijection():
    offs = sorted(wl_off(f, c) for f in range(32) for c in range(8))
    assert offs == [16 * i for i in range(256)]


def test_operand_reads_touch_sixteen_bank_groups_per_sixteen_lanes():
    # A operand: rows by x (bin 8 blk + 2 x + e), columns 4 y + m -> chunks 2 y + h; B operand: rows by y, chunks 2 x + h.
    for e, h in itertools.product(range(2), range(2)):
        for kind in "AB":
            for base in range(0, 64, 16):              # a ds_read_b128 is served 16 lanes at a time
                groups = set()
                for lane in range(base, base + 16):
                    x, blk, y = lane_xyb(lane)
                    row = 8 * blk + 2 * (x if kind == "A" else y) + e
                    chunk = 2 * (y if kind == "A" else x) + h
                    groups.add((wl_off(row, chunk) // 16) % 16)   # 16 bank groups of 16 bytes = 256 bytes = 64 banks
                assert len(groups) == 16, (kind, e, h, base, sorted(groups))


def test_kernel_lane_offsets_match_the_image():
    # the per-lane offsets the kernel precomputes: offA / offB [e][h]
    for lane in range(64):
        x, blk, y = lane_xyb(lane)
        for e, h in itertools.product(range(2), range(2)):
            gl = ((x ^ y) << 2) | (((e << 1) | h) ^ blk)
            assert (4 * blk + x) * 256 + gl * 16 == wl_off(8 * blk + 2 * x + e, 2 * y + h)
            assert (4 * blk + y) * 256 + gl * 16 == wl_off(8 * blk + 2 * y + e, 2 * x + h)


def test_partial_element_order_covers_a_bin_pair_once():
    # element `lane` of block r = (jp * 2 + e) * 4 + m  <->  bin 32 jp + 8 blk + 2 y + e, column 4 x + m
    seen = set()
    for e, m, lane in itertools.product(range(2), range(4), range(64)):
        x, blk, y = lane_xyb(lane)
        seen.add((8 * blk + 2 * y + e, 4 * x + m))
    assert seen == {(f, k) for f in range(32) for k in range(16)}


def test_lane_transposition_is_an_involution_between_the_two_tile_views():
    # W-phase view: lane (x, blk, y) holds bins 8 blk + 2 x + {0, 1} of frame y; the H phase needs bins 8 blk + 2 y + {0, 1} of
    # frame x, which lane (y, blk, x) holds
    for lane in range(64):
        x, blk, y = lane_xyb(lane)
        src = y + 4 * blk + 16 * x
        assert lane_xyb(src) == (y, blk, x)          # the lane that holds (bins 8 blk + 2 y + e, frame x) in the W-phase view
        sx, sblk, sy = lane_xyb(src)
        assert sy + 4 * sblk + 16 * sx == lane        # applied twice: back where it started


def test_quads_are_dealt_contiguously_and_evenly():
    k_nq = 6
    for T in (1, 4, 5, 173, 862, 887, 2584, 5168, 6144, 10336, 51680):
        nq = (T + 3) // 4
        nwg = max(min(256, nq), (nq + k_nq - 1) // k_nq)           # nmf_strip_workgroups
        base, rem = divmod(nq, nwg)
        nxt = 0
        for g in range(nwg):
            beg = g * base + min(g, rem)
            end = beg + base + (1 if g < rem else 0)
            assert beg == nxt and 1 <= end - beg <= k_nq
            nxt = end
        assert nxt == nq
