"""The work lists of the factor updates (api_corpus.hip build_list_plan, reached through fluhip_debug_plan_lists: pure host code,
no GPU): whatever strip width, cutting and grouping the planner picks, the lists must be a valid schedule --
every (buffer, column group) covered over exactly its contraction range, once; the wavefronts that split a strip
consecutive in ONE workgroup with one leader; output slots (result / partial / denominator / statistics) each written by
exactly one wavefront and consistent with the finalize's per-buffer table.  The kernels trust these descriptors blindly."""
import ctypes

import numpy as np
import pytest


def _plan(lib, frames, bins, K, which):
    fr = np.ascontiguousarray(frames, dtype=np.int64)
    info = (ctypes.c_int32 * 8)()
    n = lib.fluhip_debug_plan_lists(len(fr), fr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), bins, K, which, None, 0, info)
    assert n >= 4 and n % 4 == 0
    desc = np.zeros((n, 12), dtype=np.int32)
    assert lib.fluhip_debug_plan_lists(len(fr), fr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), bins, K, which,
                                       desc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n, info) == n
    keys = ("wgs", "ng", "partial", "max_split", "pieces", "n_partials", "side", "stat_parts")
    return desc, dict(zip(keys, [int(v) for v in info]))


def _check(desc, info, frames, bins, which):
    B = len(frames)
    side = info["side"]
    if which == 0:   # W update: columns = bins (without the Nyquist side column), contraction over a buffer's frames
        groups = [((bins - side) + 15) // 16] * B
        steps = [(t + 3) // 4 for t in frames]
    else:            # H update: columns = a buffer's frames, contraction over the bins
        groups = [(t + 15) // 16 for t in frames]
        steps = [(bins + 3) // 4] * B
    assert desc.shape[0] == 4 * info["wgs"]
    cover = {}       # (buffer, group) -> list of step ranges
    part_writers, d_writers, stat_writers, result_writers = {}, {}, {}, {}
    for wgi in range(info["wgs"]):
        wg = desc[4 * wgi:4 * wgi + 4]
        live = [d for d in wg if d[2] > 0]
        any_group = any(((d[8] >> 8) & 15) > 1 for d in live)
        w = 0
        while w < 4:
            d = wg[w]
            buf, g0, ng, s0, s1, part, stat, dslot, grp = (int(x) for x in d[:9])
            if ng == 0:
                w += 1
                continue
            leader, rank, size = grp & 15, (grp >> 4) & 15, (grp >> 8) & 15
            assert ((grp >> 16) & 1) == (1 if any_group else 0)      # every live wavefront of a grouping workgroup meets the barrier
            assert rank == 0 and leader == w and 1 <= size <= 4 and w + size <= 4, (wgi, w, grp)
            assert 0 <= buf < B and 0 <= g0 and g0 + ng <= groups[buf] and ng <= info["ng"]
            for r in range(size):                                     # the members: same strip, consecutive wavefronts, disjoint ranges
                m = wg[w + r]
                assert (int(m[0]), int(m[1]), int(m[2])) == (buf, g0, ng)
                assert (int(m[8]) & 15, (int(m[8]) >> 4) & 15, (int(m[8]) >> 8) & 15) == (leader, r, size)
                assert 0 <= m[3] <= m[4] <= steps[buf]
                if r > 0:
                    assert m[5] == -1 and m[7] == -1                  # only the leader owns output slots
                for g in range(g0, g0 + ng):
                    cover.setdefault((buf, g), []).append((int(m[3]), int(m[4])))
            if info["partial"]:
                assert part >= 0
                part_writers.setdefault(part, []).append((buf, g0))
                if dslot >= 0:
                    assert dslot == part
                    d_writers[dslot] = d_writers.get(dslot, 0) + 1
            else:
                assert part == -1 and dslot == -1
                result_writers[(buf, g0)] = result_writers.get((buf, g0), 0) + 1
                if which == 0:
                    stat_writers[stat] = stat_writers.get(stat, 0) + 1
            w += size
    for b in range(B):
        for g in range(groups[b]):
            rs = sorted(r for r in cover.get((b, g), []) if r[1] > r[0])
            assert rs and rs[0][0] == 0 and rs[-1][1] == steps[b], (b, g, rs)
            assert all(rs[i][1] == rs[i + 1][0] for i in range(len(rs) - 1)), (b, g, rs)   # a partition: no gap, no overlap
    if info["partial"]:
        assert set(part_writers) == set(range(info["n_partials"]))
        assert all(v == 1 for v in d_writers.values()) and set(d_writers) == set(part_writers)   # one denominator per partial
        per_buf = {}
        for p, ws in part_writers.items():
            assert len({b for b, _ in ws}) == 1                        # a partial belongs to one buffer ...
            strips = [g0 for _, g0 in ws]
            assert len(strips) == len(set(strips))                     # ... and every strip of it writes it once
            per_buf.setdefault(ws[0][0], []).append(p)
        for b, ps in per_buf.items():
            assert sorted(ps) == list(range(min(ps), min(ps) + len(ps))) and len(ps) <= info["max_split"]
    else:
        assert all(v == 1 for v in result_writers.values())
        if which == 0:
            assert all(v == 1 for v in stat_writers.values())
            assert max(stat_writers) < B * info["stat_parts"]


def test_work_lists_are_valid_schedules(fluhip_lib_path):
    import fluhip
    lib = fluhip.load_library(fluhip_lib_path)
    rs = np.random.RandomState(12)
    cases = [([862] * 8, 1025, 32), ([862] * 64, 1025, 32), ([862] * 128, 1025, 32), ([1], 17, 1), ([5168], 1025, 16),
             ([25840, 25840], 2049, 128), ([862] * 3, 1025, 100), ([40, 1, 700, 33, 2000], 513, 5)]
    for _ in range(60):
        B = int(rs.choice([1, 2, 3, 7, 20, 64, 100, 200, 400]))
        longest = int(rs.choice([1, 9, 100, 900, 5000]))
        frames = [int(rs.randint(1, longest + 1)) for _ in range(B)]
        cases.append((frames, int(rs.choice([17, 33, 129, 513, 1025, 2049])), int(rs.choice([1, 3, 16, 17, 32, 33, 64, 65, 128]))))
    for frames, bins, K in cases:
        for which in (0, 1):
            desc, info = _plan(lib, frames, bins, K, which)
            _check(desc, info, frames, bins, which)


def test_work_list_schedule_choices(fluhip_lib_path):
    """what the planner is expected to pick where it was measured (profiles/r03)"""
    import fluhip
    lib = fluhip.load_library(fluhip_lib_path)
    # the bench shard: whole contractions at the widest strips, nothing cut, Nyquist as a side column
    _, w = _plan(lib, [862] * 128, 1025, 32, 0)
    _, h = _plan(lib, [862] * 128, 1025, 32, 1)
    assert (w["wgs"], w["ng"], w["partial"], w["pieces"], w["side"]) == (256, 8, 0, 1, 1)
    assert (h["wgs"], h["partial"], h["pieces"]) == (256, 0, 1) and h["ng"] == 7
    # eight buffers: narrow strips, pieces added up inside workgroups, no partials in memory, one round
    _, w = _plan(lib, [862] * 8, 1025, 32, 0)
    assert w["partial"] == 0 and 1 < w["pieces"] <= 4 and w["wgs"] <= 256 and w["ng"] < 8
    with pytest.raises(Exception):
        _plan(lib, [0, 5], 1025, 32, 0)


def _tail(lib, count, frames, bins, K):
    out = (ctypes.c_int64 * 4)()
    assert lib.fluhip_debug_plan_tail(count, frames, bins, K, out) == 0
    return tuple(int(v) for v in out)


def test_two_launch_plan_of_the_h_update(fluhip_lib_path):
    """api_corpus.hip plan_tail (pure host code): a whole-contraction H update whose wavefronts leave a poorly filled last round of the
    1024 SIMDs is cut into a launch of whole rounds and a split tail; exact fills and single rounds stay one launch"""
    import fluhip
    lib = fluhip.load_library(fluhip_lib_path)
    # config 3: 2 x 25840 frames, 2049 bins, rank 128 -> 2 x 808 strips of 32 frames: 2 x 512 whole, 2 x 296 in three pieces
    sp, wA, wB, cols = _tail(lib, 2, 25840, 2049, 128)
    assert (sp, wA, wB, cols) == (3, 512, 296, 512 * 32)
    # the same channel on its own: 808 strips, a single round -- nothing to cut
    assert _tail(lib, 1, 25840, 2049, 128)[0] == 0
    # the bench shard (128 x 862 frames, rank 32): 1024 strips, exactly one round
    assert _tail(lib, 128, 862, 1025, 32)[0] == 0
    # a schedule whatever the shape: the first launch fills whole rounds, both launches cover the frames once
    for count, frames, bins, K in [(2, 20000, 513, 128), (2, 17001, 513, 100), (2, 36000, 513, 64), (2, 40000, 513, 128),
                                   (1, 70000, 1025, 128), (3, 30000, 2049, 128), (2, 16384 + 32, 513, 128)]:
        sp, wA, wB, cols = _tail(lib, count, frames, bins, K)
        if sp == 0:
            continue
        assert 2 <= sp <= 8 and wA >= 1 and wB >= 1
        assert count * wA <= 3 * 1024 and count * (wA + 1) > 1024 * (count * wA // 1024) and cols % 16 == 0 and 0 < cols < frames
        width = cols // wA                       # frames per strip of the first launch: whole column groups
        assert width % 16 == 0 and wB * width >= frames - cols > (wB - 1) * width - 16
    # rank above 128: another kernel form, never two launches
    assert _tail(lib, 2, 40000, 513, 200) == (0, 0, 0, 0)


def test_schedule_family_of_the_baseline_shapes(fluhip_lib_path):
    """api_corpus.hip list_plan_pays (pure host code): the measured rules, pinned for the shapes whose records are under profiles/r03/"""
    import fluhip
    lib = fluhip.load_library(fluhip_lib_path)
    kind = lambda B, T, F, K: lib.fluhip_debug_plan_kind(B, T, F, K)  # noqa: E731
    assert kind(128, 862, 1025, 32) == 0        # the bench shard: exactly one round of wavefronts, uniform kernel
    assert kind(1024, 862, 1025, 32) == 1       # the whole config-4 corpus on one GPU
    assert kind(8, 862, 1025, 32) == 1 and kind(200, 862, 1025, 32) == 1 and kind(3, 862, 1025, 32) == 1
    assert kind(1, 862, 1025, 32) == 0 and kind(2, 862, 1025, 32) == 0          # one c4 buffer, a stereo pair of them
    assert kind(1, 25840, 1025, 32) == 1 and kind(2, 2584, 1025, 32) == 1       # 1 x 300 s, 2 x 30 s
    assert kind(1, 2584, 1025, 32) == 0                                         # 1 x 30 s
    assert kind(2, 25840, 2049, 128) == 0       # config 3: uniform schedule (with the two-launch H update)
    assert kind(40, 862, 1025, 128) == 0 and kind(4, 862, 1025, 128) == 1 and kind(4, 5168, 1025, 128) == 0
    assert kind(100, 862, 1025, 64) == 1 and kind(1, 25840, 1025, 64) == 1
    assert kind(2, 40000, 513, 200) == 0        # above rank 128: the un-fused path, no lists
    assert kind(0, 1, 1, 1) == -1


def test_what_the_h_update_takes_over(fluhip_lib_path):
    """kernels_nmf5.hip SIDEQ forms, asked through the launcher's dry run (pure host code): in steady state an iteration of the
    bench shard is two launches -- the H update forms the next W update's side column and does the norm combine in front"""
    import fluhip
    lib = fluhip.load_library(fluhip_lib_path)
    form = lambda B, T, F, K: lib.fluhip_debug_plan_h_update(B, T, F, K)  # noqa: E731
    assert form(128, 862, 1025, 32) == 3        # the bench shard (and every 128-buffer window of the config-4 corpus run round-major)
    assert form(128, 862, 1025, 16) == 3 and form(128, 862, 1025, 12) == 3
    assert form(128, 862, 1025, 64) == 0        # rank 64 at that size runs from the work lists
    assert form(128, 862, 1025, 128) == 0       # rank 128: more components than lanes, the launches stay
    assert form(1024, 862, 1025, 32) == 0       # work lists
    assert form(2, 25840, 2049, 128) == 0       # config 3
    assert form(128, 862, 1024, 32) == 0        # 1 024 bins: whole column groups, no side column
    assert form(128, 137, 1025, 32) == 3        # 70 000 samples at hop 512 (the variants test's corpus)
    # more than 64 strips per buffer in the H update (uniform plan, long buffers): the side-column partials have 64 slices per
    # buffer, so the epilogue must not take the side column over (it wrote past the area)
    kind = lambda B, T, F, K: lib.fluhip_debug_plan_kind(B, T, F, K)  # noqa: E731
    assert kind(64, 5000, 1025, 64) == 0 and form(64, 5000, 1025, 64) == 0       # 80 strips
    assert kind(64, 12800, 1025, 64) == 0 and form(64, 12800, 1025, 64) == 0     # 200 strips
    assert kind(128, 9300, 1025, 32) == 0 and form(128, 9300, 1025, 32) == 0     # 65 strips
    assert form(0, 1, 1, 1) == -1
