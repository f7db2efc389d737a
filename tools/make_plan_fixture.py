"""Mint tests/golden/plans_r06.json: the planner's whole answer (fluhip_debug_plan_shape, no device) for the BASELINE shapes and
the shapes of tools/perf_matrix.py -- the pin a schedule change has to move consciously (and re-run the perf matrix for).

    python tools/make_plan_fixture.py          # writes the fixture from the in-tree library
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip  # noqa: E402

KEYS = ("variant", "split_w", "split_h", "deferred_norm", "side_column", "strips_w", "padded_rank", "compute_rank", "strip",
        "lists", "tail_split_h", "tail_strips_h", "tail_rest_h", "tail_cols_h", "strips_h", "part", "dpart", "csum", "wscratch",
        "colpart", "strippart", "wide", "h_takes", "combine_form", "side_slices", "side_slots", "stat_doubles")

# (buffers, frames, bins, rank): BASELINE configs 1 - 4 and the rows of tools/perf_matrix.py
SHAPES = {
    "c1_bufnmf_loop_k3": (1, 887, 513, 3),
    "c2_60s_k16": (1, 5168, 1025, 16),
    "c3_2x10min_k128_fft4096": (2, 25840, 2049, 128),
    "c4_shard_128x10s_k32": (128, 862, 1025, 32),
    "c4_1024x10s_k32_one_gpu": (1024, 862, 1025, 32),
    "c4x1_10s_k32": (1, 862, 1025, 32),
    "corpus_8x10s_k32": (8, 862, 1025, 32),
    "corpus_128x10s_k20": (128, 862, 1025, 20),
    "corpus_128x10s_k40": (128, 862, 1025, 40),
    "corpus_128x10s_k96": (128, 862, 1025, 96),
    "corpus_128x10s_k128": (128, 862, 1025, 128),
    "long_60s_k64": (1, 5168, 1025, 64),
    "single_10s_k128": (1, 862, 1025, 128),
    "ragged_64x40_equal_twin": (64, 889, 1025, 32),
    "ragged_256x100_equal_twin": (256, 594, 1025, 32),
    "rank_200_any_rank_path": (4, 862, 1025, 200),
}


def plan(lib, B, T, F, K):
    out = (ctypes.c_int64 * 32)()
    assert lib.fluhip_debug_plan_shape(B, T, F, K, out) == 0
    return dict(zip(KEYS, [int(v) for v in out]))


if __name__ == "__main__":
    lib = fluhip.load_library()
    fx = {name: {"shape": list(sh), "plan": plan(lib, *sh)} for name, sh in SHAPES.items()}
    path = os.path.join(ROOT, "tests", "golden", "plans_r06.json")
    with open(path, "w") as f:
        json.dump(fx, f, indent=1)
        f.write("\n")
    print(path, len(fx), "shapes")
