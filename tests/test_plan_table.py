"""The schedule of the factor updates as DATA: `fluhip_debug_plan_shape` is the planner's whole answer for a
shape -- api_corpus.hip decide_update_plan, the one function fluhip_corpus_create plans from -- without a device.  Here:

  * the pinned plans of the BASELINE shapes and of every row of tools/perf_matrix.py (tests/golden/plans_r06.json, minted by
    tools/make_plan_fixture.py; the fields the GPU runs of profiles/r06/perf_matrix_v0.json recorded agree with it);
  * over a few hundred generated shapes: a plan exists, it is internally consistent, and every workspace it sizes bounds every
    index the launches of that plan form (the class of ADVICE r04's out-of-bounds: an H update of more than 64 strips wrote its
    side partials past their area);
  * the norm combine's form comes out of ONE table (kernels_nmf.hip kWnormForms): its regimes pinned at their boundaries.

Pure host code: no GPU.
"""
import ctypes
import itertools
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_plan_fixture import KEYS, SHAPES  # noqa: E402


@pytest.fixture(scope="module")
def lib(fluhip_lib_path):
    import fluhip
    return fluhip.load_library(fluhip_lib_path)


def plan(lib, B, T, F, K):
    out = (ctypes.c_int64 * 32)()
    assert lib.fluhip_debug_plan_shape(B, T, F, K, out) == 0, (B, T, F, K)
    return dict(zip(KEYS, [int(v) for v in out]))


def test_pinned_plans_of_the_baseline_and_perf_matrix_shapes(lib):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "plans_r06.json")))
    assert set(fx) == set(SHAPES)
    for name, e in fx.items():
        assert tuple(e["shape"]) == SHAPES[name]
        assert plan(lib, *e["shape"]) == e["plan"], name
    # the named regimes, spelled out (what DESIGN's "which shapes reach which kernels" table says)
    p = fx["c4_shard_128x10s_k32"]["plan"]
    assert (p["split_w"], p["split_h"], p["side_column"], p["strips_w"], p["strips_h"], p["lists"], p["h_takes"]) == (1, 1, 1, 8, 8, 0, 7)
    assert fx["c2_60s_k16"]["plan"]["strip"] == 1 and fx["c1_bufnmf_loop_k3"]["plan"]["strip"] == 1
    p = fx["c3_2x10min_k128_fft4096"]["plan"]
    assert (p["split_w"], p["tail_split_h"], p["lists"], p["strip"]) == (8, 3, 0, 0)
    assert fx["corpus_8x10s_k32"]["plan"]["lists"] == 1 and fx["c4_1024x10s_k32_one_gpu"]["plan"]["lists"] == 1
    assert fx["rank_200_any_rank_path"]["plan"]["variant"] == 0


def _round_up(x, m):
    return (x + m - 1) // m * m


def _max_groups(Kp):
    return 9 if Kp <= 32 else (4 if Kp <= 64 else 2)


def check_plan(p, B, T, F, K):
    Kp, Kc = p["padded_rank"], p["compute_rank"]
    Tp, Fp = _round_up(T, 32), _round_up(F, 32)
    assert Kp >= K and Kp % 16 == 0 and (Kp in (16, 32, 64, 128) or Kp > 128)
    assert K <= Kc <= Kp
    assert p["variant"] == (5 if Kp <= 128 else 0)
    if p["variant"] == 0:
        assert p["wide"] > 0 and not p["lists"] and not p["strip"] and not p["deferred_norm"]
        return
    assert p["deferred_norm"] == 1
    assert 1 <= p["split_w"] <= 64 and 1 <= p["split_h"] <= 64
    assert p["wscratch"] > 0 and p["dpart"] >= 32
    slots = p["side_slots"]
    # the layout of wscratch: statistics records, then two generations of side partials + old side rows, the pre-reduced records
    assert p["stat_doubles"] == B * p["strips_w"] * 2 * Kp
    assert p["wscratch"] >= p["stat_doubles"] + B * (4 * slots * 2 * Kp + Kp) + B
    if p["strip"]:
        assert B == 1 or os.environ.get("FLUHIP_STRIP") == "1"
        assert Kp == 16 and not p["lists"] and not p["side_column"] and p["strippart"] > 0
        return
    if p["side_column"]:
        assert F % 16 == 1 and F > 16
    C = F - p["side_column"]
    if p["lists"]:
        # (the descriptors themselves: tests/test_list_plan.py) -- every partial slot the lists announce fits the workspace
        assert p["strips_w"] >= 1 and p["part"] % (max(Fp, Tp) * Kp) == 0 and p["dpart"] >= B * Kp
        return
    # uniform forms: the strips cover the columns, within the widest strip the kernel is built for
    G, GH = (C + 15) // 16, (T + 15) // 16
    if p["split_w"] == 1:
        assert p["strips_w"] * _max_groups(Kp) >= G and p["strips_w"] <= G
    assert p["strips_h"] * _max_groups(Kp) >= GH and 1 <= p["strips_h"] <= GH
    ns = max(p["split_w"], p["split_h"], p["tail_split_h"])
    if p["split_w"] > 1:
        assert p["part"] >= B * p["split_w"] * Fp * Kp
    if p["split_h"] > 1:
        assert p["part"] >= B * p["split_h"] * Tp * Kp
    if p["tail_split_h"] > 1:
        assert p["split_h"] == 1
        assert p["tail_strips_h"] % 4 == 0 and p["tail_strips_h"] >= 4 and p["tail_rest_h"] >= 1
        assert p["tail_strips_h"] + p["tail_rest_h"] == p["strips_h"]
        assert 0 < p["tail_cols_h"] < T and p["tail_cols_h"] % 16 == 0
        assert p["part"] >= B * p["tail_split_h"] * _round_up(T - p["tail_cols_h"], 32) * Kp
    assert p["dpart"] >= B * max(ns, 1 + p["tail_split_h"]) * Kp
    assert (p["csum"] > 0) == (Kp > 64)
    takes = p["h_takes"]
    assert takes in (0, 1, 3, 7)
    if takes:
        assert p["side_column"] and p["split_h"] == 1 and p["tail_split_h"] <= 1 and Kp <= 64
    if takes & 1:
        assert p["strips_h"] <= slots, "side partials of the H update past their area"
    if takes & 2:
        assert p["strips_w"] <= 16 and p["strips_h"] <= 16 and p["combine_form"] == -1
    else:
        assert 0 <= p["combine_form"] <= 4
    if takes & 4:
        assert Kp == 32 and Kc == 32 and p["colpart"] >= B * p["strips_w"] * Kp
    if p["side_column"] and not (takes & 1):
        assert 16 <= p["side_slices"] <= 4 * slots     # the side-column launch has the whole side area: 4 x 64 records per buffer


def generated_shapes():
    Bs = (1, 2, 3, 4, 8, 16, 40, 64, 100, 128, 200, 256, 520, 1024)
    Ts = (5, 87, 173, 431, 862, 887, 2584, 5168, 25840)
    Fs = (33, 129, 513, 1025, 2049, 4097)
    Ks = (1, 3, 8, 16, 17, 20, 24, 32, 33, 40, 56, 64, 72, 100, 112, 128, 130, 200)
    rs = np.random.RandomState(606)
    grid = list(itertools.product(Bs, Ts, Fs, Ks))
    pick = [grid[i] for i in rs.choice(len(grid), 420, replace=False)]
    # keep what a 288 GB part holds (two layouts of V + factors) and what the any-rank path accepts
    out = []
    for B, T, F, K in pick:
        if 2.2 * B * _round_up(T, 32) * _round_up(F, 32) * 8 > 200e9:
            continue
        if K > 128 and max(T, F) > 65535:
            continue
        out.append((B, T, F, K))
    return out


def test_generated_shapes_have_consistent_plans_and_bounded_workspaces(lib):
    shapes = generated_shapes() + list(SHAPES.values())
    assert len(shapes) >= 250
    kinds = {"uniform": 0, "split": 0, "lists": 0, "strip": 0, "tail": 0, "wide": 0, "h_takes_7": 0}
    for B, T, F, K in shapes:
        p = plan(lib, B, T, F, K)
        try:
            check_plan(p, B, T, F, K)
        except AssertionError as e:
            raise AssertionError(f"shape B={B} T={T} F={F} K={K}: plan {p}: {e}") from e
        kinds["wide"] += p["variant"] == 0
        kinds["lists"] += p["lists"]
        kinds["strip"] += p["strip"] > 0
        kinds["tail"] += p["tail_split_h"] > 1
        kinds["split"] += (not p["lists"]) and (p["split_w"] > 1 or p["split_h"] > 1)
        kinds["uniform"] += p["variant"] == 5 and not p["lists"] and not p["strip"] and p["split_w"] == 1 and p["split_h"] == 1
        kinds["h_takes_7"] += p["h_takes"] == 7
    assert all(v > 0 for v in kinds.values()), kinds     # the grid reaches every schedule family
    print(kinds)


def test_norm_combine_forms_come_from_one_table(lib):
    f = lib.fluhip_debug_wnorm_form
    # (Kp, buffers, parts, slices, side rows, side phase, column sums wanted) -> form
    # corpora of short buffers: side column + combine in one launch; not for few buffers, many parts or long factors
    assert f(32, 128, 8, 16, 862, 0, 0) == 0
    assert f(32, 63, 8, 16, 862, 0, 0) == 4 and f(32, 128, 65, 16, 862, 0, 0) == 4 and f(128, 128, 8, 16, 1025, 0, 0) == 4
    assert f(32, 128, 8, 16, 862, 1, 0) == 1                  # the side-column launch alone
    # the bench shard's steady state when the H update does not take the combine: eight parts, eight slices, 256 threads
    assert f(32, 128, 8, 8, 862, 2, 0) == 4
    # many parts but short records (one 10 s buffer at rank 32: 129 records of 512 B): one workgroup of 1024 threads
    assert f(32, 1, 129, 0, 0, 0, 0) == 3
    # config 3: 512 parts + 256 slices at rank 128 -> pre-reduction; just under the record threshold -> 1024 threads
    assert f(128, 2, 512, 256, 25840, 2, 0) == 2
    assert f(128, 2, 200, 112, 25840, 2, 0) == 3 and f(128, 2, 200, 113, 25840, 2, 0) == 2      # (parts + slices) Kp > 40000
    # windows of a rank-128 corpus: the pre-reduction for the column sums alone
    assert f(128, 16, 32, 16, 862, 2, 1) == 2 and f(128, 16, 32, 16, 862, 2, 0) == 4
