"""csrc/recip_tree.h on the host (no GPU): the reciprocals the factor updates' quotients are built from -- one v_rcp_f64 per group
of operands through a product tree (round 6) -- stay at the single reciprocal's accuracy (the seed's error squared by one Newton step, plus a
rounding per product) for every (operand count, group size) the kernels instantiate, over the operand range the updates see, at the clamp and
at the largest Q a float input can produce (groups of up to six stay inside the double range there; eight would not)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reciprocal_trees_keep_the_single_reciprocals_accuracy(tmp_path):
    exe = str(tmp_path / "recip_tree_host")
    src = os.path.join(ROOT, "tests", "cpp", "recip_tree_host.cpp")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    rows = [line.split() for line in out if line.strip()]
    assert [int(r[0]) for r in rows] == [1, 2, 3, 4, 6]
    single = max(float(v) for v in rows[0][1:])              # one reciprocal per operand: the modelled seed (2^-22) squared by the Newton step
    assert single < 6e-14, rows[0]
    for r in rows:
        errs = [float(v) for v in r[1:]]
        assert len(errs) == 9 and max(errs) < single * 1.05 + 2e-15, r      # (a rounding per product on top, nothing more)
