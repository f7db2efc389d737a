"""What CAN be compiled of the real reference in this image (oracle/Makefile -> oracle/_ref/refcheck: the dependency-free
headers of the path -- FluidTensor_Support strides / transpose, FluidTask progress arithmetic, Result, epsilon) against
the host-side mirrors of include/flucoma_hip/Types.hpp, line by line.  The algorithms themselves (STFT, NMF) need Eigen /
HISSTools and stay "parity unpinned" (DESIGN section 5)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "refcheck")


def _ref_lines():
    if os.path.isdir("/root/reference/include/flucoma"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/refcheck is not built and /root/reference is not here to build it from")
    return subprocess.run([REF], check=True, capture_output=True, text=True).stdout.splitlines()


def _mirror_lines(tmp_path):
    exe = tmp_path / "host_types_check"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "host_types_check.cpp"), "-o", str(exe)],
                   check=True, capture_output=True)
    return subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()


def test_host_mirrors_match_the_compiled_reference_headers(tmp_path):
    ref, mine = _ref_lines(), _mirror_lines(tmp_path)
    assert len(ref) >= 20
    assert mine[:len(ref)] == ref                       # every line the reference prints, identically
    assert mine[len(ref):] == ["abi_status 0 1 2 3"]    # clients/common/Result.hpp:24 order == fluhip_status
    # the strides the boundary is specified in: row-major T x F, transpose() swaps extents and strides
    assert "slice 862 1025 strides 1025 1 size 883550 | transpose extents 1025 862 strides 1 1025 | at(1,2) 1027 tat(2,1) 1027" in ref
    # epsilon of the clamps (util/AlgorithmUtils.hpp:19) is the constant both oracles and the kernels use
    import oracle_np
    assert ref[0] == "epsilon %.17g" % oracle_np.EPS
    kern = open(os.path.join(ROOT, "flucoma-core_amd", "csrc", "fluhip_kernels.h")).read()
    assert "kEpsilon = 2.220446049250313e-16" in kern and float("2.220446049250313e-16") == oracle_np.EPS
