"""Build libflucoma_hip.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU).

    python flucoma-core_amd/build.py [--force]

Outputs flucoma-core_amd/lib/libflucoma_hip.so (git-ignored, shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libflucoma_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
SOURCES = ["kernels_stft.hip", "kernels_stft2.hip", "kernels_nmf.hip", "kernels_nmf5.hip", "kernels_nmf_strip.hip", "kernels_nmf_wide.hip", "kernels_istft.hip", "kernels_feat.hip", "kernels_svd.hip", "api.hip", "api_pool.cpp"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result"]


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "flucoma_hip.h"))
    return deps


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


# per-file extras: the factor-update kernels never see NaNs by construction (every operand is
# clamped to >= eps or is a finite product of finite inputs), and fmax() without the sNaN
# canonicalisation saves one op per quotient on the FP64 datapath the MFMAs share.
EXTRA_FLAGS = {"kernels_nmf_wide.hip": ["-fno-honor-nans"], "kernels_nmf5.hip": ["-fno-honor-nans", "-Wno-inline-asm"], "kernels_nmf_strip.hip": ["-fno-honor-nans"], "kernels_nmf.hip": ["-fno-honor-nans"]}


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if _newer(obj, [path] + _deps() + [os.path.abspath(__file__)]):
        if src.endswith(".cpp"):   # host-only C++ above the C ABI: no device code, plain g++
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread", "-c", path, "-o", obj]
        else:
            cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(src, []), "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for s in srcs:
            o = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _newer(LIB, objs):
        tmp = LIB + f".tmp{os.getpid()}"
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-pthread", "-o", tmp, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB)
    return LIB


def build_host_tests() -> str:
    """g++ build of the C++ host-client test driver (plain C++17 above the C ABI)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "client_driver.cpp")
    out = os.path.join(LIBDIR, "client_driver")
    deps = [src] + [os.path.join(root, "include", "flucoma_hip", f)
                    for f in os.listdir(os.path.join(root, "include", "flucoma_hip"))]
    deps.append(os.path.join(root, "include", "flucoma_hip.h"))
    if _newer(out, deps) or _newer(out, [LIB]):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", src, "-o", out, "-L" + LIBDIR,
               "-lflucoma_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed for client_driver:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_host_tests())
