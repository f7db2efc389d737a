"""Build libflucoma_hip.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU).

    python flucoma-core_amd/build.py [--force]

Outputs flucoma-core_amd/lib/libflucoma_hip.so (git-ignored, shipped to the GPU box by gpurun): the production library,
one schedule per shape, no experiment switches.  build_ab() makes the two measurement builds beside it:

    lib_ab/libflucoma_hip_ab.so   -DFLUHIP_AB_SWITCHES: the FLUHIP_* environment switches of DESIGN section 6b are live
                                  (csrc/fluhip_env.h); tests/test_gpu_variants.py and the tools/ A/B scripts load it
                                  (FLUHIP_AB=1 or FLUHIP_LIB=<path> for the Python binding)
    lib_ab/libflucoma_hip_qc.so   the same with -DFLUHIP_QUOTIENT_CORRECTION=1: the factor-update quotients with the residual
                                  correction (rounding-level instead of <= 2^-46), for the two-quotient comparison
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
ABDIR = os.path.join(HERE, "lib_ab")
LIB_AB = os.path.join(ABDIR, "libflucoma_hip_ab.so")
LIB_QC = os.path.join(ABDIR, "libflucoma_hip_qc.so")
# variant -> (object directory, extra defines, files compiled for it; the others are taken from `base`), base variant
VARIANTS = {
    "default": (OBJ, [], None, None),
    "ab": (os.path.join(HERE, "build_ab"), ["-DFLUHIP_AB_SWITCHES"], None, None),
    "qc": (os.path.join(HERE, "build_qc"), ["-DFLUHIP_AB_SWITCHES", "-DFLUHIP_QUOTIENT_CORRECTION=1"],
           ["kernels_nmf5.hip", "kernels_nmf5_off.hip", "kernels_nmf_strip.hip"], "ab"),
}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
SOURCES = ["kernels_stft.hip", "kernels_stft2.hip", "kernels_nmf.hip", "kernels_nmf5.hip", "kernels_nmf5_off.hip", "kernels_nmf_strip.hip", "kernels_nmf_bintile.hip", "kernels_nmf_wide.hip", "kernels_istft.hip", "kernels_feat.hip", "kernels_svd.hip", "api_core.hip", "api_corpus.hip", "api_algorithms.hip", "api_features.hip", "api_frames.hip", "api_pool.cpp"]
# kernel forms no production shape reaches are not part of the production library (kernels_nmf_bintile.hip was 415
# lines of dead weight in lib/libflucoma_hip.so): compiled into the measurement builds only
AB_ONLY = {"kernels_nmf_bintile.hip"}
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
EXTRA_FLAGS = {"kernels_nmf_wide.hip": ["-fno-honor-nans"], "kernels_nmf5.hip": ["-fno-honor-nans", "-Wno-inline-asm"], "kernels_nmf5_off.hip": ["-fno-honor-nans", "-Wno-inline-asm"], "kernels_nmf_strip.hip": ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form"], "kernels_nmf.hip": ["-fno-honor-nans"],
               "kernels_nmf_bintile.hip": ["-fno-honor-nans", "-Wno-inline-asm", "-mllvm", "-amdgpu-mfma-vgpr-form"]}


# sources a file includes beside the headers (the off-size half of the factor-update kernel's instantiation list)
INCLUDES = {"kernels_nmf5_off.hip": ["kernels_nmf5.hip"]}


def _compile(src, variant="default"):
    objdir, defines, only, base = VARIANTS[variant]
    if only is not None and src not in only:
        return _compile(src, base)
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    extra = [os.path.join(CSRC, f) for f in INCLUDES.get(src, [])]
    if _newer(obj, [path] + extra + _deps() + [os.path.abspath(__file__)]):
        if src.endswith(".cpp"):   # host-only C++ above the C ABI: no device code, plain g++
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread", *defines, "-c", path, "-o", obj]
        else:
            cmd = [HIPCC, *CXXFLAGS, *defines, *EXTRA_FLAGS.get(src, []), "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def _build_variant(variant: str, lib: str, force: bool = False) -> str:
    objdir = VARIANTS[variant][0]
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    root = variant
    while VARIANTS[root][3] is not None:
        root = VARIANTS[root][3]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s)) and not (root == "default" and s in AB_ONLY)]
    if force:
        for s in srcs:
            o = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=max(4, (os.cpu_count() or 4) - 2)) as ex:
        objs = list(ex.map(lambda s: _compile(s, variant), srcs))
    if force or _newer(lib, objs):
        tmp = lib + f".tmp{os.getpid()}"
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-pthread", "-o", tmp, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, lib)
    return lib


def build(force: bool = False) -> str:
    return _build_variant("default", LIB, force)


def build_ab(force: bool = False):
    """the two measurement builds (see the module docstring); returns (ab, qc)"""
    return _build_variant("ab", LIB_AB, force), _build_variant("qc", LIB_QC, force)


def build_exp(name: str, defines, files, base: str = "default") -> str:
    """an experiment build for one-session A/B runs: `files` recompiled with `defines` into build_<name>/, every other object
    taken from `base`; the library goes to lib_ab/libflucoma_hip_<name>.so (FLUHIP_LIB=<path> loads it)"""
    VARIANTS[name] = (os.path.join(HERE, "build_" + name), list(defines), list(files), base)
    return _build_variant(name, os.path.join(ABDIR, f"libflucoma_hip_{name}.so"))


def build_host_tests() -> str:
    """g++ build of the C++ host-client test driver (plain C++17 above the C ABI)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "client_driver.cpp")
    out = os.path.join(LIBDIR, "client_driver")
    deps = [src] + [os.path.join(root, "include", "flucoma_hip", f)
                    for f in os.listdir(os.path.join(root, "include", "flucoma_hip"))]
    deps.append(os.path.join(root, "include", "flucoma_hip.h"))
    if _newer(out, deps) or _newer(out, [LIB]):
        # -DFLUHIP_AB_SWITCHES: the test driver keeps the client's channel-by-channel switch (FLUHIP_CLIENT_SEQUENTIAL)
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-DFLUHIP_AB_SWITCHES", src, "-o", out, "-L" + LIBDIR,
               "-lflucoma_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed for client_driver:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return out


if __name__ == "__main__":
    if "--exp" in sys.argv:   # python build.py --exp NAME -DX=1 [-DY] file.hip [file.hip ...]
        i = sys.argv.index("--exp")
        rest = sys.argv[i + 2:]
        print(build_exp(sys.argv[i + 1], [a for a in rest if a.startswith("-D")], [a for a in rest if not a.startswith("-")]))
        sys.exit(0)
    print(build(force="--force" in sys.argv))
    print(build_host_tests())
    if "--ab" in sys.argv:
        print(*build_ab(force="--force" in sys.argv))
