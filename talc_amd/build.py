"""Build driver for the native parts of talc_amd (in-tree, no JIT cache).

Outputs go to talc_amd/_build/ (git-ignored, shipped to the GPU box by gpurun):
  libtalc_synth.so  host-only synthetic data generator          (g++)
  libtalc_hip.so    C-ABI library: GPU k-mer table + HIP kernels (hipcc --offload-arch=gfx950)
  talc              drop-in CLI (reference main.cpp surface)     (hipcc, links libtalc_hip.so)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _deps(*names):
    out = []
    for n in names:
        out.append(os.path.join(CSRC, n))
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hpp", ".cuh", ".hip.h")):
            out.append(os.path.join(CSRC, f))
    for f in os.listdir(INCLUDE):
        out.append(os.path.join(INCLUDE, f))
    return out


def build_synth(force=False):
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, "libtalc_synth.so")
    src = os.path.join(CSRC, "synth.cpp")
    if force or _newer(tgt, [src]):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", src, "-o", tgt])
    return tgt


def build_pure(force=False):
    """Host build of the host/device-pure product pieces for the CPU test-suite."""
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, "libtalc_pure.so")
    if force or _newer(tgt, _deps("pure_capi.cpp")):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-I", INCLUDE, "-I", CSRC,
              os.path.join(CSRC, "pure_capi.cpp"), "-o", tgt])
    return tgt


HIP_SOURCES = ["talc_capi.hip"]


def build_hip(force=False, extra_flags=(), name="libtalc_hip.so"):
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, name)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    if force or _newer(tgt, _deps(*HIP_SOURCES)):
        _run([HIPCC, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-fPIC", "-shared", "-fopenmp",
              "-ffp-contract=off", "-fgpu-rdc" if False else "-fno-gpu-rdc",
              "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-I", INCLUDE, "-I", CSRC, *extra_flags, *srcs, "-o", tgt])
    return tgt


def build_hip_prof(force=False):
    """Developer build with the in-kernel s_memtime category profiler (select it with TALC_LIB=..., print
    with TALC_PROF_PRINT=1); never loaded by default."""
    return build_hip(force, ("-DTALC_PROF",), "libtalc_hip_prof.so")


def build_cli(force=False):
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, "talc")
    src = os.path.join(CSRC, "talc_main.cpp")
    if not os.path.exists(src):
        return None
    if force or _newer(tgt, _deps("talc_main.cpp")):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-fopenmp", "-I", INCLUDE, "-I", CSRC, src, "-o", tgt,
              "-L", OUT, "-ltalc_hip", "-Wl,-rpath,$ORIGIN"])
    return tgt


def build_oracle():
    """The CPU oracle is test infrastructure; building the checker is not using it."""
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_all(force=False):
    build_synth(force)
    build_pure(force)
    build_hip(force)
    build_cli(force)
    build_oracle()


if __name__ == "__main__":
    if "--prof" in sys.argv:
        build_hip_prof(force="--force" in sys.argv)
    else:
        build_all(force="--force" in sys.argv)
