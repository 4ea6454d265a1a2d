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


def _digest(sources, cmd):
    """Content hash of every source a target depends on plus the command line that builds it."""
    import hashlib
    # (paths relative to the repo: the same tree gives the same stamp wherever it is checked out, so a binary built here
    #  is recognised on the GPU box, whose copy lives under another path)
    h = hashlib.sha256(" ".join(cmd).replace(ROOT, "$ROOT").encode())
    for s in sorted(sources):
        if os.path.exists(s):
            h.update(os.path.basename(s).encode())
            with open(s, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _build_if_stale(target, sources, cmd, force=False):
    """Rebuild `target` unless it exists and was built from exactly these sources with this command.  The stamp
    (<target>.srchash) holds content hashes, not mtimes: a snapshot copied to the GPU box keeps no useful mtimes,
    and a binary older than HEAD must never be what the tests run."""
    stamp = target + ".srchash"
    want = _digest(sources, cmd)
    if not force and os.path.exists(target) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == want:
                return False
    _run(cmd)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return True


def source_hash(target_name="libtalc_hip.so"):
    """The stamp of a built target (None if absent): recorded in bench.py's JSON line."""
    try:
        with open(os.path.join(OUT, target_name + ".srchash")) as f:
            return f.read().strip()
    except OSError:
        return None


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _deps(*names):
    out = []
    for n in names:
        out.append(os.path.join(CSRC, n))
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hpp", ".cuh", ".inc", ".hip")):   # every header and every included source
            out.append(os.path.join(CSRC, f))
    for f in os.listdir(INCLUDE):
        out.append(os.path.join(INCLUDE, f))
    return out


def build_synth(force=False):
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, "libtalc_synth.so")
    src = os.path.join(CSRC, "synth.cpp")
    _build_if_stale(tgt, [src], ["g++", "-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-Wall", src, "-o", tgt], force)
    return tgt


def build_pure(force=False):
    """Host build of the host/device-pure product pieces for the CPU test-suite."""
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, "libtalc_pure.so")
    _build_if_stale(tgt, _deps("pure_capi.cpp"),
                    ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-I", INCLUDE, "-I", CSRC,
                     os.path.join(CSRC, "pure_capi.cpp"), "-o", tgt], force)
    return tgt


HIP_SOURCES = ["talc_capi.hip"]


def build_hip(force=False, extra_flags=(), name="libtalc_hip.so"):
    os.makedirs(OUT, exist_ok=True)
    tgt = os.path.join(OUT, name)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    _build_if_stale(tgt, _deps(*HIP_SOURCES),
                    [HIPCC, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-fPIC", "-shared", "-fopenmp",
                     "-ffp-contract=off", "-fno-gpu-rdc",
                     "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-pass-failed", "-I", INCLUDE, "-I", CSRC, *extra_flags, *srcs, "-o", tgt],
                    force)
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
    _build_if_stale(tgt, _deps("talc_main.cpp"),
                    ["g++", "-std=c++17", "-O2", "-Wall", "-fopenmp", "-I", INCLUDE, "-I", CSRC, src, "-o", tgt,
                     "-L", OUT, "-ltalc_hip", "-Wl,-rpath,$ORIGIN"], force)
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


def build_hip_variants():
    """Experiment builds: other occupancy targets for k_search (select one with TALC_LIB=...)."""
    out = []
    for w in (4, 6):
        out.append(build_hip(False, ("-DTALC_SEARCH_WAVES_PER_SIMD=%d" % w,), "libtalc_hip_w%d.so" % w))
    return out


if __name__ == "__main__":
    if "--variants" in sys.argv:
        build_hip_variants()
    elif "--prof" in sys.argv:
        build_hip_prof(force="--force" in sys.argv)
    else:
        build_all(force="--force" in sys.argv)
