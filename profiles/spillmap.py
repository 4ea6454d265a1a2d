#!/usr/bin/env python3
"""Per-function register / scratch map of the device code of libtalc_hip.so.

Compiles talc_capi.hip for gfx950 to assembly (device only, the flags of talc_amd/build.py) and, for every
function in it, reports the instruction count, the scratch (private segment) loads and stores — register
spills and call-preserved saves — and the function's own resource notes from the assembler comments.
    python profiles/spillmap.py [extra hipcc flags] > profiles/rNN/spillmap.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "talc.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fopenmp", "-ffp-contract=off",
               "-fno-gpu-rdc", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-command-line-argument",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "talc_amd", "csrc"), *extra,
               os.path.join(ROOT, "talc_amd", "csrc", "talc_capi.hip"), "--cuda-device-only", "-S", "-o", asm]
        subprocess.check_call(cmd)
        lines = open(asm).read().splitlines()
    fn = None
    stats = {}
    order = []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            fn = m.group(1)
            stats[fn] = dict(insts=0, sst=0, sld=0, notes={})
            order.append(fn)
            continue
        if fn is None:
            continue
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            continue
        m = re.match(r";\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|sgpr_spill_count|vgpr_spill_count|codeLenInByte):?\s*(\d+)", s) or \
            re.match(r";\s*(SGPRs|VGPRs|ScratchSize|VGPR Spill|SGPR Spill|codeLenInByte)[^:]*:\s*(\d+)", s)
        if m:
            stats[fn]["notes"][m.group(1)] = int(m.group(2))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        stats[fn]["insts"] += 1
        op = s.split()[0]
        if op.startswith("scratch_store") or (op.startswith("buffer_store") and "offen" in s or op.startswith("buffer_store") and "s[0:3]" in s):
            stats[fn]["sst"] += 1
        if op.startswith("scratch_load") or (op.startswith("buffer_load") and "s[0:3]" in s):
            stats[fn]["sld"] += 1
    dem = subprocess.run(["c++filt"], input="\n".join(order), capture_output=True, text=True).stdout.splitlines()
    print("%-70s %8s %8s %8s  %s" % ("function", "insts", "scr_st", "scr_ld", "notes"))
    for f, d in zip(order, dem):
        st = stats[f]
        name = re.sub(r"\(.*", "", d)
        print("%-70s %8d %8d %8d  %s" % (name[:70], st["insts"], st["sst"], st["sld"],
                                       " ".join("%s=%d" % kv for kv in sorted(st["notes"].items()))))


if __name__ == "__main__":
    main()
