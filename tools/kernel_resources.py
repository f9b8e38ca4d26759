#!/usr/bin/env python3
"""Register / spill / scratch figures of every kernel in the gfx950 code object (from the AMDGPU metadata that
hipcc emits with -S). Used by tests/test_kernel_resources_cpu.py so that spills cannot creep back in unnoticed.
   python tools/kernel_resources.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd", "csrc", "nmpc_capi.hip")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except (OSError, subprocess.CalledProcessError):
        return list(names)


def kernel_resources(extra_flags=()):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                        "-fno-slp-vectorize", "-Wno-unused-function", *extra_flags, "-o", asm, SRC], check=True, capture_output=True)
        txt = open(asm).read()
    meta = txt[txt.rfind("amdhsa.kernels"):]
    rows = {}
    names = []
    for b in meta.split("\n  - ")[1:]:
        n = re.search(r"\.name:\s+(\S+)", b)
        if not n:
            continue
        g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, b) or [None, "-1"])[1])
        names.append(n.group(1))
        rows[n.group(1)] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), vgpr_spill=g("vgpr_spill_count"),
                                sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                                lds_static=g("group_segment_fixed_size"))
    return {d: rows[m] for m, d in zip(names, demangle(names))}


if __name__ == "__main__":
    for name, r in sorted(kernel_resources(sys.argv[1:]).items()):
        short = re.sub(r"void \(anonymous namespace\)::", "", name)
        short = re.sub(r"\(nmpc::.*", "", short)
        print(f"{short:58s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d}  vgpr_spill {r['vgpr_spill']:3d}  sgpr_spill {r['sgpr_spill']:3d}  scratch {r['scratch']:4d} B")
