#!/usr/bin/env python3
"""Register / spill / scratch figures of every kernel in the gfx950 code object of the SHIPPED library (the AMDGPU
metadata notes of the code object bundled in libnmpc_hip.so). tests/test_kernel_resources_cpu.py asserts on them so that
spills cannot creep back in unnoticed.    python tools/kernel_resources.py [path/to/libnmpc_hip.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd", "libnmpc_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except (OSError, subprocess.CalledProcessError):
        return list(names)


def kernel_resources(lib=DEFAULT_LIB):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "gfx950.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True,
                       capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
        syms = subprocess.run(["nm", "-S", co], check=True, capture_output=True, text=True).stdout
        # kernels that reserve a private segment without spilling a vector register: how many scratch instructions do they
        # really execute? (the backend sometimes reserves a 9-dword frame object next to the SGPR-spill lanes and never
        # touches it: private_segment_fixed_size 36, no scratch_* / buffer_* instruction in the kernel)
        phantom = {}
        for b in re.split(r"\n\s+- \.", notes[notes.find("amdhsa.kernels"):])[1:]:
            b = "." + b
            n = re.search(r"\.name:\s+(\S+)", b)
            g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, b) or [None, "-1"])[1])
            if n and g("private_segment_fixed_size") > 0 and g("vgpr_spill_count") == 0:
                dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn",
                                      f"--disassemble-symbols={n.group(1)}", co], check=True, capture_output=True, text=True).stdout
                phantom[n.group(1)] = len(re.findall(r"^\s+(scratch_|buffer_load|buffer_store)", dis, flags=re.M))
    size = {}   # mangled kernel name -> bytes of code
    for line in syms.splitlines():
        m = re.match(r"[0-9a-f]+ ([0-9a-f]+) [Tt] (\S+)$", line)
        if m and not m.group(2).endswith(".kd"):
            size[m.group(2)] = int(m.group(1), 16)
    meta = notes[notes.find("amdhsa.kernels"):]
    rows, names = {}, []
    for b in re.split(r"\n\s+- \.", meta)[1:]:
        b = "." + b
        n = re.search(r"\.name:\s+(\S+)", b)
        if not n:
            continue
        g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, b) or [None, "-1"])[1])
        names.append(n.group(1))
        rows[n.group(1)] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                                vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"),
                                scratch=g("private_segment_fixed_size"), lds_static=g("group_segment_fixed_size"),
                                code_bytes=size.get(n.group(1), -1),
                                # scratch instructions in the kernel's code; counted only where it matters (see above), else -1
                                scratch_instr=phantom.get(n.group(1), -1))
    return {d: rows[m] for m, d in zip(names, demangle(names))}


def short_name(name):
    name = re.sub(r"void \(anonymous namespace\)::", "", name)
    return re.sub(r"\(nmpc::.*", "", name)


if __name__ == "__main__":
    for name, r in sorted(kernel_resources(*(sys.argv[1:2])).items()):
        print(f"{short_name(name):58s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d}  vgpr_spill {r['vgpr_spill']:3d}"
              f"  sgpr_spill {r['sgpr_spill']:3d}  scratch {r['scratch']:4d} B  code {r['code_bytes']:6d} B")
