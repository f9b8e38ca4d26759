#!/usr/bin/env python3
"""Where in a kernel's code do the spill instructions sit? Position histogram (10 bins over the kernel's instructions) of
scratch_* (VGPR spills), v_writelane / v_readlane (SGPR spills live in VGPR lanes) and s_endpgm -- for the two-path
register-table kernels (general path + axis-aligned path in one kernel) this shows which path pays for the spills the
code object's metadata reports as a maximum over both.     usage: path_split.py [name pattern] [lib]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd", "libnmpc_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else "solve_kernel<float, 3, false, 14"
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "gfx950.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co], capture_output=True, text=True).stdout
cur, body = None, {}
for line in dis.splitlines():
    m = re.match(r"[0-9a-f]+ <(.*)>:", line)
    if m:
        cur = m.group(1) if pat in m.group(1) else None
        if cur:
            body[cur] = []
        continue
    if cur:
        t = line.strip().split()
        if t:
            body[cur].append(t[0])
for name, ops in body.items():
    n = len(ops)
    print(re.sub(r"\(nmpc::KParams.*", "", name).replace("void ", ""), f"-- {n} instructions")
    for label, test in (("scratch_*", lambda o: o.startswith("scratch_")), ("v_writelane", lambda o: o == "v_writelane_b32"),
                        ("v_readlane", lambda o: o == "v_readlane_b32"), ("s_endpgm", lambda o: o == "s_endpgm"),
                        ("v_mfma/none", lambda o: o.startswith("v_mfma"))):
        bins = [0] * 10
        for i, o in enumerate(ops):
            if test(o):
                bins[min(9, i * 10 // n)] += 1
        print(f"  {label:12s} total {sum(bins):5d}  by tenth of the code: {bins}")
