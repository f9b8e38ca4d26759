#!/usr/bin/env python3
"""Code size (bytes) and static instruction mix of the kernels in the shipped gfx950 code object.
   usage: code_size.py [pattern] [lib]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd", "libnmpc_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else "solve_kernel"
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "gfx950.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    sym = subprocess.run(["nm", "-S", "--demangle", co], capture_output=True, text=True).stdout
    names = {}
    for line in sym.splitlines():
        m = re.match(r"([0-9a-f]+) ([0-9a-f]+) [Tt] (.*)", line)
        if m and pat in m.group(3) and not m.group(3).endswith(".kd"):
            names[m.group(3)] = int(m.group(2), 16)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co], capture_output=True, text=True).stdout
    cur, mix = None, {}
    for line in dis.splitlines():
        m = re.match(r"[0-9a-f]+ <(.*)>:", line)
        if m:
            cur = m.group(1) if m.group(1) in names else None
            if cur: mix[cur] = {}
            continue
        if cur:
            t = line.strip().split()
            if not t: continue
            op = t[0]
            k = ("dpp" if "dpp" in line else "valu") if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"
            if op.startswith("scratch_"): k = "scratch"
            if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): k = "lane"
            if op == "ds_bpermute_b32": k = "bperm"
            mix[cur][k] = mix[cur].get(k, 0) + 1
    for n in sorted(names):
        short = re.sub(r"\(nmpc::KParams.*", "", n).replace("void ", "")
        print(f"{short:60s} {names[n]:7d} B  {mix.get(n)}")
