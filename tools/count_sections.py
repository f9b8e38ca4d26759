#!/usr/bin/env python3
"""Static instruction counts per section of an evaluation / of the solver in a marker build (-DNMPC_MARK: the profile
stamps become `s_nop 8+slot` / `s_sleep slot`, never run). Straight-line sections are what they cost per evaluation; loops
(path segments, polygons, two-loop recursion, t = 0 groups) count once here and run several times.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -fno-slp-vectorize -DNMPC_MARK \\
         -o build/libnmpc_mark.so dyobav-mpcnwta-warehouse_amd/csrc/nmpc_capi.hip
   python tools/count_sections.py [kernel substring]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "libnmpc_mark.so")
pat = sys.argv[1] if len(sys.argv) > 1 else "solve_kernel<float, 3, false, 14, 0>"
LLVM = "/opt/rocm/lib/llvm/bin"
names = {8: "entry (constants, counters)", 9: "rollout", 10: "polygons + fleet", 11: "path segments + group min", 12: "obstacle passes (both variants) + t = 0 groups",
         13: "padding, control terms, cost sum", 14: "adjoint"}
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co], capture_output=True, text=True).stdout
cur, prev, n = False, None, 0
for line in dis.splitlines():
    m = re.match(r"[0-9a-f]+ <(.*)>:", line)
    if m:
        cur = pat in m.group(1); prev = None
        if cur: print(m.group(1)[:100])
        continue
    if not cur: continue
    t = line.strip().split()
    if not t: continue
    mk = None
    if t[0] == "s_nop" and t[1].isdigit() and int(t[1]) >= 8: mk = int(t[1])
    if t[0] == "s_sleep": mk = 100 + int(t[1])
    if mk is None:
        n += 1; continue
    if prev is not None and mk != 8:
        print(f"   {n:5d} instructions up to marker {mk:3d}  {names.get(mk, 'solver stamp ' + str(mk - 100) if mk >= 100 else '')}")
    prev, n = mk, 0
