"""sha256 of the .text / .rodata sections of the gfx950 code object inside one or more builds of libnmpc_hip.so.

    python tools/code_object_hash.py [lib.so ...]          (default: the in-tree library)

Two builds whose hashes agree run the same device code: the check behind "moved out of the product headers without
changing a single instruction" (profiles/r06_hygiene_code_object_identity.txt) and behind any A/B claim of "same bits".
Host-side changes (layout bookkeeping, the C ABI) do not show up here by construction.
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd", "libnmpc_hip.so")


def code_object_hashes(lib):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "gfx950.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True,
                       capture_output=True)
        out = {}
        for sec in (".text", ".rodata"):
            raw = os.path.join(td, sec.strip(".") + ".bin")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", f"--only-section={sec}", co, raw], check=True)
            data = open(raw, "rb").read()
            out[sec] = (len(data), hashlib.sha256(data).hexdigest())
        # per kernel: the bytes of each function symbol of .text (a kernel whose instructions did not change keeps its hash
        # even when a neighbour grew or shrank; branch offsets inside a function are relative)
        text = open(os.path.join(td, "text.bin"), "rb").read()
        base = None
        for line in subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", co], check=True, capture_output=True,
                                   text=True).stdout.splitlines():
            m = re.search(r"\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)", line)
            if m:
                base = int(m.group(1), 16)
        kernels = {}
        for line in subprocess.run(["nm", "-S", co], check=True, capture_output=True, text=True).stdout.splitlines():
            m = re.match(r"([0-9a-f]+) ([0-9a-f]+) [Tt] (\S+)$", line)
            if m and not m.group(3).endswith(".kd") and base is not None:
                a, n = int(m.group(1), 16) - base, int(m.group(2), 16)
                kernels[m.group(3)] = (n, hashlib.sha256(text[a:a + n]).hexdigest())
        out["kernels"] = kernels
    return out


def demangle(names):
    try:
        return subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.strip().split("\n")
    except (OSError, subprocess.CalledProcessError):
        return list(names)


if __name__ == "__main__":
    libs = sys.argv[1:] or [DEFAULT_LIB]
    seen = []
    for lib in libs:
        h = code_object_hashes(lib)
        seen.append(h)
        print(lib)
        for sec in (".text", ".rodata"):
            n, digest = h[sec]
            print(f"  {sec:8s} {n:9d} B  sha256 {digest}")
        print(f"  {len(h['kernels'])} kernels")
    if len(seen) == 2:
        a, b = seen[0]["kernels"], seen[1]["kernels"]
        same = sorted(k for k in a if k in b and a[k] == b[k])
        diff = sorted(k for k in a if k in b and a[k] != b[k])
        only = sorted(set(a) ^ set(b))
        print(f"kernels byte-identical: {len(same)} of {len(set(a) | set(b))}")
        for k, d in zip(diff, demangle(diff)):
            print(f"  differs: {d}  ({a[k][0]} B -> {b[k][0]} B)")
        for k, d in zip(only, demangle(only)):
            print(f"  only in one build: {d}")
        whole = seen[0][".text"] == seen[1][".text"] and seen[0][".rodata"] == seen[1][".rodata"]
        print("IDENTICAL device code" if whole else "device code differs in the kernels listed above")
        sys.exit(0 if whole else 1)
