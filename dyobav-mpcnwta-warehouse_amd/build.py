"""Compile ``libnmpc_hip.so`` (gfx950 code object + C ABI) in-tree with hipcc.

Plays the role of ``cargo build`` behind the reference's ``solver_build.py`` (``/root/reference/src/solver_build.py:22-27``
-> ``MpcModule.build`` -> ``OpEnOptimizerBuilder.build()``): after this step a solver for *any* yaml dimension set
exists, because the dimensions are run-time arguments of the kernels rather than constants baked into generated
code.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# NMPC_HIP_LIBRARY: load an alternative build of the same C ABI (A/B tests of kernel variants)
LIB_PATH = os.environ.get("NMPC_HIP_LIBRARY") or os.path.join(PKG_DIR, "libnmpc_hip.so")
SOURCES = ("nmpc_capi.hip",)
HEADERS = ("nmpc_device.h", "nmpc_spec.h", "nmpc_assemble.h", "nmpc_hypotheses.h", "nmpc_step.h", "wave_ops.h", os.path.join("..", "..", "include", "nmpc_hip.h"))
# -fno-slp-vectorize: packed fp32 VALU ops (v_pk_fma_f32 ...) issue at half the rate of plain ones on gfx950 (measured,
# tools/mb/issue_rate.hip), so they buy nothing once two wavefronts share a SIMD, but they need their operands in aligned
# register pairs -- with the register-resident obstacle table that costs ~40 VGPRs in copies and the second wavefront.
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
               "-fno-slp-vectorize", "-Wall", "-Wno-unused-function")


def find_hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 solver library)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Build the shared library if it is missing or older than its sources; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [find_hipcc(), *HIPCC_FLAGS, "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    if verbose and proc.stderr:
        print(proc.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
