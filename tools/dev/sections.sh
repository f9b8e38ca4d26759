#!/bin/bash
# usage: tools/dev/sections.sh [extra hipcc flags]  -- compile the headline kernel alone (marker build: -DNMPC_MARK turns the
# profile stamps into `s_nop 8+slot` / `s_sleep slot`) and print the static instruction counts per section of an
# evaluation / of the solver, the instruction mix, and the register figures. Seconds, no GPU.
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -fno-slp-vectorize -DNMPC_MARK "$@" --offload-device-only -S -o /tmp/headline_mark.s $R/tools/dev/headline_only.hip 2>/dev/null
python3 - <<'PY'
import re
names = {8: "entry", 9: "rollout", 10: "polygons + fleet", 11: "path segments + group min", 12: "obstacle passes + t0 groups", 13: "padding, control terms, cost sum", 14: "adjoint"}
prev, n, tot, mix = None, 0, 0, {}
for line in open("/tmp/headline_mark.s"):
    t = line.strip().split()
    if not t or not re.match(r"^(s_|v_|ds_|global_|scratch_|buffer_|flat_)", t[0]): continue
    op = t[0]
    mk = None
    if op == "s_nop" and t[1].isdigit() and int(t[1]) >= 8: mk = int(t[1])
    if op == "s_sleep": mk = 100 + int(t[1])
    tot += 1
    if mk is None:
        n += 1
        k = "dpp" if "dpp" in line else "valu" if op.startswith("v_") else "wait/nop" if op in ("s_waitcnt", "s_nop") else "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"
        mix[k] = mix.get(k, 0) + 1
        continue
    if prev is not None and mk != 8:
        print(f"   {n:5d} instructions up to marker {mk:3d}  {names.get(mk, 'solver stamp ' + str(mk - 100) if mk >= 100 else '')}")
    prev, n = mk, 0
print("total instructions", tot, mix)
PY
grep -E "\.(vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size):" /tmp/headline_mark.s | tr -s ' ' | tr '\n' ' '; echo
