#!/bin/bash
# Diagnostic: marginal cost of an instruction in the headline kernel. Builds with -DNMPC_DBG_PAD_VALU=<n> / -DNMPC_DBG_PAD_SALU=<n>
# (n independent v_add_f32 / s_add_u32 per evaluation, dead results) next to the shipped library:
#   hipcc ... -DNMPC_DBG_PAD_VALU=40 -o build/libnmpc_v40.so dyobav-mpcnwta-warehouse_amd/csrc/nmpc_capi.hip   (v120, s120 likewise)
for v in "" v40 v120 s120; do
  if [ -n "$v" ]; then export NMPC_HIP_LIBRARY=$PWD/build/libnmpc_$v.so; else unset NMPC_HIP_LIBRARY; fi
  echo "== ${v:-shipped}"
  STAGED=-1 python tools/quick_rate.py cfg2 16384 2 2>&1 | tail -1 | sed 's/.*cfg2 B/cfg2 B/'
done
