"""CPU test: register / spill figures of the kernels in the SHIPPED gfx950 code object (read from the AMDGPU metadata
notes of the code object inside libnmpc_hip.so). Round 1 shipped 120-300 spilled SGPRs per solve kernel (v_writelane /
v_readlane traffic on the critical path); the budgets below keep that from coming back."""
import os
import re
import sys

import pytest

import dyobav_mpcnwta_warehouse_amd as nm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def resources():
    import kernel_resources
    nm.build_library()
    res = {kernel_resources.short_name(k): v for k, v in kernel_resources.kernel_resources(nm.library_path()).items()}
    assert len(res) >= 40
    return res


def _sel(res, pattern):
    out = {k: v for k, v in res.items() if re.search(pattern, k)}
    assert out, pattern
    return out


def test_fp32_throughput_kernels_do_not_spill(resources):
    # The register-table kernels are PAIRS since round 4: <.., 1> = the axis-aligned path alone -- what the reference's inputs
    # (angle = 0, main_base.py:302) take and what bench.py times -- and <.., 2> = the general (rotated-ellipse) path alone.
    # The axis-aligned members must be free of VGPR spills and scratch; the general path of the 14-slot kernel keeps a few
    # table values in scratch (written once in load(), read once per evaluation).
    # (Round 5: the 6-slot kernels -- 13..18 provisioned rows, the shipped yaml's 15 -- hold half as many table rows again at
    #  the 4-slot kernels' budget of 168 registers (three wavefronts per SIMD): 6 / 21 VGPRs in scratch, and still +18 % at
    #  B = 65 536 and +35 % at B = 1024 over the 256-register 14-slot kernels they replace there, tools/ab_nohint.py.)
    for name, r in _sel(resources, r"^solve_kernel<float").items():
        general = re.search(r"<float, 3, false, (4|14), 2>", name) is not None
        mid = re.search(r"<float, 3, false, 6, (1|2)>", name)
        assert r["sgpr_spill"] <= 16, (name, r)
        if mid:
            lim = (8, 32) if mid.group(1) == "1" else (24, 96)
            assert r["vgpr_spill"] <= lim[0] and r["scratch"] <= lim[1], (name, r)
        else:
            assert r["vgpr_spill"] <= (16 if general else 0) and r["scratch"] <= (48 if general else 0), (name, r)
    assert "solve_kernel<float, 3, false, 14, 1>" in resources and "solve_kernel<float, 3, false, 4, 1>" in resources
    assert "solve_kernel<float, 3, false, 6, 1>" in resources


def test_register_budgets_of_the_kernel_variants(resources):
    # residency follows from these: 2 / 3 wavefronts per SIMD for the 14- / 4-slot register table, 3 for the LDS table
    for only in (1, 2):
        assert resources[f"solve_kernel<float, 3, false, 14, {only}>"]["vgpr"] <= 256
        assert resources[f"solve_kernel<float, 3, false, 4, {only}>"]["vgpr"] <= 168
        assert resources[f"solve_spec_kernel<float, 3, false, 4, {only}, true>"]["vgpr"] <= 168
        assert resources[f"solve_kernel<float, 3, false, 6, {only}>"]["vgpr"] <= 168
        assert resources[f"solve_spec_kernel<float, 3, false, 6, {only}, true>"]["vgpr"] <= 168
        # (round 6: the TAIL members -- the latency kernel with the throughput kernels' gated evaluation -- keep the budgets)
        assert resources[f"solve_spec_kernel<float, 3, false, 4, {only}, false>"]["vgpr"] <= 168
        assert resources[f"solve_spec_kernel<float, 3, false, 6, {only}, false>"]["vgpr"] <= 168
        assert resources[f"solve_spec_kernel<float, 3, false, 14, {only}, false>"]["vgpr"] <= 256
    assert resources["solve_kernel<float, 3, false, 0, 0>"]["vgpr"] <= 168
    for name, r in _sel(resources, r"^solve_spec_kernel<float, \d, (true|false), 0, 0, true>").items():
        assert r["vgpr"] <= 168 and r["vgpr_spill"] == 0, (name, r)


def test_fp32_latency_and_cooperative_kernels_spill_little(resources):
    for name, r in _sel(resources, r"^solve_spec_kernel<float").items():
        if name.endswith(", false>"):
            # TAIL members (round 6): a few hundred instances per launch at most run on them; the gated evaluation and the
            # deep-park reader next to the latency kernel's solver state cost 12-22 more spilled VGPRs than the flat form
            # (axis-aligned members: 12 / 35 / 20 for 4 / 6 / 14 slots; 54 in the general 6-slot member)
            assert r["sgpr_spill"] <= 56 and r["vgpr_spill"] <= 56 and r["scratch"] <= 200, (name, r)
            continue
        general = re.search(r"<float, 3, false, (4|14), 2, true>", name) is not None
        axis = re.search(r"<float, 3, false, (4|14), 1, true>", name) is not None
        # (round 4: the solver's integer state stays in scalar registers all the way round the loop -- no v_readfirstlane per
        #  variable and evaluation any more -- and the master / worker split added the command addresses: ~30 SGPRs take
        #  the v_writelane / v_readlane route, outside the evaluation; measured +6.5 % and +5.8 % on configs[1] all the same)
        # (round 5: the options the master consults in every iteration are read once instead of per phase -- six more scalar
        #  values alive round the loop, up to 50 spilled: a v_readlane where the lone master wavefront waited ~200 cycles for a
        #  scalar load; with the candidates' FBE formed by the wavefront that evaluated them +1.8 % on configs[1])
        assert r["sgpr_spill"] <= 52, (name, r)
        # (general path of the register-table variants: up to 23 VGPRs in scratch; the axis-aligned members: 0 / 9 dwords)
        mid = re.search(r"<float, 3, false, 6, (1|2), true>", name)       # (6-slot kernels: 16 / 32 VGPRs in scratch, see above)
        # (round 6: an LDS-table member may carry the same unused 9-dword frame object as the 14-slot axis member: private
        #  segment 36 B with NO spilled vector register and NO scratch instruction in its code -- checked on the disassembly)
        phantom = r["scratch"] == 36 and r["vgpr_spill"] == 0 and r["scratch_instr"] == 0
        assert phantom or r["scratch"] <= ((72 if mid.group(1) == "1" else 168) if mid else 96 if general else 40 if axis else 0), (name, r)
    sel = _sel(resources, r"solve_coop(_reg)?_kernel<(float|true|false)")
    assert any("coop_reg" in n for n in sel)          # (the on-chip kernels are named <true> / <false>: they must not drop out)
    for name, r in sel.items():
        # (SGPR -> VGPR-lane spills only, no scratch; the segment chunk bounds and the exchange of the partial minima
        # added ~8 to the on-chip kernel in exchange for the 24 % they bought on configs[4])
        assert r["sgpr_spill"] <= 40 and r["scratch"] == 0, (name, r)


def test_evaluation_and_data_kernels_are_spill_free(resources):
    # every alternative must match at least one kernel of the shipped code object (a renamed kernel must not drop out
    # of the check silently)
    for pattern, sgpr_budget in ((r"^eval_kernel<float", 8), (r"assemble_kernel", 0), (r"hypotheses_kernel", 0),
                                 (r"hypotheses_wide_kernel", 72)):   # (3-4 points per lane: mask words held as scalars;
                                                                     #  eval: SGPR -> VGPR-lane moves in the dual-path 14-slot kernel)
        for name, r in _sel(resources, pattern).items():
            assert r["sgpr_spill"] <= sgpr_budget and r["vgpr_spill"] == 0 and r["scratch"] == 0, (name, r)


def test_every_kernel_stays_within_short_branch_range(resources):
    # s_cbranch / s_branch reach +-32 K instructions words = 128 KB. A kernel beyond that has its far branches relaxed
    # into s_getpc / s_add / s_setpc sequences with scavenged scalar registers -- the one kernel of this library that
    # ever crossed the line (fp64 evaluation with both obstacle-table paths inlined, 142 KB) computed nondeterministic
    # garbage on the far path while each path compiled alone was exact. Keep every kernel short of it.
    for name, r in resources.items():
        assert 0 < r["code_bytes"] < 124 * 1024, (name, r["code_bytes"])
