"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/nmpc_hip.h declares; argument validation works; nothing computes without a GPU."""
import ctypes
import os
import re

import pytest

import dyobav_mpcnwta_warehouse_amd as nm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "nmpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nmpc_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    lib = nm.load_library()
    declared = _header_symbols()
    assert sorted(nm.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nmpc_hip.h but not exported"


def test_config_struct_matches_header_defaults():
    cfg = nm.default_config_struct()
    assert cfg.abi_version == 5
    assert (cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs) == (20, 10, 10, 15)
    assert cfg.ts == 0.2 and cfg.lin_vel_max == 1.5 and cfg.ang_acc_max == 3.0
    assert cfg.tolerance == 1e-4 and cfg.initial_penalty == 10.0 and cfg.max_inner_iterations == 500
    assert cfg.max_outer_iterations == 10 and cfg.lbfgs_memory == 10
    assert cfg.latency_waves == 0 and cfg.akkt_form == 0 and cfg.max_solver_time_us == 0.0
    assert cfg.coop_waves == 0 and cfg.axis_aligned == 0 and cfg.reg_table == 0 and cfg.staged == 0
    assert cfg.polish == 0 and cfg.polish_max_outer_iterations == 4 and cfg.polish_max_inner_iterations == 150
    assert cfg.polish_tolerance == 1e-6 and cfg.polish_delta_tolerance == 1e-5 and cfg.staged_evals == 0
    assert cfg.max_evaluations == 0 and cfg.tail_latency == 0 and cfg.batch_invariant == 0
    # struct size and a late field's offset: ctypes mirror vs the C compiler on include/nmpc_hip.h (catches field drift)
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "nmpc_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu", '
                             'sizeof(nmpc_config), offsetof(nmpc_config, latency_waves), offsetof(nmpc_config, initial_penalty), offsetof(nmpc_config, reg_table), '
                             'offsetof(nmpc_config, polish_delta_tolerance), offsetof(nmpc_config, max_evaluations), offsetof(nmpc_config, batch_invariant), sizeof(nmpc_layout_info), sizeof(nmpc_loop_args), offsetof(nmpc_loop_args, n_hyp), '
                             'offsetof(nmpc_loop_args, hyp_radius_growth), sizeof(nmpc_assemble_args));return 0;}\n')
        exe = os.path.join(td, "sz")
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), src, "-o", exe], check=True)
        size, off_lw, off_ip, off_gram, off_pd, off_me, off_bi, size_li, size_loop, off_nh, off_hg, size_asm = map(
            int, subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert ctypes.sizeof(nm.NmpcConfigStruct) == size == 6 * 4 + 13 * 8 + 4 * 4 + 11 * 8 + 2 * 4 + 8 + 4 * 4 + 4 * 4 + 2 * 8 + 3 * 4 + 4   # (+ 4: tail padding)
    assert nm.NmpcConfigStruct.latency_waves.offset == off_lw and nm.NmpcConfigStruct.initial_penalty.offset == off_ip
    assert nm.NmpcConfigStruct.reg_table.offset == off_gram
    assert nm.NmpcConfigStruct.polish_delta_tolerance.offset == off_pd
    assert nm.NmpcConfigStruct.max_evaluations.offset == off_me and nm.NmpcConfigStruct.batch_invariant.offset == off_bi  # ABI v5
    from dyobav_mpcnwta_warehouse_amd._capi import NmpcAssembleArgs, NmpcLayoutInfo, NmpcLoopArgs
    assert ctypes.sizeof(NmpcLayoutInfo) == size_li
    # ABI v4: the hypothesis fan of nmpc_loop_args (n_hyp in the former `reserved` slot + three doubles at the end)
    assert ctypes.sizeof(NmpcLoopArgs) == size_loop and NmpcLoopArgs.n_hyp.offset == off_nh
    assert NmpcLoopArgs.hyp_radius_growth.offset == off_hg and ctypes.sizeof(NmpcAssembleArgs) == size_asm


def test_no_cpu_fallback_without_device():
    """On a box without a GPU, creating a solver fails loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(nm.NmpcError) as ei:
        nm.Handle(nm.default_config_struct())
    assert ei.value.code == -3 and "no CPU path" in str(ei.value)


def test_invalid_arguments_are_rejected_before_touching_the_device():
    lib = nm.load_library()
    cfg = nm.default_config_struct()
    h = ctypes.c_void_p()
    cfg.abi_version = 99
    assert lib.nmpc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"abi_version" in lib.nmpc_last_error()
    cfg = nm.default_config_struct()
    cfg.N_hor = 65
    assert lib.nmpc_create(ctypes.byref(cfg), ctypes.byref(h)) == -4
    cfg = nm.default_config_struct()
    cfg.lbfgs_memory = 11
    assert lib.nmpc_create(ctypes.byref(cfg), ctypes.byref(h)) == -4
    assert lib.nmpc_param_len(None) == -1
    assert lib.nmpc_destroy(None) == 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "nmpc_oracle" not in src, f


def test_layout_bookkeeping_without_a_device():
    """nmpc_layout: the dimension bookkeeping nmpc_create would do. Pins that a configuration whose
    register-table layout still exceeds LDS falls back to the global table WITH the full [row][t] entry count -- the
    kernels of that fallback index cap * (N + 1) rows of the workspace."""
    cfg = nm.default_config_struct()
    li = nm.layout_info(cfg)
    # (15 rows of the shipped yaml: the 6-slot register table since round 5; t = 0 rows padded with dummy rows to 3 x slots)
    assert li.np == 2778 and li.reg_slots_f32 == 6 and not li.global_table_f32 and li.table_entries_f32 == 3 * 6
    cfg.max_active_dynobs = 10
    assert nm.layout_info(cfg).reg_slots_f32 == 4
    cfg = nm.default_config_struct()
    cfg.Ndynobs = 40                                    # BASELINE configs[2]: 4 x 10 hypotheses -> the 14-slot table
    li = nm.layout_info(cfg)
    assert li.reg_slots_f32 == 14 and li.table_entries_f32 == 3 * 14
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Ndynobs = 40, 160                    # BASELINE configs[4]: 236 KB table -> global workspace
    li = nm.layout_info(cfg)
    assert li.np == 40968 and li.global_table_f32 and li.global_table_f64 and li.reg_slots_f32 == 0
    assert li.ws_elems_f32 == 9 * 160 * 41 == li.ws_elems_f64 and li.table_entries_f32 == 160 * 41
    # register table selected (<= 42 rows, N <= 21) but everything else does not fit LDS: so many static obstacles that
    # the polygon table alone is ~160 KB in fp32
    cfg = nm.default_config_struct()
    cfg.Nstcobs = 3300
    li = nm.layout_info(cfg)
    assert li.global_table_f32 and li.reg_slots_f32 == 0
    assert li.table_entries_f32 == 15 * 21 and li.ws_elems_f32 == 9 * 15 * 21
    cfg.N_hor = 0
    with pytest.raises(nm.NmpcError):
        nm.layout_info(cfg)
