#!/usr/bin/env python3
"""VERDICT r4 "Next round" 4(b): THREE INSTANCES PER WAVEFRONT (one lane per horizon step, 20 lanes each) costed against
the shipped one-instance-per-wavefront layout (three lanes per step) -- on paper, from measured inputs:

  * the per-instance psi-evaluation counts of the headline batch (this script solves it once and reads info[:, 4]):
    a wavefront that carries three instances runs until its SLOWEST one is done and every evaluation of the wavefront
    costs the same whether one, two or three of its instances still need it (lanes of finished instances idle): the
    straggler factor is  sum over triples of max(evals) * 3 / sum of evals;
  * the instruction counts of an evaluation by section (profiles/r04_cfg2_pmc_instruction_mix.txt, DESIGN.md "Where the
    1 051 vector instructions go"): obstacle passes ~410 of 1 051 VALU (one slot of 3 rows per lane-triple today; one
    row per lane then: 3x the slots), everything else ~640 (computed in all three lanes of a step today: redundant x3);
  * the issue cadence of one wavefront per SIMD (the table of 40 rows x 7 values per lane needs the 512-register budget)
    against two per SIMD: profiles/r02_issue_rate_microbench.txt.
usage: exp_three_per_wave.py [cfg2 family ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm

VALU_PASSES, VALU_REST = 410.0, 641.0           # per evaluation, today (3 lanes per step)
OTHER = 1510.0 - 1051.0                         # scalar / LDS / branch / wait instructions per evaluation
# cycles per instruction and SIMD: measured 3.55 on the headline kernel at 2 wavefronts/SIMD; a lone wavefront issues one
# instruction per 4.2 (independent FMAs) .. 5.7 (dependent chains, DPP) cycles -- r02_issue_rate_microbench.txt
CAD_2, CAD_1 = 3.55, (4.2, 5.7)
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg2_b65536_n20_4x10"])
lay = spec.pop("layout"); B = spec.pop("B")
for fam in (sys.argv[1:] or ["toward_robot", "passing"]):
    P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec)
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs, cfg.max_active_dynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn, 40
    with nm.Handle(cfg) as h:
        r = h.solve(P)
    ev = r["info"][:, 4].astype(np.float64)
    tri = ev[: (B // 3) * 3].reshape(-1, 3)
    straggler = float(tri.max(axis=1).sum() * 3 / tri.sum())
    # sorted by evaluation count (what a perfect oracle of the work could group): the floor of the factor
    tri_s = np.sort(ev)[: (B // 3) * 3].reshape(-1, 3)
    straggler_sorted = float(tri_s.max(axis=1).sum() * 3 / tri_s.sum())
    # instruction stream of one evaluation of a THREE-instance wavefront: the passes run 3x the slots (each lane all 40
    # rows of its step instead of a third of them), the rest once for all three; segmented reductions over 20-lane groups
    # that straddle the 16-lane DPP rows: +2 instructions per reduction, ~40 reductions per evaluation
    valu3 = 3 * VALU_PASSES + VALU_REST + 80
    per_inst_today = VALU_PASSES + VALU_REST + OTHER
    per_inst_three = (valu3 + OTHER * 1.5) / 3          # (scalar / branch glue: the pass loop is 3x as long, the rest shared)
    out = {"family": fam, "evals_mean": float(ev.mean()), "evals_max": float(ev.max()),
           "straggler_factor_index_order": straggler, "straggler_factor_sorted_by_work": straggler_sorted,
           "instructions_per_instance_evaluation": {"today": per_inst_today, "three_per_wavefront": per_inst_three,
                                                    "ratio": per_inst_today / per_inst_three},
           "net_speedup_estimate": {}}
    for name, cad in (("lone_wavefront_best_case_4.2_cycles", CAD_1[0]), ("lone_wavefront_dependent_5.7_cycles", CAD_1[1])):
        # today: 2 wavefronts per SIMD share its issue slots (3.55 cycles per instruction of EITHER); three-per-wavefront:
        # one wavefront per SIMD at `cad` cycles per instruction, three instances in it, stretched by the straggler factor
        t_today = per_inst_today * CAD_2                # SIMD cycles per instance-evaluation
        t_three = per_inst_three * cad * straggler
        out["net_speedup_estimate"][name] = t_today / t_three
    print(json.dumps(out))
