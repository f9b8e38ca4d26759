import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
key = "cfg2_b65536_n20_4x10"; B = 65536
spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")
for fam in ("passing", "toward_robot"):
    P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec)
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = 40
    h = nm.Handle(cfg)
    U = np.empty((B, 40), np.float32); it = np.empty((B, 2), np.int32); st = np.empty(B, np.int32); info = np.empty((B, 8), np.float32)
    h.solve_raw(np.float32, P, B, U, status=st, iters=it, info=info)
    h.solve_raw(np.float32, P, B, U, status=st, iters=it, info=info)
    print(fam, "kernel ms", h.last_kernel_ms())
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"evals_{fam}.npz"), evals=info[:, 4].astype(np.int32), grads=info[:, 5].astype(np.int32), inner=it[:, 1], outer=it[:, 0], status=st, ms=h.last_kernel_ms())
