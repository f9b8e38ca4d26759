"""GPU test of the latency mode (csrc/nmpc_spec.h): 2, 3 or 4 wavefronts per instance with speculative line search must
give bit-identical results to the same kernel launched with one wavefront (= the sequential algorithm) -- every
candidate is the same function of the same inputs and the acceptance logic is replayed in the sequential order. Against
the throughput kernel (a separate compilation of the same algorithm) the agreement is to rounding, checked the way
fp32 is checked against fp64."""
import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd.scenarios import ParamLayout

pytestmark = pytest.mark.gpu


def _solve(cfg, P, dtype, waves, **kw):
    cfg.latency_waves = waves
    h = nm.Handle(cfg)
    out = h.solve(P.astype(dtype), dtype=dtype, **kw)
    h.close()
    return out


def _same(a, b):
    for key in ("U", "cost", "status", "iters", "y"):
        assert np.array_equal(a[key], b[key], equal_nan=True), key
    assert np.array_equal(a["info"][:, :6], b["info"][:, :6])          # fpr, f2, dy, c, #psi, #grad (algorithmic)
    assert (a["info"][:, 7] == 1).all() and (b["info"][:, 7] > 1).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kw", [dict(seed=31), dict(seed=32, ped_mode="oncoming"), dict(seed=33, n_ped=0, n_boxes=0),
                                dict(seed=34, n_ped=3, n_hyp=5, n_boxes=2)])
def test_latency_mode_is_bit_identical(dtype, kw):
    P = nm.scenarios.make_batch(96, **kw)
    cfg = nm.default_config_struct()
    a = _solve(cfg, P, dtype, -1)
    for waves in (2, 3, 4, 6, 8):                                    # (6: what batches of at most one workgroup per CU get)
        b = _solve(cfg, P, dtype, waves)
        _same(a, b)
        assert (b["info"][:, 7] == waves).all()
        assert (b["info"][:, 6] < b["info"][:, 4]).all()        # fewer rounds than sequential evaluations
    assert (a["status"] == 0).any() or kw.get("ped_mode") is None


def test_latency_mode_with_inputs_and_capacity_hint():
    P = nm.scenarios.make_batch(64, seed=35, ped_mode="oncoming")
    rng = np.random.default_rng(0)
    u0 = rng.uniform(-0.2, 0.8, (64, 40))
    y0 = rng.uniform(-1, 1, (64, 40))
    c0 = rng.uniform(5, 200, 64)
    cfg = nm.default_config_struct()
    cfg.max_active_dynobs = 10
    for dtype in (np.float32, np.float64):
        a = _solve(cfg, P, dtype, -1, u0=u0.astype(dtype), y0=y0.astype(dtype), c0=c0.astype(dtype))
        b = _solve(cfg, P, dtype, 4, u0=u0.astype(dtype), y0=y0.astype(dtype), c0=c0.astype(dtype))
        _same(a, b)
    cfg.max_active_dynobs = 4                                            # capacity exceeded: same loud failure
    a, b = _solve(cfg, P, np.float32, -1), _solve(cfg, P, np.float32, 4)
    assert (a["status"] == 4).all() and (b["status"] == 4).all() and np.isnan(b["U"]).all()


def test_latency_mode_other_dimensions():
    # N = 30 (2 lanes per step), N = 40 with the obstacle table in the global workspace (1 lane per step)
    for N, Ndyn, B in ((30, 12, 24), (40, 160, 6)):
        lay = ParamLayout(N=N, Ndyn=Ndyn)
        P = nm.scenarios.make_batch(B, lay, seed=36, n_ped=2, n_hyp=3, ped_mode="oncoming")
        cfg = nm.default_config_struct()
        cfg.N_hor, cfg.Ndynobs = N, Ndyn
        cfg.max_inner_iterations, cfg.max_outer_iterations = 60, 4
        for dtype in (np.float32, np.float64):
            _same(_solve(cfg, P, dtype, -1), _solve(cfg, P, dtype, 4))


def test_automatic_choice_follows_batch_size():
    """latency_waves = 0: batches of at most one workgroup per CU get six wavefronts per instance (fp32, 4- or 6-slot register
    table), small batches four (round 4: with the long instances started first the faster line search wins over what stays
    resident together) -- also with the 15 rows of the shipped yaml and no capacity hint, which run on the 6-slot register
    table at the same register budget (round 5); with the 14-slot register table (more than 18 rows), whose kernels run two
    wavefronts per SIMD, six / four / two up to 256 / 768 / 1 433 instances (round 6); mid-size batches 2, large ones the throughput kernel -- from 1 434 instances on with the 14-slot table
    (info[7] = wavefronts per instance, 0 = throughput kernel)."""
    P_small = nm.scenarios.make_batch(32, seed=37)
    for hint, dtype, expect in ((10, np.float32, 6), (0, np.float32, 6), (0, np.float64, 4)):
        cfg = nm.default_config_struct()
        cfg.latency_waves = 0
        cfg.max_active_dynobs = hint
        with nm.Handle(cfg) as h:
            small = h.solve(P_small.astype(dtype), dtype=dtype)
        assert (small["info"][:, 7] == expect).all(), (hint, dtype, small["info"][0, 7])
    lay24 = nm.scenarios.ParamLayout(20, 10, 10, 24)
    cfg = nm.default_config_struct()
    cfg.latency_waves, cfg.Ndynobs = 0, 24
    with nm.Handle(cfg) as h:
        r24 = h.solve(nm.scenarios.make_batch(32, lay24, seed=37, n_ped=4, n_hyp=5).astype(np.float32), dtype=np.float32)
    assert (r24["info"][:, 7] == 6).all(), r24["info"][0, 7]
    cfg = nm.default_config_struct()
    cfg.latency_waves = 0
    cfg.max_active_dynobs = 10
    with nm.Handle(cfg) as h:                                          # more than one workgroup per CU, at most one per SIMD
        four = h.solve(nm.scenarios.make_batch(600, seed=37).astype(np.float32), dtype=np.float32)
    assert (four["info"][:, 7] == 4).all(), four["info"][0, 7]
    cfg = nm.default_config_struct()
    cfg.latency_waves = 0
    h = nm.Handle(cfg)
    mid = h.solve(nm.scenarios.make_batch(2048, seed=37, n_ped=0, n_boxes=0).astype(np.float32), dtype=np.float32)
    big = h.solve(nm.scenarios.make_batch(8192, seed=37, n_ped=0, n_boxes=0).astype(np.float32), dtype=np.float32)
    fam_big = h.last_launch_info()["family"]
    h.close()
    # (the throughput plan from one device fill on hands the drain phase of its last launch to the latency family's tail
    #  member, nmpc_config.tail_latency: the few instances finished there report its six wavefronts)
    assert (mid["info"][:, 7] == 2).all() and fam_big == "throughput" and (big["info"][:, 7] == 0).mean() > 0.9
    assert set(np.unique(big["info"][:, 7])) <= {0, 6}
    # 14-slot kernels (two wavefronts per SIMD, 2 048 resident): the throughput plan takes over at 0.7 device fills, 1 434 instances
    cfg = nm.default_config_struct()
    cfg.latency_waves, cfg.Ndynobs = 0, 24
    with nm.Handle(cfg) as h:
        for B, fam, w in ((600, "latency", 4), (1000, "latency", 2), (1400, "latency", 2), (1500, "throughput", 0), (2100, "throughput", 0)):
            r = h.solve(nm.scenarios.make_batch(B, lay24, seed=37, n_ped=4, n_hyp=5).astype(np.float32), dtype=np.float32)
            assert h.last_launch_info()["family"] == fam and (w == 0 or (r["info"][:, 7] == w).all()), (B, h.last_launch_info())


def test_latency_kernel_agrees_with_throughput_kernel_to_rounding():
    """Separate compilations of the same algorithm: same exit status and iterates to solver accuracy on instances
    that converge (fp64, Lipschitz step 1e-4: the parity protocol of DESIGN.md)."""
    P = nm.scenarios.make_batch(256, seed=38, ped_mode="oncoming")
    cfg = nm.default_config_struct()
    cfg.lip_eps_f64 = cfg.lip_delta_f64 = 1e-4
    a, b = _solve(cfg, P, np.float64, 1), _solve(cfg, P, np.float64, 4)
    conv = (a["status"] == 0) & (b["status"] == 0)
    assert conv.sum() >= 16 and (a["status"] == b["status"]).mean() > 0.9
    d = np.abs(a["U"] - b["U"]).max(axis=1)[conv]
    print("throughput vs latency kernel, converged:", conv.sum(), "median", np.median(d), "q90", np.quantile(d, 0.9), "max", d.max())
    # most instances follow the same path to solver accuracy; a minority is pushed onto another branch of the
    # (non-convex) problem by a last-bit difference early on (DESIGN.md "parity protocol")
    assert np.median(d) < 1e-8 and np.mean(d < 1e-4) >= 0.6


def test_latency_kernel_follows_the_throughput_kernel_step_for_step():
    """The tight half of the comparison above: over the first iterations -- before the solver's own
    amplification of last-bit differences sets in -- the two compilations must agree to rounding on EVERY instance, with
    identical iteration and evaluation counts; and at tolerance 1e-8 on the converging family the fixed points coincide."""
    P = nm.scenarios.make_batch(256, seed=38, ped_mode="oncoming")
    cfg = nm.default_config_struct()
    cfg.lip_eps_f64 = cfg.lip_delta_f64 = 1e-4
    cfg.max_outer_iterations, cfg.max_inner_iterations = 1, 6
    a, b = _solve(cfg, P, np.float64, 1), _solve(cfg, P, np.float64, 4)
    assert np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["info"][:, 4:6], b["info"][:, 4:6])
    d = np.abs(a["U"] - b["U"]).max(axis=1)
    print("6 inner iterations: median", np.median(d), "q90", np.quantile(d, 0.9), "max", d.max())
    assert np.quantile(d, 0.9) < 1e-9 and d.max() < 1e-6
    lay = ParamLayout(N=20, Ndyn=15)
    P = nm.scenarios.make_batch(192, lay, seed=39, n_ped=2, n_hyp=5, ped_mode="passing")
    cfg = nm.default_config_struct()
    cfg.max_active_dynobs = 10
    cfg.tolerance = cfg.initial_tolerance = cfg.delta_tolerance = 1e-8
    cfg.max_outer_iterations, cfg.max_inner_iterations = 15, 2000
    a, b = _solve(cfg, P, np.float64, 1), _solve(cfg, P, np.float64, 4)
    conv = (a["status"] == 0) & (b["status"] == 0)
    d = np.abs(a["U"] - b["U"]).max(axis=1)[conv]
    print("tolerance 1e-8, converged on both:", conv.sum(), "median", np.median(d), "q90", np.quantile(d, 0.9))
    assert conv.sum() >= 30 and np.median(d) < 1e-7 and np.quantile(d, 0.8) < 1e-4


def test_latency_plans_get_their_dispatch_order_from_one_evaluation():
    """Round 6: at about one workgroup per SIMD (512 < B <= 1 024) the order decides which long solves share a SIMD to the end. A
    pilot launch used to rank the instances; one evaluation at nominal controls -- ||F2||^2 at (2/3 v_max, 0) -- does it without
    the barrier (the 4- / 6-slot kernels above 896 instances keep the pilot: configs[1] itself). Whatever the order and the number
    of launches, the results are the same bits."""
    lay = nm.scenarios.ParamLayout(20, 10, 10, 15)
    for B, staged_expected in ((700, 0), (1000, 1)):
        P = nm.scenarios.make_batch_chunked(B, lay, seed=71, n_ped=2, n_hyp=5, dtype=np.float32)
        res = {}
        for staged in (0, -1, 1):
            cfg = nm.default_config_struct()
            cfg.max_active_dynobs, cfg.staged = 10, staged
            with nm.Handle(cfg) as h:
                res[staged] = h.solve(P)
                li = h.last_launch_info()
            assert li["family"] == "latency" and li["staged_outer_iterations"] == (staged_expected if staged == 0 else max(staged, 0)), (B, staged, li)
        for k in ("U", "y", "cost", "status", "iters"):
            assert np.array_equal(res[0][k], res[-1][k]) and np.array_equal(res[0][k], res[1][k]), (B, k)
    # the two-wavefront plans of the 4- / 6-slot kernels (1 024 < B <= 4 096) are ordered the same way; bits as under any order
    P = nm.scenarios.make_batch_chunked(2500, lay, seed=73, n_ped=2, n_hyp=5, dtype=np.float32)
    res = {}
    for staged in (0, -1):
        cfg = nm.default_config_struct()
        cfg.max_active_dynobs, cfg.staged = 10, staged
        with nm.Handle(cfg) as h:
            res[staged] = h.solve(P)
            assert h.last_launch_info()["family"] == "latency" and h.last_launch_info()["staged_outer_iterations"] == 0
    assert np.array_equal(res[0]["U"], res[-1]["U"]) and np.array_equal(res[0]["iters"], res[-1]["iters"])
    # 14-slot kernels: no exception
    lay2 = nm.scenarios.ParamLayout(20, 10, 10, 40)
    cfg = nm.default_config_struct()
    cfg.Ndynobs, cfg.max_active_dynobs = 40, 40
    with nm.Handle(cfg) as h:
        r = h.solve(nm.scenarios.make_batch_chunked(1000, lay2, seed=72, n_ped=4, n_hyp=10, dtype=np.float32))
        assert h.last_launch_info()["family"] == "latency" and h.last_launch_info()["staged_outer_iterations"] == 0
        assert (r["status"] >= 0).all()
