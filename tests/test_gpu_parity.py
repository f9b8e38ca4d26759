"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (stated per test):
  * psi / grad psi / ||F2||^2 : f64 rel 1e-11 ; f32 rel 2e-5 (psi), 2e-4 of max|grad| (grad)
  * iterate path, f64, Lipschitz step 1e-4 on both sides, <= 10 inner iterations: max|u - u_ref| < 1e-7,
    identical iteration counts and exit status
  * full solves: see each test. OpEn's own Lipschitz estimator step (1e-12) makes the very first step length
    differ by ~1e-3 between ANY two floating-point evaluation orders (measured; DESIGN.md "parity protocol"),
    so full-solve parity is asserted with the step set to 1e-4 on both sides and reported as fractions.
"""
import os

import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm
import oracle
from accuracy_protocol import HOST_THREADS
import conftest
from conftest import config_for

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module", params=[1, 4, "coop"],
                ids=["throughput-kernel", "latency-kernel", "cooperative-kernel"])
def kernel_mode(request):
    """Every test of this module runs against both solver kernels (separate compilations of the same algorithm)."""
    conftest.set_kernel_mode(request.param)
    yield request.param
    conftest.set_kernel_mode(0)


@pytest.fixture(scope="module")
def handle20(kernel_mode):
    h = nm.Handle(config_for(oracle.Problem()))
    yield h
    h.close()


def test_wave_primitives_selftest(handle20):
    assert handle20.selftest() == 0


def test_kernel_is_the_native_library(handle20):
    import os
    assert os.path.exists(nm.library_path())
    info = handle20.kernel_info()
    assert info["lanes_per_step"] == 3 and info["lds_bytes_f32"] > 0 and info["waves_per_cu_f32"] >= 1


@pytest.mark.parametrize("fixture", ["problem_n20", "problem_small"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_psi_and_gradient_match_oracle_on_golden_inputs(fixture, dtype, request):
    fx, pr = request.getfixturevalue(fixture)
    rng = np.random.default_rng(5)
    K, n = fx["P"].shape[0], 2 * pr.N
    Y = rng.normal(size=(K, n)) * 3
    C = rng.uniform(1, 300, K)
    C[:3] = 0.0                                  # c = 0 must return f (golden value from the reference)
    with nm.Handle(config_for(pr)) as h:
        r = h.eval(fx["P"], fx["U"], Y, C, dtype=dtype)
    rp, rg = (1e-11, 1e-11) if dtype == np.float64 else (2e-5, 2e-4)
    for i in range(K):
        v, g = oracle.psi(pr, fx["U"][i], C[i], Y[i], fx["P"][i])
        assert r["psi"][i] == pytest.approx(v, rel=rp)
        np.testing.assert_allclose(r["grad"][i], g, rtol=0, atol=rg * np.abs(g).max())
        f2 = float(np.sum(fx["F2"][i] ** 2))     # golden F2 from the reference's own code
        assert r["f2sq"][i] == pytest.approx(f2, rel=10 * rp, abs=1e-12)
    for i in range(3):
        assert r["psi"][i] == pytest.approx(fx["f"][i], rel=rp)


@pytest.mark.parametrize("family", ["free", "boxes", "oncoming", "toward_robot"])
def test_iterate_path_matches_oracle_f64(family):
    kw = dict(free=dict(n_ped=0, n_boxes=0), boxes=dict(n_ped=0), oncoming=dict(ped_mode="oncoming"),
              toward_robot=dict())[family]
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(48, L, seed=21, **kw)
    pr = oracle.Problem()
    for max_inner in (1, 3, 10):
        op = oracle.Options(max_outer=1, max_inner=max_inner, lip_delta=1e-4, lip_eps=1e-4)
        Uo, ro = oracle.solve_batch(pr, op, P, nthreads=HOST_THREADS)
        cfg = config_for(pr, max_outer_iterations=1, max_inner_iterations=max_inner, lip_delta_f64=1e-4,
                         lip_eps_f64=1e-4)
        with nm.Handle(cfg) as h:
            r = h.solve(P)
        assert np.array_equal(r["iters"][:, 1], ro["inner_iters"])
        assert np.array_equal(r["status"], ro["status"])
        assert np.array_equal(r["info"][:, 5].astype(int), ro["n_grad_evals"])
        # (round 6) ... and the evaluated points: info[4] = orc_result.n_points, the quantity nmpc_config.max_evaluations /
        # orc_options.max_evals budget
        assert np.array_equal(r["info"][:, 4].astype(int), ro["n_points"])
        du = np.abs(r["U"] - Uo).max(axis=1)
        assert np.quantile(du, 0.9) < 1e-9, (family, max_inner, du.max())
        assert du.max() < 1e-6, (family, max_inner, du.max())
        np.testing.assert_allclose(r["cost"], ro["cost"], rtol=1e-6)


def test_full_solve_matches_oracle_f64_free_space():
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(128, L, seed=22, n_ped=0, n_boxes=0)
    pr = oracle.Problem()
    op = oracle.Options(lip_delta=1e-4, lip_eps=1e-4)
    Uo, ro = oracle.solve_batch(pr, op, P, nthreads=HOST_THREADS)
    with nm.Handle(config_for(pr, lip_delta_f64=1e-4, lip_eps_f64=1e-4)) as h:
        r = h.solve(P)
    du = np.abs(r["U"] - Uo).max(axis=1)
    assert np.mean(r["status"] == ro["status"]) >= 0.95
    assert np.mean(du < 1e-4) >= 0.90            # rounding-noise divergence on a few ill-conditioned instances
    assert np.median(du) < 1e-8
    assert np.mean(r["iters"][:, 1] == ro["inner_iters"]) >= 0.85
    # exit bookkeeping is self-consistent
    conv = r["status"] == 0
    assert (r["info"][conv, 1] <= 1e-4 + 1e-9).all()          # ||F2|| <= delta
    assert (r["iters"][:, 0] >= 2).all() and (r["iters"][:, 0] <= 10).all()


def test_full_solve_f32_statistics_vs_oracle_f64():
    """fp32 kernel against the fp64 oracle at the default tolerance (1e-4 on ||gamma*fpr||, which bounds the
    solution error only by ~1e-4*L/0.95): the solutions agree to ~1e-3 in the median, statuses mostly agree."""
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(256, L, seed=23, n_ped=0, n_boxes=0)
    pr = oracle.Problem()
    Uo, ro = oracle.solve_batch(pr, oracle.Options(), P, nthreads=HOST_THREADS)
    with nm.Handle(config_for(pr)) as h:
        r = h.solve(P.astype(np.float32))
    assert set(np.unique(r["status"])) <= {0, 1}
    both = (r["status"] == 0) & (ro["status"] == 0)
    assert both.mean() > 0.4
    du = np.abs(r["U"].astype(np.float64) - Uo).max(axis=1)
    assert np.median(du[both]) < 5e-3
    cost_rel = np.abs(r["cost"].astype(np.float64) - ro["cost"]) / np.abs(ro["cost"])
    assert np.median(cost_rel[both]) < 1e-3
    # solutions respect the box U exactly
    U = r["U"]
    assert (U[:, 0::2] >= pr.lin_vel_min).all() and (U[:, 0::2] <= pr.lin_vel_max).all()
    assert (np.abs(U[:, 1::2]) <= pr.ang_vel_max).all()


_ORACLE_MEMO = {}     # the module runs once per kernel family: the oracle's side of a comparison is the same every time


def test_tight_tolerance_solutions_coincide_f64():
    """Run both sides to 1e-8: instances that converge on both sides reach the same KKT point."""
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(48, L, seed=24, n_ped=0, n_boxes=0)
    pr = oracle.Problem()
    kw = dict(tolerance=1e-8, initial_tolerance=1e-8, delta_tolerance=1e-8)
    if "tight48" not in _ORACLE_MEMO:      # (15 s of CPU per pass; hoist_trig: the same bits, tests/test_oracle_solver.py)
        _ORACLE_MEMO["tight48"] = oracle.solve_batch(pr, oracle.Options(max_inner=5000, max_outer=30, hoist_trig=1, **kw), P, nthreads=16)
    Uo, ro = _ORACLE_MEMO["tight48"]
    with nm.Handle(config_for(pr, max_inner_iterations=5000, max_outer_iterations=30, **kw)) as h:
        r = h.solve(P)
    both = (r["status"] == 0) & (ro["status"] == 0)
    assert both.sum() >= 8
    du = np.abs(r["U"] - Uo).max(axis=1)[both]
    assert np.median(du) < 1e-5
    assert np.mean(du < 1e-4) >= 0.5             # the rest sit in other local minima (non-convex path term)


def test_size_independent_properties_full_batch():
    """BASELINE config 1 at full size (B=1024, fp32): properties that need no oracle."""
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(1024, L, seed=0).astype(np.float32)
    pr = oracle.Problem()
    with nm.Handle(config_for(pr)) as h:
        r1 = h.solve(P)
        r2 = h.solve(P)
        # determinism: no atomics, fixed reduction trees -> bit-identical reruns
        assert np.array_equal(r1["U"], r2["U"]) and np.array_equal(r1["iters"], r2["iters"])
        # permutation equivariance: instances are independent
        perm = np.random.default_rng(1).permutation(1024)
        r3 = h.solve(np.ascontiguousarray(P[perm]))
        assert np.array_equal(r3["U"], r1["U"][perm]) and np.array_equal(r3["status"], r1["status"][perm])
        # reported cost is f(u*) of the returned controls (checked with the device's own evaluator, c = 0)
        ev = h.eval(P, r1["U"], np.zeros_like(r1["U"]), np.zeros(1024, np.float32), grad=False)
        # (fp32 rounding: the evaluator forms the t = 0 soft terms per group of identical rows with summed weights, the
        #  latency kernel row by row -- a few ulp of the total where those terms dominate)
        np.testing.assert_allclose(ev["psi"], r1["cost"], rtol=5e-6)
        np.testing.assert_allclose(np.sqrt(ev["f2sq"]), r1["info"][:, 1], rtol=1e-5, atol=1e-7)
    U = r1["U"]
    assert np.isfinite(U).all()
    assert (U[:, 0::2] >= pr.lin_vel_min).all() and (U[:, 0::2] <= pr.lin_vel_max).all()
    assert (np.abs(U[:, 1::2]) <= pr.ang_vel_max).all()
    assert ((r1["iters"][:, 0] >= 2) & (r1["iters"][:, 0] <= 10)).all()
    assert (r1["iters"][:, 1] <= 10 * 501).all()


def test_edge_cases_zero_padded_and_initial_guess():
    """All-zero obstacle / robot slots (the reference's defaults, trajectory_tracker.py:291-296) and the optional
    run() arguments (initial guess, multipliers, penalty)."""
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(16, L, seed=30, n_ped=0, n_boxes=0)
    pr = oracle.Problem()
    op = oracle.Options(lip_delta=1e-4, lip_eps=1e-4, max_outer=2, max_inner=8)
    cfg = config_for(pr, lip_delta_f64=1e-4, lip_eps_f64=1e-4, max_outer_iterations=2, max_inner_iterations=8)
    rng = np.random.default_rng(2)
    u0 = rng.uniform(-0.3, 0.8, (16, 40))
    y0 = rng.normal(size=(16, 40))
    c0 = rng.uniform(5, 50, 16)
    with nm.Handle(cfg) as h:
        r = h.solve(P, u0=u0, y0=y0, c0=c0)
    for b in range(16):
        op_b = oracle.Options(**{**op.__dict__, "initial_penalty": float(c0[b])})
        u, y, res = oracle.solve(pr, op_b, P[b], u0=u0[b], y0=y0[b])
        assert np.abs(r["U"][b] - u).max() < 1e-7
        assert np.abs(r["y"][b] - y).max() < 1e-6
        assert r["iters"][b, 1] == res["inner_iters"] and r["iters"][b, 0] == res["outer_iters"]


def test_batch_of_one_and_ragged_sizes(handle20):
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(67, L, seed=31, n_ped=0, n_boxes=0)
    full = handle20.solve(P)
    one = handle20.solve(P[:1])
    part = handle20.solve(P[5:38])
    assert np.array_equal(one["U"][0], full["U"][0])
    assert np.array_equal(part["U"], full["U"][5:38])
    with pytest.raises(ValueError):
        handle20.solve(P[:, :-1])


def test_config2_dimensions_40_obstacles():
    """BASELINE configs[2] dimensions (Ndynobs = 40), reduced batch; f64 path parity + f32 sanity."""
    lay = nm.scenarios.ParamLayout(20, 10, 10, 40)
    P = nm.scenarios.make_batch(32, lay, seed=1, n_ped=4, n_hyp=10)
    pr = oracle.Problem(20, 10, 10, 40)
    op = oracle.Options(max_outer=1, max_inner=5, lip_delta=1e-4, lip_eps=1e-4)
    Uo, ro = oracle.solve_batch(pr, op, P, nthreads=HOST_THREADS)
    cfg = config_for(pr, max_outer_iterations=1, max_inner_iterations=5, lip_delta_f64=1e-4, lip_eps_f64=1e-4)
    with nm.Handle(cfg) as h:
        r = h.solve(P)
        assert np.abs(r["U"] - Uo).max() < 1e-6
        assert np.array_equal(r["iters"][:, 1], ro["inner_iters"])
    with nm.Handle(config_for(pr)) as h:
        r32 = h.solve(P.astype(np.float32))
        assert np.isfinite(r32["U"]).all() and set(np.unique(r32["status"])) <= {0, 1}


def test_device_pointers_are_used_in_place():
    import torch
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(64, L, seed=32, n_ped=0, n_boxes=0).astype(np.float32)
    pr = oracle.Problem()
    with nm.Handle(config_for(pr)) as h:
        ref = h.solve(P)
        dP = torch.from_numpy(P).cuda()
        dU = torch.empty(64, 40, device="cuda")
        dst = torch.empty(64, dtype=torch.int32, device="cuda")
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        h.solve_raw(np.float32, dP, 64, dU, status=dst, sync=False)
        torch.cuda.synchronize()
        assert np.array_equal(dU.cpu().numpy(), ref["U"])
        assert np.array_equal(dst.cpu().numpy(), ref["status"])


def test_config4_long_horizon_obstacle_table_streamed_from_global_memory():
    """BASELINE configs[4] dimensions (N = 40, 8 x 20 = 160 obstacle slots): the 236 KB obstacle table does not fit
    in LDS and is streamed from the per-instance global workspace (GLB kernel variant). Reduced batch."""
    lay = nm.scenarios.ParamLayout(40, 10, 10, 160)
    assert lay.np_ == 40968
    P = nm.scenarios.make_batch(12, lay, seed=5, n_ped=8, n_hyp=20, ped_mode="oncoming")
    pr = oracle.Problem(40, 10, 10, 160)
    rng = np.random.default_rng(9)
    U = np.stack([rng.uniform(-0.5, 1.5, (12, 40)), rng.uniform(-0.5, 0.5, (12, 40))], axis=2).reshape(12, 80)
    Y = rng.normal(size=(12, 80))
    C = rng.uniform(1, 100, 12)
    with nm.Handle(config_for(pr)) as h:
        for dt, rp, rg in ((np.float64, 1e-11, 1e-10), (np.float32, 5e-5, 5e-4)):
            r = h.eval(P, U, Y, C, dtype=dt)
            for i in range(12):
                v, g = oracle.psi(pr, U[i], C[i], Y[i], P[i])
                assert r["psi"][i] == pytest.approx(v, rel=rp)
                np.testing.assert_allclose(r["grad"][i], g, rtol=0, atol=rg * np.abs(g).max())
    op = oracle.Options(max_outer=1, max_inner=4, lip_delta=1e-4, lip_eps=1e-4)
    Uo, ro = oracle.solve_batch(pr, op, P, nthreads=HOST_THREADS)
    cfg = config_for(pr, max_outer_iterations=1, max_inner_iterations=4, lip_delta_f64=1e-4, lip_eps_f64=1e-4)
    with nm.Handle(cfg) as h:
        r = h.solve(P)
    assert np.array_equal(r["iters"][:, 1], ro["inner_iters"])
    assert np.abs(r["U"] - Uo).max() < 1e-6


def test_streamed_table_kernel_pair_members_agree_to_rounding():
    """The two members of the streamed-table kernel pair on the SAME batch: the compressed member forms its
    inverse squared radii by v_rcp + Newton steps (not correctly rounded), the general member stores IEEE quotients -- they
    agree to rounding (1e-13 of |psi| in fp64), not bit for bit, which is why `axis_aligned = 0` (the member is chosen per
    CALL) makes an instance's last bits depend on the batch and callers that need batch-independent results pin 1 or -1
    (nmpc_hip.h). A radius so large that its square overflows is simply out of reach for both members (fast_rcp clamps its
    argument: the Newton step of an infinity would be NaN)."""
    lay = nm.scenarios.ParamLayout(40, 10, 10, 160)
    P = nm.scenarios.make_batch(8, lay, seed=6, n_ped=8, n_hyp=20, ped_mode="oncoming")
    pr = oracle.Problem(40, 10, 10, 160)
    rng = np.random.default_rng(10)
    U = np.stack([rng.uniform(-0.5, 1.5, (8, 40)), rng.uniform(-0.5, 0.5, (8, 40))], axis=2).reshape(8, 80)
    Y, C = rng.normal(size=(8, 80)), rng.uniform(1, 100, 8)
    res = {}
    for dt, big in ((np.float64, 1e200), (np.float32, 1e30)):
        Pd = P.copy()
        Pd[7, lay.od + 6 * (3 * 41 + 5) + 2] = big          # one radius whose square overflows: (rx + 1e-6)^2 = inf
        for axis in (1, -1):
            with nm.Handle(config_for(pr, coop_waves=4, latency_waves=1, reg_table=-1, axis_aligned=axis)) as h:
                res[axis] = h.eval(Pd, U, Y, C, dtype=dt)
        tol = 1e-12 if dt == np.float64 else 2e-5
        assert np.isfinite(res[1]["psi"]).all() and np.isfinite(res[-1]["psi"]).all() and np.isfinite(res[1]["grad"]).all()
        np.testing.assert_allclose(res[1]["psi"], res[-1]["psi"], rtol=tol)
        scale = np.abs(res[-1]["grad"]).max(axis=1, keepdims=True)
        assert (np.abs(res[1]["grad"] - res[-1]["grad"]) <= 10 * tol * scale).all()


@pytest.mark.parametrize("axis", [0, 1, -1])
def test_config4_compressed_global_table(axis):
    """The COMPRESSED streamed table of the cooperative global-table kernels (round 5; nmpc_device.h Instance::CMP): for
    axis-aligned ellipses an entry is streamed as (cx, cy, rx, ry) + alpha -- 5 values instead of 9 -- and the inverse squared
    radii are formed where it is used. psi / grad psi against the oracle (f64 1e-11, the bar of the general table), the
    iterate path of a short solve, and the pair mechanics: axis_aligned = 0 -> the device-side scan picks the compressed
    member, -1 -> the general table, 1 -> the promise; a rotated ellipse sends the whole call to the general member (0) or
    is refused for that instance alone (1: status 5, NaN controls)."""
    lay = nm.scenarios.ParamLayout(40, 10, 10, 160)
    P = nm.scenarios.make_batch(12, lay, seed=5, n_ped=8, n_hyp=20, ped_mode="oncoming")
    pr = oracle.Problem(40, 10, 10, 160)
    rng = np.random.default_rng(9)
    U = np.stack([rng.uniform(-0.5, 1.5, (12, 40)), rng.uniform(-0.5, 0.5, (12, 40))], axis=2).reshape(12, 80)
    Y = rng.normal(size=(12, 80))
    C = rng.uniform(1, 100, 12)
    want_axis = {0: 2, 1: 1, -1: 0}[axis]
    ov = dict(coop_waves=4, latency_waves=1, reg_table=-1, axis_aligned=axis)
    res = {}
    with nm.Handle(config_for(pr, **ov)) as h:
        for dt, rp, rg in ((np.float64, 1e-11, 1e-10), (np.float32, 5e-5, 5e-4)):
            r = h.eval(P, U, Y, C, dtype=dt)
            li = h.last_launch_info()
            assert li["family"] == "cooperative" and li["axis_aligned"] == want_axis, li
            res[dt] = r
            for i in range(12):
                v, g = oracle.psi(pr, U[i], C[i], Y[i], P[i])
                assert r["psi"][i] == pytest.approx(v, rel=rp)
                np.testing.assert_allclose(r["grad"][i], g, rtol=0, atol=rg * np.abs(g).max())
    op = oracle.Options(max_outer=1, max_inner=4, lip_delta=1e-4, lip_eps=1e-4)
    Uo, ro = oracle.solve_batch(pr, op, P, nthreads=HOST_THREADS)
    cfg = config_for(pr, max_outer_iterations=1, max_inner_iterations=4, lip_delta_f64=1e-4, lip_eps_f64=1e-4, **ov)
    with nm.Handle(cfg) as h:
        r = h.solve(P)
        assert h.last_launch_info()["axis_aligned"] == want_axis
        assert np.array_equal(r["iters"][:, 1], ro["inner_iters"])
        assert np.abs(r["U"] - Uo).max() < 1e-6
        # one rotated ellipse in instance 3
        Pr = P.copy()
        Pr[3, lay.od + 6 * (5 * 41 + 7) + 4] = 0.3          # (angle; with rx != ry: a rotated CIRCLE is still axis-aligned)
        Pr[3, lay.od + 6 * (5 * 41 + 7) + 2] *= 1.5
        Uor, _ = oracle.solve_batch(pr, op, Pr, nthreads=HOST_THREADS)
        rr = h.solve(Pr)
        if axis == 1:
            assert rr["status"][3] == 5 and np.isnan(rr["U"][3]).all()
            ok = np.arange(12) != 3
            assert np.array_equal(rr["U"][ok], r["U"][ok])
        else:
            assert np.abs(rr["U"] - Uor).max() < 1e-6 and (rr["status"] != 5).all()


def test_fp32_vs_fp64_tolerance_sweep_long_horizon():
    """BASELINE configs[4]: fp64 vs fp32 at solver tolerances 1e-3 ... 1e-6 (SURVEY.md 8d configuration 5; device vs
    device, reduced batch). Obstacle-free family: both precisions converge to the same controls, the median distance
    shrinks with the tolerance down to the floor of fp32 (1e-6 is below it: no further gain, the solves run into the
    iteration caps instead). The 8 x 20 crowd: on the contract family (pedestrians walking INTO the robot) only
    feasibility / finiteness can be asserted -- iterates of a non-convex problem stopped by iteration caps are not
    comparable across precisions; on the `passing` family the instances that converge on both sides are compared."""
    lay = nm.scenarios.ParamLayout(40, 10, 10, 160)
    pr = oracle.Problem(40, 10, 10, 160)
    P_free = nm.scenarios.make_batch(16, lay, seed=6, n_ped=0, n_boxes=0)
    P_crowd = nm.scenarios.make_batch(8, lay, seed=6, n_ped=8, n_hyp=20, ped_mode="oncoming")
    med = {}
    P_pass = nm.scenarios.make_batch(12, lay, seed=7, n_ped=8, n_hyp=20, ped_mode="passing")
    crowd = {}
    for tol in (1e-3, 1e-4, 1e-5, 1e-6):
        cfg = config_for(pr, tolerance=tol, initial_tolerance=tol)
        with nm.Handle(cfg) as h:
            r64, r32 = h.solve(P_free), h.solve(P_free.astype(np.float32))
            both = (r64["status"] == 0) & (r32["status"] == 0)
            du = np.abs(r64["U"] - r32["U"].astype(np.float64)).max(axis=1)
            med[tol] = float(np.median(du[both])) if both.sum() >= 4 else float(np.median(du))
            if tol in (1e-3, 1e-4, 1e-5):             # the crowd on the converging family: same comparison, reported
                p64, p32 = h.solve(P_pass), h.solve(P_pass.astype(np.float32))
                bp = (p64["status"] == 0) & (p32["status"] == 0)
                dp = np.abs(p64["U"] - p32["U"].astype(np.float64)).max(axis=1)
                crowd[tol] = (int(bp.sum()), float(np.median(dp[bp])) if bp.any() else None)
                assert np.isfinite(p64["U"]).all() and np.isfinite(p32["U"]).all()
            if tol == 1e-4:
                rc64, rc32 = h.solve(P_crowd), h.solve(P_crowd.astype(np.float32))
                for r in (rc64, rc32):
                    assert np.isfinite(r["U"]).all() and set(np.unique(r["status"])) <= {0, 1}
                    assert (r["U"][:, 0::2] <= pr.lin_vel_max).all() and (r["U"][:, 0::2] >= pr.lin_vel_min).all()
                    assert (np.abs(r["U"][:, 1::2]) <= pr.ang_vel_max).all()
    print("obstacle-free, median |u32 - u64| by tolerance:", med, "; 8 x 20 crowd `passing` (both converged, median):", crowd)
    # measured on MI355X: 0.125 / 0.0127 / 0.0018 -- the distance scales with the tolerance (error ~ tol / gamma) --
    # and no better at 1e-6 (fp32's floor)
    assert med[1e-3] < 0.3 and med[1e-4] < 0.04 and med[1e-5] < 0.006 and med[1e-6] < 0.02, med
    assert med[1e-5] < med[1e-4] < med[1e-3], med
    assert all(v[1] is None or v[1] < 0.3 for v in crowd.values()), crowd


def test_capacity_hint_same_results_and_safe_failure():
    """nmpc_config.max_active_dynobs: provisioning exactly the non-zero obstacle slots changes nothing; provisioning
    fewer makes the instance fail loudly (status 4 = CapacityExceeded, NaN controls) instead of reading past the table."""
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(96, L, seed=41, n_ped=2, n_hyp=5).astype(np.float32)      # 10 of 15 slots used
    pr = oracle.Problem()
    # LDS table (reg_table = -1): the hint only moves LDS offsets, the machine code is the same -> bit-identical
    with nm.Handle(config_for(pr, reg_table=-1)) as h:
        full = h.solve(P)
        lds_full = h.kernel_info()["lds_bytes_f32"]
    with nm.Handle(config_for(pr, reg_table=-1, max_active_dynobs=10)) as h:
        hint = h.solve(P)
        assert h.kernel_info()["lds_bytes_f32"] < lds_full
    assert np.array_equal(hint["U"], full["U"]) and np.array_equal(hint["status"], full["status"])
    assert np.array_equal(hint["iters"], full["iters"])
    # automatic mode: 15 provisioned rows -> 14-slot register table, 10 rows -> 4-slot one; separate compilations of the
    # same arithmetic agree to rounding (and both with the LDS-table kernel). Compared after 5 inner iterations, before
    # the iteration amplifies fp32 rounding (DESIGN.md "parity protocol"); full solves: same statuses, similar costs.
    if conftest.KERNEL_MODE["reg_table"] == 0:      # (the cooperative kernel always uses the LDS / global table)
        short = dict(max_outer_iterations=1, max_inner_iterations=5)
        with nm.Handle(config_for(pr, reg_table=-1, **short)) as h:
            s_lds = h.solve(P)
        with nm.Handle(config_for(pr, **short)) as h:
            s15 = h.solve(P)
            assert h.kernel_info()["lds_bytes_f32"] < lds_full          # only the t = 0 snapshot is in LDS
        with nm.Handle(config_for(pr, max_active_dynobs=10, **short)) as h:
            s10 = h.solve(P)
        for a, b in ((s15, s10), (s15, s_lds)):
            assert np.array_equal(a["iters"], b["iters"])
            du = np.abs(a["U"] - b["U"]).max(axis=1)
            assert np.median(du) < 2e-4 and du.max() < 5e-2, (np.median(du), du.max())
        with nm.Handle(config_for(pr)) as h:
            auto15 = h.solve(P)
        assert np.mean(auto15["status"] == full["status"]) >= 0.9 and np.isfinite(auto15["U"]).all()
    for rt in (0, -1):
        with nm.Handle(config_for(pr, max_active_dynobs=9, reg_table=rt)) as h:
            small = h.solve(P)
        assert (small["status"] == 4).all() and np.isnan(small["U"]).all()
    P2 = P.copy()
    P2[:48, L.od + 9 * 21 * 6: L.od + 10 * 21 * 6] = 0.0        # first half: only 9 non-zero slots
    with nm.Handle(config_for(pr, max_active_dynobs=9)) as h:
        mixed = h.solve(P2)
    assert (mixed["status"][:48] <= 1).all() and (mixed["status"][48:] == 4).all()
    assert np.isfinite(mixed["U"][:48]).all()


def test_fleet_terms_on_and_off_on_device(problem_n20):
    """Other-robot (fleet) terms, mpc_builder.py:86-97 / mpc_cost.py:65-76: evaluated on the device with the non-zero
    robot slots of the golden inputs (index lists in LDS, positions read from the parameter vector), with the t = 0 set
    only, with the predictive set only, and with every slot zeroed (the reference's default, where the phantom robots at
    the origin are folded into a closed form) -- each against the oracle, and the variants must differ from one another."""
    fx, pr = problem_n20
    L = nm.scenarios.ParamLayout()
    P = fx["P"].copy()
    K = P.shape[0]
    assert (P[:, L.c0:L.c0 + 30] != 0).any() and (P[:, L.c:L.c + 600] != 0).any()
    # put half of the instances next to a predicted robot position so that the predictive term is active for them
    for b in range(0, K, 2):
        P[b, L.c + 3 * (2 * 20 + 5): L.c + 3 * (2 * 20 + 5) + 2] = P[b, L.s0:L.s0 + 2] + 0.05
        P[b, L.c0 + 3 * 3: L.c0 + 3 * 3 + 2] = P[b, L.s0:L.s0 + 2] + 0.1
    variants = {"all": P.copy(), "t0_only": P.copy(), "pred_only": P.copy(), "none": P.copy()}
    variants["t0_only"][:, L.c:L.c + 600] = 0.0
    variants["pred_only"][:, L.c0:L.c0 + 30] = 0.0
    variants["none"][:, L.c0:L.c + 600] = 0.0
    rng = np.random.default_rng(9)
    Y, C = rng.normal(size=(K, 40)), rng.uniform(1, 50, K)
    U = fx["U"] * 0.05                      # short roll-outs stay near the start, where the robots were placed
    psis = {}
    with nm.Handle(config_for(pr)) as h:
        for name, Pv in variants.items():
            for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-5)):
                r = h.eval(Pv, U, Y, C, dtype=dtype)
                for i in range(K):
                    v, g = oracle.psi(pr, U[i], C[i], Y[i], Pv[i])
                    assert r["psi"][i] == pytest.approx(v, rel=tol), (name, dtype, i)
                    np.testing.assert_allclose(r["grad"][i], g, rtol=0, atol=20 * tol * max(1.0, np.abs(g).max()))
                if dtype == np.float64:
                    psis[name] = r["psi"].copy()
    assert (np.abs(psis["all"] - psis["none"]) > 1e-6).sum() >= K // 2
    assert (np.abs(psis["t0_only"] - psis["none"]) > 1e-6).any() and (np.abs(psis["pred_only"] - psis["none"]) > 1e-6).any()
    assert (np.abs(psis["all"] - psis["t0_only"]) > 1e-6).any()


def test_config2_full_batch_size_independent_properties():
    """BASELINE configs[2] at FULL size (B = 65536, Ndynobs = 40, fp32): determinism across launches, permutation
    equivariance on a slice (bit for bit for a given kernel; the automatic mode may pick different kernels for
    different batch sizes, and those agree to rounding only), reported cost = f(u*) from the device evaluator,
    feasibility of every control."""
    lay = nm.scenarios.ParamLayout(20, 10, 10, 40)
    B = 65536
    P = nm.scenarios.make_batch(B, lay, seed=1, n_ped=4, n_hyp=10).astype(np.float32)
    pr = oracle.Problem(20, 10, 10, 40)
    with nm.Handle(config_for(pr)) as h:
        r1 = h.solve(P)
        r2 = h.solve(P)
        assert np.array_equal(r1["U"], r2["U"]) and np.array_equal(r1["iters"], r2["iters"])
        idx = np.random.default_rng(3).permutation(B)[:4096]
        r3 = h.solve(np.ascontiguousarray(P[idx]))
        assert np.array_equal(r3["U"], r1["U"][idx]) and np.array_equal(r3["status"], r1["status"][idx])
        ev = h.eval(P[:8192], r1["U"][:8192], np.zeros((8192, 40), np.float32), np.zeros(8192, np.float32), grad=False)
        # (the same source compiled into two kernels: fp contraction is context-dependent, a few fp32 ulp of the total)
        np.testing.assert_allclose(ev["psi"], r1["cost"][:8192], rtol=5e-6)
    U = r1["U"]
    assert np.isfinite(U).all() and set(np.unique(r1["status"])) <= {0, 1}
    assert (U[:, 0::2] >= pr.lin_vel_min).all() and (U[:, 0::2] <= pr.lin_vel_max).all()
    assert (np.abs(U[:, 1::2]) <= pr.ang_vel_max).all()
    assert ((r1["iters"][:, 0] >= 2) & (r1["iters"][:, 0] <= 10)).all()


def test_non_finite_inputs_and_optional_outputs(handle20):
    """NaN in the parameters -> NMPC_NOT_FINITE (OpEn: SolverError::NotFiniteComputation, run() returns None); every
    optional output of the C ABI may be NULL; B = 0 is a no-op."""
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(8, L, seed=50, n_ped=0, n_boxes=0)
    P[3, L.s0] = np.nan
    r = handle20.solve(P)
    assert r["status"][3] == 3 and (np.delete(r["status"], 3) <= 1).all()
    U = np.empty((8, 40))
    handle20.solve_raw(np.float64, P, 8, U)                      # cost/status/iters/u0/y/c0/info all NULL
    assert np.array_equal(np.delete(U, 3, axis=0), np.delete(r["U"], 3, axis=0))
    handle20.solve_raw(np.float64, P, 0, U)
    from dyobav_mpcnwta_warehouse_amd.solver import Solver
    s = Solver(config_for(oracle.Problem()))
    assert s.run(P[3].tolist()) is None
    assert s.run(P[0].tolist()) is not None
    s.close()


@pytest.mark.parametrize("fixture", ["problem_n20", "problem_small"])
def test_iterate_path_on_golden_inputs_all_terms_active(fixture, request):
    """Solve (short, fixed iteration caps) the golden parameter vectors -- rotated ellipses, active fleet terms, robot
    driven through polygons, every weight non-zero -- from their own u as initial guess: GPU fp64 vs oracle."""
    fx, pr = request.getfixturevalue(fixture)
    P, U0 = fx["P"], fx["U"]
    for mo, mi in ((1, 3), (2, 12)):
        op = oracle.Options(max_outer=mo, max_inner=mi, lip_delta=1e-4, lip_eps=1e-4)
        cfg = config_for(pr, max_outer_iterations=mo, max_inner_iterations=mi, lip_delta_f64=1e-4, lip_eps_f64=1e-4)
        with nm.Handle(cfg) as h:
            r = h.solve(P, u0=U0)
        du = []
        for b in range(P.shape[0]):
            u, y, res = oracle.solve(pr, op, P[b], u0=U0[b])
            assert r["iters"][b, 1] == res["inner_iters"], (b, mo, mi)
            du.append(np.abs(r["U"][b] - u).max())
            assert r["cost"][b] == pytest.approx(res["cost"], rel=1e-3)
        du = np.array(du)
        # these inputs are extreme (robot inside obstacles, costs ~1e4-1e5): rounding noise is amplified on a few of
        # them after a dozen iterations; the bulk follows the oracle's path to 1e-6
        assert np.quantile(du, 0.8) < 1e-6 and du.max() < 5e-3, (mo, mi, np.sort(du)[-4:])
        if mi <= 3:
            assert du.max() < 1e-6
