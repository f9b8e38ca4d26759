"""CPU tests: the oracle (oracle/nmpc_oracle*.{c,h}) against everything the reference pins for this path.

* known answers of the reference's own unit tests (src/tests/test_mpc_builder.py:16-253), via fixtures produced
  by running the reference's functions (tests/golden/known_answers.json);
* f, F1, F2 of the reference's MpcModule.build() (mpc_builder.py:28-201) on random (u, p) (problem_*.npz);
* unicycle RK4 (motion_model.py:141-163);
* the hand-written adjoint against central differences of the REFERENCE's f.
"""
import json
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answers_primitives(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    # every stored reference output equals the value the reference's test asserts
    for name, case in ka.items():
        assert np.allclose(np.ravel(case["got"]), np.ravel(case["expected"]), atol=1e-3), name
    # the oracle's primitives reproduce them (inputs as in test_mpc_builder.py)
    assert oracle.dist2_to_lineseg(1, 2, 3, 2, 3, 0) == pytest.approx(2.0 ** 2)           # :28-43
    assert oracle.dist2_to_lineseg(1, 2, 3, 1, 3, 0) == pytest.approx(5.0)
    assert oracle.inside_ellipse(1, 2, 1, 2, 1, 1, 0) == pytest.approx(1.0)               # :45-60
    assert oracle.inside_ellipse(1, 2, 1, 4, 1, 1, 0) == pytest.approx(-3.0, abs=1e-3)
    b1, a0, a1 = [0, 2, 1, 3], [-1, 1, 0, 0], [0, 0, -1, 1]
    assert oracle.inside_cvx_polygon(1, 2, b1, a0, a1) == pytest.approx(3.0)              # :62-83
    assert oracle.inside_cvx_polygon(1, 2, [0, 1, 0, 1], a0, a1) == pytest.approx(0.0)
    # cost_inside_cvx_polygon weight 2 -> 18 (:123-138); cost_inside_ellipses -> [1, 0] (:140-156)
    assert 2 * oracle.inside_cvx_polygon(1, 2, b1, a0, a1) ** 2 == pytest.approx(18.0)
    assert max(0.0, oracle.inside_ellipse(1, 2, 1, 2, 1, 1, 0)) ** 2 == pytest.approx(1.0)
    assert max(0.0, oracle.inside_ellipse(1, 2, 1, 4, 1, 1, 0)) ** 2 == 0.0
    # cost_refpath_deviation: point (1,2), polyline (0,0)-(1,0)-(3,2), w 0.5 -> 1.0 (:228-240)
    d = min(oracle.dist2_to_lineseg(1, 2, 0, 0, 1, 0), oracle.dist2_to_lineseg(1, 2, 1, 0, 3, 2))
    assert 0.5 * d == pytest.approx(1.0, abs=1e-3)


def test_unicycle_rk4_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "motion_model.npz"))
    for s, a, sn in zip(fx["S"], fx["A"], fx["S_next"]):
        assert np.allclose(oracle.unicycle_rk4(float(fx["ts"]), s, a), sn, rtol=0, atol=1e-14)


@pytest.mark.parametrize("fixture", ["problem_n20", "problem_small"])
def test_problem_functions_match_reference(fixture, request):
    fx, pr = request.getfixturevalue(fixture)
    assert pr.np_ == fx["P"].shape[1]
    for i in range(fx["P"].shape[0]):
        f, F1, F2 = oracle.eval_problem(pr, fx["U"][i], fx["P"][i])
        assert f == pytest.approx(fx["f"][i], rel=1e-12)
        np.testing.assert_allclose(F1, fx["F1"][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(F2, fx["F2"][i], rtol=1e-12, atol=1e-12)
    # the fixtures exercise the penalty constraints (robot inside obstacles)
    assert (fx["F2"] > 0).any()


@pytest.mark.parametrize("fixture", ["problem_n20", "problem_small"])
def test_adjoint_gradient_matches_reference_fd(fixture, request):
    fx, pr = request.getfixturevalue(fixture)
    n = 2 * pr.N
    for i in range(fx["P"].shape[0]):
        val, g = oracle.psi(pr, fx["U"][i], 0.0, np.zeros(n), fx["P"][i])
        assert val == pytest.approx(fx["f"][i], rel=1e-12)
        scale = np.abs(fx["grad_f_fd"][i]).max()
        np.testing.assert_allclose(g, fx["grad_f_fd"][i], rtol=0, atol=2e-7 * scale)


def test_psi_gradient_finite_differences_with_penalty(problem_small):
    fx, pr = problem_small
    rng = np.random.default_rng(0)
    n = 2 * pr.N
    for i in range(4):
        u, p = fx["U"][i], fx["P"][i]
        y, c = rng.normal(size=n), 37.0
        _, g = oracle.psi(pr, u, c, y, p)
        gfd = np.zeros(n)
        for j in range(n):
            h = 1e-6
            up, um = u.copy(), u.copy()
            up[j] += h
            um[j] -= h
            gfd[j] = (oracle.psi(pr, up, c, y, p, grad=False)[0] - oracle.psi(pr, um, c, y, p, grad=False)[0]) / (2 * h)
        np.testing.assert_allclose(g, gfd, rtol=0, atol=5e-7 * np.abs(gfd).max())


def test_psi_is_f_plus_penalties(problem_n20):
    fx, pr = problem_n20
    rng = np.random.default_rng(1)
    n = 2 * pr.N
    lo = np.r_[np.full(pr.N, pr.lin_acc_min), np.full(pr.N, -pr.ang_acc_max)]
    hi = np.r_[np.full(pr.N, pr.lin_acc_max), np.full(pr.N, pr.ang_acc_max)]
    for i in range(6):
        y, c = rng.normal(size=n) * 5, float(rng.uniform(0.5, 200))
        val, _ = oracle.psi(pr, fx["U"][i], c, y, fx["P"][i])
        z = fx["F1"][i] + y / max(c, 1.0)
        d2 = np.sum((z - np.clip(z, lo, hi)) ** 2)
        expect = fx["f"][i] + 0.5 * c * (d2 + np.sum(fx["F2"][i] ** 2))
        assert val == pytest.approx(expect, rel=1e-12)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference checkout (authoring container)")
def test_fixture_recipe_regenerates_every_fixture_byte_for_byte(tmp_path):
    """tests/golden/make_golden.py imports the reference's own Python and rewrites every fixture; the committed files are
    exactly what it produces (the recipe must run as committed)."""
    import filecmp
    import subprocess
    import sys
    out = tmp_path / "regen"
    subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), str(out)], check=True, capture_output=True,
                   timeout=900)
    # (everything under tests/golden/ is a recording of the reference; what is merely RECALLED lives in tests/recalled/)
    names = sorted(f for f in os.listdir(GOLDEN) if f.endswith((".json", ".npz")))
    assert names == sorted(os.listdir(out))
    for f in names:
        assert filecmp.cmp(os.path.join(GOLDEN, f), os.path.join(out, f), shallow=False), f
