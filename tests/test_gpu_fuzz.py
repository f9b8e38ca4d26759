"""Reduced runs of the randomised differential checks (tests/fuzz_eval.py, fuzz_solve.py, fuzz_assemble.py) with fixed
seeds, so that the driver-run GPU suite -- not a text file under profiles/ -- covers every lane-map / table boundary with
obstacles ON the path: random dimensions with the table boundaries over-represented, every evaluation code path
(register / LDS / global table, axis-aligned and general variants, cooperative with 2-4 wavefronts, on-chip cooperative
with and without helper lanes), short solves with every solver kernel, the assembly kernel; all against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def _collect():
    lines = []
    return lines, lines.append


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_eval(seed):
    import fuzz_eval
    lines, out = _collect()
    rc = fuzz_eval.run(cases=500, seed=seed, out=out)
    print("\n".join(lines))
    assert rc == 0, lines[-1]


@pytest.mark.parametrize("seed,outer,inner,cases", [(21, 1, 4, 400), (22, 1, 4, 400), (23, 3, 15, 150)])
def test_fuzz_solve(seed, outer, inner, cases):
    import fuzz_solve
    lines, out = _collect()
    rc = fuzz_solve.run(cases=cases, seed=seed, n_outer=outer, n_inner=inner, out=out)
    print("\n".join(lines))
    assert rc == 0, lines[-1]


def test_fuzz_assemble():
    import fuzz_assemble
    lines, out = _collect()
    rc = fuzz_assemble.run(cases=600, seed=31, out=out)
    print("\n".join(lines))
    assert rc == 0, lines[-1]
