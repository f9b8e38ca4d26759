"""CPU tests of the f1 oracle (oracle/assemble.py) against the recording of the reference's MpcInterface.run_step."""
import json
import os

import numpy as np

from oracle import assemble as oa


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "assemble_cases.json")))


def test_assemble_matches_reference_recording(golden_dir):
    off_os, off_od = 728, 848
    for c in _cases(golden_dir):
        p_ref = np.array(c["params"])
        refs = np.array(c["ref_states"])
        p = oa.assemble(p_ref[0:2], c["state"], refs, p_ref[78], c["tuning"], None, c["map_polygons"], c["dyn"],
                        c["stc_weights"], c["dyn_weights"])
        assert p.size == p_ref.size == 2778
        np.testing.assert_allclose(p[:off_os], p_ref[:off_os], rtol=0, atol=1e-12)
        np.testing.assert_allclose(p[off_od:], p_ref[off_od:], rtol=0, atol=1e-12)
        got, want = oa.canonical_static_block(p[off_os:off_od]), oa.canonical_static_block(p_ref[off_os:off_od])
        np.testing.assert_allclose(np.array(got), np.array(want), rtol=0, atol=1e-7)
        # the selected polygons are the reference's closest set
        sel = oa.closest_polygons(np.array(c["state"][:2]), c["map_polygons"], 10)
        got_set = sorted(tuple(map(tuple, np.round(c["map_polygons"][m], 9))) for m in sel)
        want_set = sorted(tuple(map(tuple, np.round(q, 9))) for q in c["closest"])
        assert got_set == want_set


def test_halfspaces_describe_the_polygon():
    rng = np.random.default_rng(0)
    quad = np.array([[1.0, 0.5], [-1.0, 0.5], [-1.0, -0.5], [1.0, -0.5]]) + 3.0
    b, a0, a1 = oa.quad_halfspaces(quad)
    for _ in range(200):
        x = rng.uniform(0, 6, 2)
        inside = (abs(x[0] - 3) < 1) and (abs(x[1] - 3) < 0.5)
        assert bool(np.all(b - a0 * x[0] - a1 * x[1] > 0)) == inside
    assert oa.polygon_distance([3.0, 5.0], quad) == 1.5 and oa.polygon_distance([5.0, 3.0], quad) == 1.0
