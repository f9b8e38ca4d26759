"""CPU tests of the f2 oracle (oracle/hypotheses.py) against the recording of the reference's clustering functions."""
import json
import os

import numpy as np

from oracle import hypotheses as oh


def test_matches_reference_recording(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "hypotheses_cases.json")))
    assert any(c["n_obs"] > len(c["cur"]) for c in cases)          # multi-modal splits occur
    for c in cases:
        dyn, n_obs = oh.hypotheses_to_obstacles(np.array(c["cur"]), np.array(c["hypos"]))
        assert n_obs == c["n_obs"]
        want = np.array(c["dyn_obs_list"], dtype=float)             # [n_obs][N+1][6]
        np.testing.assert_allclose(dyn[:n_obs], want, rtol=0, atol=1e-12)
        assert (dyn[n_obs:] == 0).all()


def test_dbscan_min2_semantics():
    pts = np.array([[0, 0], [0.9, 0], [1.8, 0], [5, 5], [9, 9], [9.5, 9]], dtype=float)
    assert oh.dbscan_min2(pts, 1.0).tolist() == [0, 0, 0, -1, 1, 1]     # chain, isolated noise, second cluster
    assert oh.dbscan_min2(np.array([[0.0, 0], [1.0, 0]]), 1.0).tolist() == [0, 0]   # distance == eps is a neighbour
