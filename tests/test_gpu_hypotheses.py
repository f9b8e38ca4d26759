"""GPU tests of the f2 path (nmpc_hypotheses_to_ellipses_*) through the C ABI: against the recording of the reference's
fit_DBSCAN / fit_cluster2gaussian (sklearn) and against the numpy oracle on random batches; then the device-resident
chain f2 -> f1 -> solve."""
import json
import os

import numpy as np
import pytest
import torch

import dyobav_mpcnwta_warehouse_amd as nm
import oracle
from conftest import config_for
from oracle import hypotheses as oh

pytestmark = pytest.mark.gpu


def _run(h, dt, hypos, cur):
    tdt = torch.float32 if dt == np.float32 else torch.float64
    B = hypos.shape[0]
    dyn = torch.full((B, 15, 21, 6), float("nan"), dtype=tdt, device="cuda")
    nobs = torch.zeros(B, dtype=torch.int32, device="cuda")
    h.hypotheses_to_ellipses(dt, torch.from_numpy(hypos.astype(dt)).cuda(), torch.from_numpy(cur.astype(dt)).cuda(), dyn, nobs)
    torch.cuda.synchronize()
    return dyn.cpu().numpy().astype(np.float64), nobs.cpu().numpy()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_matches_reference_recording(golden_dir, dt):
    cases = json.load(open(os.path.join(golden_dir, "hypotheses_cases.json")))
    tol = 1e-11 if dt == np.float64 else 5e-5
    with nm.Handle(config_for(oracle.Problem())) as h:
        for c in cases:
            dyn, nobs = _run(h, dt, np.array(c["hypos"])[None], np.array(c["cur"])[None])
            assert nobs[0] == c["n_obs"]
            want = np.array(c["dyn_obs_list"], dtype=float)
            np.testing.assert_allclose(dyn[0, :c["n_obs"]], want, rtol=tol, atol=tol)
            assert (dyn[0, c["n_obs"]:] == 0).all()


def test_random_batch_matches_oracle_and_chain_to_solver():
    rng = np.random.default_rng(8)
    B, N, H, K = 192, 20, 2, 10
    cur = rng.uniform(-4, 4, (B, H, 2))
    vel = rng.uniform(-1, 1, (B, H, 2))
    t = np.arange(1, N + 1)[None, :, None, None, None]
    modes = rng.normal(0, 0.9, (B, 1, H, 3, 2))
    which = rng.integers(0, 3, (B, N, H, K))
    ctr = cur[:, None, :, None, :] + vel[:, None, :, None, :] * 0.2 * t + modes
    pts = np.take_along_axis(np.broadcast_to(ctr, (B, N, H, 3, 2)), which[..., None].repeat(2, -1), axis=3)
    hypos = (pts + rng.normal(0, 0.15, pts.shape)).reshape(B, N, H * K, 2)
    with nm.Handle(config_for(oracle.Problem())) as h:
        dyn, nobs = _run(h, np.float64, hypos, cur)
        for b in range(0, B, 7):
            want, n = oh.hypotheses_to_obstacles(cur[b], hypos[b])
            assert nobs[b] == n
            np.testing.assert_allclose(dyn[b], want, rtol=0, atol=1e-11)
        # chain on the device: hypotheses -> ellipses -> parameter vectors -> solve
        dt, tdt = np.float32, torch.float32
        d_dyn = torch.empty(B, 15, 21, 6, dtype=tdt, device="cuda")
        h.hypotheses_to_ellipses(dt, torch.from_numpy(hypos.astype(dt)).cuda(), torch.from_numpy(cur.astype(dt)).cuda(), d_dyn)
        state = np.c_[rng.uniform(-6, 6, (B, 2)), rng.uniform(-3, 3, B)]
        refs = np.concatenate([state[:, None, :2] + (np.arange(1, N + 1) * 0.24)[None, :, None] *
                               np.stack([np.cos(state[:, 2]), np.sin(state[:, 2])], 1)[:, None, :],
                               np.tile(state[:, 2][:, None, None], (1, N, 1))], axis=2)
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=dt)).cuda()
        P = torch.empty(B, h.np_, dtype=tdt, device="cuda")
        polys = dev(np.array([[[9, 9], [8, 9], [8, 8], [9, 8]]] * 12, dtype=float) + np.arange(12)[:, None, None])
        h.assemble_params(dt, B, P, dev(np.zeros((B, 2))), dev(state), dev(refs), dev(np.full(B, 1.2)),
                          dev(nm.scenarios.WORK_MODE_Q), dev(np.full(N, 10.0)), dev(np.full(N, 10.0)), polys, d_dyn)
        U = torch.empty(B, 40, dtype=tdt, device="cuda")
        st = torch.empty(B, dtype=torch.int32, device="cuda")
        h.solve_raw(dt, P, B, U, status=st, sync=True)
        assert torch.isfinite(U).all() and set(st.cpu().numpy().tolist()) <= {0, 1}
        # the assembled o_d block is the f2 output
        od = P[:, 848:848 + 1890].cpu().numpy().reshape(B, 15, 21, 6)
        assert np.array_equal(od, d_dyn.cpu().numpy())


def test_edge_cases_single_points_and_overflow():
    with nm.Handle(config_for(oracle.Problem())) as h:
        # every point isolated -> all noise -> only the current positions remain
        hyp = (np.arange(8)[None, None, :, None] * 5.0 + np.zeros((1, 20, 8, 2)))
        dyn, nobs = _run(h, np.float64, hyp, np.array([[[1.0, 2.0]]]))
        assert nobs[0] == 1 and np.allclose(dyn[0, 0, 0], [1, 2, 0.2, 0.2, 0, 1])
        assert (dyn[0, 0, 1:, :5] == 0).all() and (dyn[0, 0, 1:, 5] == 1).all() and (dyn[0, 1:] == 0).all()
        # 20 well separated pairs -> 20 clusters > Ndynobs = 15: truncated, n_obs reports the overflow
        pairs = np.repeat(np.arange(20) * 4.0, 2)[None, None, :, None] + np.zeros((1, 20, 40, 2))
        pairs[..., 1::2, 0] += 0.3
        dyn, nobs = _run(h, np.float64, pairs, np.array([[[0.0, 0.0]]]))
        assert nobs[0] == 20 and (dyn[0, :, 1:, 5] == 1).all()
        with pytest.raises(nm.NmpcError):
            _run(h, np.float64, np.zeros((1, 20, 257, 2)), np.zeros((1, 1, 2)))


@pytest.mark.parametrize("P", [1, 2, 7, 20, 21, 32, 33, 40, 64, 65, 100, 128, 129, 160, 192, 193, 256])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_point_counts_and_chain_clusters(P, dt):
    """Every lane grouping of the kernels (64 // P time offsets per pass, 32- and 64-bit masks; above 64 points the wide
    kernel with 2, 3 or 4 points per lane -- 160 = the 8 pedestrians x 20 hypotheses of BASELINE configs[4], which the
    reference clusters together, main_base.py:192-196) against the numpy oracle;
    half of the instances are chains (neighbours 0.9 eps apart in shuffled order: the component is only found through
    paths as long as the point count), the rest blobs with noise points."""
    rng = np.random.default_rng(100 + P)
    B, N = 24, 20
    hypos = np.empty((B, N, P, 2))
    for b in range(B):
        for t in range(N):
            if b % 2 == 0:
                n1 = int(rng.integers(1, P + 1))
                chain = np.c_[np.arange(n1) * 0.9, np.zeros(n1)] + rng.uniform(-3, 3, 2)
                rest = rng.uniform(20, 60, (P - n1, 2))
                pts = np.r_[chain, rest]
            else:
                ctr = rng.uniform(-6, 6, (3, 2))
                pts = ctr[rng.integers(0, 3, P)] + rng.normal(0, 0.3, (P, 2))
            hypos[b, t] = pts[rng.permutation(P)]
    cur = rng.uniform(-4, 4, (B, 2, 2))
    if dt == np.float32:  # the oracle sees the same rounded inputs
        hypos, cur = hypos.astype(np.float32).astype(np.float64), cur.astype(np.float32).astype(np.float64)
    tol = 1e-11 if dt == np.float64 else 2e-4
    with nm.Handle(config_for(oracle.Problem())) as h:
        dyn, nobs = _run(h, dt, hypos, cur)
    for b in range(B):
        want, n = oh.hypotheses_to_obstacles(cur[b], hypos[b])
        assert nobs[b] == n
        np.testing.assert_allclose(dyn[b], want, rtol=0, atol=tol)
