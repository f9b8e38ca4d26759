"""CPU ORACLE (test infrastructure) for the "next" row f2: multi-hypothesis motion predictions -> obstacle ellipses.

numpy restatement of
* ``fit_DBSCAN(data, eps=1, min_sample=2)``                    /root/reference/src/utils_test.py:133-143
  (sklearn DBSCAN; with min_samples = 2 every point that has another point within eps is a core point, so the
  clusters are the connected components of the "distance <= eps" graph, numbered in order of their smallest point
  index; points without a neighbour are noise and dropped)
* ``fit_cluster2gaussian(clusters, enlarge=2, extra_margin=0)``  utils_test.py:145-151 (mean, population std * enlarge)
* the obstacle-list assembly of ``MainBase.run_one_step``        main_base.py:293-302
Pinned by ``tests/golden/hypotheses_cases.json`` (recorded from the reference's functions with sklearn).
"""
from __future__ import annotations

import numpy as np


def dbscan_min2(points: np.ndarray, eps: float) -> np.ndarray:
    """Labels (-1 = noise) of DBSCAN(eps, min_samples=2)."""
    n = len(points)
    d2 = ((points[:, None, :] - points[None, :, :]) ** 2).sum(axis=2)
    adj = d2 <= eps * eps
    labels = np.full(n, -1)
    nxt = 0
    for i in range(n):
        if labels[i] != -1 or adj[i].sum() < 2:
            continue
        stack, labels[i] = [i], nxt
        while stack:
            j = stack.pop()
            for m in np.nonzero(adj[j])[0]:
                if labels[m] == -1:
                    labels[m] = nxt
                    stack.append(m)
        nxt += 1
    return labels


def hypotheses_to_obstacles(cur: np.ndarray, hypos: np.ndarray, human_size=0.2, eps=1.0, enlarge=2.0,
                            extra_margin=0.0, Ndyn=15):
    """cur [H][2], hypos [N][P][2] -> (dyn [Ndyn][N+1][6], n_obs). Slots >= n_obs stay zero (the MPC interface pads them)."""
    N = hypos.shape[0]
    rows = [[(c[0], c[1], human_size, human_size) for c in cur]]
    for t in range(N):
        lab = dbscan_min2(hypos[t], eps)
        cl = []
        for c in range(lab.max() + 1 if lab.size else 0):
            pts = hypos[t][lab == c]
            mu, sd = pts.mean(axis=0), pts.std(axis=0) * enlarge + extra_margin
            cl.append((mu[0], mu[1], sd[0], sd[1]))
        rows.append(cl)
    n_obs = max(len(r) for r in rows)
    dyn = np.zeros((Ndyn, N + 1, 6))
    dyn[:min(n_obs, Ndyn), :, 5] = 1.0
    for t, r in enumerate(rows):
        for c, v in enumerate(r[:Ndyn]):
            dyn[c, t, :4] = v
    return dyn, n_obs
