"""Stand-ins for the two third-party functions the reference's map pipeline needs and this image lacks -- fixture tooling
only (like ``_numeric_casadi.py``), used by ``make_golden.py`` to run the REFERENCE's own ``MapInterface`` /
``OccupancyMap.get_geometric_map`` / ``BlobBounding`` / ``GeometricMap`` on ``data/warehouse_sim_original/mymap.pgm``:

* ``skimage.measure.find_contours(image)`` -- iso-contours at the mid level of the image by marching squares with linear
  interpolation, low-valued corners connected at saddle points (skimage's default ``fully_connected='low'``), one array of
  (row, col) points per closed contour. The reference only takes the convex hull of each contour
  (``map_tools/blob_bounding.py:97-99``), so the order of the points does not matter, only the set.
* ``pyclipper.PyclipperOffset`` with ``JT_MITER`` / ``ET_CLOSEDPOLYGON`` -- offset of a simple polygon by moving every edge
  along its outward normal and intersecting neighbours (what a mitre join is while the mitre limit is not hit; the
  rectangles and the 4-corner boundary of this pipeline never hit it), on pyclipper's integer grid (2**31 per unit).

What comes out is therefore the reference's pipeline up to the differences between these two restatements and the real
libraries (sub-pixel: the polygons are bounding rectangles of blobs of >= 5 px at 0.1 m per px)."""
from __future__ import annotations

import sys
import types

import numpy as np


def find_contours(image, level=None):
    img = np.asarray(image, dtype=float)
    if level is None:
        level = (img.max() + img.min()) / 2.0
    H, W = img.shape
    hi = img > level
    # segments of every 2x2 cell, keyed by their end points on the cell edges; end point id = (edge kind, r, c)
    def pt(a, b):       # interpolated point between grid points a, b (one above, one below the level)
        (r0, c0), (r1, c1) = a, b
        t = (level - img[r0, c0]) / (img[r1, c1] - img[r0, c0])
        return (r0 + t * (r1 - r0), c0 + t * (c1 - c0))
    segs = []
    for r in range(H - 1):
        for c in range(W - 1):
            tl, tr, bl, br = hi[r, c], hi[r, c + 1], hi[r + 1, c], hi[r + 1, c + 1]
            code = (tl << 3) | (tr << 2) | (br << 1) | int(bl)
            if code in (0, 15):
                continue
            T = ("T", r, c)
            R = ("R", r, c)
            Bm = ("B", r, c)
            L = ("L", r, c)
            P = {T: lambda: pt((r, c), (r, c + 1)), R: lambda: pt((r, c + 1), (r + 1, c + 1)),
                 Bm: lambda: pt((r + 1, c), (r + 1, c + 1)), L: lambda: pt((r, c), (r + 1, c))}
            table = {1: [(L, Bm)], 2: [(Bm, R)], 3: [(L, R)], 4: [(T, R)], 6: [(T, Bm)], 7: [(L, T)], 8: [(L, T)],
                     9: [(T, Bm)], 11: [(T, R)], 12: [(L, R)], 13: [(Bm, R)], 14: [(L, Bm)],
                     5: [(L, T), (Bm, R)],      # tr and bl high (saddle): low corners connected -> the high ones cut off
                     10: [(T, R), (L, Bm)]}     # tl and br high (saddle)
            for a, b in table[code]:
                segs.append((a, b, P[a](), P[b]()))
    # canonical ids: an edge shared by two cells has two names
    def canon(e):
        k, r, c = e
        if k == "B":
            return ("T", r + 1, c)
        if k == "R":
            return ("L", r, c + 1)
        return e
    adj = {}
    for a, b, pa, pb in segs:
        a, b = canon(a), canon(b)
        adj.setdefault(a, []).append((b, pa, pb))
        adj.setdefault(b, []).append((a, pb, pa))
    contours, seen = [], set()
    for start in sorted(adj):
        if start in seen:
            continue
        pts, cur, prev = [], start, None
        while cur not in seen:
            seen.add(cur)
            nxt = [x for x in adj[cur] if x[0] != prev] or adj[cur]
            n, p_here, _ = nxt[0]
            pts.append(p_here)
            prev, cur = cur, n
        pts.append(pts[0])
        contours.append(np.array(pts, dtype=float))
    return contours


class PyclipperOffset:
    def __init__(self):
        self._paths = []

    def Clear(self):
        self._paths = []

    def AddPath(self, path, join_type, end_type):
        self._paths.append(np.asarray(path, dtype=float))

    def Execute(self, delta):
        out = []
        for P in self._paths:
            n = len(P)
            area2 = float(np.sum(P[:, 0] * np.roll(P[:, 1], -1) - np.roll(P[:, 0], -1) * P[:, 1]))
            sgn = 1.0 if area2 > 0 else -1.0                     # counter-clockwise: outward normal = (dy, -dx)
            lines = []
            for i in range(n):
                a, b = P[i], P[(i + 1) % n]
                d = b - a
                nrm = sgn * np.array([d[1], -d[0]]) / np.hypot(*d)
                lines.append((a + delta * nrm, d))
            Q = []
            for i in range(n):
                (p0, d0), (p1, d1) = lines[i - 1], lines[i]
                det = d0[0] * (-d1[1]) - (-d1[0]) * d0[1]
                t = ((p1[0] - p0[0]) * (-d1[1]) - (-d1[0]) * (p1[1] - p0[1])) / det
                Q.append(np.rint(p0 + t * d0))
            out.append([[int(v[0]), int(v[1])] for v in Q])
        return out


_SCALE = float(2 ** 31)


def scale_to_clipper(x):
    if isinstance(x, (int, float)):
        return int(round(x * _SCALE))
    return [[int(round(v[0] * _SCALE)), int(round(v[1] * _SCALE))] for v in x]


def scale_from_clipper(x):
    if isinstance(x, (int, float)):
        return x / _SCALE
    return [[[v[0] / _SCALE, v[1] / _SCALE] for v in path] for path in x]


def install():
    """Put the stand-in ``skimage`` / ``pyclipper`` modules into sys.modules (no-op where the real ones import)."""
    try:
        import skimage.measure  # noqa: F401
    except ImportError:
        sk = types.ModuleType("skimage")
        sk.__path__ = []
        for name in ("util", "color", "filters", "measure", "morphology"):
            m = types.ModuleType("skimage." + name)
            setattr(sk, name, m)
            sys.modules["skimage." + name] = m
        sk.measure.find_contours = find_contours
        sk.color.rgb2gray = lambda im: np.asarray(im, dtype=float) @ np.array([0.2125, 0.7154, 0.0721])
        sys.modules["skimage"] = sk
    try:
        import pyclipper  # noqa: F401
    except ImportError:
        pc = types.ModuleType("pyclipper")
        pc.PyclipperOffset, pc.JT_MITER, pc.ET_CLOSEDPOLYGON = PyclipperOffset, 2, 0
        pc.scale_to_clipper, pc.scale_from_clipper = scale_to_clipper, scale_from_clipper
        sys.modules["pyclipper"] = pc
