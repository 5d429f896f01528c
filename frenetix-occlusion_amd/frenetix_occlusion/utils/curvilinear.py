"""Polyline curvilinear coordinate system -- the three methods of pycrccosy.CurvilinearCoordinateSystem [ext] that the
spawn rules call (ref: spawn_locator.py:229,385,398,449,536,549,554): Cartesian <-> (s, d) on a reference polyline,
d positive to the left.  Points outside the projection domain (before the first / after the last vertex) raise
ValueError like the C++ original; callers skip them (spawn_locator.py:228-231)."""
import numpy as np


def pathlength(polyline):
    p = np.asarray(polyline, dtype=np.float64)
    return np.concatenate(([0.0], np.cumsum(np.hypot(np.diff(p[:, 0]), np.diff(p[:, 1])))))


def curvature(polyline):
    """signed curvature per vertex (gradients w.r.t. path length, like commonroad_dc.geometry.util
    compute_curvature_from_polyline [ext])"""
    p = np.asarray(polyline, dtype=np.float64)
    keep = np.concatenate(([True], np.hypot(np.diff(p[:, 0]), np.diff(p[:, 1])) > 0.0))   # repeated vertices
    p = p[keep]
    if len(p) < 3:
        return np.zeros(len(p))
    s = pathlength(p)
    xd, yd = np.gradient(p[:, 0], s), np.gradient(p[:, 1], s)
    xdd, ydd = np.gradient(xd, s), np.gradient(yd, s)
    return (xd * ydd - xdd * yd) / np.maximum((xd * xd + yd * yd) ** 1.5, 1e-300)


class PolylineCS:
    def __init__(self, reference_path):
        self.path = np.asarray(reference_path, dtype=np.float64)
        if self.path.ndim != 2 or len(self.path) < 2:
            raise ValueError("reference path needs at least two points")
        self.s = pathlength(self.path)
        seg = np.diff(self.path, axis=0)
        self.seg_len = np.hypot(seg[:, 0], seg[:, 1])
        self.tangent = seg / np.maximum(self.seg_len, 1e-300)[:, None]

    def convert_to_curvilinear_coords(self, x, y):
        p = np.array([x, y], dtype=np.float64)
        rel = p[None] - self.path[:-1]
        t = np.sum(rel * self.tangent, axis=1)
        tc = np.clip(t, 0.0, self.seg_len)
        foot = self.path[:-1] + tc[:, None] * self.tangent
        d2 = np.sum((p[None] - foot) ** 2, axis=1)
        k = int(np.argmin(d2))
        if (k == 0 and t[0] < 0.0) or (k == len(t) - 1 and t[-1] > self.seg_len[-1]):
            raise ValueError("point outside the projection domain")
        n = np.array([-self.tangent[k, 1], self.tangent[k, 0]])
        return np.array([self.s[k] + tc[k], float(np.dot(p - foot[k], n))])

    def convert_to_cartesian_coords(self, s, d):
        if s < self.s[0] or s > self.s[-1]:
            raise ValueError("s outside the reference path")
        k = int(min(np.searchsorted(self.s, s, side="right") - 1, len(self.seg_len) - 1))
        base = self.path[k] + (s - self.s[k]) * self.tangent[k]
        return base + d * np.array([-self.tangent[k, 1], self.tangent[k, 0]])

    def convert_list_of_points_to_curvilinear_coords(self, points, num_threads=1):
        out = []
        for q in points:
            q = np.asarray(q, dtype=np.float64).reshape(-1)
            out.append(self.convert_to_curvilinear_coords(q[0], q[1]))
        return out
