#!/usr/bin/env python3
"""Generates tests/golden/independent_kat.npz: known answers for the O(1) arithmetic of the ICP chain computed with
numpy / scipy ONLY -- neither the oracle (oracle/icp_oracle.c) nor the product is imported here.

Why: where the big kernels are concerned the oracle and the product are different algorithms (kd-tree vs voxel
pyramid, nth_element vs radix select), so agreeing with each other means something.  The small arithmetic, however,
is the same restatement written twice by the same hand (box normal of the SamplingSurfaceNormal filter, the 6x6 solve
of the point-to-plane minimiser, the rotation metric of the differential checker, the trim index): a shared misreading
would pass every oracle-vs-product test.  These vectors come from independent library code:

  box normals    numpy.linalg.eigh of the box covariance in float64 (normal = eigenvector of the smallest eigenvalue,
                 compared up to sign), boxes that the filter must drop = exactly collinear points
  6x6 solve      numpy.linalg.cholesky + two triangular solves, in float32 and in float64, on the committed normal
                 matrix of tests/golden/icp_pair4k.npz
  rotation metric  scipy.spatial.transform.Rotation: magnitude of R_a R_b^T, float64
  trim limit     numpy.partition at index floor(float32(n) * float32(ratio)), clamped to n - 1

Replayed by tests/test_oracle.py::test_independent_known_answers (oracle + the product's host code, CPU) and
tests/test_gpu_parity.py::test_independent_known_answers_on_device (device filter, device select, device solve).
This does NOT pin parity with libpointmatcher (its source is not here): it pins the restatement to the mathematics.

    python tests/golden/make_golden_independent.py
"""
import os

import numpy as np
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))


def box_cloud(rng, n_clusters=256, pts=8):
    """n_clusters boxes of `pts` points, 1 m apart along x: the filter's median splits (always along x, the widest
    extent by far, exact halvings of a power-of-two count) end with one cluster per box when knn == pts."""
    cloud = np.ones((n_clusters * pts, 4), np.float32)
    normals = np.zeros((n_clusters, 3))
    kept = np.ones(n_clusters, bool)
    for c in range(n_clusters):
        centre = np.array([c * 1.0, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)])   # x stays the widest extent of every segment
        if c % 16 == 7:                       # exactly collinear: covariance of rank 1 -> the filter drops the box
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            t = np.linspace(-0.03, 0.03, pts)
            p = centre + t[:, None] * d[None, :]
            # keep the points exactly collinear in float32: move along one axis only
            p = centre + np.outer(t, np.eye(3)[c % 3])
            kept[c] = False
        else:                                 # noisy planar patch, 6 cm across, 1 mm out-of-plane noise
            n = rng.normal(size=3); n /= np.linalg.norm(n)
            u = np.cross(n, [1.0, 0.3, 0.2]); u /= np.linalg.norm(u)
            v = np.cross(n, u)
            ab = rng.uniform(-0.03, 0.03, (pts, 2))
            p = centre + ab[:, :1] * u + ab[:, 1:] * v + rng.normal(0, 0.001, (pts, 1)) * n
        cloud[c * pts:(c + 1) * pts, :3] = p.astype(np.float32)
    # expected normals from the float32 coordinates the filters will see, in float64
    for c in range(n_clusters):
        p = cloud[c * pts:(c + 1) * pts, :3].astype(np.float64)
        d = p - p.mean(0)
        w, v = np.linalg.eigh(d.T @ d)
        normals[c] = v[:, 0]
        if kept[c]:
            assert w[0] < 0.2 * w[1], "patch not clearly planar"
        else:
            assert np.linalg.matrix_rank(d.T @ d) == 1
    perm = rng.permutation(cloud.shape[0])   # the caller's point order is arbitrary
    return cloud[perm], (perm // pts).astype(np.int32), normals, kept


def main():
    rng = np.random.default_rng(20260927)
    out = {}
    cloud, cluster_of_point, normals, kept = box_cloud(rng)
    out.update(box_cloud=cloud, box_cluster_of_point=cluster_of_point, box_normals=normals, box_kept=kept)

    g = np.load(os.path.join(HERE, "icp_pair4k.npz"))
    A, b = g["A0"].astype(np.float64), g["b0"].astype(np.float64)
    A32, b32 = A.astype(np.float32), b.astype(np.float32)
    L = np.linalg.cholesky(A32)
    y = np.linalg.solve(L, b32).astype(np.float32)
    out["solve_x_f32"] = np.linalg.solve(L.T, y).astype(np.float32)
    L64 = np.linalg.cholesky(A)
    out["solve_x_f64"] = np.linalg.solve(L64.T, np.linalg.solve(L64, b))
    out["solve_cond"] = np.float64(np.linalg.cond(A))

    # rotation metric: pairs of nearly equal rotations (what the checker sees) and a few large ones, incl. traces < 0
    Ta, Tb, ang = [], [], []
    for i in range(64):
        Ra = Rotation.random(random_state=int(rng.integers(1 << 31)))
        mag = 10.0 ** rng.uniform(-5, -1) if i < 48 else rng.uniform(0.5, 3.0)
        Rd = Rotation.from_rotvec(mag * (lambda v: v / np.linalg.norm(v))(rng.normal(size=3)))
        Rb = Rd * Ra
        for R, store in ((Ra, Ta), (Rb, Tb)):
            T = np.eye(4); T[:3, :3] = R.as_matrix(); T[:3, 3] = rng.uniform(-5, 5, 3)
            store.append(np.ascontiguousarray(T.astype(np.float32).T).reshape(16))     # column major
        # expected from the float32 matrices actually handed over
        Ma = np.array(Ta[-1], np.float64).reshape(4, 4).T[:3, :3]; Mb = np.array(Tb[-1], np.float64).reshape(4, 4).T[:3, :3]
        ang.append((Rotation.from_matrix(Ma) * Rotation.from_matrix(Mb).inv()).magnitude())
    out.update(rot_Ta=np.stack(Ta), rot_Tb=np.stack(Tb), rot_angle=np.array(ang))

    # trim limit: ties, infinities (unmatched), ratio * n exactly integral and not
    trims = []
    for n, ratio, ninf in ((1000, 0.75, 0), (1001, 0.75, 7), (4096, 0.85, 100), (17, 0.5, 0), (5, 1.0, 0), (333, 0.999, 3)):
        d2 = rng.gamma(2.0, 0.002, n).astype(np.float32)
        d2[rng.choice(n, n // 10, replace=False)] = d2[0]            # ties
        if ninf:
            d2[rng.choice(n, ninf, replace=False)] = np.inf
        vals = d2[np.isfinite(d2)]                                   # the filter looks at the matched pairs only
        k = int(np.float32(vals.size) * np.float32(ratio))
        k = min(max(k, 0), vals.size - 1)
        trims.append((d2, np.float32(ratio), np.float32(np.partition(vals, k)[k])))
    out["trim_n"] = np.int32(len(trims))
    for i, (d2, ratio, lim) in enumerate(trims):
        out[f"trim{i}_d2"], out[f"trim{i}_ratio"], out[f"trim{i}_limit"] = d2, ratio, lim

    path = os.path.join(HERE, "independent_kat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; cond(A) =", float(out["solve_cond"]))


if __name__ == "__main__":
    main()
