"""Design study (dev only, CPU, numpy): what a per-lane search walks next to a wall, today and with local refinement.

DESIGN.md section 8, item 2: on a sub-map with a wall 0.8 m from the sensor a handful of 12.5 cm cells hold tens of thousands
of points; a query whose ball touches one of them walks every 64-point chunk box of the cell (lane_ball_search /
rowq_search / k_knn_fallback all cull a cell's chunks linearly).  This script counts, for every query of the reading, the
chunk boxes in the level-0 cells its ball touches -- as the grid is built now, and if every cell above `--split` points
were given children (octree, down to h0 / 8) whose own chunk ranges the search could address.
   python devtools/sim_dense_cells.py [scan index (18)] [--radius 0.01] [--split 1024]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from laser_slam_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("scan", nargs="?", type=int, default=18)
ap.add_argument("--radius", type=float, default=0.01)
ap.add_argument("--split", type=int, default=1024)
ap.add_argument("--n-az", type=int, default=16384)
args = ap.parse_args()
i, H0 = args.scan, 0.125
E = synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))
pose = lambda k: synth.se3(0.8 * k, 0.05 * k, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * k))
scans = {k: synth.scan_job((1234, pose(k), args.n_az, 10 + k)) for k in range(i - 3, i + 1)}
a = i - 1
parts = [scans[a]] + [(scans[k] @ (np.linalg.inv(pose(a)) @ pose(k)).astype(np.float32).T) for k in (i - 2, i - 3)]
ref = np.concatenate(parts)[::2, :3].astype(np.float64)      # stand-in for the surface-normal filter's ratio 0.5
T = np.linalg.inv(pose(a)) @ pose(i)
q = (scans[i][::2, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3])   # reading (prob 0.5), at the true pose
lo = ref.min(0) - 1.0


def cell_keys(p, h):
    c = np.floor((p - lo) / h).astype(np.int64)
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def chunks_walked(levels):
    """levels: list of (h, dict cell key -> chunk count) from coarse to fine; a cell present at a finer level replaces its
    parent.  For every query: chunk boxes in the cells its ball's bounding box touches, at the finest level available."""
    out = np.zeros(len(q), np.int64)
    R = args.radius
    # corners of the ball's bounding box: <= 8 distinct cells per level
    offs = np.array([[sx, sy, sz] for sx in (-R, R) for sy in (-R, R) for sz in (-R, R)])
    for h, table, parents_split in levels:
        keys = np.stack([cell_keys(q + o, h) for o in offs], 1)          # n x 8
        keys.sort(1)
        first = np.ones_like(keys, bool); first[:, 1:] = keys[:, 1:] != keys[:, :-1]
        uk, inv = np.unique(keys, return_inverse=True)
        cnt = np.array([table.get(int(k), 0) for k in uk])[inv.reshape(keys.shape)]
        out += (cnt * first).sum(1)
    return out


def build(h, pts):
    k, c = np.unique(cell_keys(pts, h), return_counts=True)
    return k, c


k0, c0 = build(H0, ref)
now = [(H0, {int(k): int((c + 63) // 64) for k, c in zip(k0, c0)}, None)]
# refinement: cells above the split count are replaced by their children, recursively (3 levels)
levels, cur_pts, h = [], ref, H0
table0 = {int(k): int((c + 63) // 64) for k, c in zip(k0, c0)}
split_keys = set(int(k) for k, c in zip(k0, c0) if c > args.split)
for k in split_keys: table0[k] = 0          # a split cell's chunks are reached through its children
levels.append((H0, table0, None))
pts, parent_h, parents = ref, H0, split_keys
for depth in range(3):
    if not parents: break
    inside = np.isin(cell_keys(pts, parent_h), np.fromiter(parents, np.int64))
    pts = pts[inside]
    h = parent_h / 2
    k, c = build(h, pts)
    last = depth == 2
    tab = {int(kk): int((cc + 63) // 64) for kk, cc in zip(k, c)}
    nxt = set() if last else set(int(kk) for kk, cc in zip(k, c) if cc > args.split)
    for kk in nxt: tab[kk] = 0
    levels.append((h, tab, None))
    parent_h, parents = h, nxt
w_now, w_ref = chunks_walked(now), chunks_walked(levels)
print("scan %d: reference %d points in %d level-0 cells (largest %d points), %d cells above %d points; reading %d queries, ball radius %.3f m" % (
    i, len(ref), len(k0), c0.max(), len(split_keys), args.split, len(q), args.radius))
for name, w in (("as built", w_now), ("with children", w_ref)):
    tiles = w[: len(w) // 64 * 64].reshape(-1, 64).max(1)       # a spread tile in per-lane mode is as slow as its slowest lane
    print("%-14s chunk boxes walked per query: mean %.1f p99 %d p99.9 %d max %d; slowest lane of a 64-query tile: p99 %d max %d" % (
        name, w.mean(), np.percentile(w, 99), np.percentile(w, 99.9), w.max(), np.percentile(tiles, 99), tiles.max()))
