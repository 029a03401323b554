#!/usr/bin/env python3
"""Design study (dev only, CPU): cached per-row candidate lists for the settled kNN launches.

Row = 16 consecutive queries (query order of k_query_keys).  A row's list = every reference point inside
AABB(row queries) dilated by (cap * 1.05 + m).  A tile (4 rows) rebuilds when a searching lane's ball leaves its row's
box.  Reports per iteration: tiles rebuilt, list lengths (max over the 4 rows of a tile, in blocks of 16).
"""
import sys, os, pickle
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

n_az = 16384
M = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
START = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ROW = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ref, nrm, rd, T_init, Ts, limits = pickle.load(open(f"/tmp/sim/pair_{n_az}.pkl", "rb"))
mean = ref[:, :3].astype(np.float64).mean(0)
refc = (ref[:, :3] - mean).astype(np.float64)
Tm = np.eye(4); Tm[:3, 3] = -mean
rdc = (rd[:, :3].astype(np.float64) @ (Tm @ T_init)[:3, :3].T) + (Tm @ T_init)[:3, 3]
tree = cKDTree(refc)
p = rd[:, :3].astype(np.float64)
rng_ = np.linalg.norm(p, axis=1)
el = np.degrees(np.arcsin(p[:, 2] / rng_)); az = np.degrees(np.arctan2(p[:, 1], p[:, 0])) % 360.0
eb = np.floor((el + 90) / 0.57).astype(np.int64); sb = np.floor(az / 0.25).astype(np.int64); rb = np.floor(rng_).astype(np.int64)
order = np.lexsort((az, rb, sb, eb))
rdc = rdc[order]
nq = rdc.shape[0]
GAP = 0.002
rs = np.random.default_rng(0)
nt = nq // 64
tiles = np.sort(rs.choice(nt, 3000, replace=False))
sel = (tiles[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)     # sampled queries
rdc = rdc[sel]; nq = rdc.shape[0]; nrow = nq // ROW; RPT = 64 // ROW
box_lo = np.zeros((nrow, 3)); box_hi = np.zeros((nrow, 3)); have = np.zeros(nrow, bool); cnt = np.zeros(nrow, np.int64)
lb = np.zeros(nq); match = np.full(nq, -1); q_prev = None
tot_eval = 0; tot_rebuild = 0
for k, T in enumerate([np.eye(4)] + Ts[:-1]):
    q = rdc @ T[:3, :3].T + T[:3, 3]
    dd, ii = tree.query(q, k=2, workers=8)
    if k == 0:
        match = ii[:, 0].copy(); lb = dd[:, 1].copy(); q_prev = q; continue
    lim = limits[k - 1]; cap = np.sqrt(1.1 * lim)
    delta = np.linalg.norm(q - q_prev, axis=1)
    lbn = np.maximum(lb - delta, 0)
    ub = np.linalg.norm(q - refc[match], axis=1)
    keep = ub < lbn
    far = np.minimum(ub, lbn) > cap
    search = ~(keep | far)
    R = np.minimum(ub + GAP, cap * 1.05)
    if k >= START:
        qr = q.reshape(nrow, ROW, 3); Rr = R.reshape(nrow, ROW); sr = search.reshape(nrow, ROW)
        inside = ((qr - Rr[:, :, None] >= box_lo[:, None, :]) & (qr + Rr[:, :, None] <= box_hi[:, None, :])).all(2)
        row_ok = have & (inside | ~sr).all(1)
        tile_ok = row_ok.reshape(-1, RPT).all(1)
        rebuild_rows = np.repeat(~tile_ok, RPT)
        for r in np.nonzero(rebuild_rows)[0]:
            lo = qr[r].min(0) - (cap * 1.05 + M); hi = qr[r].max(0) + (cap * 1.05 + M)
            c = (lo + hi) / 2; rad = np.linalg.norm(hi - lo) / 2
            idx = np.array(tree.query_ball_point(c, rad), dtype=np.int64)
            pts = refc[idx] if idx.size else np.zeros((0, 3))
            cnt[r] = ((pts >= lo) & (pts <= hi)).all(1).sum()
            box_lo[r] = lo; box_hi[r] = hi; have[r] = True
        active = sr.any(1)
        blocks = np.where(active, (cnt + 15) // 16, 0).reshape(-1, RPT)
        tmax = blocks.max(1)
        tot_eval += tmax.mean(); tot_rebuild += (~tile_ok).mean()
        print(f"it {k:2d} cap {cap*100:5.2f}cm rebuilt {(~tile_ok).mean()*100:5.1f}% tiles | list len mean {cnt.mean():6.1f} p50 {np.median(cnt):5.0f} p90 {np.percentile(cnt,90):5.0f} p99 {np.percentile(cnt,99):5.0f} max {cnt.max()} | "
              f"blocks/tile(max of rows) mean {tmax.mean():5.2f} p90 {np.percentile(tmax,90):4.0f}  sum-of-rows {blocks.sum(1).mean():5.2f}")
    found = dd[:, 0] <= cap * 1.05
    newmatch = np.where(search & found, ii[:, 0], match)
    same = newmatch == match
    other = np.where(ii[:, 0] == newmatch, dd[:, 1], dd[:, 0])
    nb = np.minimum(other, R)
    nb = np.where(search, np.where(same, np.maximum(nb, lbn), nb), lbn)
    match = newmatch; lb = nb; q_prev = q
n_it = len(Ts) - START
print(f"margin {M*100:.1f} cm rows of {ROW}: mean blocks/tile/iter {tot_eval/n_it:.2f}, rebuild fraction {tot_rebuild/n_it:.3f}")
