#!/usr/bin/env python3
"""Design study (dev only, CPU): per-query cached need masks over a per-tile cached chunk list, row-level evaluation.

At a (re)build every lane of a tile gets reach = R + s (R = its search radius, s = slack) and a mask of the chunks
(<= CH points) within reach; the tile's list = union over its lanes.  Each iteration the slack shrinks by the lane's
displacement; a searching lane whose radius no longer fits rebuilds its tile.  A round evaluates one chunk per 16-lane
row: rounds = max over the 4 rows of |OR of the searching lanes' masks|.
usage: sim_masks.py CH slack_mm slack_factor start_iter
"""
import sys, os, pickle
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S_MIN = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.002
S_FAC = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
START = int(sys.argv[4]) if len(sys.argv) > 4 else 3
NT = int(sys.argv[5]) if len(sys.argv) > 5 else 800
n_az = 16384
ref, nrm, rd, T_init, Ts, limits = pickle.load(open(f"/tmp/sim/pair_{n_az}.pkl", "rb"))
mean = ref[:, :3].astype(np.float64).mean(0)
refc = (ref[:, :3] - mean).astype(np.float64)
Tm = np.eye(4); Tm[:3, 3] = -mean
rdc = (rd[:, :3].astype(np.float64) @ (Tm @ T_init)[:3, :3].T) + (Tm @ T_init)[:3, 3]
p = rd[:, :3].astype(np.float64)
rng_ = np.linalg.norm(p, axis=1)
el = np.degrees(np.arcsin(p[:, 2] / rng_)); az = np.degrees(np.arctan2(p[:, 1], p[:, 0])) % 360.0
eb = np.floor((el + 90) / 0.57).astype(np.int64); sb = np.floor(az / 0.25).astype(np.int64); rb = np.floor(rng_).astype(np.int64)
order = np.lexsort((az, rb, sb, eb))
rdc = rdc[order]
h0 = 0.125; fine = 5; hf = h0 / 32
o = refc.min(0)
fc = np.floor((refc - o) / hf).astype(np.int64)
def spread3(v):
    x = v & 0x1FFFFF
    x = (x | x << 32) & 0x1F00000000FFFF
    x = (x | x << 16) & 0x1F0000FF0000FF
    x = (x | x << 8) & 0x100F00F00F00F00F
    x = (x | x << 4) & 0x10C30C30C30C30C3
    x = (x | x << 2) & 0x1249249249249249
    return x
key = spread3(fc[:, 0]) | (spread3(fc[:, 1]) << 1) | (spread3(fc[:, 2]) << 2)
perm = np.argsort(key, kind="stable")
pts = refc[perm]; key = key[perm]
cell0 = key >> (3 * fine)
tree = cKDTree(pts)
n = pts.shape[0]
newcell = np.ones(n, bool); newcell[1:] = cell0[1:] != cell0[:-1]
idx = np.arange(n)
flag = newcell | ((idx % CH) == 0)
st = np.nonzero(flag)[0]; en = np.append(st[1:], n); ccnt = en - st
clo = np.minimum.reduceat(pts, st, axis=0); chi = np.maximum.reduceat(pts, st, axis=0)
cid = np.cumsum(flag) - 1
print("CH", CH, "chunks", len(st), "mean fill", ccnt.mean())
def ptboxdist(lo, hi, q):
    g = np.maximum(np.maximum(lo - q, q - hi), 0)
    return np.sqrt((g * g).sum(-1))

GAP = 0.002
rs = np.random.default_rng(0)
nq = rdc.shape[0]; nt = nq // 64
tiles = np.sort(rs.choice(nt, NT, replace=False))
sel = (tiles[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
rq = rdc[sel]; n = rq.shape[0]
lb = np.zeros(n); match = np.full(n, -1); q_prev = None
LOOSE = float(sys.argv[6]) if len(sys.argv) > 6 else 1.5
reach0 = np.zeros(n); slack = np.zeros(n); masks = [None] * n; tlist = [None] * NT; built = np.zeros(NT, bool)
tot_r = 0; tot_b = 0; its = 0
for k, T in enumerate([np.eye(4)] + Ts[:-1]):
    q = rq @ T[:3, :3].T + T[:3, 3]
    dd, ii = tree.query(q, k=2, workers=8)
    if k == 0:
        match = ii[:, 0].copy(); lb = dd[:, 1].copy(); q_prev = q; continue
    lim = limits[k - 1]; cap = np.sqrt(1.1 * lim)
    delta = np.linalg.norm(q - q_prev, axis=1)
    lbn = np.maximum(lb - delta, 0)
    ub = np.linalg.norm(q - pts[match], axis=1)
    keep = ub < lbn; far = np.minimum(ub, lbn) > cap; search = ~(keep | far)
    R = np.minimum(ub + GAP, cap * 1.05)
    if k >= START:
        slack = slack - delta
        rebuilt = 0; rounds = []; cands = []; lists = []; over = 0; evals_tile = []
        for t in range(NT):
            sl = slice(t * 64, t * 64 + 64); s = search[sl]
            want = R[sl] + np.maximum(S_MIN, S_FAC * delta[sl])
            bad = (not built[t]) or (s & (R[sl] > slack[sl])).any() or (reach0[sl].mean() > LOOSE * want.mean())
            if bad:
                rebuilt += 1
                reach = R[sl] + np.maximum(S_MIN, S_FAC * delta[sl])
                qs = q[sl]
                blo = (qs - reach[:, None]).min(0); bhi = (qs + reach[:, None]).max(0)
                c = (blo + bhi) / 2; rad = np.linalg.norm(bhi - blo) / 2 + 0.3
                near = np.array(tree.query_ball_point(c, rad), dtype=np.int64)
                cs = np.unique(cid[near]) if near.size else np.zeros(0, np.int64)
                g = np.maximum(np.maximum(clo[cs] - bhi, blo - chi[cs]), 0)
                cs = cs[(g == 0).all(1)]
                tlist[t] = cs
                for j in range(64):
                    masks[t * 64 + j] = ptboxdist(clo[cs], chi[cs], qs[j]) <= reach[j]
                slack[sl] = reach; reach0[sl] = reach
                built[t] = True
            cs = tlist[t]; lists.append(len(cs))
            if len(cs) > 64: over += 1
            if not s.any(): rounds.append(0); cands.append(0); evals_tile.append(0); continue
            rr = []; rc = []
            anym = np.zeros(len(cs), bool)
            for r in range(4):
                m = np.zeros(len(cs), bool)
                for j in range(16):
                    if s[r * 16 + j]: m |= masks[t * 64 + r * 16 + j]
                rr.append(m.sum()); rc.append((((ccnt[cs][m] + 3) // 4) * 4).sum()); anym |= m
            rounds.append(max(rr)); cands.append(max(rc)); evals_tile.append((((ccnt[cs][anym] + 3) // 4) * 4).sum())
        print(f"it {k:2d} search {search.mean()*100:4.1f}% rebuilt {rebuilt/NT*100:5.1f}% | list mean {np.mean(lists):5.1f} p90 {np.percentile(lists,90):4.0f} p99 {np.percentile(lists,99):4.0f} >64: {over/NT*100:4.1f}% | "
              f"rounds mean {np.mean(rounds):5.2f} p90 {np.percentile(rounds,90):3.0f} | cand/lane(max row) {np.mean(cands):6.1f} | tile-union cand {np.mean(evals_tile):6.1f}", flush=True)
        tot_r += np.mean(rounds); tot_b += rebuilt / NT; its += 1
    found = dd[:, 0] <= cap * 1.05
    newmatch = np.where(search & found, ii[:, 0], match)
    same = newmatch == match
    other = np.where(ii[:, 0] == newmatch, dd[:, 1], dd[:, 0])
    nb = np.minimum(other, R)
    nb = np.where(search, np.where(same, np.maximum(nb, lbn), nb), lbn)
    match = newmatch; lb = nb; q_prev = q
print(f"CH {CH} slack min {S_MIN*1e3} mm factor {S_FAC}: mean rounds {tot_r/its:.2f}, rebuild fraction {tot_b/its:.3f}")
