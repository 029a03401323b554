#!/usr/bin/env python3
"""Design study (dev only, CPU): per-QUERY cached candidate ranges, searched by a quad of lanes; a tile whose
searching lane outgrew its cache goes through the broadcast search again and refreshes every lane's cache.

Cache of a query = the chunks (<= CH points) within reach = R + s of it (R = its search radius, also for keep / far
lanes: the radius they would search with), s = max(s_min, f * displacement of this iteration).
usage: sim_qc.py CH s_min_mm f slots start_iter ntiles
"""
import sys, os, pickle
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S_MIN = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.003
S_FAC = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
SLOTS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
START = int(sys.argv[5]) if len(sys.argv) > 5 else 3
NT = int(sys.argv[6]) if len(sys.argv) > 6 else 600
ref, nrm, rd, T_init, Ts, limits = pickle.load(open("/tmp/sim/pair_16384.pkl", "rb"))
mean = ref[:, :3].astype(np.float64).mean(0)
refc = (ref[:, :3] - mean).astype(np.float64)
Tm = np.eye(4); Tm[:3, 3] = -mean
rdc = (rd[:, :3].astype(np.float64) @ (Tm @ T_init)[:3, :3].T) + (Tm @ T_init)[:3, 3]
p = rd[:, :3].astype(np.float64)
rng_ = np.linalg.norm(p, axis=1)
el = np.degrees(np.arcsin(p[:, 2] / rng_)); az = np.degrees(np.arctan2(p[:, 1], p[:, 0])) % 360.0
eb = np.floor((el + 90) / 0.57).astype(np.int64); sb = np.floor(az / 0.25).astype(np.int64); rb = np.floor(rng_).astype(np.int64)
order = np.lexsort((az, rb, sb, eb)); rdc = rdc[order]
h0 = 0.125; fine = 5; hf = h0 / 32
o = refc.min(0); fc = np.floor((refc - o) / hf).astype(np.int64)
def spread3(v):
    x = v & 0x1FFFFF
    x = (x | x << 32) & 0x1F00000000FFFF; x = (x | x << 16) & 0x1F0000FF0000FF
    x = (x | x << 8) & 0x100F00F00F00F00F; x = (x | x << 4) & 0x10C30C30C30C30C3
    x = (x | x << 2) & 0x1249249249249249
    return x
key = spread3(fc[:, 0]) | (spread3(fc[:, 1]) << 1) | (spread3(fc[:, 2]) << 2)
perm = np.argsort(key, kind="stable"); pts = refc[perm]; key = key[perm]
cell0 = key >> (3 * fine); tree = cKDTree(pts); n = pts.shape[0]
newcell = np.ones(n, bool); newcell[1:] = cell0[1:] != cell0[:-1]
flag = newcell | ((np.arange(n) % CH) == 0)
st = np.nonzero(flag)[0]; en = np.append(st[1:], n); ccnt = en - st
clo = np.minimum.reduceat(pts, st, axis=0); chi = np.maximum.reduceat(pts, st, axis=0)
cid = np.cumsum(flag) - 1
GAP = 0.002
rs = np.random.default_rng(0)
nq = rdc.shape[0]; nt = nq // 64
tiles = np.sort(rs.choice(nt, NT, replace=False))
sel = (tiles[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
rq = rdc[sel]; n = rq.shape[0]
lb = np.zeros(n); match = np.full(n, -1); q_prev = None
slack = np.full(n, -1.0); ccand = np.zeros(n); cchunks = np.zeros(n, np.int64)
tot = dict(reb=0, q=0, it=0, cost=0.0)
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
        reb = 0; ns_fast = []; passes = []; steps = []; over = 0
        for t in range(NT):
            sl = slice(t * 64, t * 64 + 64); s = search[sl]
            bad = (s & (R[sl] > slack[sl])).any()
            if bad:
                reb += 1
                reach = R[sl] + np.maximum(S_MIN, S_FAC * delta[sl])
                qs = q[sl]
                blo = (qs - reach[:, None]).min(0); bhi = (qs + reach[:, None]).max(0)
                c = (blo + bhi) / 2; rad = np.linalg.norm(bhi - blo) / 2 + 0.3
                near = np.array(tree.query_ball_point(c, rad), dtype=np.int64)
                cs = np.unique(cid[near]) if near.size else np.zeros(0, np.int64)
                g = np.maximum(np.maximum(clo[cs][None, :, :] - qs[:, None, :], qs[:, None, :] - chi[cs][None, :, :]), 0)
                dist = np.sqrt((g * g).sum(-1))                    # [64, nchunks]
                inm = dist <= reach[:, None]
                cchunks[sl] = inm.sum(1); ccand[sl] = (inm * ccnt[cs][None, :]).sum(1)
                slack[sl] = np.where(cchunks[sl] <= SLOTS, reach, -1.0)   # too many chunks: never valid
                over += (cchunks[sl] > SLOTS).sum()
            else:
                # fast path: quads walk their own candidates, 16 queries per pass in lane order
                js = np.nonzero(s)[0]
                ns_fast.append(len(js))
                np_ = 0; stp = 0
                for p0 in range(0, len(js), 16):
                    grp = js[p0:p0 + 16]
                    np_ += 1
                    # a quad takes its chunks one after the other, 4 points per step: steps = max over quads of sum ceil(cnt/4)
                    stp += int(np.ceil(ccand[sl][grp] / 4 + 0.5 * cchunks[sl][grp]).max())
                passes.append(np_); steps.append(stp)
        fast = NT - reb
        mean_steps = np.mean(steps) if steps else 0; mean_pass = np.mean(passes) if passes else 0
        cost = (reb * 1843 * 1.3 + fast * 110 + sum(steps) * 13 + sum(passes) * 50) / NT
        print(f"it {k:2d} search {search.mean()*100:4.1f}% tiles rebuilt {reb/NT*100:5.1f}% | cache: chunks/query {cchunks.mean():4.2f} p99 {np.percentile(cchunks,99):3.0f} cand/query {ccand.mean():6.1f} p90 {np.percentile(ccand,90):5.0f} overflow {over} | "
              f"fast tiles: passes {mean_pass:4.2f} steps {mean_steps:6.1f} | est. instr/tile {cost:6.0f}", flush=True)
        tot["reb"] += reb / NT; tot["it"] += 1; tot["cost"] += cost
    found = dd[:, 0] <= cap * 1.05
    newmatch = np.where(search & found, ii[:, 0], match)
    same = newmatch == match
    other = np.where(ii[:, 0] == newmatch, dd[:, 1], dd[:, 0])
    nb = np.minimum(other, R)
    nb = np.where(search, np.where(same, np.maximum(nb, lbn), nb), lbn)
    match = newmatch; lb = nb; q_prev = q
print(f"CH {CH} s_min {S_MIN*1e3} mm f {S_FAC} slots {SLOTS}: rebuild fraction {tot['reb']/tot['it']:.3f}, est. instr/tile {tot['cost']/tot['it']:.0f} (today 1843)")
