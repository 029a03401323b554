#!/usr/bin/env python3
"""Design study (dev only, CPU): per-row evaluation over small chunks against today's tile-level evaluation.

Emulates the reference grid (Morton-sorted points, level-0 cells of 12.5 cm, chunks of <= CH consecutive points inside
one cell) and, for sampled tiles of settled iterations, counts what a wave would evaluate:
  tile64 : today -- chunks (<= 64 points) needed by ANY searching lane of the tile, evaluated by all 64 lanes
  rowCH  : chunks (<= CH points) whose box is within reach of a 16-lane row's box; rounds = max over the 4 rows
"""
import sys, os, pickle
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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

# ---- grid emulation
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

def chunks_for(CH):
    n = pts.shape[0]
    newcell = np.ones(n, bool); newcell[1:] = cell0[1:] != cell0[:-1]
    # position inside the cell run
    idx = np.arange(n)
    start_of_cell = np.maximum.accumulate(np.where(newcell, idx, 0))
    flag = newcell | (((idx - start_of_cell) % CH) == 0)   # (the device cuts at global multiples of CH; cell-relative is the same idea)
    st = np.nonzero(flag)[0]; en = np.append(st[1:], n)
    lo = np.minimum.reduceat(pts, st, axis=0); hi = np.maximum.reduceat(pts, st, axis=0)
    cid = np.cumsum(flag) - 1
    return st, en - st, lo, hi, cid

CHS = [64, 16, 8]
chunk_sets = {c: chunks_for(c) for c in CHS}
for c in CHS:
    print("CH", c, "chunks", len(chunk_sets[c][0]), "mean fill", chunk_sets[c][1].mean())

def boxdist2(lo, hi, blo, bhi):   # box (lo,hi) [n,3] vs box (blo,bhi)
    g = np.maximum(np.maximum(lo - bhi, blo - hi), 0)
    return (g * g).sum(-1)
def ptboxdist2(lo, hi, q):
    g = np.maximum(np.maximum(lo - q, q - hi), 0)
    return (g * g).sum(-1)

GAP = 0.002
rs = np.random.default_rng(0)
nq = rdc.shape[0]; nt = nq // 64
tiles = np.sort(rs.choice(nt, 1200, replace=False))
sel = (tiles[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
rq = rdc[sel]; n = rq.shape[0]
lb = np.zeros(n); match = np.full(n, -1); q_prev = None
REPORT = {4, 6, 8, 12, 16, 20, 24, 28, 31}
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
    if k in REPORT:
        out = {}
        for t in range(len(tiles)):
            sl = slice(t * 64, t * 64 + 64); s = search[sl]
            if not s.any(): continue
            qs = q[sl]; Rs = R[sl]
            tlo = qs[s].min(0); thi = qs[s].max(0); Rmax = Rs[s].max()
            ext = (thi - tlo).max()
            c = (tlo + thi) / 2; rad = np.linalg.norm(thi - tlo) / 2 + Rmax + 0.3
            near = np.array(tree.query_ball_point(c, rad), dtype=np.int64)
            for CH in CHS:
                st, cnt, lo, hi, cid = chunk_sets[CH]
                cs = np.unique(cid[near]) if near.size else np.zeros(0, np.int64)
                clo, chi, ccnt = lo[cs], hi[cs], cnt[cs]
                d = out.setdefault(CH, dict(tile_ch=[], tile_cand=[], row_rounds=[], row_cand=[], row_rounds_x=[], row_cand_x=[], tested=[]))
                # tile level: chunk passes the tile box test, and some lane needs it
                tp = boxdist2(clo, chi, tlo, thi) <= Rmax ** 2
                d["tested"].append(tp.sum())
                need = np.zeros(len(cs), bool)
                for j in np.nonzero(s)[0]:
                    need |= ptboxdist2(clo, chi, qs[j]) <= Rs[j] ** 2
                need &= tp
                d["tile_ch"].append(need.sum()); d["tile_cand"].append((((ccnt[need] + 3) // 4) * 4).sum())
                rr = []; rc = []; rrx = []; rcx = []
                for r in range(4):
                    sr = s[r * 16:(r + 1) * 16]
                    if not sr.any(): rr.append(0); rc.append(0); rrx.append(0); rcx.append(0); continue
                    qr = qs[r * 16:(r + 1) * 16][sr]; Rr = Rs[r * 16:(r + 1) * 16][sr]
                    bl = qr.min(0); bh = qr.max(0)
                    m = boxdist2(clo, chi, bl, bh) <= Rr.max() ** 2
                    rr.append(m.sum()); rc.append((((ccnt[m] + 3) // 4) * 4).sum())
                    mx = np.zeros(len(cs), bool)
                    for qq, r_ in zip(qr, Rr):
                        mx |= ptboxdist2(clo, chi, qq) <= r_ ** 2
                    rrx.append(mx.sum()); rcx.append((((ccnt[mx] + 3) // 4) * 4).sum())
                d["row_rounds"].append(max(rr)); d["row_cand"].append(max(rc)); d["row_rounds_x"].append(max(rrx)); d["row_cand_x"].append(max(rcx))
        line = f"it {k:2d} search {search.mean()*100:4.1f}% |"
        for CH in CHS:
            d = out[CH]
            line += (f" CH{CH}: tile chunks {np.mean(d['tile_ch']):5.1f} cand {np.mean(d['tile_cand']):6.1f}; row(box) rounds {np.mean(d['row_rounds']):5.1f} p90 {np.percentile(d['row_rounds'],90):3.0f} cand {np.mean(d['row_cand']):5.1f};"
                     f" row(exact) rounds {np.mean(d['row_rounds_x']):5.1f} cand {np.mean(d['row_cand_x']):5.1f}; tested {np.mean(d['tested']):5.1f} |")
        print(line, flush=True)
    found = dd[:, 0] <= cap * 1.05
    newmatch = np.where(search & found, ii[:, 0], match)
    same = newmatch == match
    other = np.where(ii[:, 0] == newmatch, dd[:, 1], dd[:, 0])
    nb = np.minimum(other, R)
    nb = np.where(search, np.where(same, np.maximum(nb, lbn), nb), lbn)
    match = newmatch; lb = nb; q_prev = q
