#!/usr/bin/env python3
"""Design study (dev only, CPU, scipy): what a settled kNN launch of the benchmark pair has to look at.

Runs the oracle's ICP on the benchmark pair to get T_iter per iteration, replays the keep / far bookkeeping of
k_knn_tile with exact neighbours from scipy's cKDTree, and reports per iteration: the fraction of queries that
search, their ball radii, and the number of reference points inside (i) each searching query's own ball, (ii) the
dilated box of its 16-query row, (iii) the dilated box of its 64-query tile.
"""
import sys, os, time, pickle
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laser_slam_amd import synth
from oracle import oracle_py as O

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cache = f"/tmp/sim/pair_{n_az}.pkl"
if os.path.exists(cache):
    ref, nrm, rd, T_init, Ts, limits = pickle.load(open(cache, "rb"))
else:
    ref0, rd, T_true, T_init = synth.scan_pair(n_az, noise_seeds=(1, 2), guess_seed=7)
    ref, nrm = O.sampling_surface_normal(ref0, 10, 1.0, 0)
    cfg = O.config_yaml(min_diff_rot=1e-5, min_diff_trans=1e-4, num_threads=8, reading_sampling_prob=1.0)
    t0 = time.time()
    rc, T, st, tr = O.icp_compute(cfg, rd, ref, nrm, synth.colmajor(T_init), trace_cap=64)
    print("oracle icp", time.time() - t0, "s", st.iterations, "iterations")
    Ts = [np.array(t['T_iter'], np.float64).reshape(4, 4).T for t in tr]
    limits = [t['limit'] for t in tr]
    pickle.dump((ref, nrm, rd, T_init, Ts, limits), open(cache, "wb"))

# the oracle iterates in the frame centred on the reference mean
mean = ref[:, :3].astype(np.float64).mean(0)
refc = (ref[:, :3] - mean).astype(np.float64)
Tm = np.eye(4); Tm[:3, 3] = -mean
rdc = (rd[:, :3].astype(np.float64) @ (Tm @ T_init)[:3, :3].T) + (Tm @ T_init)[:3, 3]
tree = cKDTree(refc)

# query order: spherical cells in the reading's own frame (k_query_keys): elevation bin, azimuth sector, range bin
p = rd[:, :3].astype(np.float64)
rng_ = np.linalg.norm(p, axis=1)
el = np.degrees(np.arcsin(p[:, 2] / rng_)); az = np.degrees(np.arctan2(p[:, 1], p[:, 0])) % 360.0
eb = np.floor((el + 90) / 0.57).astype(np.int64); sb = np.floor(az / 0.25).astype(np.int64); rb = np.floor(rng_).astype(np.int64)
order = np.lexsort((az, rb, sb, eb))
rdc = rdc[order]
nq = rdc.shape[0]
print("nq", nq, "nr", refc.shape[0], "iterations", len(Ts))

GAP = 0.002
lb = np.zeros(nq); match = np.full(nq, -1); q_prev = None
rs = np.random.default_rng(0)
for k, T in enumerate([np.eye(4)] + Ts[:-1]):   # T_iter used by iteration k's search
    q = rdc @ T[:3, :3].T + T[:3, 3]
    dd, ii = tree.query(q, k=2, workers=8)
    if k == 0:
        match = ii[:, 0].copy(); lb = dd[:, 1].copy(); q_prev = q; continue
    lim = limits[k - 1]; cap2 = 1.1 * lim; cap = np.sqrt(cap2)
    delta = np.linalg.norm(q - q_prev, axis=1)
    lbn = np.maximum(lb - delta, 0)
    ub = np.linalg.norm(q - refc[match], axis=1)
    keep = ub < lbn
    far = np.minimum(ub, lbn) > cap
    search = ~(keep | far)
    R = np.minimum(ub + GAP, cap * 1.05)
    # after the search: new match = exact NN if inside cap*1.05, else unchanged
    found = dd[:, 0] <= cap * 1.05
    newmatch = np.where(search & found, ii[:, 0], match)
    same = newmatch == match
    # new lb: second nearest or search radius
    other = np.where(ii[:, 0] == newmatch, dd[:, 1], dd[:, 0])
    nb = np.minimum(other, R)
    nb = np.where(search, np.where(same, np.maximum(nb, lbn), nb), lbn)
    # ---- statistics on a sample of tiles
    nt = nq // 64
    tiles = rs.choice(nt, 1500, replace=False)
    ball_cnt = []; row_cnt = []; tile_cnt = []; row_any = 0; tile_any = 0; rows_tot = 0
    for t in tiles:
        sl = slice(t * 64, t * 64 + 64)
        s = search[sl]
        if not s.any(): continue
        tile_any += 1
        qs = q[sl]; Rs = R[sl]
        lo = (qs[s] - Rs[s, None]).min(0); hi = (qs[s] + Rs[s, None]).max(0)
        c = (lo + hi) / 2; rad = np.linalg.norm(hi - lo) / 2
        idx = np.array(tree.query_ball_point(c, rad), dtype=np.int64)
        pts = refc[idx] if idx.size else np.zeros((0, 3))
        inb = ((pts >= lo) & (pts <= hi)).all(1)
        tile_cnt.append(inb.sum())
        pts = pts[inb]
        for r in range(4):
            rows_tot += 1
            sr = s[r * 16:(r + 1) * 16]
            if not sr.any(): continue
            row_any += 1
            qr = qs[r * 16:(r + 1) * 16][sr]; Rr = Rs[r * 16:(r + 1) * 16][sr]
            lo_r = (qr - Rr[:, None]).min(0); hi_r = (qr + Rr[:, None]).max(0)
            row_cnt.append(((pts >= lo_r) & (pts <= hi_r)).all(1).sum())
            for qq, rr in zip(qr, Rr):
                ball_cnt.append((np.linalg.norm(pts - qq, axis=1) <= rr).sum())
    pc = lambda a, p_: np.percentile(a, p_) if len(a) else 0
    print(f"it {k:2d} lim {np.sqrt(lim)*100:5.2f}cm search {search.mean()*100:5.1f}% keep {keep.mean()*100:5.1f}% far {far.mean()*100:5.1f}% "
          f"delta med {np.median(delta)*1000:6.2f}mm R med {np.median(R[search])*100:5.2f} p90 {pc(R[search],90)*100:5.2f}cm | "
          f"ball mean {np.mean(ball_cnt):6.1f} p90 {pc(ball_cnt,90):5.0f} | row mean {np.mean(row_cnt):6.1f} p90 {pc(row_cnt,90):5.0f} rows-active {row_any/max(rows_tot,1)*100:4.0f}% | "
          f"tile mean {np.mean(tile_cnt):6.1f} p90 {pc(tile_cnt,90):5.0f} tiles-active {tile_any/len(tiles)*100:4.0f}%")
    match = newmatch; lb = nb; q_prev = q
