#!/usr/bin/env python3
"""Dev only (CPU): the window arithmetic of k_knn_cone (lsgpu_cone.hip.h) restated operation by operation in float32
numpy, checked for exactness on the benchmark pair: for a sample of searching queries in every settled iteration, every
reference point within the search radius (brute force through scipy, float64) must fall into one of the query's windows.
Hardware rcp / rsq / sqrt are 1 ulp; here they are correctly rounded -- the margins of the kernel are 20x that.
usage: sim_cone_f32.py [rows=128] [cols=8192] [n_az=16384] [samples=20000]
"""
import sys, os, pickle
import numpy as np
from scipy.spatial import cKDTree
f32 = np.float32
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
A = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n_az = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
NS = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
ref, nrm, rd, T_init, Ts, limits = pickle.load(open(f"/tmp/sim/pair_{n_az}.pkl", "rb"))   # written by sim_cone.py
mean = ref[:, :3].astype(np.float64).mean(0).astype(f32)
refc = (ref[:, :3] - mean).astype(f32)                      # what k_ref_gather stores
Tm = np.eye(4); Tm[:3, 3] = -mean.astype(np.float64)
rdc = ((rd[:, :3].astype(np.float64) @ (Tm @ T_init)[:3, :3].T) + (Tm @ T_init)[:3, 3]).astype(f32)
tree = cKDTree(refc.astype(np.float64))
O = (-mean).astype(f32)

def cone_dir(v):
    vx, vy, vz = v[:, 0], v[:, 1], v[:, 2]
    r2 = (vy * vy + vx * vx).astype(f32)
    inv_rho = (f32(1) / np.sqrt((vz * vz + r2).astype(f32))).astype(f32)
    rxy = np.sqrt(r2).astype(f32)
    zeta = (vz * inv_rho).astype(f32)
    inv_h = (f32(1) / (np.abs(vx) + np.abs(vy)).astype(f32)).astype(f32)
    t = (vy * inv_h).astype(f32)
    pa = np.where(vx >= 0, np.where(vy >= 0, t, f32(4) + t), f32(2) - t).astype(f32)
    return inv_rho, zeta, pa, rxy, inv_h

# ---- index (k_ref_stats range from the raw points, k_cone_keys bins from the centred ones)
raw = ref[:, :3].astype(f32)
zraw = (raw[:, 2] / np.sqrt((raw * raw).sum(1))).astype(f32)
zr = max(f32(zraw.max() - zraw.min()), f32(1e-3))
z0 = f32(zraw.min() - f32(1e-5) - f32(1e-4) * zr)
rs = f32(E / (zr * f32(1.0002) + f32(2e-5)))
cs = f32(A * 0.25)
_, zeta_r, pa_r, _, _ = cone_dir((refc - O).astype(f32))
row_r = np.clip(np.floor(((zeta_r - z0) * rs).astype(f32)), 0, E - 1).astype(np.int64)
col_r = np.clip(np.floor((pa_r * cs).astype(f32)), 0, A - 1).astype(np.int64)
rzmin = np.full(E, np.inf, f32); rzmax = np.full(E, -np.inf, f32)
np.minimum.at(rzmin, row_r, zeta_r); np.maximum.at(rzmax, row_r, zeta_r)
print("rows occupied", (rzmin <= rzmax).sum(), "of", E, "z0", z0, "rs", rs)

rng = np.random.default_rng(1)
GAP = 0.002
missed = 0; checked = 0; fbl = 0
for k, T in enumerate([np.eye(4)] + Ts[:-1]):
    if k < 3: continue
    q = (rdc.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(f32)
    lim = limits[k - 1]; cap2s = f32(1.1 * lim * 1.05 * 1.05)
    smp = rng.choice(q.shape[0], NS, replace=False)
    qs = q[smp]
    dd, _ = tree.query(qs.astype(np.float64), k=1, workers=8)
    # radii as the kernel sees them: from tiny (keep-like) to the cap
    ub = (dd * rng.uniform(1.0, 3.0, NS)).astype(f32) ** 2
    lim0 = np.minimum((ub + f32(2 * GAP) * np.sqrt(ub) + f32(GAP * GAP)).astype(f32), cap2s)
    R = (np.sqrt(lim0) * f32(1 + 1e-5) + f32(1e-7)).astype(f32)
    inv_rho, zeta, pa, rxy, inv_h = cone_dir((qs - O).astype(f32))
    s = (R * inv_rho * f32(1 + 1e-5) + f32(4e-6)).astype(f32)
    ce = (rxy * inv_rho).astype(f32)
    cone = (s <= 0.5) & (ce > 0) & (ce <= 1.5)
    alpha = (s * (f32(1) + f32(0.2) * s * s) + f32(2e-6)).astype(f32)
    dz = (ce * s + np.abs(zeta) * s * s + f32(2e-6)).astype(f32)
    gq = ((rxy * inv_h) * (rxy * inv_h) * f32(1 + 1e-6)).astype(f32)
    r_lo = np.clip(np.floor(((zeta - dz - z0) * rs).astype(f32)), 0, E - 1).astype(np.int64)
    r_hi = np.clip(np.floor(((zeta + dz - z0) * rs).astype(f32)), 0, E - 1).astype(np.int64)
    for n_ in range(NS):
        if not cone[n_]: fbl += 1; continue
        inb = np.array(tree.query_ball_point(qs[n_].astype(np.float64), float(np.sqrt(lim0[n_]))), dtype=np.int64)
        if inb.size == 0: continue
        wins = []; bad = False
        for r in range(r_lo[n_], r_hi[n_] + 1):
            if not (rzmin[r] <= rzmax[r]): continue
            dzr = f32(max(max(max(rzmin[r] - zeta[n_], zeta[n_] - rzmax[r]), f32(0)) - f32(1e-6), f32(0)))
            if not (dzr <= alpha[n_]): continue
            zm = max(abs(rzmin[r]), abs(rzmax[r]))
            cep = f32(np.sqrt(max(f32(1) - zm * zm, f32(0))) * f32(1 - 1e-5))
            u2 = f32((alpha[n_] * alpha[n_] - dzr * dzr) / (f32(4) * ce[n_] * cep))
            if not (u2 <= 0.25): bad = True; break
            u = f32(np.sqrt(max(u2, f32(0))))
            da = f32(f32(2) * u * (f32(1) + f32(0.2) * u * u))
            dp = f32(gq[n_] * da + f32(1.42) * da * da + f32(4e-6))
            clo = int(np.floor(f32((pa[n_] - dp) * cs))); chi = int(np.floor(f32((pa[n_] + dp) * cs)))
            wins.append((r, clo, chi))
        if bad: fbl += 1; continue
        for pnt in inb:
            hit = False
            for r, clo, chi in wins:
                if row_r[pnt] != r: continue
                c_ = col_r[pnt]
                if clo <= c_ <= chi or clo <= c_ - A <= chi or clo <= c_ + A <= chi: hit = True; break
            checked += 1
            if not hit:
                missed += 1
                print("MISSED it", k, "query", smp[n_], "point", pnt, "row", row_r[pnt], "col", col_r[pnt], wins, flush=True)
    print(f"it {k}: checked {checked} in-ball points, missed {missed}, lanes without the index {fbl}", flush=True)
print("RESULT missed", missed, "of", checked)
