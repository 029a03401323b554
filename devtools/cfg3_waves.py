"""dev helper (GPU box): per-tile cycle distribution of the voxel search (k_knn_tile, LAST launch) for BASELINE configs[3] on one GPU --
8 aggregated scans (8.4 M-point local map) vs one 1 M-point scan (the workload of devtools/config4_shape.py).  Needs
devtools/liblsgpu_stats.so.      python devtools/cfg3_waves.py [iterations=12] [n_az=16384]"""
import ctypes as C, sys, os
if os.environ.get("CFG3_PHASES"): os.environ["LSGPU_KNN_DBG"] = str(4096)   # second record per tile: prologue | chunk loop | fetch + evaluate | epilogue
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_az = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
scene = synth.Scene(1234)
poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(9)]
parts = []
T_ref = poses[7]
for i in range(8):
    s = synth.hdl64_scan(scene, poses[i], n_az, 20 + i)
    Trel = np.linalg.inv(T_ref) @ poses[i]
    p = s.copy(); p[:, :3] = (s[:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
    parts.append(p)
ref = np.concatenate(parts)
rd = synth.hdl64_scan(scene, poses[8], n_az, 40)
T_true = np.linalg.inv(T_ref) @ poses[8]
T_init = synth.se3(0.25, -0.1, 0.05, yaw=np.deg2rad(1.2)) @ T_true
hf = icp.IcpHandle()
d_rf, d_rn = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0); rf, rn = d_rf.cpu().numpy(), d_rn.cpu().numpy(); hf.close()
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
cfg.max_iterations = iters; cfg.profile_kernels = 1
h = icp.IcpHandle(cfg)
nw = (rd.shape[0] + 255) // 256 * 4
nt = (rd.shape[0] + 63) // 64
phases = bool(os.environ.get("CFG3_PHASES"))
lib().lsgpu_dev_knn_wave_stats(h._h, None, 2 * nw if phases else nw)
h.set_reference(rf, rn)
T, st = h.align(rd, T_init)
buf = np.zeros((2 * nw if phases else nw, 4), np.uint32)
lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), buf.shape[0])
if phases:
    b = buf[nt:2 * nt].astype(np.float64); a0 = buf[:nt].astype(np.float64)
    na0 = buf[:nt, 3] >> 16
    for name, v in (("total", a0[:, 0]), ("prologue (loads, reductions, lookup)", b[:, 0]), ("chunk loop", b[:, 1]), ("  of which fetch + evaluate", b[:, 2]), ("epilogue", b[:, 3])):
        print("%-40s mean %8.0f  p50 %8.0f  p90 %8.0f  share %.2f" % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.sum() / a0[:, 0].sum()))
    for lo, hi in ((1, 8), (9, 16), (17, 32), (33, 64)):
        m = (na0 >= lo) & (na0 <= hi)
        if m.any(): print("tiles with %2d-%2d searching lanes: %5d  total %7.0f  loop %7.0f  fetch+eval %7.0f  evals %.1f" % (lo, hi, m.sum(), a0[m, 0].mean(), b[m, 1].mean(), b[m, 2].mean(), a0[m, 1].mean()))
    buf = buf[:nw]
cyc, ev, sv, gl = buf[:, 0].astype(np.float64), buf[:, 1], buf[:, 2], buf[:, 3]
grp, lvl, nact = (gl >> 8) & 255, gl & 255, gl >> 16
print("reference", rf.shape[0], "reading", rd.shape[0], "iterations", st.iterations, "knn avg us %.1f" % (st.t_knn_ms / max(st.knn_launches, 1) * 1e3),
      "tiles recorded", int((cyc > 0).sum()), "of", nw)
try:
    tr = h.trace()
    print("per-iteration kNN us (main + hand-over):", [round(float(t["knn_main_us"] + t["knn_fallback_us"]), 1) for t in tr])
    print("per-iteration searching queries:", [int(t["searching"]) for t in tr], "handed over:", [int(t["stragglers"]) for t in tr])
except Exception as e:
    print("(no per-iteration times: %s)" % e)
rec = cyc > 0
for name, v in (("cycles", cyc[rec]), ("chunk evals", ev[rec]), ("proxy survivors", sv[rec]), ("groups", grp[rec]), ("level", lvl[rec]), ("searching lanes", nact[rec])):
    print("%-16s mean %.1f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % ((name, v.mean()) + tuple(np.percentile(v, [50, 90, 99, 99.9, 100]))))
print("tiles with a searching lane: %.3f; searching lanes overall: %.4f of the queries" % ((nact[rec] > 0).mean(), nact[rec].sum() / rd.shape[0]))
tot = cyc.sum()
order = np.argsort(-cyc)
for frac in (0.001, 0.01, 0.05, 0.1, 0.25):
    k = max(int(len(cyc) * frac), 1)
    o = order[:k]
    print("slowest %5.1f %% of the tiles: %.0f %% of the cycles; mean cycles %.0f evals %.0f survivors %.0f searching lanes %.1f groups %.1f level %.1f" % (
        100 * frac, 100 * cyc[o].sum() / tot, cyc[o].mean(), ev[o].mean(), sv[o].mean(), nact[o].mean(), grp[o].mean(), lvl[o].mean()))
idle = rec & (nact == 0)
print("tiles without a searching lane: mean cycles %.0f (p50 %.0f p99 %.0f)" % (cyc[idle].mean(), *np.percentile(cyc[idle], [50, 99])))
print("sum cycles / 1024 SIMDs = %.0f cycles (%.0f us at 2.4 GHz); max single wave %.0f cycles" % (tot / 1024, tot / 1024 / 2400, cyc.max()))
