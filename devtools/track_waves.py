"""dev helper (GPU box): per-tile cycle distribution of the voxel search (k_knn_tile, last iteration) for one scan-to-sub-map
registration of the track drive (needs devtools/liblsgpu_stats.so; run with LSGPU_NO_CONE=1).
   python devtools/track_waves.py <scan index> [iterations]"""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
i = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_az = 16384
E = synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))
pose = lambda k: synth.se3(0.8 * k, 0.05 * k, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * k))
scans = {k: synth.scan_job((1234, pose(k), n_az, 10 + k)) for k in range(i - 3, i + 1)}
M = {k: pose(k) @ E for k in scans}
a = i - 1
parts = [scans[a]]
for k in (i - 2, i - 3):
    T = (np.linalg.inv(M[a]) @ M[k]).astype(np.float32)
    parts.append((scans[k] @ T.T).astype(np.float32))
sub = np.ascontiguousarray(np.concatenate(parts, 0)); sub[:, 3] = 1.0
Ti = np.linalg.inv(M[a]) @ M[i]
rf, rn = icp.sampling_surface_normal(sub, 10, 0.5, 7)
keep = icp.random_sampling(scans[i].shape[0], 0.5, -1)
rd = np.ascontiguousarray(scans[i][keep])
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.max_iterations = iters; cfg.cell_size = float(os.environ.get('CELL', '0'))
h = icp.IcpHandle(cfg)
nw = (rd.shape[0] + 255) // 256 * 4
lib().lsgpu_dev_knn_wave_stats(h._h, None, nw)
h.set_reference(rf, rn)
T, st = h.align(rd, Ti)
buf = np.zeros((nw, 4), np.uint32)
lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), nw)
cyc, ev, sv, gl = buf[:, 0].astype(np.float64), buf[:, 1], buf[:, 2], buf[:, 3]
grp, lvl, nact = (gl >> 8) & 255, gl & 255, gl >> 16
print("scan", i, "reference", rf.shape[0], "reading", rd.shape[0], "iterations", st.iterations, "tiles", (cyc > 0).sum())
for name, v in (("cycles", cyc), ("chunk evals", ev), ("proxy survivors", sv), ("groups", grp), ("level", lvl), ("active lanes", nact)):
    print("%-16s mean %.1f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % ((name, v.mean()) + tuple(np.percentile(v, [50, 90, 99, 99.9, 100]))))
tot = cyc.sum()
order = np.argsort(-cyc)
for frac in (0.01, 0.05, 0.1, 0.25):
    k = int(len(cyc) * frac)
    print("slowest %4.0f %% of the tiles: %.0f %% of the cycles; mean evals %.0f, groups==64 share %.2f, mean level %.1f" % (100 * frac, 100 * cyc[order[:k]].sum() / tot, ev[order[:k]].mean(), (grp[order[:k]] == 64).mean(), lvl[order[:k]].mean()))
print("sum cycles / 1024 SIMDs = %.0f cycles (%.0f us at 2.4 GHz); max single wave %.0f" % (tot / 1024, tot / 1024 / 2400, cyc.max()))
