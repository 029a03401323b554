"""dev helper: per-wave cycle distribution of k_knn_tile (needs devtools/liblsgpu_stats.so)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from laser_slam_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ref, rd, Tt, Ti = synth.scan_pair(n_az)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
cfg.max_iterations = iters
h = icp.IcpHandle(cfg)
nw = (rd.shape[0] + 255) // 256 * 4
lib().lsgpu_dev_knn_wave_stats(h._h, None, nw)
h.set_reference(rf, rn)
T, st = h.align(rd, Ti)
buf = np.zeros((nw, 4), np.uint32)
lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), nw)
cyc, ev, sv, gl = buf[:, 0].astype(np.float64), buf[:, 1], buf[:, 2], buf[:, 3]
grp, lvl, nact = (gl >> 8) & 255, gl & 255, gl >> 16
print('active lanes per tile: mean %.1f' % nact.mean(), 'hist(0,1-4,5-8,9-16,17-32,33-64):', [int(((nact >= a) & (nact <= b)).sum()) for a, b in ((0, 0), (1, 4), (5, 8), (9, 16), (17, 32), (33, 64))])
for a, b in ((0, 0), (1, 4), (5, 8), (9, 16), (17, 32), (33, 64)):
    m = (nact >= a) & (nact <= b)
    if m.any(): print('  tiles with', a, '-', b, 'active: mean cycles %.0f evals %.1f' % (cyc[m].mean(), ev[m].mean()))
print("waves", nw, "last-iteration per-wave stats")
for name, a in (("cycles", cyc), ("chunk evals", ev), ("proxy survivors", sv), ("groups", grp), ("level", lvl)):
    print("%-16s mean %.1f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % ((name, a.mean()) + tuple(np.percentile(a, [50, 90, 99, 99.9, 100]))))
print("sum cycles / (1024 SIMD) = %.0f cycles; max single wave = %.0f" % (cyc.sum() / 1024, cyc.max()))
top = np.argsort(-cyc)[:8]
for t in top: print("wave", t, "cycles", cyc[t], "evals", ev[t], "surv", sv[t], "groups", grp[t], "lvl", lvl[t])
print("corr cycles~evals", np.corrcoef(cyc, ev)[0, 1], "cycles per eval (fit)", np.polyfit(ev, cyc, 1))
# which tiles are the slow ones: dump query extents of the top waves
rdq = rd  # original order unknown here; only stats
