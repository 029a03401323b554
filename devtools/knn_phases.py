"""dev helper: per-phase wave-cycle totals of k_knn_tile (stats build)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from laser_slam_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
prev = None
for iters in (8, 9):
    cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
    cfg.max_iterations = iters
    h = icp.IcpHandle(cfg)
    out = (C.c_ulonglong * 8)()
    lib().lsgpu_dev_knn_counters(h._h, out)
    h.set_reference(rf, rn)
    T, st = h.align(rd, Ti)
    lib().lsgpu_dev_knn_counters(h._h, out)
    v = np.array(list(out), np.float64)
    if prev is not None:
        d = v - prev  # the 9th launch alone
        nw = rd.shape[0] / 64
        print("per wave (cycles, elapsed under contention): loads %.0f | reductions %.0f | level+lookup %.0f | rest %.0f ; evals %.1f survivors %.1f" % (d[1]/nw, d[2]/nw, d[5]/nw, d[6]/nw, d[4]/nw, d[3]/nw))
    prev = v
    h.close()
