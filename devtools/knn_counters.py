"""dev helper: kNN work counters per iteration (needs devtools/liblsgpu_stats.so = -DLSGPU_KNN_STATS build)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref, rd, Tt, Ti = synth.scan_pair(n_az)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
for iters in (1, 2, 8):
    cfg.max_iterations = iters
    h = icp.IcpHandle(cfg)
    out = (C.c_ulonglong * 8)()
    lib().lsgpu_dev_knn_counters(h._h, out)
    h.set_reference(rf, rn)
    T, st = h.align(rd, Ti)
    lib().lsgpu_dev_knn_counters(h._h, out)
    v = list(out)
    print("iters=%d cumulative: groups %d (per wave %.2f) cells/group %.1f chunks_in_cells/group %.1f proxy_survivors/group %.1f evaluated_chunks/group %.1f cand/evaluated_chunk %.1f lanes_needing/eval %.1f avg_level %.2f" % (
        iters, v[0], v[0]/ (iters*rd.shape[0]/64), v[1]/v[0], v[2]/v[0], v[3]/v[0], v[4]/v[0], v[5]/max(v[4],1), v[6]/max(v[4],1), v[7]/v[0]))
    h.close()
