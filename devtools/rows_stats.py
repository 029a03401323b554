"""dev helper: per-iteration kNN time + searching-query count of one align, and (stats build) the per-wave picture of
the LAST k_knn_rows launch.  usage: rows_stats.py [n_az] [max_iterations]   (LSGPU_KNN_ROWS etc. from the environment)"""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from laser_slam_amd import _lib
stats_so = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
use_stats = os.environ.get("LSGPU_STATS_SO", "1") != "0" and os.path.exists(stats_so)
if use_stats:
    _lib.SO_PATH = stats_so
import torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ref, rd, Tt, Ti = synth.scan_pair(n_az)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
cfg.max_iterations = iters
cfg.profile_kernels = 1
h = icp.IcpHandle(cfg)
dref, dn = h.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
dref, dn = dref.clone(), dn.clone()
drd = torch.from_numpy(rd).cuda()
nw = (rd.shape[0] + 63) // 64
if use_stats:
    lib().lsgpu_dev_knn_wave_stats(h._h, None, nw)
for rep in range(2):
    h.set_reference(dref, dn)
    T, st = h.align(drd, Ti)
tr = h.trace(64)
print("iterations", st.iterations, "align ms %.3f" % st.t_total_ms, "knn ms %.3f" % st.t_knn_ms, "select ms %.3f ne ms %.3f" % (st.t_select_ms, st.t_ne_ms),
      "committed-select iterations", st.committed_select_iterations, "select misses", st.pad_, "cap retries", st.cap_retries)
print(" it   knn_us  fb_us  searching      n_used   limit")
for i, t in enumerate(tr):
    print("%3d %8.1f %6.1f %10d %11d %9.3e" % (i, t["knn_main_us"], t["knn_fallback_us"], t["searching"], t["n_used"], t["limit"]))
if use_stats:
    buf = np.zeros((nw, 4), np.uint32)
    lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), nw)
    last = tr[-1]["searching"]
    nt = (last + 63) // 64 if last else nw
    b = buf[:nt]
    cyc = b[:, 0].astype(np.float64)
    rounds, needs = b[:, 1] >> 16, b[:, 1] & 0xFFFF
    groups, batches = b[:, 2] >> 8, b[:, 2] & 0xFF
    nls, nbmax = b[:, 3] >> 16, b[:, 3] & 0xFFFF
    print("last launch: %d waves" % nt)
    for name, a in (("cycles", cyc), ("rounds", rounds), ("needed chunks (4 rows)", needs), ("eval groups of 4", groups),
                    ("cull batches", batches), ("list length max row", nbmax), ("lanes searching alone", nls)):
        print("%-26s mean %9.1f p50 %7.0f p90 %7.0f p99 %7.0f max %7.0f" % ((name, a.mean()) + tuple(np.percentile(a, [50, 90, 99, 100]))))
    print("sum cycles / 1024 SIMDs = %.0f ; max wave %.0f" % (cyc.sum() / 1024, cyc.max()))
    if nt > 8:
        print("corr cycles~groups %.2f  cycles~rounds %.2f  cycles~batches %.2f" % (np.corrcoef(cyc, groups)[0, 1], np.corrcoef(cyc, rounds)[0, 1], np.corrcoef(cyc, batches)[0, 1]))
        A = np.stack([groups, rounds, batches, np.ones(nt)], 1).astype(np.float64)
        coef = np.linalg.lstsq(A, cyc, rcond=None)[0]
        print("fit cycles = %.0f * groups + %.0f * rounds + %.0f * batches + %.0f" % tuple(coef))
