"""dev helper (stats build): where a k_knn_tile wave spends its cycles in the LAST launch of an align:
prologue (loads, reductions, cell lookup) | chunk loop (cull + need tests + fetch + evaluate) of which fetch + evaluate | epilogue."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSGPU_SO"] = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
os.environ["LSGPU_KNN_DBG"] = str(4096)
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ref, rd, Tt, Ti = synth.scan_pair(n_az)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
cfg.max_iterations = iters
h = icp.IcpHandle(cfg)
dref, dn = h.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
dref, dn = dref.clone(), dn.clone()
drd = torch.from_numpy(rd).cuda()
nw = (rd.shape[0] + 63) // 64
lib().lsgpu_dev_knn_wave_stats(h._h, None, 2 * nw)
h.set_reference(dref, dn); T, st = h.align(drd, Ti)
buf = np.zeros((2 * nw, 4), np.uint32)
lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), 2 * nw)
a, b = buf[:nw].astype(np.float64), buf[nw:].astype(np.float64)
tot, ev = a[:, 0], a[:, 1]
nact = (buf[:nw, 3] >> 16)
pro, loop, evc, epi = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
print("iterations", st.iterations, "waves", nw, "mean active lanes %.1f" % nact.mean())
for name, v in (("total", tot), ("prologue (loads, reductions, lookup)", pro), ("chunk loop", loop), ("  of which fetch + evaluate", evc), ("epilogue", epi)):
    print("%-40s mean %8.0f  p50 %8.0f  p90 %8.0f  share %.2f" % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.sum() / tot.sum()))
m = nact > 0
print("active tiles: %d; their mean total %.0f, evals %.1f; idle tiles mean total %.0f" % (m.sum(), tot[m].mean(), ev[m].mean(), tot[~m].mean() if (~m).any() else 0))
