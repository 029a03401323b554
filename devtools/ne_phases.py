"""dev helper (stats build): where k_normal_eq_loop spends its time, averaged over the launches of one align."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSGPU_SO"] = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref, rd, Tt, Ti = synth.scan_pair(n_az)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, dn = h.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
dref, dn = dref.clone(), dn.clone()
drd = torch.from_numpy(rd).cuda()
L = lib()
L.lsgpu_dev_ne_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 24)()
h.set_reference(dref, dn); T, st = h.align(drd, Ti)
L.lsgpu_dev_ne_phases(h._h, buf); a0 = np.array(list(buf), np.float64)
h.set_reference(dref, dn); T, st = h.align(drd, Ti)
L.lsgpu_dev_ne_phases(h._h, buf); a1 = np.array(list(buf), np.float64)
d = a1 - a0
n, blocks = d[0], d[4]
print("launches %d, blocks per launch %.0f, iterations %d" % (n, blocks / n, st.iterations))
print("per block (shader cycles): prologue %.0f | main loop %.0f | wave+block reduce, hand-off %.0f" % (d[1] / blocks, d[2] / blocks, d[3] / blocks))
print("wall (us, 100 MHz clock): first start -> first loop start %.2f | first start -> last loop end %.2f | last loop end -> kernel end %.2f (update lane %.2f)"
      % (d[9] / n / 100, d[7] / n / 100, d[8] / n / 100, d[11] / n / 100))
print("tail (us): last loop end -> last block knows it is last %.2f | partials + state + set-aside loads, first sums %.2f | set-aside ranking + sums %.2f | final sums, publish %.2f" % tuple(d[16:20] / n / 100))
print("update lane (shader cycles per launch): unpack + LLT %.0f | delta, T update %.0f | trace record %.0f | checker %.0f" % tuple(d[12:16] / n))
print("distances set aside per iteration (stats build):", [t["searching"] for t in h.trace()])
