"""dev helper (stats build): where workgroup 0 of k_ssn_tree spends its time (shader-clock stamps, lsgpu_ssn_tree.hip.h)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSGPU_SO"] = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cache = "/tmp/lsgpu_filter_time_scans.npy"
ref = list(np.load(cache, allow_pickle=True))[0] if os.path.exists(cache) and n_az == 16384 else synth.scan_pair(n_az)[0]
h = icp.IcpHandle()
d = torch.from_numpy(ref).cuda()
L = lib()
L.lsgpu_dev_tree_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 64)()
for rep in range(3):
    h.filter_reference(d, 10, 1.0, 0)
L.lsgpu_dev_tree_phases(h._h, buf)
t = np.array(list(buf), np.float64)
us = lambda a, b: (t[b] - t[a]) / 100.0   # clock64 = 100 MHz constant clock
print("workgroup 0 of k_ssn_tree, %d points in the cloud; us:" % ref.shape[0])
print("  load + key range %.1f" % us(0, 1))
for dd in range(3):
    print("  axis %d: %d radix passes %.1f, dense ranks %.1f" % (dd, int(t[40 + dd]), us(1 + 2 * dd, 2 + 2 * dd), us(2 + 2 * dd, 3 + 2 * dd)))
lv = [l for l in range(22) if t[8 + l] > 0]
print("  tree init %.1f" % us(7, 8))
for a, b in zip(lv[:-1], lv[1:]):
    print("  level %d %.1f" % (a, us(8 + a, 8 + b)))
print("  level %d %.1f" % (lv[-1], us(8 + lv[-1], 30)))
print("  output %.1f   total %.1f" % (us(30, 31), us(0, 31)))
print("first level's k_gs_select (%d candidates); us: bins %.1f, candidates to LDS %.1f, median + left counts %.1f, prefix %.1f, children %.1f" % (int(t[54]), us(48, 49), us(49, 50), us(50, 51), us(51, 52), us(52, 53)))
