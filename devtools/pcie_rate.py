"""dev helper: the benchmark step with HOST buffers handed to set_reference / align (PCIe-inclusive rate)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, dn = h.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
rf, rn = dref.cpu().numpy().copy(), dn.cpu().numpy().copy()
drd = torch.from_numpy(rd).cuda()
for name, a, b, c in (("device-resident", dref, dn, drd), ("host (pageable numpy)", rf, rn, rd)):
    ts = []
    for rep in range(6):
        t = time.perf_counter(); h.set_reference(a, b); T, st = h.align(c, Ti); ts.append(time.perf_counter() - t)
    print("%-22s %.2f ms per step  (%.1f scans/s)" % (name, np.median(ts[1:]) * 1e3, 1.0 / np.median(ts[1:])))
