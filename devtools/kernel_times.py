"""dev helper: per-kernel average durations of one align (rocprofv3 must wrap this script)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import _lib as _l
if os.environ.get("LSGPU_SO"): _l.SO_PATH = os.environ["LSGPU_SO"]
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, dn = h.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
drd = torch.from_numpy(rd).cuda()
import time
for rep in range(3):
    h.set_reference(dref, dn); t = time.perf_counter(); T, st = h.align(drd, Ti); dt = time.perf_counter() - t
print("align ms %.3f" % (dt * 1e3))
