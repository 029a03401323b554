"""dev helper: the differential checker's smoothed pose change per iteration of the benchmark alignment (how predictable is the end?)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
h.set_reference(torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda())
T, st = h.align(torch.from_numpy(rd).cuda(), Ti)
Ts = [np.eye(4)] + [np.asarray(t["T_iter"], np.float64).reshape(4, 4).T for t in h.trace()]
dt = [np.linalg.norm(Ts[i][:3, 3] - Ts[i - 1][:3, 3]) for i in range(1, len(Ts))]
dr = [np.arccos(np.clip((np.trace(Ts[i][:3, :3] @ Ts[i - 1][:3, :3].T) - 1) / 2, -1, 1)) for i in range(1, len(Ts))]
sm = 4
for i in range(len(dt)):
    if i + 1 >= sm:
        print("it %2d  dt %.2e dr %.2e | mean4 dt %.2e (x lim %.2f) dr %.2e (x lim %.2f)" % (i, dt[i], dr[i], np.mean(dt[i - sm + 1:i + 1]), np.mean(dt[i - sm + 1:i + 1]) / 1e-4, np.mean(dr[i - sm + 1:i + 1]), np.mean(dr[i - sm + 1:i + 1]) / 1e-5))
