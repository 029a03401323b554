"""dev helper: kNN time per launch of an align against a local map of K scans (K = 1, 3, 8) -- is the direction index
(k_knn_cone) or the voxel grid (k_knn_tile) the faster search for the settled iterations?  Run under LSGPU_NO_CONE=1 and
without.   usage: cone_density.py [n_az=16384] [K ...]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Ks = [int(a) for a in sys.argv[2:]] or [1, 3, 8]
scene = synth.Scene(1234)
poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(9)]
scans = {}
def scan(i):
    if i not in scans: scans[i] = synth.hdl64_scan(scene, poses[i], n_az, 20 + i)
    return scans[i]
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4; cfg.profile_kernels = 1
for K in Ks:
    last = K - 1
    parts = []
    for i in range(K):
        Trel = np.linalg.inv(poses[last]) @ poses[i]
        p = scan(i).copy(); p[:, :3] = (scan(i)[:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
        parts.append(p)
    ref = np.concatenate(parts)
    rd = scan(K)
    T_init = synth.se3(0.25, -0.1, 0.05, yaw=np.deg2rad(1.2)) @ (np.linalg.inv(poses[last]) @ poses[K])
    with icp.IcpHandle() as hf:
        d_rf, d_rn = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
    d_rf, d_rn = d_rf.contiguous().clone(), d_rn.contiguous().clone()
    drd = torch.from_numpy(rd).cuda()
    with icp.IcpHandle(cfg) as h:
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter(); h.set_reference(d_rf, d_rn); torch.cuda.synchronize(); t1 = time.perf_counter()
            T, st = h.align(drd, T_init); t2 = time.perf_counter()
        tr = h.trace()
        us = [t["knn_main_us"] + t["knn_fallback_us"] for t in tr]
        print("K=%d ref %d pts (occupancy %.2f, %d index launches): set_reference %.2f ms align %.2f ms, %d iterations, stragglers %d; kNN us: first three %s, then mean %.1f (min %.1f max %.1f)"
              % (K, d_rf.shape[0], st.direction_index_occupancy, st.direction_index_launches, (t1 - t) * 1e3, (t2 - t1) * 1e3, st.iterations, st.stragglers, ["%.0f" % u for u in us[:3]], np.mean(us[3:]), np.min(us[3:]), np.max(us[3:])))
