"""dev helper: time of the device reference filter (SamplingSurfaceNormal) and of a whole compute's filters + grid, for the
1 M-point benchmark scan and a 3-scan sub-map."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
scene = synth.Scene(1234)
poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(4)]
cache = "/tmp/lsgpu_filter_time_scans.npy"   # (the ray casting takes longer than everything timed here)
if os.path.exists(cache):
    scans = list(np.load(cache, allow_pickle=True))
else:
    scans = [synth.hdl64_scan(scene, poses[i], 16384, 10 + i) for i in range(4)]
    arr = np.empty(4, object)
    for i in range(4):
        arr[i] = scans[i]
    np.save(cache, arr, allow_pickle=True)
parts = []
for k in (2, 1, 0):
    Trel = np.linalg.inv(poses[2]) @ poses[k]
    p = scans[k].copy(); p[:, :3] = (scans[k][:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
    parts.append(p)
sub = np.concatenate(parts)
h = icp.IcpHandle()
for name, cloud in (("scan 1M", scans[0]), ("sub-map 3M", sub)):
    d = torch.from_numpy(cloud).cuda()
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t = time.perf_counter(); a, b = h.filter_reference(d, 10, 1.0, 0); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("%-11s reference filter %.3f ms (min %.3f) -> %d points" % (name, np.median(ts[1:]), min(ts[1:]), a.shape[0]))
T_g = (np.linalg.inv(poses[2]) @ poses[3]) @ synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))
for name, ref, prob, ratio in (("F 1M", scans[2], 1.0, 1.0), ("F sub-map", sub, 1.0, 1.0), ("P sub-map", sub, 0.5, 0.5)):
    d_ref, d_rd = torch.from_numpy(ref).cuda(), torch.from_numpy(scans[3]).cuda()
    Tg = T_g if ref is sub else np.linalg.inv(poses[2]) @ poses[3] @ synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))
    ts, fg = [], []
    for rep in range(6):
        t = time.perf_counter(); T, st = h.compute(d_rd, d_ref, Tg, prob, 10, ratio, seed=0); ts.append((time.perf_counter() - t) * 1e3); fg.append(st.t_reserved[0])
    print("%-10s compute %.3f ms, filters + grid %.3f ms, %d iterations" % (name, np.median(ts[1:]), np.median(fg[1:]), st.iterations))
