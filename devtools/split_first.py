"""dev helper (GPU box): the first searches of `bench.py --split`'s workload (configs[3]: 8-scan local map, configs[1]'s guess) on a
plain handle -- per-iteration main / hand-over times and hand-over counts.   python devtools/split_first.py [iterations=6]"""
import ctypes as C, sys, os, time
if os.environ.get("PHASES"):   # stats build: per-tile cycle records of the LAST launch (run with iterations=1 for the first search)
    os.environ["LSGPU_KNN_DBG"] = str(4096)
    os.environ["LSGPU_SO"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "liblsgpu_icp_stats.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_az = 16384
scene = synth.Scene(1234)
step_T = synth.se3(0.8, 0.05, 0.0, yaw=np.deg2rad(2.0), pitch=np.deg2rad(0.2))
poses = [synth.se3(0.0, 0.0, synth.SENSOR_HEIGHT)]
for _ in range(8):
    poses.append(poses[-1] @ step_T)
clouds = []
for i in range(8):
    s = synth.hdl64_scan(scene, poses[i], n_az, 100 + i)
    Trel = np.linalg.inv(poses[7]) @ poses[i]
    s[:, :3] = (s[:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
    clouds.append(s)
ref = np.concatenate(clouds)
rd = synth.hdl64_scan(scene, poses[8], n_az, 200)
T_init = synth.scan_pair(64)[3]
print("guess error vs truth:", synth.pose_error(T_init.astype(np.float64), step_T))
hf = icp.IcpHandle()
d_ref, d_nrm = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
d_ref, d_nrm = d_ref.contiguous().clone(), d_nrm.contiguous().clone(); hf.close()
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
cfg.max_iterations = iters; cfg.profile_kernels = 1
h = icp.IcpHandle(cfg)
d_rd = torch.from_numpy(rd).cuda()
nt = (rd.shape[0] + 63) // 64
if os.environ.get("PHASES"): lib().lsgpu_dev_knn_wave_stats(h._h, None, 2 * nt + 8)
for rep in range(2):
    h.set_reference(d_ref, d_nrm); torch.cuda.synchronize()
    t = time.perf_counter(); T, st = h.align(d_rd, T_init); t1 = time.perf_counter()
tr = h.trace()
print("align ms %.2f iterations %d" % ((t1 - t) * 1e3, st.iterations))
print("main us     :", [round(float(x["knn_main_us"]), 1) for x in tr])
print("hand-over us:", [round(float(x["knn_fallback_us"]), 1) for x in tr])
print("handed over :", [int(x["stragglers"]) for x in tr])
print("limit       :", [float(x["limit"]) for x in tr])

if os.environ.get("PHASES"):
    buf = np.zeros((2 * nt + 8, 4), np.uint32)
    lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), buf.shape[0])
    a0, b = buf[:nt].astype(np.float64), buf[nt:2 * nt].astype(np.float64)
    gl = buf[:nt, 3]; nact, grp = gl >> 16, (gl >> 8) & 255
    tot = a0[:, 0]
    print("tiles", nt, "recorded", int((tot > 0).sum()), "sum cycles / 1024 SIMDs = %.0f us at 2.4 GHz" % (tot.sum() / 1024 / 2400))
    for name, v in (("total", tot), ("prologue", b[:, 0]), ("chunk loop", b[:, 1]), ("  fetch + evaluate", b[:, 2]), ("epilogue", b[:, 3]), ("chunk evals", a0[:, 1]), ("survivors", a0[:, 2])):
        print("%-20s mean %9.0f p50 %9.0f p90 %9.0f p99 %9.0f max %9.0f" % ((name, v.mean()) + tuple(np.percentile(v, [50, 90, 99, 100]))))
    order = np.argsort(-tot)
    for frac in (0.01, 0.05, 0.2):
        o = order[:max(int(nt * frac), 1)]
        print("slowest %4.0f %%: %2.0f %% of the cycles; mean total %8.0f loop %8.0f fetch+eval %8.0f evals %5.0f survivors %5.0f lanes %4.1f groups %4.1f" % (
            100 * frac, 100 * tot[o].sum() / tot.sum(), tot[o].mean(), b[o, 1].mean(), b[o, 2].mean(), a0[o, 1].mean(), a0[o, 2].mean(), nact[o].mean(), grp[o].mean()))
