"""dev helper (GPU box): per-iteration kNN times of ONE scan-to-sub-map registration of bench.py's value_track drive.
   python devtools/track_iter.py <scan index> [n_az]      (environment switches apply: run once per variant)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib

i = int(sys.argv[1]); n_az = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
E = synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))
pose = lambda k: synth.se3(0.8 * k, 0.05 * k, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * k))
scans = {k: synth.scan_job((1234, pose(k), n_az, 10 + k)) for k in range(i - 3, i + 1)}
M = {k: pose(k) @ E for k in scans}
a = i - 1
parts = [scans[a]]
for k in (i - 2, i - 3):
    T = (np.linalg.inv(M[a]) @ M[k]).astype(np.float32)
    parts.append((scans[k] @ T.T).astype(np.float32))
sub = np.ascontiguousarray(np.concatenate(parts, 0)); sub[:, 3] = 1.0
Ti = np.linalg.inv(M[a]) @ M[i]
if os.environ.get("GUESS"):   # a guess off by GUESS metres / 5*GUESS degrees (a long alignment on the same clouds)
    g = float(os.environ["GUESS"])
    Ti = Ti @ synth.se3(0.6 * g, -0.7 * g, 0.39 * g, yaw=np.deg2rad(5.0 * g))
if os.environ.get("GUESS_E"):   # bench.py's compute_variants: the true relative pose off by 10 cm / 0.5 deg
    Ti = (np.linalg.inv(pose(a)) @ pose(i)) @ E
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.profile_kernels = 1
if os.environ.get('TIGHT'): cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
PR = (1.0, 1.0) if os.environ.get('CHAIN') == 'F' else (0.5, 0.5)
_unused = 0; cfg.cell_size = float(os.environ.get('CELL', '0'))
h = icp.IcpHandle(cfg)
dsub, drd = torch.from_numpy(sub).cuda(), torch.from_numpy(scans[i]).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); T, st = h.compute(drd, dsub, Ti, PR[0], 10, PR[1], 7); t1 = time.perf_counter()
print("scan", i, "compute ms %.3f filters %.3f align %.3f iters %d cone launches %d occupancy %.2f heavy %.4f" % ((t1 - t) * 1e3, st.t_reserved[0], st.t_total_ms, st.iterations,
      st.direction_index_launches, st.direction_index_occupancy, st.direction_index_heavy_share))
info = h.info(); print('chunks', info.n_chunks, 'cells', list(info.cells)[:12], 'h0', info.cell_size)
print('knn us:', ' '.join('%.0f+%.0f/%d' % (tr['knn_main_us'], tr['knn_fallback_us'], tr['searching']) for tr in h.trace()))
for j, tr in enumerate(h.trace()[:0]):
    print(j, "limit %.5f used %d searching %d knn_main %.1f us fb %.1f us strag %d" % (tr["limit"], tr["n_used"], tr["searching"], tr["knn_main_us"], tr["knn_fallback_us"], tr["stragglers"]))
