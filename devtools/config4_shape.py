"""BASELINE configs[3] shape on one GPU -- 8 aggregated scans (8.4 M-point sub-map) vs one 1 M-point scan.
usage: config4_shape.py [n_az=16384] [oracle|-] [out.json]   (writes the JSON artifact kept under profiles/)"""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
scene = synth.Scene(1234)
poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(9)]
t = time.time()
parts = []
T_ref = poses[7]
for i in range(8):
    s = synth.hdl64_scan(scene, poses[i], n_az, 20 + i)
    Trel = np.linalg.inv(T_ref) @ poses[i]
    p = s.copy(); p[:, :3] = (s[:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
    parts.append(p)
ref = np.concatenate(parts)
rd = synth.hdl64_scan(scene, poses[8], n_az, 40)
T_true = np.linalg.inv(T_ref) @ poses[8]
T_init = synth.se3(0.25, -0.1, 0.05, yaw=np.deg2rad(1.2)) @ T_true
print("gen", time.time() - t, ref.shape, rd.shape)
hf = icp.IcpHandle()
t = time.time(); d_rf, d_rn = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0); rf, rn = d_rf.cpu().numpy(), d_rn.cpu().numpy(); hf.close(); print("normals (device filter)", time.time() - t)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4; cfg.profile_kernels = 1; cfg.cell_size = float(os.environ.get('CELL', '0'))
h = icp.IcpHandle(cfg)
dref, dn, drd = torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda()
for rep in range(3):
    t = time.perf_counter(); h.set_reference(dref, dn); torch.cuda.synchronize(); t1 = time.perf_counter()
    T, st = h.align(drd, T_init); t2 = time.perf_counter()
print("set_reference ms %.2f align ms %.2f iters %d knn avg us %.1f cap_retries %d" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, st.iterations, st.t_knn_ms / max(st.knn_launches, 1) * 1e3, st.cap_retries))
if os.environ.get("PER_ITER"):
    tr = h.trace()
    print("per-iteration kNN us:", [round(float(t["knn_main_us"] + t["knn_fallback_us"]), 1) for t in tr])
    print("per-iteration searching queries (k):", [int(t["searching"]) // 1000 for t in tr])
print("err vs truth", synth.pose_error(T.astype(np.float64), T_true), "info chunks", h.info().n_chunks, list(h.info().cells)[:6])
if len(sys.argv) > 2 and sys.argv[2] == "oracle":
    from oracle import oracle_py as O
    ocfg = O.config_yaml(accum_double=1, min_diff_rot=1e-5, min_diff_trans=1e-4, num_threads=32)
    t = time.time()
    rc, To, sto, tro = O.icp_compute(ocfg, rd, rf, rn, synth.colmajor(T_init), 40)
    print("oracle rc", rc, "iters", sto.iterations, "time", time.time() - t, "err vs truth", synth.pose_error(synth.from_colmajor(To), T_true))
    print("GPU vs oracle", synth.pose_error(synth.from_colmajor(To), T.astype(np.float64)))
    trg = h.trace()
    print("limits equal:", [np.float32(a["limit"]) == np.float32(b["limit"]) for a, b in zip(trg, tro)].count(True), "of", len(tro),
          "| n_used equal:", [a["n_used"] == b["n_used"] for a, b in zip(trg, tro)].count(True))

if len(sys.argv) > 3:
    import json
    et, er = synth.pose_error(T.astype(np.float64), T_true)
    json.dump({"workload": "configs[3] on ONE GPU: %d-point aggregated local map (8 scans) vs %d-point scan, chain F, differential checker 1e-4 m / 1e-5 rad" % (rf.shape[0], rd.shape[0]),
               "set_reference_ms": (t1 - t) * 1e3, "align_ms": (t2 - t1) * 1e3, "iterations": int(st.iterations),
               "knn_avg_us_per_launch": st.t_knn_ms / max(st.knn_launches, 1) * 1e3, "select_avg_us": st.t_select_ms / max(st.knn_launches, 1) * 1e3,
               "ne_avg_us": st.t_ne_ms / max(st.knn_launches, 1) * 1e3, "cap_retries": int(st.cap_retries), "chunks": int(h.info().n_chunks),
               "trans_err_vs_truth_m": et, "rot_err_vs_truth_rad": er, "command": "python devtools/config4_shape.py %d - %s" % (n_az, sys.argv[3])},
              open(sys.argv[3], "w"), indent=1)
