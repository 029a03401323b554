"""dev helper (GPU box): BASELINE configs[4] at KITTI scale -- the 2000-pose figure-eight with 64 x 2048-ray scans (~125 k points
each), device ICP alone through LaserTrack::processPoseAndLaserScan + IncrementalEstimator::estimate / processLoopClosure,
scans resident in HBM -- with the per-pose stage breakdown the driver prints (round-5 verdict, item 6).
usage: python devtools/config4_kitti.py [n_poses] [n_az] [out.json]"""
import json, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_sequence as TS
from laser_slam_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_az = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "config4_kitti.json")
import pathlib
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
    exe = TS._build(pathlib.Path(d))
    lcs = {n // 4: [0], n // 2: [0], (3 * n) // 4: [n // 4], n - 1: [n // 2 - 1]}
    stream = os.path.join(d, "stream.bin")
    t0 = time.perf_counter()
    truth, odom = TS.make_stream(stream, n, n_az, lcs, min(64, os.cpu_count() or 1))
    t_gen = time.perf_counter() - t0
    import subprocess
    with open(stream, "rb") as f:
        r = subprocess.run([exe, TS.YAML_TIGHT, "3", "2", "dev", "1", "16"], stdin=f, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    stages, poses, total_ms, pts = [], [], None, []
    for line in r.stdout.splitlines():
        t = line.split()
        if t[0] == "stage":
            stages.append({t[k]: float(t[k + 1]) for k in range(3, len(t) - 1, 2)})
        elif t[0] == "pose" and t[1] == "dev":
            poses.append(TS._T([float(v) for v in t[5:12]]))
        elif t[0] == "time" and t[1] == "dev":
            total_ms = float(t[2])
    med = lambda k, sel=slice(None): float(np.median([s[k] for s in stages[sel]]))
    steady = slice(50, None)
    rep = {"poses": n, "rays_per_scan": 64 * n_az, "points_per_scan_median": int(np.median([0] + [0])), "ms_total": total_ms,
           "poses_per_s": n / (total_ms * 1e-3), "scan_generation_s": t_gen,
           "per_pose_ms_median": {k: med(k, steady) for k in ("track", "copy", "upload", "icp", "graph", "update")},
           "per_pose_ms_p95": {k: float(np.percentile([s[k] for s in stages[steady]], 95)) for k in ("track", "icp", "graph", "update")},
           "active_variables_median": med("active", steady), "active_variables_max": max(s["active"] for s in stages),
           "graph_ms_max": max(s["graph"] for s in stages),
           "stages_are": "track = processPoseAndLaserScan (copy = the call's working copy of the scan + input filters; upload = H2D of sub-map scans "
                         "that are not resident; icp = icp_.compute incl. the new scan's upload); graph = IncrementalEstimator::estimate (pose-graph "
                         "steps over the variables that still move); update = LaserTrack::updateFromValues",
           "rmse_vs_truth_m": TS._rmse(poses, truth), "rmse_dead_reckoning_m": TS._rmse(odom, truth), "loop_closures": len(lcs)}
    del rep["points_per_scan_median"]
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(rep, open(out_path, "w"), indent=1)
    print(json.dumps(rep, indent=1))
