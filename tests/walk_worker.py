"""Worker of test_wall_scan_search_walk_is_bounded: one scan-to-sub-map registration of the track drive (bench.py's value_track:
scans 0.8 m / 2 deg apart, yaml chain) through the -DLSGPU_KNN_STATS build of the library (tests/liblsgpu_icp_stats.so, which
records per 64-query tile of the LAST voxel-grid launch how many chunk boxes survived the tile-level cull and how many chunks
were fetched and evaluated); prints the distribution.   walk_worker.py <scan index> [iterations] [n_az]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from laser_slam_amd import _lib
    _lib.SO_PATH = os.path.join(ROOT, "tests", "liblsgpu_icp_stats.so")
    from laser_slam_amd import synth, icp
    from laser_slam_amd._lib import IcpConfig, lib
    i = int(sys.argv[1])
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_az = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
    E = synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))      # odometry: truth off by 10 cm / 0.5 deg (bench.py value_track)
    pose = lambda k: synth.se3(0.8 * k, 0.05 * k, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * k))
    scans = {k: synth.scan_job((1234, pose(k), n_az, 10 + k)) for k in range(i - 3, i + 1)}
    M = {k: pose(k) @ E for k in scans}
    a = i - 1
    parts = [scans[a]]
    for k in (i - 2, i - 3):
        T = (np.linalg.inv(M[a]) @ M[k]).astype(np.float32)
        parts.append((scans[k] @ T.T).astype(np.float32))
    sub = np.ascontiguousarray(np.concatenate(parts, 0))
    sub[:, 3] = 1.0
    Ti = np.linalg.inv(M[a]) @ M[i]
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.max_iterations = iters
    with icp.IcpHandle(cfg) as h:
        rf, rn = h.filter_reference(sub, 10, 0.5, 7)
        rf, rn = np.ascontiguousarray(rf), np.ascontiguousarray(rn)
        keep = icp.random_sampling(scans[i].shape[0], 0.5, -1)
        rd = np.ascontiguousarray(scans[i][keep])
        nw = (rd.shape[0] + 255) // 256 * 4
        assert lib().lsgpu_dev_knn_wave_stats(h._h, None, nw) == 0
        h.set_reference(rf, rn)
        T, st = h.align(rd, Ti)
        buf = np.zeros((nw, 4), np.uint32)
        assert lib().lsgpu_dev_knn_wave_stats(h._h, buf.ctypes.data_as(C.POINTER(C.c_uint)), nw) == 0
        tr = h.trace()
    rec = buf[:, 0] > 0
    ev, sv = buf[rec, 1].astype(np.int64), buf[rec, 2].astype(np.int64)
    nearest_wall = float(np.min(np.linalg.norm(scans[i][:, :2], axis=1)))
    print("WALK_RESULT " + json.dumps({
        "tiles": int(rec.sum()), "of": int((rd.shape[0] + 63) // 64), "iterations": int(st.iterations),
        "n_reference": int(rf.shape[0]), "n_reading": int(rd.shape[0]),
        "evals_mean": float(ev.mean()), "evals_p99": float(np.percentile(ev, 99)), "evals_max": int(ev.max()),
        "survivors_mean": float(sv.mean()), "survivors_p99": float(np.percentile(sv, 99)), "survivors_max": int(sv.max()),
        "handed_over": [int(t["stragglers"]) for t in tr], "nearest_return_m": nearest_wall}))


if __name__ == "__main__":
    main()
