"""Worker of test_experiment_switches_do_not_change_results: one long alignment of a 64x4096-ray pair with whatever
LSGPU_* switches the environment carries; prints a digest of everything the alignment returns."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from laser_slam_amd import icp, synth
    from laser_slam_amd._lib import IcpConfig, lib
    n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ref, rd, T_true, T_init = synth.scan_pair(n_az)
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
    with icp.IcpHandle(cfg) as h:
        rf, rn = h.filter_reference(ref, 10, 1.0, 0)
        rf, rn = np.ascontiguousarray(rf), np.ascontiguousarray(rn)
        h.set_reference(rf, rn)
        T, st = h.align(rd, T_init)
        tr = h.trace()
        ids, d2 = h.knn(rd, synth.colmajor(T_init))
        # the whole ICP::compute on the raw clouds (both filters with one draw stream, grid, loop)
        Tc, stc = h.compute(rd, ref, T_init, 0.6, 10, 0.7, seed=5)
        trc = h.trace()
    dg = hashlib.sha256()
    dg.update(np.ascontiguousarray(T).tobytes())
    for t in tr:
        dg.update(np.float32(t["limit"]).tobytes()); dg.update(np.int64(t["n_used"]).tobytes())
        dg.update(np.ascontiguousarray(t["A"]).tobytes()); dg.update(np.ascontiguousarray(t["T_iter"]).tobytes())
    dg.update(np.ascontiguousarray(d2).tobytes())
    dg.update(np.ascontiguousarray(rf).tobytes()); dg.update(np.ascontiguousarray(rn).tobytes())
    dg.update(np.ascontiguousarray(Tc).tobytes()); dg.update(np.int64(stc.iterations).tobytes())
    for t in trc:
        dg.update(np.float32(t["limit"]).tobytes()); dg.update(np.int64(t["n_used"]).tobytes())
        dg.update(np.ascontiguousarray(t["A"]).tobytes())
    # what does not depend on the order in which the normal equations are summed: the search from T_init and the first
    # iteration's order statistic and inlier count
    di = hashlib.sha256()
    di.update(np.ascontiguousarray(d2).tobytes()); di.update(np.ascontiguousarray(ids).tobytes())
    di.update(np.float32(tr[0]["limit"]).tobytes()); di.update(np.int64(tr[0]["n_used"]).tobytes())
    print("SWITCH_RESULT " + json.dumps({"digest": dg.hexdigest(), "digest_order_free": di.hexdigest(),
                                         "T": [float(v) for v in np.asarray(T, np.float64).ravel()],
                                         "iterations": int(st.iterations),
                                         "committed": int(st.committed_select_iterations), "sel_retries": int(st.pad_), "spread_tiles": int(st.spread_tiles),
                                         "n_ref": int(rf.shape[0])}))


if __name__ == "__main__":
    main()
