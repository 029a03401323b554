"""Worker of test_direction_index_returns_after_a_price_off: one alignment of a 64 x n_az-ray pair under the LSGPU_* switches of
the environment; prints what the alignment says about the direction index."""
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
        h.set_reference(np.ascontiguousarray(rf), np.ascontiguousarray(rn))
        T, st = h.align(rd, T_init)
        tr = h.trace()
    dg = hashlib.sha256()
    dg.update(np.ascontiguousarray(T).tobytes())
    for t in tr:
        dg.update(np.float32(t["limit"]).tobytes()); dg.update(np.int64(t["n_used"]).tobytes()); dg.update(np.ascontiguousarray(t["A"]).tobytes())
    print("PRICE_RESULT " + json.dumps({"digest": dg.hexdigest(), "iterations": int(st.iterations),
                                        "index_launches": int(st.direction_index_launches),
                                        "heavy_share": float(st.direction_index_heavy_share),
                                        "occupancy": float(st.direction_index_occupancy)}))


if __name__ == "__main__":
    main()
