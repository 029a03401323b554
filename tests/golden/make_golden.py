#!/usr/bin/env python3
"""Generates tests/golden/icp_pair4k.npz: inputs + expected outputs of the configured ICP chain on a
~4k-point synthetic HDL-64E pair, produced by the CPU oracle (oracle/icp_oracle.c).

The reference ships no golden vectors (laser_slam/test/test_empty.cpp:3-5 is its only test) and its
arithmetic lives in libpointmatcher, which cannot be built here, so these vectors pin the ORACLE
(against drift) and the HIP path (against the oracle); "parity unpinned" w.r.t. upstream remains
(DESIGN.md §oracle).  Re-run only when the oracle's definition changes:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from laser_slam_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def main():
    ref, rd, T_true, T_init = synth.scan_pair(64)
    rf, rn = O.sampling_surface_normal(ref, 10, 1.0, 11)
    Ti = synth.colmajor(T_init)
    # kernel-level vectors in the reference-mean frame at the initial guess
    mean = (rf[:, :3].astype(np.float64).sum(0) / rf.shape[0]).astype(np.float32)
    ref_c = rf.copy()
    ref_c[:, :3] -= mean
    Tm = Ti.copy()
    Tm[12:15] -= mean
    q = O.transform_points(Tm, rd)
    ids, d2 = O.KdTree(ref_c).nn(q)
    rc, limit = O.trim_limit(d2, 0.75)
    rc, A, b, x, dT, used = O.point_to_plane(q, ref_c, rn, ids, d2, limit, 1)
    out = dict(ref=rf, nrm=rn, rd=rd, T_init=Ti, T_true=T_true, mean=mean, nn_ids=ids, nn_d2=d2,
               limit0=np.float32(limit), A0=A, b0=b, x0=x, used0=np.int64(used))
    for tag, kw in (("yaml", {}), ("tight", dict(min_diff_rot=1e-5, min_diff_trans=1e-4))):
        for acc in (0, 1):
            cfg = O.config_yaml(accum_double=acc, **kw)
            rc, T, st, tr = O.icp_compute(cfg, rd, rf, rn, Ti, 40)
            assert rc == 0
            k = f"{tag}_acc{acc}"
            out[k + "_T"] = T
            out[k + "_iters"] = np.int32(st.iterations)
            out[k + "_converged"] = np.int32(st.converged)
            out[k + "_limits"] = np.array([t["limit"] for t in tr], np.float32)
            out[k + "_used"] = np.array([t["n_used"] for t in tr], np.int64)
            out[k + "_Titer"] = np.stack([t["T_iter"] for t in tr])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_pair4k.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: out[k] for k in out if k.endswith("_iters")})


if __name__ == "__main__":
    main()
