#!/usr/bin/env python3
"""Generates tests/golden/filters_4k.npz: a raw ~4k-point synthetic HDL-64E scan and what the CPU oracle
(oracle/icp_oracle.c) makes of it with the chain's two sampling filters (icp_default.yaml:1-7) and the two
local-map filters of the ROS worker.  Pins the oracle against drift and the host / device filters against the
oracle (bit for bit).  Re-run only when the oracle's definition changes:

    python tests/golden/make_golden_filters.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from laser_slam_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def main():
    scan = synth.scan_pair(64)[0]
    out = dict(scan=scan)
    out["ssn_xyz"], out["ssn_nrm"] = O.sampling_surface_normal(scan, 10, 0.5, 5)      # yaml:5-7, seeded
    out["keep_after_ssn"] = O.random_sampling(len(scan), 0.5, -1)                      # yaml:1-3, continuing the stream
    out["ssn_full_xyz"], out["ssn_full_nrm"] = O.sampling_surface_normal(scan, 7, 1.0, 0)
    out["keep_seed7"] = O.random_sampling(3000, 0.75, 7)
    out["voxel_0p5"] = O.voxel_grid(scan, [0.5, 0.5, 0.5], 1)
    out["voxel_1p0_min3"] = O.voxel_grid(scan, [1.0, 1.0, 1.0], 3)
    out["cyl_in"] = O.cylinder_filter(scan, [0.5, -0.5, 0.0], 10.0, 40.0, False)
    out["cyl_out"] = O.cylinder_filter(scan, [0.5, -0.5, 0.0], 10.0, 40.0, True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "filters_4k.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
