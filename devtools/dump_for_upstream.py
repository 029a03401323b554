#!/usr/bin/env python3
"""Everything somebody WITH libpointmatcher needs to diff the restatement this library was written against
(oracle/icp_oracle.c; "parity unpinned": SURVEY.md 8c) against upstream -- CPU only, no GPU, deterministic.

For a synthetic HDL-64E pair (default: the 4 k-point pair of the parity tests, `--n-az 1024` = the 64 k pair) it writes

  reading.vtk / .csv, reference.vtk / .csv   the RAW clouds `icp_.compute(reading, reference, T_init)` is handed
                                             (laser_slam/src/laser_track.cpp:496), as DataPoints::load reads them
  icp.yaml                                   the module chain (tests/golden/icp_chain*.yaml, loadable by
                                             PointMatcher::ICP::loadFromYaml; checked here with this repo's own loader)
  T_init.txt                                 4 x 4, row major, "%.9g"
  reference_filtered.csv                     what SamplingSurfaceNormalDataPointsFilter leaves (x,y,z,nx,ny,nz): choices 1, 2, 8, 10, 11
  reading_filtered.csv                       what RandomSamplingDataPointsFilter leaves: choice 8 (the draws continue the
                                             reference filter's; the process starts at srand(1) like an unseeded one)
  input_filtered.csv                         reading.csv through tests/golden/input_filters.yaml (srand(1) again): choice 9
  oracle_trace.csv                           per iteration: iter, limit (squared metres), n_used, T_iter (16, column major)
  oracle_result.txt                          rc, iterations, converged, final T (4 x 4 row major)
  README.txt                                 what is what, and which file / column each restatement choice would change

With --submap the dump has the REAL call shape of LaserTrack::localScanToSubMap (laser_slam/src/laser_track.cpp:466-519)
instead of a bare pair: four scans along a trajectory go through the input filter chain (input_filters_.apply, :146), the
sub-map is assembled from the three older ones in the frame of the newest of them (RigidTransformation::compute with the
float relative pose + concatenate, :474-486), the guess is T_a^-1 T_b of the odometry poses (:489-491):
  scan{0..3}.vtk / .csv                      the raw scans           scan{0..3}_input_filtered.csv   after the input chain (one
                                                                     draw stream over the four scans, srand(1) before the first)
  poses.txt                                  the four odometry poses T_w_scan (4 x 4 row major each)
  T_rel{1,2}.txt                             the float relative poses the two older scans are moved by
  reading.csv / reference.csv / T_init.txt   what icp_.compute is handed: scan 3 filtered, the assembled sub-map, the guess
  reference_filtered.csv, reading_filtered.csv, oracle_trace.csv, oracle_result.txt   as above (the ICP's own draw stream
                                             continues the input filters': one process, one rand())

The files regenerate byte for byte (tests/test_oracle.py::test_upstream_dump_regenerates).  INTEGRATION.md has the
C++ program that replays them on a real PointMatcher<float>::ICP and prints a trace in the same format.
usage: dump_for_upstream.py OUT_DIR [--n-az 64] [--chain tests/golden/icp_chain_tight.yaml] [--submap]
"""
import argparse
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import cloud_io, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CHOICES = """Restatement choices of oracle/icp_oracle.h and where a different upstream behaviour shows:
 1 box rank test (FullPivHouseholderQR, eps * 3)      reference_filtered.csv: number of rows (thin boxes kept / dropped)
 2 box split (stable sort, split at the median)      reference_filtered.csv: normals of boxes with tied coordinates; for ratio < 1 WHICH rows
 3 kd-tree ties (any nearest point)                  nothing: oracle_trace.csv is the same for every valid tie order
 4 TrimmedDist: index floor(n * ratio), d2 <= limit  oracle_trace.csv: columns limit and n_used
 5 reference mean in double, rounded to float        oracle_trace.csv: last digits of T_iter (1e-6 level)
 6 minimiser: float J, LLT solve in float            oracle_trace.csv: T_iter from the first iteration on
 7 differential checker: 2 atan2(|vec|, |w|)         oracle_trace.csv: number of rows (the iteration it stops at)
 8 rand(): glibc sequence, draw < prob keeps         reading_filtered.csv (and reference_filtered.csv for ratio < 1): WHICH rows
 9 MaxDist signed on one axis, MinDist absolute      input_filtered.csv: rows
10 kept points in ascending ORIGINAL index             reference_filtered.csv: row order (upstream sorts indicesToKeep before it compacts)
11 box eigenvectors: Jacobi in double on the float C   reference_filtered.csv: last digits of nx, ny, nz (upstream: EigenSolver in float)
"""


def g(v) -> str:
    return "%.9g" % float(v)


def parse_chain(path):
    """The chain file through this repo's own loader (laser_slam_amd.icp.ICP.load_from_yaml): the values the oracle runs."""
    from laser_slam_amd import icp
    m = icp.ICP()
    m.load_from_yaml(path)
    return m.chain


def input_filter_chain(path):
    """tests/golden/input_filters.yaml -> oracle PointFilter array (the five module types of the golden chain)."""
    import yaml
    doc = yaml.safe_load(open(path).read()) or []
    arr = (O.PointFilter * len(doc))()
    for a, item in zip(arr, doc):
        (name, p), = item.items() if isinstance(item, dict) else ((item, {}),)
        p = p or {}
        a.state = 0.0
        if name == "BoundingBoxDataPointsFilter":
            a.type, a.flag = 3, int(p.get("removeInside", 1))
            for i, k in enumerate(("xMin", "xMax", "yMin", "yMax", "zMin", "zMax")):
                a.v[i] = float(p.get(k, -1.0 if i % 2 == 0 else 1.0))
        elif name == "MaxDistDataPointsFilter":
            a.type, a.dim = 1, int(p.get("dim", -1)); a.v[0] = float(p.get("maxDist", 1.0))
        elif name == "MinDistDataPointsFilter":
            a.type, a.dim = 2, int(p.get("dim", -1)); a.v[0] = float(p.get("minDist", 1.0))
        elif name == "FixStepSamplingDataPointsFilter":
            a.type = 4; a.v[0] = float(p.get("startStep", 10)); a.v[1] = float(p.get("endStep", 10)); a.v[2] = float(p.get("stepMult", 1))
        elif name == "RandomSamplingDataPointsFilter":
            a.type = 5; a.v[0] = float(p.get("prob", 0.75))
        elif name == "RemoveNaNDataPointsFilter":
            a.type = 6
        else:
            raise SystemExit("input filter %s is not in the restatement" % name)
    return arr


def write_mat(path, T):
    with open(path, "w") as f:
        for r in range(4):
            f.write(" ".join(g(T[r, c]) for c in range(4)) + "\n")


def submap_inputs(out_dir, n_az):
    """laser_track.cpp:146, 466-496 on four synthetic scans: input filters, sub-map of three in the frame of scan 2, the
    odometry guess.  Returns (reading, reference, T_init 4x4 float64, seed for the ICP's first draw)."""
    scene = synth.Scene(1234)
    truth = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(4)]
    odom = [T @ synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5)) for T in truth]     # (the drive of bench.py's value_track)
    flt = input_filter_chain(os.path.join(ROOT, "tests", "golden", "input_filters.yaml"))
    filtered = []
    with open(os.path.join(out_dir, "poses.txt"), "w") as f:
        for i in range(4):
            raw = synth.hdl64_scan(scene, truth[i], n_az, 10 + i)
            cloud_io.save_vtk(os.path.join(out_dir, "scan%d.vtk" % i), raw)
            cloud_io.save_csv(os.path.join(out_dir, "scan%d.csv" % i), raw)
            kept = O.apply_point_filters(flt, raw, seed=1 if i == 0 else -1)       # one process, one rand() stream
            kept = kept if kept is not None else raw[:0]
            cloud_io.save_csv(os.path.join(out_dir, "scan%d_input_filtered.csv" % i), kept)
            filtered.append(kept)
            for r in range(4):
                f.write(" ".join(g(odom[i][r, c]) for c in range(4)) + "\n")
    a = 2                                                                           # the sub-map's frame: the newest of its scans
    parts = [filtered[a]]
    for n, k in enumerate((1, 0), start=1):
        Trel = (np.linalg.inv(odom[a]) @ odom[k]).astype(np.float32)               # TransformationParameters are float
        write_mat(os.path.join(out_dir, "T_rel%d.txt" % n), Trel)
        parts.append(O.transform_points(synth.colmajor(Trel), filtered[k]))        # RigidTransformation::compute
    return filtered[3], np.ascontiguousarray(np.concatenate(parts, 0)), np.linalg.inv(odom[a]) @ odom[3], -1


def dump(out_dir, n_az, chain_path, submap=False):
    os.makedirs(out_dir, exist_ok=True)
    ch = parse_chain(chain_path)
    shutil.copyfile(chain_path, os.path.join(out_dir, "icp.yaml"))
    if submap:
        rd, ref, T_init, icp_seed = submap_inputs(out_dir, n_az)
    else:
        ref, rd, T_true, T_init = synth.scan_pair(n_az)
        icp_seed = 1
        cloud_io.save_vtk(os.path.join(out_dir, "reading.vtk"), rd)
        cloud_io.save_vtk(os.path.join(out_dir, "reference.vtk"), ref)
    cloud_io.save_csv(os.path.join(out_dir, "reading.csv"), rd)
    cloud_io.save_csv(os.path.join(out_dir, "reference.csv"), ref)
    T16 = synth.colmajor(T_init)
    write_mat(os.path.join(out_dir, "T_init.txt"), T16.reshape(4, 4).T)
    # ICP::compute, steps 1 and 4 with ONE draw stream that starts where an unseeded process starts (srand(1)) -- or, in the
    # sub-map dump, where the input filters of the four scans left it
    rf, rn = O.sampling_surface_normal(ref, ch.surface_normal_knn, ch.surface_normal_ratio, icp_seed)
    if ch.reading_sampling_prob >= 0:
        rdf = rd[O.random_sampling(rd.shape[0], ch.reading_sampling_prob, -1)]
    else:
        rdf = rd
    cloud_io.save_csv(os.path.join(out_dir, "reference_filtered.csv"), rf, rn)
    cloud_io.save_csv(os.path.join(out_dir, "reading_filtered.csv"), rdf)
    if not submap:
        flt = input_filter_chain(os.path.join(ROOT, "tests", "golden", "input_filters.yaml"))
        kept = O.apply_point_filters(flt, rd, seed=1)
        cloud_io.save_csv(os.path.join(out_dir, "input_filtered.csv"), kept if kept is not None else rd[:0])
    cfg = O.config_yaml(accum_double=0, trim_ratio=ch.trim_ratio, max_iterations=ch.max_iterations,
                        min_diff_rot=ch.min_diff_rot, min_diff_trans=ch.min_diff_trans, smooth_length=ch.smooth_length)
    rc, T, st, tr = O.icp_compute(cfg, rdf, rf, rn, T16, trace_cap=ch.max_iterations)
    with open(os.path.join(out_dir, "oracle_trace.csv"), "w") as f:
        f.write("iter,limit,n_used," + ",".join("T%d%d" % (r, c) for c in range(4) for r in range(4)) + "\n")
        for k, t in enumerate(tr):
            f.write("%d,%s,%d,%s\n" % (k, g(t["limit"]), t["n_used"], ",".join(g(v) for v in t["T_iter"])))
    Tf = np.asarray(T, np.float32).reshape(4, 4).T
    with open(os.path.join(out_dir, "oracle_result.txt"), "w") as f:
        f.write("rc %d iterations %d converged %d\n" % (rc, st.iterations, st.converged))
        for r in range(4):
            f.write(" ".join(g(Tf[r, c]) for c in range(4)) + "\n")
    with open(os.path.join(out_dir, "README.txt"), "w") as f:
        f.write(("Synthetic HDL-64E drive, LaserTrack::localScanToSubMap's call shape (input filters, sub-map of three scans), "
                 if submap else "Synthetic HDL-64E pair, ") +
                "%d azimuth steps: reading %d points, reference %d points; chain %s\n"
                "(reading_sampling_prob %s, surface normal knn %d ratio %s, trim %s, checkers %d / %s / %s / %d).\n"
                "Written by devtools/dump_for_upstream.py; the float accumulation of libpointmatcher (accum_double 0).\n"
                "Replay on libpointmatcher: INTEGRATION.md, \"Diffing against a real libpointmatcher\".\n\n%s"
                % (n_az, rd.shape[0], ref.shape[0], os.path.relpath(chain_path, ROOT), g(ch.reading_sampling_prob),
                   ch.surface_normal_knn, g(ch.surface_normal_ratio), g(ch.trim_ratio), ch.max_iterations,
                   g(ch.min_diff_rot), g(ch.min_diff_trans), ch.smooth_length, CHOICES))
    return rc, st.iterations


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--n-az", type=int, default=64)
    ap.add_argument("--chain", default=os.path.join(ROOT, "tests", "golden", "icp_chain.yaml"))
    ap.add_argument("--submap", action="store_true", help="the call shape of LaserTrack::localScanToSubMap (four scans, input filters, three-scan sub-map)")
    a = ap.parse_args()
    rc, it = dump(a.out_dir, a.n_az, a.chain, a.submap)
    print("wrote %s: oracle rc %d, %d iterations" % (a.out_dir, rc, it))
