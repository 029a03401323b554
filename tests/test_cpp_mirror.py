"""The C++ host-side mirror (laser_slam_amd/cpp): ICP (loadFromYaml / setDefault / compute) and the
LaserTrack facade (processPoseAndLaserScan -> localScanToSubMap), built with g++ against the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from laser_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, src, name):
    out = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-DLSGPU_TEST_SEAMS", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "laser_slam_amd", "cpp", "include"), os.path.join(ROOT, "tests", "cpp", src),
           "-o", out, "-L", os.path.join(ROOT, "laser_slam_amd"), "-llsgpu_icp",
           "-Wl,-rpath," + os.path.join(ROOT, "laser_slam_amd")]
    subprocess.check_call(cmd)
    return out


def test_cpp_host_checks(tmp_path):
    import torch
    exe = _build(tmp_path, "host_checks.cpp", "host_checks")
    args = [exe] + ([] if torch.cuda.is_available() else ["--expect-no-gpu"])
    r = subprocess.run(args, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, LSGPU_GOLDEN_DIR=os.path.join(ROOT, "tests", "golden"), LSGPU_TEST_DUMP_DIR=str(tmp_path)))
    assert r.returncode == 0 and "host_checks: ok" in r.stdout, r.stdout + r.stderr
    # save_icp_results (laser_track.cpp:504-513): the four .vtk files of the last ICP (scan 2 against scans 1 + 0 in the frame
    # of scan 1, the fake ICP returns its guess), byte for byte what the Python writer emits, and loadable again
    from laser_slam_amd import cloud_io
    scan = lambda i: np.array([[i, 0, 0, 1], [0.1 * i, 1.0 / 3.0, -2.5e-7, 1]], np.float32)
    assert open(tmp_path / "last_scan.vtk").read() == cloud_io.vtk_text(scan(2))
    back, nrm = cloud_io.load_vtk(str(tmp_path / "last_scan.vtk"))
    assert nrm is None and np.array_equal(back.view(np.uint32), scan(2).view(np.uint32))
    sub, _ = cloud_io.load_vtk(str(tmp_path / "sub_map.vtk"))
    moved0 = scan(0).copy()
    moved0[:, 0] -= np.float32(0.8)                       # scan 0 in the frame of scan 1 (poses 0.8 m apart along x)
    assert np.allclose(sub, np.concatenate([scan(1), moved0]), atol=1e-6)
    by_guess, _ = cloud_io.load_vtk(str(tmp_path / "last_scan_alligned_by_initial_guess.vtk"))
    by_solution, _ = cloud_io.load_vtk(str(tmp_path / "last_scan_alligned_by_solution.vtk"))
    assert np.array_equal(by_guess, by_solution) and np.allclose(by_guess[:, 0], scan(2)[:, 0] + 0.8, atol=1e-6)


def test_launch_policy_state_machine(tmp_path):
    """csrc/lsgpu_policy.h decides what lsgpu_icp_align enqueues next (no HIP in it): the sequence of an alignment, the
    hand-over to the direction index, pricing / re-pricing (an alignment priced off the index returns to it once a look
    finds it cheap -- the defect round 4 shipped), the repeat paths, the split-scan mode.  tests/cpp/policy_check.cpp
    drives the state machine against a scripted device; no GPU, no library."""
    exe = str(tmp_path / "policy_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "policy_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "policy_check ok" in r.stdout, r.stdout + r.stderr


def test_gtsam_overlay_parses_and_resolves_the_ros_worker_calls():
    """PARSE check of integration/gtsam/laser_slam_gtsam_overlay.hpp (the `namespace laser_slam` types laser_slam_ros
    compiles against): g++ -fsyntax-only against the declaration-only stand-ins of tests/cpp/mock/ (GTSAM, minkindr,
    libpointmatcher, Eigen are not installed here), together with tests/cpp/overlay_worker_calls.cpp, which makes every
    call of laser_slam_ros/src/laser_slam_worker.cpp:47-600 plus the rest of the public surface of laser_track.hpp:20-144 /
    incremental_estimator.hpp:20-53 under the include names laser_slam_ros uses.  It pins NO behaviour: what the overlay
    delegates to (LaserTrack, WorkerLinks, ICP of the mirror) is tested by test_cpp_host_checks and the GPU tests."""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "tests", "cpp", "mock"),
           "-I", os.path.join(ROOT, "integration", "gtsam"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "laser_slam_amd", "cpp", "include"),
           os.path.join(ROOT, "tests", "cpp", "overlay_worker_calls.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    # the reference's public members, by name, must all be declared by the overlay
    src = open(os.path.join(ROOT, "integration", "gtsam", "laser_slam_gtsam_overlay.hpp")).read()
    for name in ("processPose", "processLaserScan", "processPoseAndLaserScan", "getLastPointCloud", "getPointCloudOfTimeInterval",
                 "getLocalCloudInWorldFrame", "getLaserScans", "getTrajectory", "getOdometryTrajectory", "getCovariances",
                 "getCurrentPose", "getPreviousPose", "getMinTime", "getMaxTime", "getLaserScansTimes", "appendPriorFactors",
                 "appendOdometryFactors", "appendICPFactors", "appendLoopClosureFactors", "initializeGTSAMValues",
                 "updateFromGTSAMValues", "updateCovariancesFromGTSAMValues", "getNumScans", "printTrajectory", "findNearestPose",
                 "buildSubMapAroundTime", "getValueExpression", "evaluate", "getScanMatchingTimes", "saveTrajectory",
                 "processLoopClosure", "getLaserTrack", "getAllLaserTracks", "estimate", "estimateAndRemove", "registerPrior"):
        assert name + "(" in src, name


@pytest.mark.gpu
def test_cpp_laser_track_registers_scans(tmp_path):
    """Four scans along a straight drive: LaserTrack must emit prior / odometry / ICP factors with the
    reference's bookkeeping, and each ICP factor must recover the true relative motion although the
    odometry it starts from is off by 20 cm / 1 deg."""
    exe = _build(tmp_path, "track_driver.cpp", "track_driver")
    scene = synth.Scene(1234)
    n = 4
    truth, odom = [], []
    rng = np.random.default_rng(3)
    for i in range(n):
        T = synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i))
        truth.append(T)
        scan = synth.hdl64_scan(scene, T, 256, 10 + i)
        scan.tofile(tmp_path / f"scan{i}.bin")
        # drifting odometry: each step is off by ~20 cm / 1 deg
        drift = synth.se3(0.2 * i * rng.uniform(0.5, 1), -0.1 * i, 0, yaw=np.deg2rad(1.0 * i))
        odom.append(T @ drift)
    with open(tmp_path / "poses.txt", "w") as f:
        for i, T in enumerate(odom):
            R = T[:3, :3]
            qw = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            q = [qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)]
            f.write("%d %s\n" % (100000000 * i, " ".join(repr(float(v)) for v in [*q, *T[:3, 3]])))
    yaml = os.path.join(ROOT, "tests", "golden", "icp_chain.yaml")
    r = subprocess.run([exe, str(tmp_path), str(n), yaml, "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    scans = [l.split() for l in lines if l.startswith("scan ")]
    assert [int(s[3]) for s in scans] == [1, 0, 0, 0]            # only the first scan yields a prior
    assert [int(s[5]) for s in scans] == [1, 2, 2, 2]            # prior | odometry + ICP
    assert [int(s[9]) for s in scans] == [1, 2, 3, 4]
    factors = [l.split() for l in lines if l.startswith("factor ")]
    icp = [f for f in factors if f[1] == "2"]
    assert len(icp) == n - 1
    for i, f in enumerate(icp):
        q = np.array(f[6:10], float)
        p = np.array(f[11:14], float)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = p
        want = np.linalg.inv(truth[i]) @ truth[i + 1]
        et, er = synth.pose_error(T, want)
        assert et < 0.03 and er < 3e-3, (i, et, er)
        assert int(f[3]) + 1 == int(f[4])                           # consecutive node keys
    its = [int(l.split()[1]) for l in lines if l.startswith("icp_iterations")]
    assert len(its) == n - 1 and all(2 <= k <= 40 for k in its)
    last = lines[-1].split()
    assert last[0] == "world_cloud" and int(last[1]) > 10000 and int(last[3]) > 30000


def _pose_line(t_ns, T):
    R = T[:3, :3]
    qw = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = [qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)]
    return "%d %s\n" % (t_ns, " ".join(repr(float(v)) for v in [*q, *T[:3, 3]]))


def _parse_poses(lines):
    out = []
    for l in lines:
        v = [float(x) for x in l.split()[2:]]
        w, x, y, z = v[:4]
        T = np.eye(4)
        T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
        T[:3, 3] = v[4:]
        out.append(T)
    return out


@pytest.mark.gpu
def test_cpp_incremental_estimator_closes_a_loop(tmp_path):
    """BASELINE config 5 in miniature: a closed drive, odometry with a steady drift,
    ICP factors from the device, pose graph on the host, then one loop closure whose relative pose comes from
    the sub-map vs sub-map ICP (the second icp_.compute call site).  The estimate must beat dead reckoning by
    a wide margin, and the loop closure must pull the end of the trajectory onto its start."""
    exe = _build(tmp_path, "slam_driver.cpp", "slam_driver")
    scene = synth.Scene(1234)
    n, radius = 18, 4.0
    truth = []
    for i in range(n):                       # one lap of a circle, ending where it started (overlapping views)
        a = 2 * np.pi * i / (n - 1)
        truth.append(synth.se3(radius * np.sin(a), radius * (1 - np.cos(a)), synth.SENSOR_HEIGHT, yaw=a))
    bias = synth.se3(0.04, 0.01, 0.0, yaw=np.deg2rad(0.8))   # per-step odometry error
    odom = [truth[0]]
    for i in range(1, n):
        odom.append(odom[-1] @ np.linalg.inv(truth[i - 1]) @ truth[i] @ bias)
    with open(tmp_path / "poses.txt", "w") as f:
        for i in range(n):
            synth.hdl64_scan(scene, truth[i], 256, 50 + i).tofile(tmp_path / f"scan{i}.bin")
            f.write(_pose_line(100000000 * (i + 1), odom[i]))
    yaml = os.path.join(ROOT, "tests", "golden", "icp_chain.yaml")
    r = subprocess.run([exe, str(tmp_path), str(n), yaml, "3", "0", str(n - 1), "1"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    ib, ia = lines.index("before_lc"), lines.index("after_lc")
    before = _parse_poses(lines[ib + 1:ib + 1 + n])
    after = _parse_poses(lines[ia + 1:ia + 1 + n])

    def rmse(est):
        return float(np.sqrt(np.mean([np.sum((e[:3, 3] - t[:3, 3]) ** 2) for e, t in zip(est, truth)])))

    dead = rmse(odom)
    assert dead > 0.5                                         # the odometry alone drifts by more than half a metre
    assert rmse(before) < 0.1 and rmse(before) < dead / 8     # ICP factors hold the trajectory together
    assert rmse(after) <= rmse(before) + 0.01
    end_gap_before = np.linalg.norm(before[-1][:3, 3] - truth[-1][:3, 3])
    end_gap_after = np.linalg.norm(after[-1][:3, 3] - truth[-1][:3, 3])
    assert end_gap_after < 0.03 and end_gap_after <= end_gap_before + 0.005
    lc = [l for l in lines if l.startswith("loop_closure ")][0].split()
    assert 1 <= int(lc[2]) <= 40
    m = _parse_poses(["x x " + " ".join([l for l in lines if l.startswith("lc_measurement ")][0].split()[1:])])[0]
    want = np.linalg.inv(truth[0]) @ truth[-1]                # ~identity: the lap ends where it began
    et, er = synth.pose_error(m, want)
    assert et < 0.03 and er < 3e-3, (et, er)


@pytest.mark.gpu
def test_cpp_submap_on_device_equals_host_assembly(tmp_path):
    """LaserTrack with scans resident in HBM (sub-map assembled by lsgpu_icp_compute_clouds) must produce the
    same ICP factors, bit for bit, as the host assembly of laser_track.cpp:474-486 -- with the scan's host copy made
    beside the registration (the default with an empty input chain) and in front of it, and the stored scans must be
    complete either way (the clouds built from laser_scans_ at the end of the run have the same sizes)."""
    exe = _build(tmp_path, "track_driver.cpp", "track_driver")
    scene = synth.Scene(1234)
    n = 5
    with open(tmp_path / "poses.txt", "w") as f:
        for i in range(n):
            T = synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i))
            synth.hdl64_scan(scene, T, 256, 10 + i).tofile(tmp_path / f"scan{i}.bin")
            f.write(_pose_line(100000000 * i, T @ synth.se3(0.1 * i, -0.05 * i, 0, yaw=np.deg2rad(0.5 * i))))
    yaml = os.path.join(ROOT, "tests", "golden", "icp_chain.yaml")
    outs = []
    for on_device, env in (("16", {}), ("0", {}), ("16", {"LSGPU_TRACK_NO_OVERLAP": "1"})):
        r = subprocess.run([exe, str(tmp_path), str(n), yaml, "3", on_device], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([l for l in r.stdout.splitlines() if l.startswith(("factor ", "icp_iterations", "world_cloud "))])
    strip = lambda ls: [" ".join(l.split()[:5]) if l.startswith("icp_iterations") else l for l in ls]
    assert strip(outs[0]) == strip(outs[1]) == strip(outs[2]) and len(outs[0]) > 2 * (n - 1)
    assert any(l.startswith("world_cloud ") for l in outs[0])


@pytest.mark.gpu
def test_cpp_laser_track_applies_the_input_filter_chain(tmp_path):
    """LaserTrack loads `icp_input_filters_file` (laser_track.cpp:24-30) and filters every scan before it is stored or
    matched (:146): with the test chain the stored clouds shrink, and the ICP factors still recover the motion."""
    exe = _build(tmp_path, "track_driver.cpp", "track_driver")
    scene = synth.Scene(1234)
    n = 4
    truth = []
    with open(tmp_path / "poses.txt", "w") as f:
        for i in range(n):
            T = synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i))
            truth.append(T)
            synth.hdl64_scan(scene, T, 512, 10 + i).tofile(tmp_path / f"scan{i}.bin")
            f.write(_pose_line(100000000 * i, T @ synth.se3(0.1 * i, -0.05 * i, 0, yaw=np.deg2rad(0.5 * i))))
    yaml = os.path.join(ROOT, "tests", "golden", "icp_chain.yaml")
    sizes = {}
    mild = tmp_path / "input_filters_mild.yaml"     # (the golden chain keeps a sixth of the points: too few to judge the ICP)
    mild.write_text("- BoundingBoxDataPointsFilter: {xMin: -6, xMax: 6, yMin: -4, yMax: 4, zMin: -2.5, zMax: 0.5, removeInside: 1}\n"
                    "- MaxDistDataPointsFilter:\n    maxDist: 50\n- RandomSamplingDataPointsFilter: {prob: 0.6}\n")
    for name, flt in (("none", os.path.join(ROOT, "tests", "golden", "input_filters_none.yaml")), ("chain", str(mild))):
        env = dict(os.environ, LSGPU_TEST_INPUT_FILTERS=flt)
        r = subprocess.run([exe, str(tmp_path), str(n), yaml, "3"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = r.stdout.strip().splitlines()
        sizes[name] = int(lines[-1].split()[1])
        icp = [l.split() for l in lines if l.startswith("factor 2 ")]
        assert len(icp) == n - 1
        for i, fct in enumerate(icp):
            T = _parse_poses(["x x " + " ".join(fct[6:10] + fct[11:14])])[0]
            et, er = synth.pose_error(T, np.linalg.inv(truth[i]) @ truth[i + 1])
            assert et < 0.05 and er < 5e-3, (name, i, et, er)
    assert 0.3 * sizes["none"] < sizes["chain"] < 0.65 * sizes["none"]      # near box + far field removed, 60 % of the rest      # box + ranges + every ~4th point x 0.8
    env = dict(os.environ, LSGPU_TEST_INPUT_FILTERS=str(tmp_path / "does_not_exist.yaml"))
    r = subprocess.run([exe, str(tmp_path), "1", yaml, "3"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "input filters" in (r.stdout + r.stderr)


def test_integration_shim_compiles_against_the_mirror_types(tmp_path):
    """integration/lsgpu_icp_shim.hpp (the drop-in for laser_track.hpp:217 / incremental_estimator.hpp:70) is real
    code: instantiated with the in-tree mirror types it compiles and links on a CPU-only box."""
    exe = _build(tmp_path, "shim_check.cpp", "shim_check")
    r = subprocess.run([exe, "--compile-only"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "compiled" in r.stdout


@pytest.mark.gpu
def test_integration_shim_equals_the_mirror_icp(tmp_path):
    """The shim and laser_slam_amd::ICP give bit-identical transforms on the same clouds; its DataPointsFilters twin
    keeps the same points as the mirror's and thins the descriptors with them."""
    exe = _build(tmp_path, "shim_check.cpp", "shim_check")
    ref, rd, T_true, T_init = synth.scan_pair(512)
    ref.tofile(tmp_path / "ref.bin")
    rd.tofile(tmp_path / "rd.bin")
    g = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([exe, os.path.join(g, "icp_chain.yaml"), os.path.join(g, "input_filters.yaml"),
                        str(tmp_path / "ref.bin"), str(tmp_path / "rd.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shim_check: ok" in r.stdout, r.stdout + r.stderr
