"""dev helper: per-scan time of LaserTrack::processPoseAndLaserScan (C++ mirror) on 1M-ray scans,
sub-map assembled on the device (scans resident in HBM) vs on the host."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from laser_slam_amd import synth
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 12
d = tempfile.mkdtemp()
exe = os.path.join(d, "track_driver")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", ROOT + "/include", "-I", ROOT + "/laser_slam_amd/cpp/include",
                       ROOT + "/tests/cpp/track_driver.cpp", "-o", exe, "-L", ROOT + "/laser_slam_amd", "-llsgpu_icp",
                       "-Wl,-rpath," + ROOT + "/laser_slam_amd"])
scene = synth.Scene(1234)
with open(d + "/poses.txt", "w") as f:
    for i in range(n):
        T = synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i))
        synth.hdl64_scan(scene, T, n_az, 10 + i).tofile(d + f"/scan{i}.bin")
        R = (T @ synth.se3(0.1, -0.05, 0, yaw=np.deg2rad(0.5)))
        qw = np.sqrt(1 + R[0, 0] + R[1, 1] + R[2, 2]) / 2
        q = [qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)]
        f.write("%d %s\n" % (100000000 * i, " ".join(repr(float(v)) for v in [*q, *R[:3, 3]])))
for on_device in ("16", "0"):
    r = subprocess.run([exe, d, str(n), ROOT + "/tests/golden/icp_chain.yaml", "3", on_device], capture_output=True, text=True)
    ms = [float(l.split()[-1]) for l in r.stdout.splitlines() if l.startswith("icp_iterations")]
    its = [int(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("icp_iterations")]
    print("scans_on_device=%s: per-scan ms %s  iterations %s" % (on_device, ["%.1f" % m for m in ms], its), r.stderr[-300:])
