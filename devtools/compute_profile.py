"""dev helper: the whole ICP::compute on the device (filters + set_reference + align), 1M-point pair."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref, rd, Tt, Ti = synth.scan_pair(n_az)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, drd = torch.from_numpy(ref).cuda(), torch.from_numpy(rd).cuda()
torch.cuda.synchronize()
for name, prob, ratio in (("P (yaml: 0.5 / 0.5)", 0.5, 0.5), ("F (1.0 / 1.0)", 1.0, 1.0)):
    for rep in range(3):
        t = time.perf_counter(); T, st = h.compute(drd, dref, Ti, prob, 10, ratio, seed=1); dt = time.perf_counter() - t
    print("%s: compute %.2f ms (filters+set_reference %.2f ms, align %.2f ms, %d iterations) n_ref %d  err %.4f m" % (
        name, dt * 1e3, st.t_reserved[0], st.t_total_ms, st.iterations, h.info().n_reference, synth.pose_error(T, Tt)[0]))
for rep in range(3):
    t = time.perf_counter(); f, n = h.filter_reference(dref, 10, 1.0, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("filter_reference alone (device in/out): %.2f ms" % (dt * 1e3))
for rep in range(3):
    t = time.perf_counter(); f = h.filter_reading(drd, 0.5, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("filter_reading alone: %.2f ms" % (dt * 1e3))
