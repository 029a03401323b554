"""dev helper: lsgpu_icp_compute of the benchmark pair from HOST buffers (pageable / pinned), chain F: ms per scan."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
p_ref, p_rd = torch.from_numpy(ref).pin_memory(), torch.from_numpy(rd).pin_memory()
d_ref, d_rd = torch.from_numpy(ref).cuda(), torch.from_numpy(rd).cuda()
for name, a_rd, a_ref in (("resident", d_rd, d_ref), ("pageable", rd, ref), ("pinned", p_rd.numpy(), p_ref.numpy())):
    ts = []
    for rep in range(8):
        t = time.perf_counter(); T, st = h.compute(a_rd, a_ref, Ti, 1.0, 10, 1.0, seed=0); ts.append((time.perf_counter() - t) * 1e3)
    print("%-9s median %.3f ms  min %.3f  (%d iterations, filters+grid %.2f ms)" % (name, np.median(ts[2:]), min(ts[2:]), st.iterations, st.t_reserved[0]))
