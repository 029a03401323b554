"""dev helper: how the trim limit was found per alignment of the benchmark pair (committed / retries), align time, and
the per-kernel times of one profiled alignment from rocprofv3 when run under it."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, dn, drd = torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda()
h.set_reference(dref, dn)
for rep in range(3):
    T, st = h.align(drd, Ti)
ts = []
for rep in range(10):
    torch.cuda.synchronize(); t = time.perf_counter(); T, st = h.align(drd, Ti); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("align ms median %.3f min %.3f  iters %d committed %d sel_retries %d cap_retries %d" % (np.median(ts), min(ts), st.iterations, st.committed_select_iterations, st.pad_, st.cap_retries))
print("limits", " ".join("%.5f" % t["limit"] for t in h.trace()))
