"""dev helper: a few whole computes of the benchmark pair (for rocprofv3 timelines: devtools/timeline.py)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cache = "/tmp/lsgpu_pair_%d.npz" % n_az
if os.path.exists(cache):
    z = np.load(cache); ref, rd, Ti = z["ref"], z["rd"], z["Ti"]
else:
    ref, rd, Tt, Ti = synth.scan_pair(n_az)
    np.savez(cache, ref=ref, rd=rd, Ti=Ti)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, drd = torch.from_numpy(ref).cuda(), torch.from_numpy(rd).cuda()
torch.cuda.synchronize()
for rep in range(reps):
    t = time.perf_counter(); T, st = h.compute(drd, dref, Ti, 1.0, 10, 1.0, seed=0); dt = time.perf_counter() - t
    print("compute %.3f ms, filters+grid %.3f, %d iterations" % (dt * 1e3, st.t_reserved[0], st.iterations))
    time.sleep(0.02)   # (a visible gap between the steps in the trace)
