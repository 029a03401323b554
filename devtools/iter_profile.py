"""dev helper (not a test): per-iteration kNN timing of one 1M align."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import _lib as _l
if os.environ.get("LSGPU_SO"): _l.SO_PATH = os.environ["LSGPU_SO"]
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref, rd, Tt, Ti = synth.scan_pair(n_az)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4; cfg.profile_kernels = 1; cfg.cell_size = float(os.environ.get("CELL", "0"))
h = icp.IcpHandle(cfg)
dref, dn, drd = torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda()
for rep in range(2):
    t=time.perf_counter(); h.set_reference(dref, dn); t1=time.perf_counter(); T, st = h.align(drd, Ti); t2=time.perf_counter()
print("set_reference ms", (t1-t)*1e3, "align ms", (t2-t1)*1e3, "iters", st.iterations, "retries", st.cap_retries, "sel_retries", st.pad_, "heavy %.4f" % st.direction_index_heavy_share)
info = h.info(); print("chunks", info.n_chunks, "cells", list(info.cells)[:12], "h0", info.cell_size)
for i, tr in enumerate(h.trace()):
    print(i, "limit %.5f used %d knn_main %.1f us fb %.1f us strag %d" % (tr["limit"], tr["n_used"], tr["knn_main_us"], tr["knn_fallback_us"], tr["stragglers"]))
tr = h.trace()
tot = lambda a, b: sum(t["knn_main_us"] + t["knn_fallback_us"] for t in tr[a:b])
print("SUM knn us: all %.0f  it0-2 %.0f  it3-15 %.0f  it16-end %.0f  last8 avg %.1f" % (tot(0, len(tr)), tot(0, 3), tot(3, 16), tot(16, len(tr)), tot(len(tr) - 8, len(tr)) / 8))
