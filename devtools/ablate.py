"""dev helper: k_knn_tile time under ablation flags (stats build; results are wrong by design)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "devtools", "liblsgpu_stats.so")
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4; cfg.profile_kernels = 1
cfg.max_iterations = 12
h = icp.IcpHandle(cfg)
dref, dn, drd = torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda()
h.set_reference(dref, dn)
for rep in range(2):
    T, st = h.align(drd, Ti)
print(os.environ.get("LSGPU_KNN_DBG"), [round(t["knn_main_us"]) for t in h.trace()])
