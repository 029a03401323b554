cd /root/repo
for g in 2048 512 0; do
  echo "LSGPU_FRONT_GUESS=$g"
  LSGPU_FRONT_GUESS=$g timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'knn', round(r['avg_main_us'],1), round(r['avg_fallback_us'],1), 'ne', round(d['roofline_ne']['avg_us'],1), 'sel', round(d['roofline_select']['avg_us'],1), d['final_error_vs_truth'])"
done
timeout 900 python -m pytest tests -m gpu -q -x -k "not config4_sequence and not config3_full" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
python - <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, ctypes as C
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
ref, rd, Tt, Ti = synth.scan_pair(16384)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
h = icp.IcpHandle(cfg)
dref, dn = h.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
h.set_reference(dref.clone(), dn.clone()); T, st = h.align(torch.from_numpy(rd).cuda(), Ti)
print("spread tiles", st.spread_tiles, "of", (rd.shape[0]+63)//64, "iterations", st.iterations)
PY
