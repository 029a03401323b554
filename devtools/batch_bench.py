"""BASELINE configs[2] shape on one GPU: independent 200 k-point scan pairs through lsgpu_icp_align_batch, pairs/s against
the number of handles (streams) -- the per-GPU share of "256 pairs over 8 GPUs" is 32 pairs.
usage: batch_bench.py [n_az=3125] [pairs=32] [out.json]      (writes the JSON artifact kept under profiles/)"""
import json, sys, os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (see bench.py: streams that share a hardware queue serialise)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 3125
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out_path = sys.argv[3] if len(sys.argv) > 3 else None
NU = int(os.environ.get("LSGPU_BATCH_UNIQ", "4"))
uniq = []
for i in range(NU):
    ref, rd, Tt, Ti = synth.scan_pair(n_az, noise_seeds=(1000 + i, 2000 + i), guess_seed=1000 + i)
    rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
    uniq.append((torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda(), Ti, Tt))
torch.cuda.synchronize()
# every pair owns its buffers (equal pointers would let align_batch keep a shared reference's structures)
pairs = [(uniq[i % NU][0].clone(), uniq[i % NU][1].clone(), uniq[i % NU][2].clone(), uniq[i % NU][3], uniq[i % NU][4]) for i in range(B)]
refs, nrms, rds, Tis, Tts = map(list, zip(*pairs))
print("points per cloud", rds[0].shape[0], "pairs", B)
rows = []
for pool in [int(x) for x in os.environ.get("LSGPU_BATCH_POOLS", "1,2,3,4,5,6,8,16").split(",")]:
    hs = [icp.IcpHandle() for _ in range(pool)]
    icp.align_batch(hs, refs, nrms, rds, Tis)
    best = None
    for rep in range(3):
        t = time.perf_counter()
        T, st, rc = icp.align_batch(hs, refs, nrms, rds, Tis)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    err = max(synth.pose_error(T[i], Tts[i])[0] for i in range(B))
    print("pool %2d: %.1f ms  %.1f pairs/s  iters %s  max |t err| %.4f m  rc %s" % (pool, best * 1e3, B / best, st[0].iterations, err, set(rc.tolist())))
    rows.append({"handles": pool, "ms_per_batch": best * 1e3, "pairs_per_s": B / best, "iterations_per_pair": int(st[0].iterations),
                 "max_trans_err_m": err})
    for h in hs: h.close()
if out_path:
    json.dump({"workload": "configs[2] per-GPU share: %d independent pairs of %d x %d points (64 x %d rays), default yaml checker, chain F clouds resident in HBM"
                           % (B, rds[0].shape[0], refs[0].shape[0], n_az), "entry_point": "lsgpu_icp_align_batch", "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "rows": rows,
               "command": "python devtools/batch_bench.py %d %d %s" % (n_az, B, out_path)}, open(out_path, "w"), indent=1)
