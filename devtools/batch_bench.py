"""dev helper (not a test): BASELINE config 3 shape -- 32 pairs of 200 k points on one GPU, pairs/s vs pool size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_slam_amd import synth, icp
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 3125
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
uniq = []
for i in range(4):
    ref, rd, Tt, Ti = synth.scan_pair(n_az, noise_seeds=(1000 + i, 2000 + i), guess_seed=1000 + i)
    rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
    uniq.append((torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda(), Ti, Tt))
torch.cuda.synchronize()
pairs = [uniq[i % 4] for i in range(B)]
refs, nrms, rds, Tis, Tts = map(list, zip(*pairs))
print("points per cloud", rds[0].shape[0], "pairs", B)
for pool in (1, 2, 4, 8, 16):
    hs = [icp.IcpHandle() for _ in range(pool)]
    icp.align_batch(hs, refs, nrms, rds, Tis)
    t = time.perf_counter()
    T, st, rc = icp.align_batch(hs, refs, nrms, rds, Tis)
    dt = time.perf_counter() - t
    err = max(synth.pose_error(T[i], Tts[i])[0] for i in range(B))
    print("pool %2d: %.1f ms  %.1f pairs/s  iters %s  max |t err| %.4f m  rc %s" % (pool, dt * 1e3, B / dt, st[0].iterations, err, set(rc.tolist())))
    for h in hs: h.close()
