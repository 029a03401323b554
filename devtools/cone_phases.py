"""dev helper (stats build): where a wave of k_knn_cone spends its cycles, per ICP iteration of the benchmark align."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from laser_slam_amd import _lib
_lib.SO_PATH = os.environ.get("LSGPU_SO", os.path.join(ROOT, "devtools", "liblsgpu_stats.so"))
from laser_slam_amd import synth, icp
from laser_slam_amd._lib import IcpConfig, lib
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref, rd, Tt, Ti = synth.scan_pair(n_az)
rf, rn = icp.sampling_surface_normal(ref, 10, 1.0, 0)
cfg = IcpConfig(); lib().lsgpu_icp_config_yaml(C.byref(cfg)); cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
cfg.profile_kernels = 1
h = icp.IcpHandle(cfg)
dref, dn, drd = torch.from_numpy(rf).cuda(), torch.from_numpy(rn).cuda(), torch.from_numpy(rd).cuda()
nt = (rd.shape[0] + 63) // 64
ITERS = 48
f = lib().lsgpu_dev_cone_phases
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for rep in range(2):
    h.set_reference(dref, dn)
    assert f(h._h, None, nt) == 0
    T, st = h.align(drd, Ti)
out = np.zeros(ITERS * nt * 16, np.uint32)
assert f(h._h, out.ctypes.data, nt) == 0
v = out.reshape(ITERS, nt, 16).astype(np.float64)
tr = h.trace()
def pct(x, q): return np.percentile(x, q) if len(x) else 0.0
print("cycles per wave (mean; p50/p99 where given)")
print("iter | knn us | ph1 load | ph1 comp | pack | p2 waves | lanes | rows+tab wait (p50/p99) | stage (p50/p99) | eval (p50/p99) | rowslots | chunks | steps | groups | lane eff | body p50/p99 | tail p50/p99 | p2 total p50/p99/max | non-p2 total")
for i in range(st.iterations):
    r = v[i]
    live = r[:, 3] > 0
    if not live.any(): continue
    p2 = (r[:, 3].astype(np.int64) & 2) > 0
    a, b = r[live], r[p2]
    lanes = (b[:, 3].astype(np.int64) >> 8) & 0xFF
    if not len(b): continue
    print("%2d | %6.1f | %6.0f | %6.0f | %6.0f | %6d | %5.1f | %6.0f (%6.0f/%6.0f) | %6.0f (%6.0f/%6.0f) | %6.0f (%6.0f/%6.0f) | %4.2f | %4.2f | %5.1f | %6.1f | %4.2f | %6.0f/%6.0f | %6.0f/%6.0f | %6.0f/%6.0f/%6.0f | %6.0f" % (
        i, tr[i]["knn_main_us"], a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), len(b), lanes.mean(),
        b[:, 4].mean(), pct(b[:, 4], 50), pct(b[:, 4], 99), b[:, 15].mean(), pct(b[:, 15], 50), pct(b[:, 15], 99),
        b[:, 5].mean(), pct(b[:, 5], 50), pct(b[:, 5], 99), b[:, 13].mean(), b[:, 6].mean(), b[:, 7].mean(), b[:, 8].mean(),
        b[:, 8].sum() / max(64 * b[:, 7].sum(), 1), pct(b[:, 14], 50), pct(b[:, 14], 99),
        pct(b[:, 9], 50), pct(b[:, 9], 99), pct(b[:, 10], 50), pct(b[:, 10], 99), b[:, 10].max(),
        a[~p2[live]][:, 10].mean() if (~p2[live]).any() else 0))
for it in (3, 10, 20, st.iterations - 1):
    r = v[it]
    p2 = (r[:, 3].astype(np.int64) & 2) > 0
    idx = np.flatnonzero(p2)
    if not len(idx): continue
    order = idx[np.argsort(-r[idx, 10])][:10]
    print("iteration %d: slowest waves" % it)
    for t in order:
        print("  tile %6d total %6d ph1 %5d rows %6d stage %6d eval %6d rowslots %2d chunks %3d steps %3d groups %4d lanes %2d fb %2d tail %5d body %6d" % (
            t, r[t, 10], r[t, 0] + r[t, 1] + r[t, 2], r[t, 4], r[t, 15], r[t, 5], r[t, 13], r[t, 6], r[t, 7], r[t, 8], (int(r[t, 3]) >> 8) & 0xff, (int(r[t, 3]) >> 16) & 0xff, r[t, 9], r[t, 14]))

# the launch's timeline on the device-wide 100 MHz clock: when waves start and end, and who ends last
print("timeline (us from the first wave's start): iter | length | starts p50/p90/max | searching waves' starts p50/p90/max | the ten last to end: tile start dur(us) steps")
raw = out.reshape(ITERS, nt, 16)
for it in range(2, st.iterations):
    r = raw[it]
    live = r[:, 3] > 0
    if not live.any(): continue
    s0 = r[:, 11].astype(np.int64); e0 = r[:, 12].astype(np.int64)
    base = s0[live].min()
    s_us = (s0 - base) / 100.0; e_us = (e0 - base) / 100.0
    p2 = ((r[:, 3].astype(np.int64) & 2) > 0) & live
    last = np.argsort(-np.where(live, e_us, -1))[:10]
    print("%2d | %5.1f | %5.1f/%5.1f/%5.1f | %5.1f/%5.1f/%5.1f | %s" % (
        it, e_us[live].max(), pct(s_us[live], 50), pct(s_us[live], 90), s_us[live].max(),
        pct(s_us[p2], 50), pct(s_us[p2], 90), s_us[p2].max() if p2.any() else 0,
        " ".join("%d:%.1f+%.1f(%d)" % (t, s_us[t], e_us[t] - s_us[t], r[t, 7]) for t in last)))
    if it in (3, 10, 20, st.iterations - 1):
        d = e_us - s_us
        heavy = np.argsort(-np.where(p2, d, -1))[:256]
        print("     the 256 longest waves: durations p50 %.1f max %.1f us, starts p10 %.1f p50 %.1f p90 %.1f; if they had started at 0 the launch would end at max(%.1f, the rest)" % (
            pct(d[heavy], 50), d[heavy].max(), pct(s_us[heavy], 10), pct(s_us[heavy], 50), pct(s_us[heavy], 90), d[heavy].max()))
        hist, edges = np.histogram(s_us[live], bins=12)
        print("     starts histogram:", " ".join("%.0f-%.0f:%d" % (edges[k], edges[k + 1], hist[k]) for k in range(12)))
        hist, edges = np.histogram(e_us[live], bins=12)
        print("     ends histogram:  ", " ".join("%.0f-%.0f:%d" % (edges[k], edges[k + 1], hist[k]) for k in range(12)))
