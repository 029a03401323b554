#!/usr/bin/env python3
"""bench.py -- scans/sec + ms/ICP-iteration of the ICP hot path on 1 M-point Velodyne scans.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--batch | --split]

Default workload = BASELINE.json configs[1]: one synthetic HDL-64E pair of 64 x 16384 rays, full-density chain (F)
(SURVEY.md §8d).  A step = one WHOLE `icp_.compute(reading, reference, T_init)` as laser_slam calls it
(laser_slam/src/laser_track.cpp:496) = lsgpu_icp_compute: reference filter (SamplingSurfaceNormal), centring + voxel
grid, reading filter, then {transform, exact 1-NN, trimmed weights, point-to-plane 6x6} until the (tightened, 1e-4 m /
1e-5 rad) differential checker stops it -- on RAW clouds that are resident in HBM when the timed region starts.
`value` = scans/s of that step.  Reported beside it in the same line:
  value_loop  the resident loop alone (set_reference + align on already filtered clouds: the north-star kernels)
  value_e2e   the same compute handed HOST buffers (H2D + D2H inclusive; pageable / pinned) -- never `value`
With N > 1 every rank registers its own pair (embarrassingly parallel, no data-path collective; "weak" scaling).

--batch  BASELINE configs[2]: ONE step = 256 independent 200 k-point pairs (64 x 3125 rays), pair i -> rank i mod N
         (sharding.pairs_of_rank), each rank runs its share through lsgpu_icp_align_batch on a pool of handles; no
         collective; value = pairs/s over all ranks ("strong": the batch is fixed).
--split  BASELINE configs[3]: ONE 8.4 M-point local map (8 scans in one frame) vs ONE 1 M-point scan per step, the
         reading sharded over the ranks, RCCL all-reduce of the select tables + 29 f64 per iteration; value = scans/s
         ("strong").  --split-pair uses the configs[1] pair instead (what round 2 measured on one rank).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP maps a process' streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round robin, and streams that share a queue
# serialise.  A handle has four streams of its own (loop, side chain, upload, draws): with 4 queues the upload shared one with
# the filters -- the whole reason a compute from PINNED host buffers was slower than from pageable ones in rounds 3-5 (198 vs
# 207 scans/s; with 8 queues 210 vs 205, same box, alternating runs) -- and four handles' loops shared theirs (--batch: 1 504
# pairs/s with 4 queues against 2 170 with 8; five or more busy queues collapse again, so --batch runs four handles per rank).
# Read by the runtime when it starts: it has to be in the environment before torch / the library touch the device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def csrc_digest() -> str:
    """sha256 over the kernel sources: artifacts measured out of band (PMC passes) carry it and are refused if stale."""
    d = hashlib.sha256()
    base = os.path.join(ROOT, "laser_slam_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".h", ".hip", ".cpp")):
            d.update(name.encode())
            d.update(open(os.path.join(base, name), "rb").read())
    return d.hexdigest()[:16]


def _cpu_model() -> str:
    """Model name of the host CPU (SURVEY.md 8d asks for it next to the CPU baseline)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def relaunch_argv(argv, n_gpus: int, port: int):
    """`python bench.py --gpus N ...` started WITHOUT a launcher (no WORLD_SIZE in the environment): the command line that
    runs the same arguments as N ranks of one node, one rank per GPU over RCCL -- exactly the shape the driver uses
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`).
    Pure function of its arguments (tests/test_sharding.py checks it on the CPU)."""
    if n_gpus < 2:
        raise ValueError("a single rank needs no launcher")
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__), *argv]


def needs_relaunch(n_gpus: int, environ) -> bool:
    """True when --gpus asks for more ranks than this process is part of (no launcher environment)."""
    return n_gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def track_section(n_az: int, n_scans: int, cpu_threads: int):
    """SURVEY.md 8d, the reference's OWN timed region: `scan_matching_times_` of LaserTrack::processPoseAndLaserScan
    (laser_slam/src/laser_track.cpp:128, 208-209 -- the clock runs from the top of the call to the end of
    computeICPTransformations) for a drive of `n_scans` synthetic 64 x n_az-ray scans through the C++ mirror
    (tests/cpp/track_driver.cpp): nscan_in_sub_map 3, the yaml chain (tests/golden/icp_chain.yaml = icp_default.yaml's
    modules), scans resident in HBM (scans_on_device 16).  A second run of the same drive hands ONE scan's ICP inputs --
    the reading, the assembled 3-scan sub-map, the guess -- to the CPU oracle as well: its wall time for that
    `icp_.compute` is the CPU leg (everything else processPoseAndLaserScan does is O(1) beside it), the difference of
    the two transforms the per-call parity figure."""
    import multiprocessing as mp
    import shutil
    import subprocess
    import tempfile
    from laser_slam_amd import synth
    from oracle import oracle_py
    oracle_py.build()
    d = tempfile.mkdtemp(prefix="lsgpu_track_")
    try:
        exe = os.path.join(d, "track_driver")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-DLSGPU_TRACK_SHADOW", "-DLSGPU_TEST_SEAMS", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "laser_slam_amd", "cpp", "include"), os.path.join(ROOT, "tests", "cpp", "track_driver.cpp"),
                               "-o", exe, "-L", os.path.join(ROOT, "laser_slam_amd"), "-llsgpu_icp", "-L", os.path.join(ROOT, "oracle"), "-llsoracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "laser_slam_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
        poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(n_scans)]
        jobs = [(1234, poses[i], n_az, 10 + i) for i in range(n_scans)]
        with mp.get_context("spawn").Pool(min(n_scans, os.cpu_count() or 1)) as pool:
            scans = pool.map(synth.scan_job, jobs)
        with open(os.path.join(d, "poses.txt"), "w") as f:
            for i, (T, sc) in enumerate(zip(poses, scans)):
                sc.tofile(os.path.join(d, "scan%d.bin" % i))
                q = synth.quat_wxyz(T @ synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5)))   # odometry: truth off by 10 cm / 0.5 deg
                f.write("%d %s\n" % (100000000 * i, " ".join(repr(float(v)) for v in [*q, *(T @ synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5)))[:3, 3]])))
        yaml = os.path.join(ROOT, "tests", "golden", "icp_chain.yaml")
        shadow_scan = n_scans - 1

        def run(extra):
            r = subprocess.run([exe, d, str(n_scans), yaml, "3", "16", *extra], capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, LSGPU_TRACK_STAGES="1"))
            if r.returncode != 0:
                raise RuntimeError("track_driver failed: " + r.stdout[-500:] + r.stderr[-500:])
            if os.environ.get("LSGPU_GS_DEBUG"):   # (dev: the library's diagnostics of the driver process)
                sys.stderr.write(r.stderr)
            return r.stdout.splitlines()
        lines = run([])
        ms = [float(l.split()[-1]) for l in lines if l.startswith("icp_iterations")]          # scans 1 .. n-1
        its = [int(l.split()[1]) for l in lines if l.startswith("icp_iterations")]
        # steady state: the sub-map holds 3 scans from scan 4 on, but the track keeps allocating HBM slots for new scans until
        # `scans_on_device` (16) of them are resident -- a robot drives thousands of scans, the first sixteen are start-up
        steady = ms[16:] if len(ms) > 18 else ms[3:]
        stages = [dict(zip(l.split()[1::2], map(float, l.split()[2::2]))) for l in lines if l.startswith("stages ")]
        all_stages = stages
        stages = stages[16:] if len(stages) > 18 else stages[3:]
        sh = [l.split() for l in run([str(shadow_scan), str(cpu_threads)]) if l.startswith("shadow ")]
        out = {"value": 1e3 / float(np.median(steady)), "unit": "scans/s", "ms_per_scan_median": float(np.median(steady)),
               "ms_per_scan": [round(m, 3) for m in ms], "icp_iterations": its, "n_scans": n_scans, "points_per_scan": int(scans[0].shape[0]),
               "workload": "LaserTrack::processPoseAndLaserScan through the C++ mirror, %d scans of 64 x %d rays 0.8 m / 2 deg apart, nscan_in_sub_map 3 "
                           "(sub-map of ~%.1f M points), yaml chain (prob 0.5 / ratio 0.5), scans_on_device 16; timed region = scan_matching_times_ "
                           "(laser_track.cpp:128, 208-209); steady state = scans %d .. %d (every HBM slot allocated)" % (n_scans, n_az, 3 * scans[0].shape[0] / 1e6, n_scans - len(steady), n_scans - 1)}
        if stages:
            out["stages_ms_median"] = {k[:-3]: round(float(np.median([st[k] for st in stages])), 3) for k in stages[0] if k.endswith("_ms")}
            out["stages_ms"] = {k[:-3] if k.endswith("_ms") else k: [round(st[k], 2) for st in all_stages] for k in all_stages[0]}
            out["stages_are"] = ("copy = the call's working copy of the scan + input filters (host); upload = H2D of sub-map scans that are not resident "
                                 "(none in steady state; the NEW scan's H2D goes out with icp_.compute -- lsgpu_icp_compute_clouds_upload -- and crosses PCIe "
                                 "while the sub-map is filtered); icp = icp_.compute (device_filters / device_total: the C ABI's own clocks inside it)")
        if sh:
            t = sh[0]
            kv = {t[i]: t[i + 1] for i in range(1, len(t) - 1, 2)}
            out["cpu_baseline_track"] = {"value": 1e3 / float(kv["oracle_ms"]), "unit": "scans/s", "ms_per_scan": float(kv["oracle_ms"]), "cores": int(kv["threads"]),
                                         "kind": "port", "sample": "the icp_.compute of scan %s of the same drive (reading %s points, sub-map %s points, same guess, same draws) on the CPU oracle"
                                                                   % (kv["scan"], kv["reading"], kv["reference"]),
                                         "oracle_iterations": int(kv["oracle_iterations"]), "device_iterations": int(kv["device_iterations"])}
            out["gpu_vs_cpu_transform"] = {"trans_m": float(kv["dt"]), "rot_rad": float(kv["dr"]), "calls_compared": 1}
        return out, scans, poses
    finally:
        shutil.rmtree(d, ignore_errors=True)


def dry_run(args, rank, world):
    """`--dry-run`: everything of a multi-rank run that is NOT the device -- the rendezvous the driver's launcher sets up, the
    shards every mode derives from (rank, world), the unique-id broadcast and the entry handshake of the split-scan mode, the
    per-iteration exchange pattern (sums of per-shard tables -> the same decision on every rank), the max-over-ranks clock and
    the JSON line -- on the gloo backend with a 4 k-point host workload.  No search, no ICP: nothing here is a measurement.
    Returns the process' exit code (non-zero: a peer was missing and this rank gave up in bounded time)."""
    import datetime
    import torch
    import torch.distributed as dist
    from laser_slam_amd import synth, sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    wait_s = float(os.environ.get("LSGPU_COMM_TIMEOUT_MS", "5000")) / 1e3   # (the library bounds its stream waits with the same variable)
    if rank == args.dry_run_dead_rank:
        return 0                                   # a peer that died before the first exchange
    out = {"dry_run": True, "backend": "gloo", "n_gpus": world, "mode": "batch" if args.batch else "split" if args.split else "default"}

    def bounded(work):                             # a collective that a dead peer cannot turn into a hang
        try:
            work.wait(datetime.timedelta(seconds=wait_s))
            return True
        except Exception as e:                     # (gloo raises on the timeout; RCCL's counterpart is the bounded stream wait)
            print("bench.py --dry-run rank %d: peer missing, giving up (%s)" % (rank, str(e)[:80]), file=sys.stderr)
            return False

    def allreduce(t, op=None):
        if world == 1:
            return True
        return bounded(dist.all_reduce(t, op=op or dist.ReduceOp.SUM, async_op=True))

    t0 = time.perf_counter()
    if args.batch:
        mine = sharding.pairs_of_rank(args.batch_pairs, rank, world)
        owner = torch.zeros(args.batch_pairs, dtype=torch.int64)
        owner[mine] = 1
        if not allreduce(owner):
            return 3
        out["pairs_owned_exactly_once"] = bool((owner == 1).all().item())
        out["pairs_of_rank0"] = len(sharding.pairs_of_rank(args.batch_pairs, 0, world))
    else:
        data_rank = 0 if args.split else rank
        ref, rd, T_true, T_init = synth.scan_pair(64, noise_seeds=(1 + 2 * data_rank, 2 + 2 * data_rank), guess_seed=7 + data_rank)
        if args.split:
            sl = sharding.split_shard(rd.shape[0], rank, world)

            class _Handle:                         # stands where an IcpHandle would: records what comm_init was given
                def comm_init(self, r, w, uid):
                    self.got = (r, w, bytes(uid))
            hnd = _Handle()
            import laser_slam_amd.icp as icp_mod
            real_uid = icp_mod.comm_unique_id
            icp_mod.comm_unique_id = lambda: bytes((7 * i + 3) % 256 for i in range(128))   # (no librccl here)
            try:
                sharding.init_split_comm(hnd, device=None)
            finally:
                icp_mod.comm_unique_id = real_uid
            uid_sum = torch.tensor([float(sum(hnd.got[2]))], dtype=torch.float64)
            uid_max = uid_sum.clone()
            # entry handshake of lsgpu_icp_align in the split-scan mode: {shard size, cannot-start flag} summed over the ranks
            hs = torch.tensor([sl.stop - sl.start, 0], dtype=torch.int64)
            if not (allreduce(hs) and allreduce(uid_max, dist.ReduceOp.MAX) and allreduce(uid_sum)):
                return 3
            out["handshake"] = {"points_total": int(hs[0]), "cannot_start": int(hs[1]), "covers_the_reading": int(hs[0]) == rd.shape[0]}
            out["unique_id_equal_on_all_ranks"] = bool(abs(uid_sum.item() - world * uid_max.item()) < 0.5) and hnd.got[:2] == (rank, world)
            # the exchange pattern of an iteration: per-shard tables summed, every rank derives the same trim rank / decision
            decisions = []
            for it in range(3):
                x = rd[sl, :3].astype(np.float64) + 0.01 * it
                d2 = (x ** 2).sum(1).astype(np.float32)
                hist = torch.from_numpy(np.bincount(d2.view(np.uint32) >> 20, minlength=4096).astype(np.int64))
                if not allreduce(hist):
                    return 3
                k = int(np.float32(int(hs[0])) * np.float32(0.75))
                b1 = int(np.searchsorted(np.cumsum(hist.numpy()), k, side="right"))
                decisions.append(b1)
            dec = torch.tensor(decisions, dtype=torch.int64)
            dmin, dmax = dec.clone(), dec.clone()
            if not (allreduce(dmin, dist.ReduceOp.MIN) and allreduce(dmax, dist.ReduceOp.MAX)):
                return 3
            out["decisions_equal_on_all_ranks"] = bool((dmin == dmax).all().item())
        else:
            out["pair_of_this_rank_differs_from_rank0s"] = bool(rank == 0 or not np.array_equal(rd, synth.scan_pair(64)[1]))
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if not allreduce(el, dist.ReduceOp.MAX if world > 1 else None):
        return 3
    out["max_elapsed_s_over_ranks"] = float(el.item())
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-az", type=int, default=16384, help="azimuth steps (16384 -> 1 M rays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--no-compute-e2e", action="store_true",
                    help="skip the value_loop / value_e2e sections (used for the rocprofv3 runs, so that the kernel\n"
                         "averages of the profile cover the timed workload only)")
    ap.add_argument("--no-track", action="store_true", help="skip value_track (LaserTrack::processPoseAndLaserScan through the C++ mirror, with the oracle-driven facade beside it)")
    ap.add_argument("--track-scans", type=int, default=22)
    ap.add_argument("--batch", action="store_true", help="BASELINE configs[2]: 256 x 200 k-point pairs sharded over the ranks")
    ap.add_argument("--batch-pairs", type=int, default=256)
    ap.add_argument("--batch-handles", type=int, default=4,
                    help="handles (= host threads = busy streams) per rank: 4 with GPU_MAX_HW_QUEUES=8 is the measured optimum\n"
                         "(2 170 pairs/s per GPU; 5 and more fall back to 1 300-1 700, profiles/r06_batch200k.json)")
    ap.add_argument("--split", action="store_true", help="BASELINE configs[3]: one 8.4 M local map vs one 1 M scan, reading sharded, RCCL")
    ap.add_argument("--split-pair", action="store_true", help="(with --split) the configs[1] pair instead of the 8-scan local map")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the launch / rendezvous / sharding / handshake / reduction plumbing of the chosen mode on the gloo\n"
                         "backend with a small host workload, so that a multi-GPU lease is not spent finding a typo in it")
    ap.add_argument("--dry-run-dead-rank", type=int, default=-1, help="(with --dry-run) this rank leaves before the first exchange: the others must give up, not hang")
    args = ap.parse_args()

    if needs_relaunch(args.gpus, os.environ):
        # `python bench.py --gpus 8` (no launcher): become N ranks of this node, default / --batch / --split alike
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(relaunch_argv(sys.argv[1:], args.gpus, _free_port()), env=env))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); n_gpus reports the ranks that ran" % (args.gpus, world), file=sys.stderr)
    if args.dry_run:
        try:
            code = dry_run(args, rank, world)
        except Exception as e:   # (a peer that vanished inside a blocking call of the rendezvous: same verdict, bounded time)
            print("bench.py --dry-run rank %d: %s" % (rank, repr(e)[:200]), file=sys.stderr)
            code = 3
        raise SystemExit(code)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the ICP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from laser_slam_amd import synth, icp, sharding
    from laser_slam_amd._lib import IcpConfig, lib

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4        # configs[1]: "to 1e-4 m tolerance"
    cfg.profile_kernels = 0

    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "csrc_sha": csrc_digest(),
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}

    # ------------------------------------------------------------------------------------------ configs[2]: --batch
    if args.batch:
        cfgy = IcpConfig()
        lib().lsgpu_icp_config_yaml(C.byref(cfgy))          # the yaml checker (1e-3 rad / 1e-2 m), as configs[2] does not tighten it
        mine = sharding.pairs_of_rank(args.batch_pairs, rank, world)
        # (the pool's streams first: HIP hands hardware queues to streams in creation order, and streams created behind the
        # filter handle's -- which come and go -- ended up two to a queue: 945 pairs/s against 2 200 for the same four handles)
        hs = [icp.IcpHandle(cfgy, local_rank) for _ in range(args.batch_handles)]
        uniq = {}
        with icp.IcpHandle(None, local_rank) as hf:
            for u in sorted({i % 16 for i in mine}):        # 16 distinct scenes' worth of scans, cycled over the batch
                ref, rd, Tt, Ti = synth.scan_pair(3125, noise_seeds=(1000 + u, 2000 + u), guess_seed=1000 + u)
                rf, rn = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)
                uniq[u] = (rf.contiguous().clone(), rn.contiguous().clone(), torch.from_numpy(rd).cuda(), Ti, Tt)
        # every pair owns its buffers: equal POINTERS would let align_batch keep a shared reference's structures, which
        # independent pairs do not have
        pairs = [(uniq[i % 16][0].clone(), uniq[i % 16][1].clone(), uniq[i % 16][2].clone(), uniq[i % 16][3], uniq[i % 16][4]) for i in mine]
        refs, nrms, rds, Tis, Tts = map(list, zip(*pairs)) if pairs else ([], [], [], [], [])
        torch.cuda.synchronize()

        def step_batch():
            return icp.align_batch(hs, refs, nrms, rds, Tis) if pairs else (None, [], np.zeros(0, np.int32))

        for _ in range(args.warmup):
            step_batch()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            T, st, rc = step_batch()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        err = max((synth.pose_error(T[i].astype(np.float64), Tts[i])[0] for i in range(len(pairs))), default=0.0)
        out = dict(base, metric="scan_pairs_per_sec", value=args.batch_pairs * args.steps / elapsed, unit="pairs/s",
                   ms_per_step=elapsed / args.steps * 1e3, scaling="strong",
                   config={"workload": "configs[2]: batch of %d independent 200 k-point scan pairs (64 x 3125 rays, chain F clouds "
                                       "resident in HBM, yaml checker), lsgpu_icp_align_batch on %d handles per rank"
                                       % (args.batch_pairs, args.batch_handles),
                           "pairs_per_rank": len(mine), "points_per_cloud": int(rds[0].shape[0]) if pairs else 0,
                           "distinct_scan_pairs": 16, "sharding": "pair i -> rank i mod N (sharding.pairs_of_rank), no collective"},
                   iterations_per_pair=float(np.mean([s.iterations for s in st])) if pairs else 0.0,
                   max_trans_err_m_rank0=err, not_converged_rank0=int((rc != 0).sum()))
        if rank == 0:
            print(json.dumps(out))
        for h in hs:
            h.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------------------------------ workload clouds (host)
    data_rank = 0 if args.split else rank   # split: every rank works on the SAME pair
    ref, rd, T_true, T_init = synth.scan_pair(args.n_az, noise_seeds=(1 + 2 * data_rank, 2 + 2 * data_rank),
                                              guess_seed=7 + data_rank)
    raw_ref, raw_rd = ref, rd
    workload = ("configs[1]: single 1M-point HDL-64E scan pair (64x%d rays), full-density chain F, point-to-plane ICP, "
                "differential checker 1e-4 m / 1e-5 rad" % args.n_az)
    if args.split and not args.split_pair:
        # configs[3]: the local map = 8 scans along the trajectory in the frame of the newest one, the reading = the next scan
        scene = synth.Scene(1234)
        step_T = synth.se3(0.8, 0.05, 0.0, yaw=np.deg2rad(2.0), pitch=np.deg2rad(0.2))
        poses = [synth.se3(0.0, 0.0, synth.SENSOR_HEIGHT)]
        for _ in range(8):
            poses.append(poses[-1] @ step_T)
        clouds = []
        for i in range(8):
            s = synth.hdl64_scan(scene, poses[i], args.n_az, 100 + i)
            Trel = np.linalg.inv(poses[7]) @ poses[i]
            s[:, :3] = (s[:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
            clouds.append(s)
        ref = np.concatenate(clouds)
        rd = synth.hdl64_scan(scene, poses[8], args.n_az, 200)
        T_true = step_T
        T_init = synth.scan_pair(64)[3]   # the same perturbed guess as configs[1] (same step between the poses)
        raw_ref, raw_rd = ref, rd
        workload = ("configs[3]: 8-scan local map (%d points after the reference filter's input) vs one 64x%d-ray scan, reading "
                    "sharded over the ranks" % (ref.shape[0], args.n_az))
    with icp.IcpHandle(None, local_rank) as hf:               # chain (F): ratio 1.0, knn 10; the device filter
        d_ref, d_nrm = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)  # == host filter == oracle
    d_ref, d_nrm = d_ref.contiguous().clone(), d_nrm.contiguous().clone()
    rd_whole = rd
    if args.split:
        rd = rd[sharding.split_shard(rd.shape[0], rank, world)]
    d_rd = torch.from_numpy(rd).cuda()
    d_raw_ref, d_raw_rd = torch.from_numpy(raw_ref).cuda(), torch.from_numpy(raw_rd).cuda()
    torch.cuda.synchronize()
    nq, nr = rd.shape[0], int(d_ref.shape[0])

    h = icp.IcpHandle(cfg, local_rank)
    # second handle, identical but with a HIP-event pair around every kNN launch: used for a few extra
    # steps right after the timed region (event records inside the timed region cost ~5 % throughput)
    cfg_p = IcpConfig()
    C.memmove(C.byref(cfg_p), C.byref(cfg), C.sizeof(cfg))
    cfg_p.profile_kernels = 1
    hp = icp.IcpHandle(cfg_p, local_rank)
    if args.split:
        for hh in (h, hp):
            sharding.init_split_comm(hh, device="cuda")

    def step_loop(hh=h):       # the resident loop: steps 2-7 of ICP::compute on filtered clouds
        hh.set_reference(d_ref, d_nrm)
        return hh.align(d_rd, T_init)

    def step_compute():        # the whole ICP::compute on raw clouds resident in HBM (both filters included)
        return h.compute(d_raw_rd, d_raw_ref, T_init, 1.0, 10, 1.0, seed=0)

    step = step_loop if args.split else step_compute
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    iters = 0
    filt_ms = 0.0
    T = None
    step_s = []
    n_converged = 0
    for _ in range(args.steps):
        ts = time.perf_counter()
        T, st = step()
        step_s.append(time.perf_counter() - ts)
        iters += st.iterations
        filt_ms += st.t_reserved[0]
        n_converged += int(st.converged)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # ---- the resident loop alone (value_loop), then the profiled steps for the rooflines
    loop_steps = max(1, min(args.steps, 10))
    value_loop = None
    align_ms, loop_iters = 0.0, 0
    if not args.split and not args.no_compute_e2e:
        step_loop(); step_loop()
        torch.cuda.synchronize()
        per_step = []
        tl = time.perf_counter()
        for _ in range(loop_steps):
            ts = time.perf_counter()
            Tl, stl = step_loop()          # (align returns once the loop state says done: at most one queued launch that exits at once is behind it)
            per_step.append(time.perf_counter() - ts)
            align_ms += stl.t_total_ms
            loop_iters += stl.iterations
        torch.cuda.synchronize()
        tl = time.perf_counter() - tl
        value_loop = {"value": loop_steps / tl, "unit": "scans/s", "ms_per_scan": tl / loop_steps * 1e3,
                      "ms_per_scan_median": float(np.median(per_step)) * 1e3, "ms_per_scan_max": float(np.max(per_step)) * 1e3,
                      "ms_per_icp_iteration": align_ms / max(loop_iters, 1), "iterations": loop_iters / loop_steps,
                      "workload": "set_reference + align on the FILTERED clouds resident in HBM (steps 2-7 of ICP::compute: the north-star kernels)"}
    prof_steps = max(1, min(args.steps, 3))
    sel_ms = ne_ms = knn_ms = knn_main_ms = knn_fb_ms = comm_ms = 0.0
    knn_launches = strag = comm_calls = 0
    step_loop(hp)
    for _ in range(prof_steps):
        Tp, stp = step_loop(hp)
        knn_ms += stp.t_knn_ms
        knn_main_ms += stp.t_knn_main_ms
        knn_fb_ms += stp.t_knn_fallback_ms
        knn_launches += stp.knn_launches
        strag += stp.stragglers
        sel_ms += stp.t_select_ms
        ne_ms += stp.t_ne_ms
        comm_ms += stp.t_comm_ms
        comm_calls += stp.comm_calls
    # the last profiled step's searches one by one (search kernel + hand-over pass, us): the first launches and the settled
    # ones are different kernels with different costs, the average alone hides which is which
    knn_per_iter = [round(float(t["knn_main_us"] + t["knn_fallback_us"]), 1) for t in hp.trace()]
    loop_equals_compute = bool(np.array_equal(Tp, T))   # (chain F keeps every point: the loop on filtered clouds is the same alignment)

    # ---- the same compute handed HOST buffers (H2D + D2H inclusive), chains F and P, pageable and pinned
    value_e2e = None
    variants = None
    if not args.split and not args.no_compute_e2e and world == 1:   # (the PCIe-inclusive side figures: one rank only)
        variants = {}
        for name, prob, ratio in (("P_yaml_chain", 0.5, 0.5),):
            ts = []
            for rep in range(4):
                tc0 = time.perf_counter()
                Te, ste = h.compute(d_raw_rd, d_raw_ref, T_init, prob, 10, ratio, seed=0)
                ts.append((time.perf_counter() - tc0) * 1e3)
            variants[name] = {"ms_per_compute": float(np.median(ts[1:])), "scans_per_s": 1e3 / float(np.median(ts[1:])),
                              "filters_and_grid_ms": ste.t_reserved[0], "iterations": ste.iterations,
                              "n_reference_after_filter": int(h.info().n_reference),
                              "trans_err_m": synth.pose_error(Te.astype(np.float64), T_true)[0]}
        value_e2e = {"workload": "lsgpu_icp_compute on HOST buffers (raw 1M-point clouds): H2D + reference filter + grid + "
                                 "reading filter + loop + D2H", "unit": "scans/s"}
        p_ref, p_rd = torch.from_numpy(raw_ref).pin_memory(), torch.from_numpy(raw_rd).pin_memory()
        for chain, prob, ratio in (("F_full_density", 1.0, 1.0), ("P_yaml_chain", 0.5, 0.5)):
            for mem, (a_rd, a_ref) in (("pageable", (raw_rd, raw_ref)), ("pinned", (p_rd.numpy(), p_ref.numpy()))):
                ts = []
                for rep in range(5):
                    tc0 = time.perf_counter()
                    Te, ste = h.compute(a_rd, a_ref, T_init, prob, 10, ratio, seed=0)
                    ts.append((time.perf_counter() - tc0) * 1e3)
                ms = float(np.median(ts[1:]))
                value_e2e[f"{chain}_{mem}"] = {"ms_per_scan": ms, "scans_per_s": 1e3 / ms, "iterations": ste.iterations}
        value_e2e["value"] = value_e2e["F_full_density_pageable"]["scans_per_s"]
        # what the link itself takes for one raw cloud (16.7 MB), from either kind of host memory: the pinned path's copies
        # are true DMA from the caller's buffer, the pageable path's go through the runtime's own staging buffers in chunks
        # on the calling / helper thread -- on this box the latter is no slower, and the compute overlaps more of it
        # (the reading's copy runs on a helper thread beside the reference filter either way)
        try:
            d_tmp = torch.empty_like(d_raw_ref)
            pc = {}
            for mem, src_t in (("pageable", torch.from_numpy(raw_ref)), ("pinned", p_ref)):
                ts = []
                for rep in range(5):
                    torch.cuda.synchronize(); tc0 = time.perf_counter()
                    d_tmp.copy_(src_t, non_blocking=False); torch.cuda.synchronize()
                    ts.append((time.perf_counter() - tc0) * 1e3)
                pc["h2d_ms_" + mem] = float(np.median(ts[1:]))
                pc["h2d_GBps_" + mem] = raw_ref.nbytes / (pc["h2d_ms_" + mem] * 1e-3) / 1e9
            pc["bytes"] = int(raw_ref.nbytes)
            value_e2e["host_to_device_copy_of_one_raw_cloud"] = pc
        except Exception as e:   # (a side figure: never fails the bench)
            value_e2e["host_to_device_copy_of_one_raw_cloud"] = {"error": str(e)}
        h.set_reference(d_ref, d_nrm)

    info = h.info()
    ncell = int(info.cells[0])
    # algorithmic bytes of one kNN launch: 24 Nq + 16 Nr + 8 Ncell
    b_knn = 24 * nq + 16 * nr + 8 * ncell
    t_knn = knn_ms / max(knn_launches, 1) * 1e-3
    achieved = b_knn / t_knn / 1e9 if t_knn > 0 else 0.0
    # HBM traffic of the kernel: PMC counters cannot be read from inside the process, so the figure comes from the
    # artifact of the separate rocprofv3 --pmc passes (devtools/gpu_check.sh pmc) -- only if it was measured on THESE
    # kernel sources (csrc_sha) and this workload; otherwise null
    traffic, traffic_note = None, "no PMC artifact for these kernel sources (profiles/knn_traffic.json csrc_sha mismatch or absent)"
    tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")
    if os.path.exists(tpath) and not args.split:
        try:
            tj = json.load(open(tpath))
            if tj.get("n_az") == args.n_az and tj.get("csrc_sha") == base["csrc_sha"]:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_note = "profiles/knn_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes on these kernel sources)"
        except Exception:
            traffic = None

    et, er = synth.pose_error(T.astype(np.float64), T_true)
    units = 1 if args.split else world
    out = dict(base, metric="scans_per_sec", value=units * args.steps / elapsed, unit="scans/s",
               ms_per_step=elapsed / args.steps * 1e3, ms_per_step_median_rank0=float(np.median(step_s)) * 1e3,
               ms_per_step_max_rank0=float(np.max(step_s)) * 1e3, icp_iterations_per_scan=iters / args.steps,
               scaling="strong" if args.split else "weak")
    # stopped by the differential checker (1) or by the counter at max_iterations (0): configs[3]'s street does NOT settle
    # within 40 iterations (the CPU oracle agrees at reduced size) -- a step there is 40 capped iterations
    out["registration_converged"] = {"steps_stopped_by_the_differential_checker": n_converged, "of": args.steps,
                                     "max_iterations": int(cfg.max_iterations)}
    if args.split:
        out["ms_per_icp_iteration"] = elapsed / max(iters, 1) * 1e3
        out["value_is"] = "set_reference + align of one pair whose reading is sharded over the ranks (filtered clouds resident in HBM)"
    else:
        out["filters_and_grid_ms_per_step"] = filt_ms / args.steps
        out["ms_per_icp_iteration"] = (elapsed / args.steps * 1e3 - filt_ms / args.steps) / max(iters / args.steps, 1)
        out["value_is"] = ("the whole ICP::compute (lsgpu_icp_compute: reference filter + grid + reading filter + loop) on RAW clouds "
                           "resident in HBM; value_loop = the loop alone on filtered clouds; value_e2e = the same compute from host "
                           "buffers (PCIe inclusive)")
    if args.split:
        # The differential checker at 1e-4 m / 1e-5 rad (configs[1]'s tolerance) does not stop this registration before
        # max_iterations: against an 8-scan map -- eight noisy samples of every surface -- the trimmed point-to-plane steps
        # shrink to ~1e-4 m and stay there for dozens of iterations, whatever the guess (3 cm: 35 iterations at 0.5 M points
        # per scan) and whatever the scene (street or open field; CPU oracle).  With the reference's OWN
        # thresholds (icp_default.yaml:24-27: 1e-3 rad / 1e-2 m) it stops after a handful: the same step once more that way,
        # so that the line also carries a registration that ends by its checker.
        cfg_y = IcpConfig()
        lib().lsgpu_icp_config_yaml(C.byref(cfg_y))
        with icp.IcpHandle(cfg_y, local_rank) as hy:
            sharding.init_split_comm(hy, device="cuda")
            ty, ity, cvy = [], 0, 0
            for rep in range(4):
                barrier()
                tq = time.perf_counter()
                Ty, sty = step_loop(hy)
                barrier()
                ty.append(max_over_ranks(time.perf_counter() - tq) * 1e3)
                if rep:
                    ity += sty.iterations; cvy += int(sty.converged)
            out["yaml_checker"] = {"ms_per_step": float(np.median(ty[1:])), "scans_per_s": 1e3 / float(np.median(ty[1:])),
                                   "iterations": ity / 3.0, "steps_stopped_by_the_differential_checker": cvy, "of": 3,
                                   "thresholds": "icp_default.yaml:24-27 (minDiffRotErr 0.001 rad, minDiffTransErr 0.01 m, smoothLength 4)",
                                   "trans_err_m": synth.pose_error(Ty.astype(np.float64), T_true)[0]}
        # what the exchange costs ("all-reduce us / iteration"): HIP events around every RCCL call of
        # the profiled steps; and, on rank 0, the SAME map, reading and guess through a plain handle (no communicator,
        # the whole reading on this GPU) -- split mode against the plain path on identical inputs, side by side
        out["split_exchange"] = {"allreduce_us_per_iteration": comm_ms / max(knn_launches, 1) * 1e3,
                                 "rccl_calls_per_iteration": comm_calls / max(knn_launches, 1),
                                 "timed_in": "%d profiled steps (HIP events around every collective of the loop)" % prof_steps,
                                 "knn_us_per_launch": knn_ms / max(knn_launches, 1) * 1e3,
                                 "select_us_per_iteration": sel_ms / max(knn_launches, 1) * 1e3,
                                 "ne_update_us_per_iteration": ne_ms / max(knn_launches, 1) * 1e3}
        if rank == 0:
            cfg_q = IcpConfig()
            C.memmove(C.byref(cfg_q), C.byref(cfg_p), C.sizeof(cfg_p))
            d_whole = torch.from_numpy(rd_whole).cuda()
            with icp.IcpHandle(cfg_q, local_rank) as hq:
                ts, kn, se, ne, its, kl = [], 0.0, 0.0, 0.0, 0, 0
                for rep in range(prof_steps + 1):
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    hq.set_reference(d_ref, d_nrm)
                    Tq, stq = hq.align(d_whole, T_init)
                    ts.append((time.perf_counter() - tq) * 1e3)
                    if rep:
                        kn += stq.t_knn_ms; se += stq.t_select_ms; ne += stq.t_ne_ms; its += stq.iterations; kl += stq.knn_launches
            out["split_exchange"]["plain_same_inputs"] = {
                "ms_per_step": float(np.median(ts[1:])), "iterations": its / prof_steps, "knn_us_per_launch": kn / max(kl, 1) * 1e3,
                "select_us_per_iteration": se / max(kl, 1) * 1e3, "ne_update_us_per_iteration": ne / max(kl, 1) * 1e3,
                "transform_equals_split": bool(np.allclose(Tq, T, atol=1e-6)),
                "what": "set_reference + align of the whole reading on one GPU without a communicator (profiled handle: events cost a few per cent)"}
    out["config"] = {"workload": workload, "n_reading": nq, "n_reference": nr, "pairs_per_gpu_per_step": 1,
                     "sharding": ("one scan pair per step, reading sharded over ranks, RCCL all-reduce of the select tables + 29 f64 "
                                  "per iteration" if args.split else "one scan pair per rank, no collective")}
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                       "kernel": "k_knn_cone (direction-indexed search, iterations >= 2) / k_knn_tile + k_knn_fallback (voxel-grid search, iterations 0-1) -- exact 1-NN correspondence search",
                       "per_iteration_us": knn_per_iter,
                       "algorithmic_bytes_per_launch": b_knn,
                       "avg_launch_us": t_knn * 1e6,
                       "avg_main_us": knn_main_ms / max(knn_launches, 1) * 1e3,
                       "avg_fallback_us": knn_fb_ms / max(knn_launches, 1) * 1e3,
                       "launches": knn_launches, "timed_in": "%d extra profiled steps of the resident loop after the timed region (HIP events on the handle's stream)" % prof_steps,
                       "occupied_cells": ncell,
                       "stragglers_per_launch": strag / max(knn_launches, 1)}
    if traffic and t_knn > 0:   # SURVEY.md 8d "roofline.measured": the literal rocprof HBM GB/s of the launch
        out["roofline"]["measured"] = traffic / t_knn / 1e9
        out["roofline"]["measured_frac"] = traffic / t_knn / 1e9 / HBM_PEAK_GBS
        out["roofline"]["traffic_over_algorithmic"] = traffic / b_knn
    # second yardstick for a kernel that is vector-issue bound, not byte bound: the distance evaluations an ideal
    # per-query search would need (a libnabo kd-tree with bucket size 8 visits 20-30 points per query) x the
    # 7.75 vector operations one evaluation costs (6 for the defined arithmetic, 1.75 for minimum / runner-up / index),
    # against what the chip's vector units could issue during the launch (256 CUs x 4 SIMDs x 16 lanes at 2.4 GHz)
    if t_knn > 0:
        ideal_ops = nq * 25 * 7.75
        peak_lane_ops = 256 * 4 * 16 * 2.4e9
        out["roofline"]["valu_frac"] = ideal_ops / (t_knn * peak_lane_ops)
        out["roofline"]["valu_frac_is"] = ("(Nq x 25 kd-tree-equivalent distance evaluations x 7.75 lane-ops) / (launch time x 256 CUs x 4 SIMDs x "
                                           "16 lanes x 2.4 GHz): the share of the launch's vector issue slots an ideal search would need")
    out["final_error_vs_truth"] = {"trans_m": et, "rot_rad": er}
    out["profiled_loop_transform_equals_timed_compute"] = loop_equals_compute
    # the other two per-iteration kernel groups against the same HBM roofline (B_trim = 4 Nq, B_ne = 52 Nq)
    n_it = max(knn_launches, 1)
    t_sel, t_ne = sel_ms / n_it * 1e-3, ne_ms / n_it * 1e-3
    if t_sel > 0 and t_ne > 0:
        out["roofline_select"] = {"bound": "hbm", "kernel": "k_hist1 + k_hist_refine<2> + k_hist_refine<3> (exact radix select of the trim limit)",
                                  "algorithmic_bytes_per_iteration": 4 * nq, "avg_us": t_sel * 1e6,
                                  "achieved": 4 * nq / t_sel / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": 4 * nq / t_sel / 1e9 / HBM_PEAK_GBS}
        out["roofline_ne"] = {"bound": "hbm", "kernel": "k_normal_eq_loop (point-to-plane normal equations + solve + checkers)",
                              "algorithmic_bytes_per_iteration": 52 * nq, "avg_us": t_ne * 1e6,
                              "achieved": 52 * nq / t_ne / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": 52 * nq / t_ne / 1e9 / HBM_PEAK_GBS}
    if value_loop is not None:
        out["value_loop"] = value_loop
    if value_e2e is not None:
        out["value_e2e"] = value_e2e
        # SURVEY.md 8d words scans/sec as "end-to-end compute calls incl. H2D ... D2H"; the task statement keeps PCIe out of
        # `value` ("inputs already resident in HBM when the timed region starts ... the PCIe-inclusive rate ... is never
        # `value`").  Both figures therefore stand next to each other at the top level: `value` (resident) and this one.
        out["value_host_buffers"] = {"value": value_e2e["value"], "unit": "scans/s",
                                     "is": "SURVEY.md 8d's scans/sec: the same lsgpu_icp_compute handed HOST buffers (pageable), H2D of both raw "
                                           "clouds and D2H of the result inside the timed call; details in value_e2e"}
    if variants is not None:
        out["compute_variants"] = variants

    # ---- CPU baseline: the oracle (port) on this box's host cores, same workload and SAME region as `value`
    # (both filters + kd-tree build + loop), rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.split:
        from oracle import oracle_py as O
        ocfg = O.config_yaml(accum_double=0, min_diff_rot=1e-5, min_diff_trans=1e-4, num_threads=args.cpu_threads,
                             reading_sampling_prob=1.0, surface_normal_knn=10, surface_normal_ratio=1.0)
        tc = time.perf_counter()
        rc, To, sto = O.icp_compute_full(ocfg, raw_rd, raw_ref, synth.colmajor(T_init), seed=0)
        cpu_s = time.perf_counter() - tc
        dt, dr = synth.pose_error(synth.from_colmajor(To), T.astype(np.float64))
        out["cpu_baseline"] = {
            "value": 1.0 / cpu_s, "unit": "scans/s", "cores": args.cpu_threads, "kind": "port",
            "sample": "1 scan pair of the same workload, same region as `value` (both filters + kd-tree build + %d ICP iterations); "
                      "host has %d cores" % (sto.iterations, os.cpu_count()),
            "ms_filters": sto.t_filter_ms, "ms_per_icp_iteration": sto.t_loop_ms / max(sto.iterations, 1),
            "iterations": sto.iterations,
            "gpu_vs_cpu_transform": {"trans_m": dt, "rot_rad": dr},
            "cpu_model": _cpu_model(),
        }
        # second row of SURVEY.md §8d: the same oracle with OpenMP over the queries on all host cores (what a
        # libnabo built with OpenMP does); reported next to the single-thread figure, never as the baseline value
        # The query loop is the only parallel part of the oracle (filters, kd-tree build, select and the 6x6 sums are serial,
        # like libpointmatcher's).  Tried at 64 threads and at every hardware thread of the host; the better one is
        # `all_threads` -- on the 2 x 64-core / 256-thread host of round 6 the 256-thread run was the SLOWER one
        # (0.16 against 0.49 scans/s: the dynamic schedule's 2048-query blocks over SMT siblings), both are kept.
        tried = {}
        for nthr in sorted({min(os.cpu_count() or 1, 64), os.cpu_count() or 1}):
            if nthr <= args.cpu_threads:
                continue
            ocfg.num_threads = nthr
            tc = time.perf_counter()
            rc_mt, _To, sto_mt = O.icp_compute_full(ocfg, raw_rd, raw_ref, synth.colmajor(T_init), seed=0)
            tried[nthr] = {"value": 1.0 / (time.perf_counter() - tc), "unit": "scans/s", "cores": nthr, "iterations": sto_mt.iterations}
        if tried:
            best = max(tried.values(), key=lambda r: r["value"])
            out["cpu_baseline"]["all_threads"] = dict(best, tried={str(k): round(v["value"], 4) for k, v in tried.items()},
                                                      host_threads=os.cpu_count())
    # ---- the reference's own timed region (value_track) and the real call shape of localScanToSubMap: a reference of
    # three scans (compute_variants.F_submap3 / P_submap3)
    if rank == 0 and world == 1 and not args.no_track and not args.no_compute_e2e and not args.split:
        try:
            trk, scans, poses = track_section(args.n_az, args.track_scans, args.cpu_threads)
            out["value_track"] = trk
            # reading = scan 3, reference = scans 2, 1, 0 in the frame of scan 2 (what localScanToSubMap assembles), raw clouds in HBM
            Ta = poses[2]
            parts = []
            for k in (2, 1, 0):
                Trel = np.linalg.inv(Ta) @ poses[k]
                p = scans[k].copy()
                p[:, :3] = (scans[k][:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
                parts.append(p)
            d_sub = torch.from_numpy(np.concatenate(parts)).cuda()
            d_rd3 = torch.from_numpy(scans[3]).cuda()
            T_g = (np.linalg.inv(Ta) @ poses[3]) @ synth.se3(0.1, -0.05, 0.0, yaw=np.deg2rad(0.5))
            torch.cuda.synchronize()
            for name, prob, ratio in (("F_submap3", 1.0, 1.0), ("P_submap3", 0.5, 0.5)):
                ts, fg = [], []
                for rep in range(5):
                    tc0 = time.perf_counter()
                    Te, ste = h.compute(d_rd3, d_sub, T_g, prob, 10, ratio, seed=0)
                    ts.append((time.perf_counter() - tc0) * 1e3)
                    fg.append(ste.t_reserved[0])
                out.setdefault("compute_variants", {})[name] = {
                    "ms_per_compute": float(np.median(ts[1:])), "scans_per_s": 1e3 / float(np.median(ts[1:])),
                    "filters_and_grid_ms": float(np.median(fg[1:])), "iterations": ste.iterations, "n_reference_raw": int(d_sub.shape[0]),
                    "n_reference_after_filter": int(h.info().n_reference),
                    "trans_err_m": synth.pose_error(Te.astype(np.float64), np.linalg.inv(Ta) @ poses[3])[0]}
        except Exception as e:   # (a side figure must not take the headline down with it)
            out["value_track"] = {"error": repr(e)[:400]}
    if rank == 0:
        print(json.dumps(out))
    h.close()
    hp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
