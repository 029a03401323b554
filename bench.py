#!/usr/bin/env python3
"""bench.py -- scans/sec + ms/ICP-iteration of the ICP hot path on 1 M-point Velodyne scans.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one scan pair: steps 2-7 of PointMatcher::ICP::compute as
called at laser_slam/src/laser_track.cpp:496 -- centre the reference + build the voxel grid, then
iterate {transform, exact 1-NN, trimmed weights, point-to-plane 6x6} until the (tightened, 1e-4 m /
1e-5 rad) differential checker stops it.  Inputs (filtered reading, filtered reference + normals) are
resident in HBM when the timed region starts.  Workload = BASELINE.json configs[1]: one synthetic
HDL-64E pair of 64 x 16384 rays, full-density chain (F).  With N > 1 every rank
registers its own pair (embarrassingly parallel, no data-path collective; "weak" scaling).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-az", type=int, default=16384, help="azimuth steps (16384 -> 1 M rays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--no-compute-e2e", action="store_true",
                    help="skip the compute_with_filters section (used for the rocprofv3 run, so that the kernel\n"
                         "averages of the profile cover the benchmark workload only)")
    ap.add_argument("--split", action="store_true",
                    help="BASELINE config 4 layout: ONE scan pair per step, its reading sharded over the ranks, "
                         "RCCL all-reduce of the select histograms + 6x6 sums (strong scaling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the ICP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from laser_slam_amd import synth, icp
    from laser_slam_amd._lib import IcpConfig, lib

    # ---- synthetic workload (host), then resident in HBM
    data_rank = 0 if args.split else rank   # split: every rank works on the SAME pair
    ref, rd, T_true, T_init = synth.scan_pair(args.n_az, noise_seeds=(1 + 2 * data_rank, 2 + 2 * data_rank),
                                              guess_seed=7 + data_rank)
    raw_ref, raw_rd = ref, rd
    with icp.IcpHandle(None, local_rank) as hf:               # chain (F): ratio 1.0, knn 10; the device filter
        d_ref, d_nrm = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)  # == host filter == oracle
    d_ref, d_nrm = d_ref.contiguous().clone(), d_nrm.contiguous().clone()
    rf, rn = d_ref.cpu().numpy(), d_nrm.cpu().numpy()
    if args.split:
        from laser_slam_amd import sharding
        rd = rd[sharding.split_shard(rd.shape[0], rank, world)]
    d_rd = torch.from_numpy(rd).cuda()
    torch.cuda.synchronize()
    nq, nr = rd.shape[0], rf.shape[0]

    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4        # configs[1]: "to 1e-4 m tolerance"
    cfg.profile_kernels = 0
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

    def step(hh=h):
        hh.set_reference(d_ref, d_nrm)
        return hh.align(d_rd, T_init)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    iters = 0
    sel_ms = ne_ms = 0.0
    knn_ms = knn_main_ms = knn_fb_ms = 0.0
    knn_launches = 0
    align_ms = 0.0
    strag = 0
    T = None
    for _ in range(args.steps):
        T, st = step()
        iters += st.iterations
        align_ms += st.t_total_ms
    barrier()
    elapsed = time.perf_counter() - t0
    # kernel timing for the roofline: same workload, same kernels, HIP events on the handle's stream
    prof_steps = max(1, min(args.steps, 3))
    step(hp)
    for _ in range(prof_steps):
        Tp, stp = step(hp)
        knn_ms += stp.t_knn_ms
        knn_main_ms += stp.t_knn_main_ms
        knn_fb_ms += stp.t_knn_fallback_ms
        knn_launches += stp.knn_launches
        strag += stp.stragglers
        sel_ms += stp.t_select_ms
        ne_ms += stp.t_ne_ms
    assert np.array_equal(Tp, T)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the whole ICP::compute (both filters + set_reference + align) on the raw clouds, SURVEY.md §8d
    # variants P (icp_default.yaml chain: prob 0.5 / ratio 0.5) and F (full density): reported, not `value`
    end_to_end = None
    value_e2e = None
    if not args.split and not args.no_compute_e2e:
        d_raw_ref, d_raw_rd = torch.from_numpy(raw_ref).cuda(), torch.from_numpy(raw_rd).cuda()
        torch.cuda.synchronize()
        end_to_end = {}
        for name, prob, ratio in (("P_yaml_chain", 0.5, 0.5), ("F_full_density", 1.0, 1.0)):
            ts = []
            for rep in range(4):
                tc0 = time.perf_counter()
                Te, ste = h.compute(d_raw_rd, d_raw_ref, T_init, prob, 10, ratio, seed=0)
                ts.append((time.perf_counter() - tc0) * 1e3)
            end_to_end[name] = {"ms_per_compute": float(np.median(ts[1:])), "filters_and_grid_ms": ste.t_reserved[0],
                                "iterations": ste.iterations, "n_reference_after_filter": int(h.info().n_reference),
                                "trans_err_m": synth.pose_error(Te.astype(np.float64), T_true)[0]}
        # SURVEY.md §8d's inclusive figure: the whole ICP::compute (lsgpu_icp_compute) handed HOST buffers, i.e. H2D of
        # both raw clouds + both filters + grid + loop + D2H of the transform, from pageable and from pinned memory
        value_e2e = {"workload": "lsgpu_icp_compute on host buffers (raw 1M-point clouds): H2D + reference filter + grid + "
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
        h.set_reference(d_ref, d_nrm)

    info = h.info()
    ncell = int(info.cells[0])
    # algorithmic bytes of one kNN launch: 24 Nq + 16 Nr + 8 Ncell
    b_knn = 24 * nq + 16 * nr + 8 * ncell
    t_knn = knn_ms / max(knn_launches, 1) * 1e-3
    achieved = b_knn / t_knn / 1e9 if t_knn > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("n_az") == args.n_az:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    et, er = synth.pose_error(T.astype(np.float64), T_true)
    out = {
        "metric": "scans_per_sec",
        "value": (1 if args.split else world) * args.steps / elapsed,
        "unit": "scans/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_icp_iteration": align_ms / max(iters, 1),
        "icp_iterations_per_scan": iters / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if args.split else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "configs[1]: single 1M-point HDL-64E scan pair (64x%d rays), full-density "
                               "chain F, point-to-plane ICP, differential checker 1e-4 m / 1e-5 rad" % args.n_az,
                   "n_reading": nq, "n_reference": nr, "pairs_per_gpu_per_step": 1,
                   "sharding": ("one scan pair per step, reading sharded over ranks, RCCL all-reduce of 3x2048 u32 + 29 f64 "
                                "per iteration" if args.split else "one scan pair per rank, no collective")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "k_knn_tile (+ k_knn_fallback / k_knn_rowq where a launch hands queries over: the first three iterations) -- exact 1-NN correspondence search",
                     "algorithmic_bytes_per_launch": b_knn,
                     "avg_launch_us": t_knn * 1e6,
                     "avg_main_us": knn_main_ms / max(knn_launches, 1) * 1e3,
                     "avg_fallback_us": knn_fb_ms / max(knn_launches, 1) * 1e3,
                     "launches": knn_launches, "timed_in": "%d extra profiled steps after the timed region" % prof_steps,
                     "occupied_cells": ncell,
                     "stragglers_per_launch": strag / max(knn_launches, 1)},
        "final_error_vs_truth": {"trans_m": et, "rot_rad": er},
    }
    out["value_is"] = ("resident-input loop: set_reference + align on filtered clouds already in HBM (the north_star kernels); "
                       "value_e2e = the whole ICP::compute from host buffers")
    # the other two per-iteration kernels groups against the same HBM roofline (B_trim = 4 Nq, B_ne = 52 Nq)
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
    if value_e2e is not None:
        out["value_e2e"] = value_e2e
    if end_to_end is not None:
        out["compute_with_filters"] = end_to_end

    # ---- CPU baseline: the oracle (port) on this box's host cores, same workload, rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O
        ocfg = O.config_yaml(accum_double=0, min_diff_rot=1e-5, min_diff_trans=1e-4,
                             num_threads=args.cpu_threads)
        tc = time.perf_counter()
        rc, To, sto, _ = O.icp_compute(ocfg, rd, rf, rn, synth.colmajor(T_init), 0)
        cpu_s = time.perf_counter() - tc
        dt, dr = synth.pose_error(synth.from_colmajor(To), T.astype(np.float64))
        out["cpu_baseline"] = {
            "value": 1.0 / cpu_s, "unit": "scans/s", "cores": args.cpu_threads, "kind": "port",
            "sample": "1 scan pair of the same workload (kd-tree build + %d ICP iterations), "
                      "host has %d cores" % (sto.iterations, os.cpu_count()),
            "ms_per_icp_iteration": sto.t_loop_ms / max(sto.iterations, 1),
            "iterations": sto.iterations,
            "gpu_vs_cpu_transform": {"trans_m": dt, "rot_rad": dr},
        }
        # second row of SURVEY.md §8d: the same oracle with OpenMP over the queries on all host cores (what a
        # libnabo built with OpenMP does); reported next to the single-thread figure, never as the baseline value
        nthr = min(os.cpu_count() or 1, 64)
        if nthr > args.cpu_threads:
            ocfg_mt = O.config_yaml(accum_double=0, min_diff_rot=1e-5, min_diff_trans=1e-4, num_threads=nthr)
            tc = time.perf_counter()
            rc_mt, _To, sto_mt, _ = O.icp_compute(ocfg_mt, rd, rf, rn, synth.colmajor(T_init), 0)
            out["cpu_baseline"]["all_threads"] = {"value": 1.0 / (time.perf_counter() - tc), "unit": "scans/s", "cores": nthr,
                                                  "iterations": sto_mt.iterations}
    if rank == 0:
        print(json.dumps(out))
    h.close()
    hp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
