"""N > 1 path on CPU: world_size-2 gloo run of the scan-pair sharding (no GPU needed).  The per-pair
engine is injected; here it is the oracle, on the GPU box bench.py injects the HIP path."""
import os
import socket
import time
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch.distributed as dist
    from laser_slam_amd import sharding, synth
    from oracle import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def align_pair(i):
        ref, rd, T_true, T_init = synth.scan_pair(32, noise_seeds=(1 + 2 * i, 2 + 2 * i), guess_seed=7 + i)
        rf, rn = O.sampling_surface_normal(ref, 10, 1.0, 0)
        rc, T, st, _ = O.icp_compute(O.config_yaml(accum_double=1), rd, rf, rn, synth.colmajor(T_init), 0)
        return (rc, T.tolist(), st.iterations)

    mine = sharding.pairs_of_rank(n_pairs, rank, world)
    t0 = time.perf_counter()
    local = sharding.run_shard(mine, align_pair)
    el = time.perf_counter() - t0 + 0.01 * (rank + 1)
    units, tmax, rate = sharding.aggregate_throughput(len(local), el)
    merged = sharding.gather_results(local)
    if rank == 0:
        q.put((units, tmax, rate, el, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_pairs_of_rank_partition():
    from laser_slam_amd import sharding
    for n, w in ((256, 8), (5, 2), (1, 4), (0, 3)):
        parts = [sharding.pairs_of_rank(n, r, w) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        sharding.pairs_of_rank(4, 2, 2)


def test_two_rank_gloo_run_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from laser_slam_amd import synth
    from oracle import oracle_py as O
    n_pairs, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    units, tmax, rate, el0, merged = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert units == n_pairs and tmax >= el0 and abs(rate - units / tmax) < 1e-9
    assert [i for i, _ in merged] == list(range(n_pairs))
    # identical to running every pair in one process
    for i, (rc, T, iters) in merged:
        ref, rd, T_true, T_init = synth.scan_pair(32, noise_seeds=(1 + 2 * i, 2 + 2 * i), guess_seed=7 + i)
        rf, rn = O.sampling_surface_normal(ref, 10, 1.0, 0)
        rc1, T1, st1, _ = O.icp_compute(O.config_yaml(accum_double=1), rd, rf, rn, synth.colmajor(T_init), 0)
        assert rc == rc1 == 0 and iters == st1.iterations
        assert np.array_equal(np.array(T, np.float32), T1)


def _split_worker(rank, world, port, q):
    """One ICP iteration of the split-scan scheme on CPU: queries sharded, reference replicated, the
    three select histograms and the 29 normal-equation sums all-reduced (gloo) -- the arithmetic the
    HIP path performs with RCCL (include/lsgpu_icp.h, lsgpu_icp_comm_init)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from laser_slam_amd import sharding, synth
    from oracle import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref, rd, T_true, T_init = synth.scan_pair(64)
    rf, rn = O.sampling_surface_normal(ref, 10, 1.0, 0)
    q_all = O.transform_points(synth.colmajor(T_init), rd)
    sl = sharding.split_shard(q_all.shape[0], rank, world)
    qs = q_all[sl]
    ids, d2 = O.KdTree(rf).nn(qs)
    n_total = torch.tensor([qs.shape[0]], dtype=torch.int64)
    dist.all_reduce(n_total)
    k = min(int(np.float32(int(n_total)) * np.float32(0.75)), int(n_total) - 1)
    # exact radix select on float bits, 12 + 11 + 9, histograms all-reduced after every pass
    bits = d2.view(np.uint32).astype(np.int64)
    prefix, shift_hi = 0, 32
    for nb, shift in ((12, 20), (11, 9), (9, 0)):
        sel = bits >> shift_hi == prefix if shift_hi < 32 else np.ones(bits.shape, bool)
        h = torch.from_numpy(np.bincount((bits[sel] >> shift) & ((1 << nb) - 1), minlength=1 << nb).astype(np.int64))
        dist.all_reduce(h)
        c = np.cumsum(h.numpy())
        b = int(np.searchsorted(c, k, side="right"))
        k -= int(c[b - 1]) if b > 0 else 0
        prefix = (prefix << nb) | b
        shift_hi = shift
    limit = np.array([prefix], np.uint32).view(np.float32)[0]
    # The committed exchange of the settled iterations: with the previous iteration's limit known,
    # every shard fills {count below its 12-bit bin, 11-bit histogram inside it, 9-bit histograms of a window of 128
    # second-level bins around it}; ONE all-reduce of the tables, then every rank reads the exact global order statistic.
    k0 = min(int(np.float32(int(n_total)) * np.float32(0.75)), int(n_total) - 1)
    last = int(np.array([np.float32(limit) * np.float32(1.0005)], np.float32).view(np.uint32)[0])   # "last iteration's" limit
    bin1, bin2_last = last >> 20, (last >> 9) & 0x7FF
    inside = bits >> 20 == bin1
    tables = np.zeros(1 + 2048 + 128 * 512, np.int64)
    tables[0] = int((bits >> 20 < bin1).sum())
    tables[1:2049] = np.bincount((bits[inside] >> 9) & 0x7FF, minlength=2048)
    drow = ((bits[inside] >> 9) & 0x7FF) - bin2_last + 64
    inwin = (drow >= 0) & (drow < 128)
    tables[2049:] = np.bincount(drow[inwin] * 512 + (bits[inside][inwin] & 0x1FF), minlength=128 * 512)
    tt = torch.from_numpy(tables)
    dist.all_reduce(tt)
    tables = tt.numpy()
    kk = k0 - int(tables[0])
    assert 0 <= kk < int(tables[1:2049].sum())
    c2 = np.cumsum(tables[1:2049])
    b2 = int(np.searchsorted(c2, kk, side="right"))
    kk -= int(c2[b2 - 1]) if b2 > 0 else 0
    d = b2 - bin2_last + 64
    assert 0 <= d < 128
    c3 = np.cumsum(tables[2049 + d * 512: 2049 + (d + 1) * 512])
    b3 = int(np.searchsorted(c3, kk, side="right"))
    committed_limit = np.array([(bin1 << 20) | (b2 << 9) | b3], np.uint32).view(np.float32)[0]
    assert np.float32(committed_limit) == np.float32(limit), (committed_limit, limit)
    rc, A, b_, x, dT, used = O.point_to_plane(qs, rf, rn, ids, d2, float(limit), 1)
    t = torch.from_numpy(np.concatenate([A.ravel(), b_, [float(used)]]))
    dist.all_reduce(t)
    if rank == 0:
        q.put((float(limit), t.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_split_scan_allreduce_scheme_matches_unsharded():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from laser_slam_amd import sharding, synth
    from oracle import oracle_py as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    limit, t = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref, rd, T_true, T_init = synth.scan_pair(64)
    rf, rn = O.sampling_surface_normal(ref, 10, 1.0, 0)
    qa = O.transform_points(synth.colmajor(T_init), rd)
    ids, d2 = O.KdTree(rf).nn(qa)
    rc, want = O.trim_limit(d2, 0.75)
    assert np.float32(limit) == np.float32(want)                     # global order statistic, exact
    rc, A, b_, x, dT, used = O.point_to_plane(qa, rf, rn, ids, d2, want, 1)
    assert int(t[-1]) == used
    assert np.allclose(t[:36].reshape(6, 6), A, rtol=1e-12) and np.allclose(t[36:42], b_, rtol=1e-11)
    parts = [sharding.split_shard(10, r, 3) for r in range(3)]
    assert [(s.start, s.stop) for s in parts] == [(0, 4), (4, 7), (7, 10)]
    parts = [sharding.split_shard(5, r, 4) for r in range(4)]          # every rank owns at least one point
    assert [(s.start, s.stop) for s in parts] == [(0, 2), (2, 3), (3, 4), (4, 5)]
    with pytest.raises(ValueError):
        sharding.split_shard(3, 0, 4)


def test_bench_relaunches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus N` without a launcher must become N ranks under torch.distributed.run (the driver's own
    command shape); with a launcher environment, or N = 1, it must not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert not bench.needs_relaunch(1, {})
    assert bench.needs_relaunch(8, {})
    assert not bench.needs_relaunch(8, {"WORLD_SIZE": "8", "RANK": "3"})
    argv = bench.relaunch_argv(["--gpus", "8", "--steps", "5", "--warmup", "1", "--split"], 8, 29511)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == "29511"
    script = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1", "--split"]   # the user's arguments, verbatim, after the script
    with pytest.raises(ValueError):
        bench.relaunch_argv([], 1, 1)
    # the launcher's own module parses that command line (no process is started)
    from torch.distributed.run import get_args_parser
    ns = get_args_parser().parse_args(argv[3:])
    assert ns.nproc_per_node == "8" and ns.training_script == os.path.join(ROOT, "bench.py")
    assert ns.training_script_args == ["--gpus", "8", "--steps", "5", "--warmup", "1", "--split"]


def test_bench_dry_run_two_ranks_on_gloo():
    """`bench.py --dry-run` (round-5 verdict, multi-GPU readiness without hardware): the launch / rendezvous / sharding /
    unique-id broadcast / entry handshake / per-iteration exchange / max-over-ranks plumbing of all three modes with two ranks
    on the gloo backend, through the very relaunch path `python bench.py --gpus N` takes on a GPU node; and a rank that dies
    before the first exchange makes its peer give up in bounded time instead of hanging."""
    import json
    import subprocess
    env = dict(os.environ, LSGPU_COMM_TIMEOUT_MS="3000")
    for mode, want in ((["--split"], ("handshake", "unique_id_equal_on_all_ranks", "decisions_equal_on_all_ranks")),
                       (["--batch"], ("pairs_owned_exactly_once",)), ([], ("pair_of_this_rank_differs_from_rank0s",))):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"] + mode,
                           capture_output=True, text=True, timeout=180, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-800:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(line) == 1, p.stdout[-400:]
        out = json.loads(line[0])
        assert out["dry_run"] is True and out["n_gpus"] == 2 and out["backend"] == "gloo"
        for k in want:
            v = out[k]
            assert (v["covers_the_reading"] and v["cannot_start"] == 0) if isinstance(v, dict) else v is True, (k, v)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--split", "--dry-run-dead-rank", "1"],
                       capture_output=True, text=True, timeout=180, env=env, cwd=ROOT)
    assert p.returncode != 0 and time.perf_counter() - t0 < 90
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]     # no result line from a run that lost a rank
