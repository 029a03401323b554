"""N > 1 path on CPU: world_size-2 gloo run of the scan-pair sharding (no GPU needed).  The per-pair
engine is injected; here it is the oracle, on the GPU box bench.py injects the HIP path."""
import os
import socket
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
