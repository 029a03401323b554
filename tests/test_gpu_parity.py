"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): ids / squared distances / trim limit / weights bit-exact (ties in
distance are equivalent neighbours, libnabo's tie order is implementation defined); final transform
within 1e-4 m / 1e-5 rad of the CPU path on identical filtered clouds.
"""
import os

import numpy as np
import pytest

from laser_slam_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_T = 1e-4    # m
TOL_R = 1e-5    # rad


@pytest.fixture(scope="module")
def icp_mod():
    from laser_slam_amd import icp
    return icp


def _filtered(icp_mod, pair, ratio=1.0, seed=11):
    rf, rn = icp_mod.sampling_surface_normal(pair["ref"], 10, ratio, seed)
    return rf, rn


def _check_nn(oracle, ref_c, q, ids, d2):
    """ids/d2 from the GPU for queries q against centred reference ref_c."""
    kd = oracle.KdTree(ref_c)
    oid, od2 = kd.nn(q)
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32)), \
        f"d2 differs at {np.flatnonzero(d2 != od2)[:5]}"
    neq = np.flatnonzero(ids != oid)
    if neq.size:  # ties: the GPU's neighbour must be at exactly the same distance
        diff = q[neq, :3] - ref_c[ids[neq], :3]
        # same arithmetic as the definition: fma(dz,dz,fma(dy,dy,dx*dx)) in float32
        dx, dy, dz = (diff[:, k].astype(np.float32) for k in range(3))
        dd = np.float32(dx * dx)
        dd = (dy.astype(np.float64) * dy + dd).astype(np.float32)
        dd = (dz.astype(np.float64) * dz + dd).astype(np.float32)
        assert np.array_equal(dd, od2[neq])
    return oid, od2


def test_knn_exact_small(icp_mod, oracle, pair4k):
    rf, rn = _filtered(icp_mod, pair4k)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        mean = h.reference_mean()
        ref_c = rf.copy()
        ref_c[:, :3] -= mean
        T = synth.colmajor(pair4k["T_init"]).copy()
        T[12:15] -= mean
        ids, d2 = h.knn(pair4k["rd"], T)
        q = oracle.transform_points(T, pair4k["rd"])
        _check_nn(oracle, ref_c, q, ids, d2)
        # brute force agrees as well (independent of the kd-tree)
        bi, bd = oracle.brute_nn(ref_c, q[:512])
        assert np.array_equal(bd, d2[:512])


def test_knn_exact_64k(icp_mod, oracle, pair64k):
    rf, rn = _filtered(icp_mod, pair64k)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        mean = h.reference_mean()
        ref_c = rf.copy()
        ref_c[:, :3] -= mean
        T = synth.colmajor(pair64k["T_init"]).copy()
        T[12:15] -= mean
        ids, d2 = h.knn(pair64k["rd"], T)
        q = oracle.transform_points(T, pair64k["rd"])
        _check_nn(oracle, ref_c, q, ids, d2)


def test_knn_far_and_outside_queries(icp_mod, oracle, pair4k):
    """Queries far outside the reference bounding box and in empty space: fallback path, exact."""
    rf, rn = _filtered(icp_mod, pair4k)
    rng = np.random.default_rng(5)
    q = np.ones((3000, 4), np.float32)
    q[:1000, :3] = rng.uniform(-300, 300, (1000, 3))
    q[1000:2000, :3] = rng.uniform(-30, 30, (1000, 3))
    q[2000:, :3] = rng.uniform(-3, 3, (1000, 3))
    q[0, :3] = (1e4, -1e4, 5e3)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        mean = h.reference_mean()
        ref_c = rf.copy()
        ref_c[:, :3] -= mean
        ids, d2 = h.knn(q, None)
        _check_nn(oracle, ref_c, q, ids, d2)


def test_knn_tiny_reference_and_duplicates(icp_mod, oracle):
    ref = np.ones((5, 4), np.float32)
    ref[:, :3] = [[0, 0, 0], [1, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]]
    nrm = np.tile(np.float32([0, 0, 1]), (5, 1))
    q = np.ones((4, 4), np.float32)
    q[:, :3] = [[0.9, 0.1, 0], [0, 0, 0], [5, 5, 5], [-1, -1, -1]]
    with icp_mod.IcpHandle() as h:
        h.set_reference(ref, nrm)
        mean = h.reference_mean()
        ref_c = ref.copy()
        ref_c[:, :3] -= mean
        qc = q.copy()
        qc[:, :3] -= mean
        ids, d2 = h.knn(qc, None)
        _check_nn(oracle, ref_c, qc, ids, d2)
        assert ids[0] in (1, 2)
        # empty query set is fine
        e_ids, e_d2 = h.knn(np.zeros((0, 4), np.float32), None)
        assert e_ids.size == 0


def test_trim_limit_exact(icp_mod, oracle):
    rng = np.random.default_rng(3)
    with icp_mod.IcpHandle() as h:
        for n in (1, 2, 7, 1000, 65537, 300001):
            d2 = (rng.gamma(2.0, 0.01, n) ** 2).astype(np.float32)
            if n > 10:
                d2[::7] = d2[3]  # heavy ties
            for ratio in (0.75, 0.85, 1.0, 0.001):
                rc, want = oracle.trim_limit(d2, ratio)
                got = h.trim_limit(d2, ratio)
                k = min(int(np.float32(n) * np.float32(ratio)), n - 1)
                assert np.float32(got) == np.float32(want) == np.partition(d2, k)[k], (n, ratio)


def test_normal_eq_matches_oracle(icp_mod, oracle, pair4k):
    rf, rn = _filtered(icp_mod, pair4k)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        mean = h.reference_mean()
        ref_c = rf.copy()
        ref_c[:, :3] -= mean
        T = synth.colmajor(pair4k["T_init"]).copy()
        T[12:15] -= mean
        ids, d2 = h.knn(pair4k["rd"], T)
        limit = h.trim_limit(d2, 0.75)
        A, b, used, r2 = h.normal_eq(pair4k["rd"], T, ids, d2, limit)
        p = oracle.transform_points(T, pair4k["rd"])
        rc, Ao, bo, xo, dTo, used_o = oracle.point_to_plane(p, ref_c, rn, ids, d2, limit, 1)
        assert used == used_o == int((d2 <= limit).sum())
        assert np.linalg.norm(A - Ao) / np.linalg.norm(Ao) < 1e-12
        assert np.linalg.norm(b - bo) / np.linalg.norm(bo) < 1e-10
        # float accumulation (libpointmatcher's own) agrees to float round-off
        rc, Af, bf, *_ = oracle.point_to_plane(p, ref_c, rn, ids, d2, limit, 0)
        assert np.linalg.norm(A - Af) / np.linalg.norm(Af) < 1e-5


def test_transform_points_bit_exact(icp_mod, oracle, pair4k):
    T = synth.colmajor(pair4k["T_init"])
    with icp_mod.IcpHandle() as h:
        got = h.transform_points(T, pair4k["rd"])
    want = oracle.transform_points(T, pair4k["rd"])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _run_both(icp_mod, oracle, pair, tight, accum_double):
    rf, rn = _filtered(icp_mod, pair)
    kw = dict(min_diff_rot=1e-5, min_diff_trans=1e-4) if tight else {}
    ocfg = oracle.config_yaml(accum_double=accum_double, **kw)
    rc, To, sto, tro = oracle.icp_compute(ocfg, pair["rd"], rf, rn, synth.colmajor(pair["T_init"]), 40)
    assert rc == 0
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    if tight:
        cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
    with icp_mod.IcpHandle(cfg) as h:
        h.set_reference(rf, rn)
        Tg, stg = h.align(pair["rd"], pair["T_init"])
        trg = h.trace()
    return synth.from_colmajor(To), sto, tro, Tg.astype(np.float64), stg, trg


@pytest.mark.parametrize("tight", [False, True])
def test_align_matches_oracle_trace(icp_mod, oracle, pair4k, tight):
    To, sto, tro, Tg, stg, trg = _run_both(icp_mod, oracle, pair4k, tight, accum_double=1)
    assert stg.iterations == sto.iterations
    assert stg.converged == sto.converged
    for a, b in zip(trg, tro):  # per-iteration: same limit, same weights, same system
        assert np.float32(a["limit"]) == np.float32(b["limit"])
        assert a["n_used"] == b["n_used"]
        assert np.linalg.norm(a["A"] - b["A"]) / np.linalg.norm(b["A"]) < 1e-5
    dt, dr = synth.pose_error(To, Tg)
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    # and ICP did its job: closer to the truth than the initial guess
    et, er = synth.pose_error(Tg, pair4k["T_true"])
    it, ir = synth.pose_error(pair4k["T_init"], pair4k["T_true"])
    assert et < 0.2 * it and er < 0.2 * ir


def test_align_vs_float_accumulating_oracle(icp_mod, oracle, pair64k):
    """Against libpointmatcher's own float accumulation: 1e-4 m / 1e-5 rad at the tightened checker."""
    To, sto, tro, Tg, stg, trg = _run_both(icp_mod, oracle, pair64k, True, accum_double=0)
    dt, dr = synth.pose_error(To, Tg)
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


def test_align_device_resident_inputs(icp_mod, oracle, pair4k):
    import torch
    rf, rn = _filtered(icp_mod, pair4k)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        T_host, _ = h.align(pair4k["rd"], pair4k["T_init"])
        dref = torch.from_numpy(rf).cuda()
        dnrm = torch.from_numpy(rn).cuda()
        drd = torch.from_numpy(pair4k["rd"]).cuda()
        torch.cuda.synchronize()
        h.set_reference(dref, dnrm)
        T_dev, _ = h.align(drd, pair4k["T_init"])
    assert np.array_equal(T_host, T_dev)  # same arithmetic whichever memory the caller used


def test_align_is_deterministic(icp_mod, pair4k):
    rf, rn = _filtered(icp_mod, pair4k)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        a, _ = h.align(pair4k["rd"], pair4k["T_init"])
        b, _ = h.align(pair4k["rd"], pair4k["T_init"])
    assert np.array_equal(a, b)


def test_errors_are_loud(icp_mod):
    from laser_slam_amd._lib import ConvergenceError, LsgpuError
    with icp_mod.IcpHandle() as h:
        with pytest.raises(ConvergenceError):  # no reference yet == empty reference cloud
            h.align(np.ones((10, 4), np.float32), np.eye(4))
        with pytest.raises(LsgpuError):
            h.set_reference(np.zeros((0, 4), np.float32), np.zeros((0, 3), np.float32))
        ref = np.ones((100, 4), np.float32)
        ref[:, :3] = np.random.default_rng(0).normal(size=(100, 3))
        h.set_reference(ref, np.tile(np.float32([0, 0, 1]), (100, 1)))
        with pytest.raises(ConvergenceError):  # empty reading
            h.align(np.zeros((0, 4), np.float32), np.eye(4))
        bad = np.eye(4)
        bad[0, 0] = 1.1                        # not rigid: PointMatcher's TransformationError out of ICP::compute, step 5
        with pytest.raises(LsgpuError):
            h.align(ref[:50].copy(), bad)


def test_icp_class_compute_recovers_pose(icp_mod, pair64k):
    """ICP.compute == icp_.compute(reading, reference, T_init) incl. both filter chains."""
    import os
    icp = icp_mod.ICP()
    yaml_path = os.path.join(os.path.dirname(__file__), "golden", "icp_chain.yaml")
    icp.load_from_yaml(yaml_path)
    icp.chain.seed = 4
    T = icp.compute(pair64k["rd"], pair64k["ref"], pair64k["T_init"])
    et, er = synth.pose_error(T.astype(np.float64), pair64k["T_true"])
    assert et < 0.02 and er < 2e-3, (et, er)
    assert icp.last_stats.iterations >= 4


def test_golden_vectors(icp_mod):
    """Committed fixture (tests/golden/make_golden.py): inputs + oracle outputs of a 4k pair."""
    import os
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "icp_pair4k.npz"))
    with icp_mod.IcpHandle() as h:
        h.set_reference(g["ref"], g["nrm"])
        assert np.array_equal(h.reference_mean(), g["mean"])
        Tm = g["T_init"].copy()
        Tm[12:15] -= g["mean"]
        ids, d2 = h.knn(g["rd"], Tm)
        assert np.array_equal(d2, g["nn_d2"])
        # ids may differ from the fixture only where two reference points are at exactly the same distance
        # (libnabo's tie order is implementation defined): the tie rule of _check_nn
        neq = np.flatnonzero(ids != g["nn_ids"])
        if neq.size:
            ref_c = g["ref"][:, :3] - g["mean"]
            q = h.transform_points(Tm, g["rd"])
            diff = q[neq, :3] - ref_c[ids[neq]]
            dx, dy, dz = (diff[:, k].astype(np.float32) for k in range(3))
            dd = np.float32(dx * dx)
            dd = (dy.astype(np.float64) * dy + dd).astype(np.float32)
            dd = (dz.astype(np.float64) * dz + dd).astype(np.float32)
            assert np.array_equal(dd, g["nn_d2"][neq])
        lim = h.trim_limit(d2, 0.75)
        assert np.float32(lim) == g["limit0"]
        A, b, used, _ = h.normal_eq(g["rd"], Tm, g["nn_ids"], g["nn_d2"], lim)
        assert used == g["used0"]
        assert np.linalg.norm(A - g["A0"]) / np.linalg.norm(g["A0"]) < 1e-12
    for tag, kw in (("yaml", None), ("tight", (1e-5, 1e-4))):
        cfg = IcpConfig()
        lib().lsgpu_icp_config_yaml(C.byref(cfg))
        if kw:
            cfg.min_diff_rot, cfg.min_diff_trans = kw
        with icp_mod.IcpHandle(cfg) as h:
            h.set_reference(g["ref"], g["nrm"])
            T, st = h.align(g["rd"], g["T_init"])
            tr = h.trace()
        k = f"{tag}_acc1"
        assert st.iterations == g[k + "_iters"] and st.converged == g[k + "_converged"]
        assert np.array_equal(np.array([t["limit"] for t in tr], np.float32), g[k + "_limits"])
        assert np.array_equal(np.array([t["n_used"] for t in tr]), g[k + "_used"])
        dt, dr = synth.pose_error(T.astype(np.float64), synth.from_colmajor(g[k + "_T"]))
        assert dt <= TOL_T and dr <= TOL_R
        dt, dr = synth.pose_error(T.astype(np.float64), synth.from_colmajor(g[f"{tag}_acc0_T"]))
        assert dt <= TOL_T and dr <= TOL_R


def test_radius_cap_does_not_change_results(icp_mod, pair64k):
    """The trimmed-radius cap is an exact optimisation: same trace with it disabled."""
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    rf, rn = _filtered(icp_mod, pair64k)
    out = []
    for disable in (0, 1):
        cfg = IcpConfig()
        lib().lsgpu_icp_config_yaml(C.byref(cfg))
        cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
        cfg.reserved[0] = disable
        with icp_mod.IcpHandle(cfg) as h:
            h.set_reference(rf, rn)
            T, st = h.align(pair64k["rd"], pair64k["T_init"])
            out.append((T, st.iterations, [(t["limit"], t["n_used"], t["A"].tobytes()) for t in h.trace()]))
    assert out[0][1] == out[1][1]
    assert out[0][2] == out[1][2]
    assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.timeout(600)
def test_experiment_switches_do_not_change_results():
    """DESIGN.md's claim about the LSGPU_* switches -- they move work between exact paths and never change a result -- on
    a 14-iteration alignment of a 262 k-point pair: front rows / separate row pass / per-lane search for spread tiles,
    committed / predicted / plain select, own radix sort / library sort, the row-wise experiment, 4-wave tile blocks, no
    first-iteration cap.  Every variant runs in its own process (the switches are read once) and must return
    bit-identical transform, per-iteration limit / inlier count / normal matrix / T, distances and filtered reference.
    LSGPU_QUERY_ORDER changes the order in which the 29 double sums of the normal equations are added, hence their last
    bits: for it the search results, the first iteration's limit and inlier count are bit-identical, the transform
    agrees to 1e-6 and the iteration count is the same.  So do LSGPU_NO_FUSED_SELECT and LSGPU_SEL_AMB_CAP: the sum is
    DEFINED with the inliers of the limit's slice added last (lsgpu_common.hip.h), those two switches change that
    definition; predicted / committed / plain select under the same definition stay bit-identical."""
    import json
    import subprocess
    import sys
    # settled launches search the direction index (k_knn_cone) by default; LSGPU_NO_CONE sends them back to the voxel
    # grid (k_knn_tile), whose own switches only act there
    tile = {"LSGPU_NO_CONE": "1"}
    variants = [{}, tile, dict(tile, LSGPU_NO_SPLIT="1"),   # (the settled voxel searches share a tile's candidates out over its idle lanes: k_knn_tile<1, false, true>; without: every lane looks at every candidate)
                dict(tile, LSGPU_NO_FRONT="1"), dict(tile, LSGPU_NO_FRONT="1", LSGPU_NO_ROWQ="1"),
                dict(tile, LSGPU_NO_FRONT="1", LSGPU_NO_ROUTE_ALL="1"), {"LSGPU_NO_COMMIT": "1"}, dict(tile, LSGPU_NO_COMMIT="1"),
                {"LSGPU_NO_PREDICT": "1"}, dict(tile, LSGPU_NO_PREDICT="1"),
                {"LSGPU_CONE_ROWS": "32", "LSGPU_CONE_COLS": "1024"}, {"LSGPU_CONE_ROWS": "512", "LSGPU_CONE_COLS": "32768"},
                {"LSGPU_NO_CONE_PROBE": "1"}, {"LSGPU_CONE_FROM": "1"}, {"LSGPU_CONE_HEAVY_SHARE": "2"}, dict(tile, LSGPU_ROUTE_DENSE="16"), dict(tile, LSGPU_ROUTE_DENSE="1073741824"),
                {"LSGPU_CONE_HEAVY_STEPS": "8", "LSGPU_CONE_HEAVY_SHARE": "0.5"},   # (too dear at first, priced again before every look)
                {"LSGPU_ROUTE_HEAVY_MAX": "-1"}, {"LSGPU_ROUTE_HEAVY_MAX": "0"}, {"LSGPU_ROUTE_HEAVY_MAX": "3", "LSGPU_ROUTE_CHUNKS": "64"},   # heavy tiles of the wide launches: all / none / the first three to the wave-per-query pass
                {"LSGPU_SORT_ITEMS": "4"}, {"LSGPU_NO_SEED_CAP": "1"}, {"LSGPU_NO_LAZY": "1"}, {"LSGPU_NO_SIDE_STREAM": "1"}, {"LSGPU_NO_LOOKAHEAD": "1"},
                dict(tile, LSGPU_FRONT_GUESS="8"), {"LSGPU_SSN_GLOBAL": "1"}, {"LSGPU_SSN_FULL_SORT": "1"}, {"LSGPU_SSN_FULL_SORT": "1", "LSGPU_SSN_GLOBAL": "1"},
                {"LSGPU_SSN_OLD_FINISH": "1"}, {"LSGPU_SSN_ROOT": "2048"}, {"LSGPU_SSN_ROOT": "4096"},   # k_ssn_finish / smaller roots of k_ssn_tree
                {"LSGPU_SSN_SORT_LEVELS": "1"}, {"LSGPU_SSN_SORT_LEVELS": "1", "LSGPU_SSN_OLD_FINISH": "1"},   # a segmented sort per upper level (round 4) / all of round 4's filter
                {"LSGPU_THREE_PASS_SELECT": "1"},   # (the select's third pass instead of the normal equations' set-aside ranking: same limit, same sums)
                {"LSGPU_QUERY_ORDER": "0"},
                # the fused select (round 6): without it (select kernels / window table), and with room for only 3 distances of
                # the limit's slice -- fuller slices void the fused iteration, which is repeated with the select in full
                {"LSGPU_NO_FUSED_SELECT": "1"}, {"LSGPU_SEL_AMB_CAP": "3"}, {"LSGPU_SEL_AMB_CAP": "0"}]
    # the measured-slower variants only exist in the -DLSGPU_EXPERIMENTS build (devtools/build.sh); when that build is
    # around it has to give the same bits as the product, switch by switch
    fenced_so = os.path.join(ROOT, "tests", "liblsgpu_icp_fenced.so")   # built by `make -C laser_slam_amd/csrc` (build())
    assert os.path.exists(fenced_so), "run __graft_entry__.build() first"
    variants.append({"LSGPU_SO": fenced_so})        # release / acquire fences instead of the fence-free hand-off: same bits
    exp_so = os.path.join(ROOT, "devtools", "liblsgpu_exp.so")
    if os.path.exists(exp_so) and os.path.getmtime(exp_so) >= os.path.getmtime(os.path.join(ROOT, "laser_slam_amd", "liblsgpu_icp.so")) - 600:   # (a stale build says nothing)
        variants += [dict(v, LSGPU_SO=exp_so) for v in ({}, dict(tile, LSGPU_KNN_ROWS="1"), dict(tile, LSGPU_TILE_WAVES="4"),
                                                        dict(tile, LSGPU_NO_FRONT="1", LSGPU_SPARSE_LANES="16"),
                                                        {"LSGPU_ROCPRIM_SORT": "1"})]          # (the library sort as a cross-check of lsgpu_sort.hip.h)
    results = []
    for env_add in variants:
        env = dict(os.environ)
        env.update(env_add)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py"), "4096"], env=env,
                           capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SWITCH_RESULT ")]
        assert r.returncode == 0 and line, (env_add, r.stdout[-1500:], r.stderr[-1500:])
        results.append((env_add, json.loads(line[0][len("SWITCH_RESULT "):])))
    base = results[0][1]
    assert base["iterations"] >= 10 and base["committed"] > 0 and base["spread_tiles"] > 0, base
    for env_add, res in results[1:]:
        assert res["iterations"] == base["iterations"] and res["digest_order_free"] == base["digest_order_free"], (env_add, res, base)
        if "LSGPU_QUERY_ORDER" in env_add or "LSGPU_NO_FUSED_SELECT" in env_add or "LSGPU_SEL_AMB_CAP" in env_add:
            assert max(abs(a - b) for a, b in zip(res["T"], base["T"])) < 1e-6, (env_add, res["T"], base["T"])
            if "LSGPU_SEL_AMB_CAP" in env_add:   # (the voided iterations really happened, and were repeated)
                assert res["sel_retries"] > 0, (env_add, res)
        else:
            assert res["digest"] == base["digest"], (env_add, res, base)


@pytest.mark.timeout(300)
def test_direction_index_returns_after_a_price_off():
    """Behaviour, not results (results cannot show it): an alignment whose first price keeps it off the direction index
    prices the index again with the last launch in front of every look at the loop state and goes BACK to it once a look
    finds it cheap -- stats.direction_index_launches must grow after that look.  (Round 4 shipped a defect there: the
    iteration enqueued behind the look consumed the count, the index never came back; csrc/lsgpu_policy.h, whose state
    machine tests/cpp/policy_check.cpp drives on the CPU.)  Run 1 never refuses the index and reports the share of heavy
    lanes the first price found (with LSGPU_CONE_HEAVY_STEPS lowered so that the 262 k-point pair has heavy lanes at
    all); run 2 refuses it at half that share: the balls shrink by more than that within the first group of iterations."""
    import json
    import subprocess
    import sys

    def run(env_add):
        env = dict(os.environ)
        env.update(env_add)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "price_worker.py"), "4096"], env=env, capture_output=True, text=True, timeout=200)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("PRICE_RESULT ")]
        assert r.returncode == 0 and line, (env_add, r.stdout[-1500:], r.stderr[-1500:])
        return json.loads(line[0][len("PRICE_RESULT "):])

    never_off = run({"LSGPU_CONE_HEAVY_STEPS": "8", "LSGPU_CONE_HEAVY_SHARE": "1.9"})
    s0 = never_off["heavy_share"]
    assert never_off["iterations"] >= 12 and never_off["index_launches"] >= never_off["iterations"] - 3, never_off
    assert s0 > 0.02, never_off                                   # (otherwise nothing would be refused below)
    back = run({"LSGPU_CONE_HEAVY_STEPS": "8", "LSGPU_CONE_HEAVY_SHARE": repr(0.5 * s0)})
    assert back["digest"] == never_off["digest"] and back["iterations"] == never_off["iterations"]   # the same alignment, bit for bit
    assert 0 < back["index_launches"] < never_off["index_launches"], (back, never_off)            # refused at first, back later
    assert back["heavy_share"] <= 0.5 * s0, (back, never_off)     # the share the look found when it let the index back in


def test_wall_scan_search_walk_is_bounded():
    """Behaviour, not results (results cannot show it): next to a wall 0.8 m from the sensor (scan 18 of the track drive:
    a 3-scan sub-map of 1.57 M points, thousands of reference points per level-0 cell) no 64-query tile of the voxel-grid
    search may walk or evaluate more than a bounded number of 64-point chunks.  Round 4 shipped without that bound: 64
    lanes of a spread wave each walked 1 700 chunk boxes and held one wave for 550 us (csrc/lsgpu_tuning.h,
    LSGPU_ROUTE_DENSE); since then dense spread waves are handed to the row-per-query pass and a lane's walk goes through
    the 16-chunk group boxes.  Measured with the -DLSGPU_KNN_STATS build (tests/liblsgpu_icp_stats.so: per tile of the LAST
    launch the chunk boxes that survived the tile-level cull and the chunks fetched and evaluated): 8.1 chunks evaluated per
    tile in the mean, 28 at p99, 182 at most; 465 survivors at most.  The bounds below leave a factor of two."""
    import json
    import subprocess
    import sys

    so = os.path.join(ROOT, "tests", "liblsgpu_icp_stats.so")
    assert os.path.exists(so), "tests/liblsgpu_icp_stats.so is missing: run __graft_entry__.build()"

    def run(env_add):
        env = dict(os.environ)
        env.update(env_add)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "walk_worker.py"), "18", "4"], env=env, capture_output=True, text=True, timeout=400)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("WALK_RESULT ")]
        assert r.returncode == 0 and line, (env_add, r.stdout[-1500:], r.stderr[-1500:])
        return json.loads(line[0][len("WALK_RESULT "):])

    for env_add in ({"LSGPU_NO_CONE": "1"}, {}):     # the voxel grid in every launch; the product's own choice of kernels
        w = run(env_add)
        assert w["nearest_return_m"] < 1.0, w                       # (the wall is there)
        assert w["iterations"] == 4 and w["tiles"] >= 0.9 * w["of"], w
        assert w["evals_mean"] <= 16 and w["evals_p99"] <= 64 and w["evals_max"] <= 400, (env_add, w)
        assert w["survivors_mean"] <= 32 and w["survivors_max"] <= 1000, (env_add, w)
        # the wave-per-query / row-per-query pass is for the first wide launches: nothing is handed over once the balls are small
        assert all(n <= 0.08 * w["n_reading"] for n in w["handed_over"]) and w["handed_over"][-1] <= 0.01 * w["n_reading"], (env_add, w)


def test_sort_free_levels_keep_walls_and_lattices():
    """Behaviour beside results: the box tree's sort-free upper levels (csrc/lsgpu_ssn_select.hip.h) must KEEP clouds that
    put thousands of equal coordinates around a median -- a wall square to a frame axis (scans 18-19 of the track drive put
    12 246 points into one bin and a fixed capacity of 2048 candidates sent both filters to the segmented sorts, at twice the
    time), a lattice -- and still give the oracle's bits.  The library reports a hand-over on stderr under LSGPU_GS_DEBUG;
    with LSGPU_SSN_SORT_LEVELS the same clouds go through the segmented sorts and give the same result."""
    import json
    import subprocess
    import sys

    def run(kind, env_add):
        env = dict(os.environ, LSGPU_GS_DEBUG="1")
        env.update(env_add)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "levels_worker.py"), kind], env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("LEVELS_RESULT ")]
        assert r.returncode == 0 and line, (kind, r.stdout[-1500:], r.stderr[-1500:])
        return json.loads(line[0][len("LEVELS_RESULT "):]), r.stderr

    for kind in ("wall", "lattice"):
        res, err = run(kind, {})
        assert res["equal"] and res["kept"] > 0, (kind, res)     # (the lattice's boxes are mostly lines in x, y: few survive the rank test)
        assert "gave up" not in err, (kind, err[-800:])            # the sort-free levels did the whole tree
        res2, _ = run(kind, {"LSGPU_SSN_SORT_LEVELS": "1"})
        assert res2 == res, (kind, res, res2)


def test_full_size_properties(icp_mod):
    """BASELINE configs[1] size (1M-point pair): size-independent properties instead of the oracle.
    (a) kNN distances are self-consistent with the returned ids and no sampled brute-force distance
    beats them; (b) a reference matched against itself returns identity ids and zero distances;
    (c) alignment converges to the known synthetic motion."""
    ref, rd, T_true, T_init = synth.scan_pair(16384)
    rf, rn = icp_mod.sampling_surface_normal(ref, 10, 1.0, 0)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        mean = h.reference_mean()
        ref_c = rf.copy()
        ref_c[:, :3] -= mean
        T = synth.colmajor(T_init).copy()
        T[12:15] -= mean
        ids, d2 = h.knn(rd, T)
        q = h.transform_points(T, rd)
        diff = q[:, :3] - ref_c[ids, :3]
        dd = (diff.astype(np.float64) ** 2).sum(1)
        assert np.allclose(dd, d2, rtol=1e-5, atol=1e-12)
        rng = np.random.default_rng(0)
        pick = rng.choice(rd.shape[0], 256, replace=False)
        D = ((q[pick, None, :3].astype(np.float64) - ref_c[None, ::1, :3]) ** 2).sum(-1)
        assert (D.min(1) >= d2[pick] * (1 - 1e-5)).all()
        ids_self, d2_self = h.knn(ref_c[:200000], None)
        assert (d2_self == 0).all()
        assert (np.abs(ref_c[ids_self, :3] - ref_c[:200000, :3]).max() == 0)
        Tg, st = h.align(rd, T_init)
    et, er = synth.pose_error(Tg.astype(np.float64), T_true)
    assert et < 0.02 and er < 1e-3 and st.iterations < 40


def test_split_scan_world1_equals_plain(icp_mod, pair64k):
    """Split-scan mode with a one-rank RCCL communicator: identical result, exercises every collective."""
    rf, rn = _filtered(icp_mod, pair64k)
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        T0, st0 = h.align(pair64k["rd"], pair64k["T_init"])
        tr0 = [(t["limit"], t["n_used"]) for t in h.trace()]
        h.comm_init(0, 1, icp_mod.comm_unique_id())
        T1, st1 = h.align(pair64k["rd"], pair64k["T_init"])
        tr1 = [(t["limit"], t["n_used"]) for t in h.trace()]
    assert st0.iterations == st1.iterations and tr0 == tr1
    assert np.array_equal(T0, T1)


@pytest.mark.gpu
def test_split_scan_world1_committed_exchange(icp_mod, pair64k):
    """Long alignment (tight checker) in the split-scan mode: once the trim limit is steady the three select passes and
    their three all-reduces are replaced by ONE grouped exchange of the search kernels' tables; a one-rank communicator
    runs every one of those collectives and must reproduce the plain loop bit for bit, committed iterations included."""
    import ctypes as C
    from laser_slam_amd._lib import IcpConfig, lib
    rf, rn = _filtered(icp_mod, pair64k)
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans, cfg.max_iterations = 0.0, 0.0, 40
    with icp_mod.IcpHandle(cfg) as h:
        h.set_reference(rf, rn)
        T0, st0 = h.align(pair64k["rd"], pair64k["T_init"])
        tr0 = [(t["limit"], t["n_used"]) for t in h.trace()]
        h.comm_init(0, 1, icp_mod.comm_unique_id())
        T1, st1 = h.align(pair64k["rd"], pair64k["T_init"])
        tr1 = [(t["limit"], t["n_used"]) for t in h.trace()]
    assert st0.iterations == st1.iterations == 40 and tr0 == tr1
    assert np.array_equal(T0, T1)
    assert st0.committed_select_iterations > 0, "the plain loop never committed its select on a 40-iteration alignment"
    assert st1.committed_select_iterations > 0, "the split-scan loop never used the fused exchange"


def test_submap_vs_scan_matches_oracle(icp_mod, oracle):
    """BASELINE config 4 shape at reduced size: an aggregated 8-scan sub-map (the reference of
    localScanToSubMap with nscan_in_sub_map = 8, laser_track.cpp:474-486) against one scan."""
    scene = synth.Scene(1234)
    poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(9)]
    parts = []
    for i in range(8):
        s = synth.hdl64_scan(scene, poses[i], 1024, 20 + i)
        Trel = np.linalg.inv(poses[7]) @ poses[i]
        p = s.copy()
        p[:, :3] = (s[:, :3].astype(np.float64) @ Trel[:3, :3].T + Trel[:3, 3]).astype(np.float32)
        parts.append(p)
    ref = np.concatenate(parts)
    rd = synth.hdl64_scan(scene, poses[8], 1024, 40)
    T_true = np.linalg.inv(poses[7]) @ poses[8]
    T_init = synth.se3(0.25, -0.1, 0.05, yaw=np.deg2rad(1.2)) @ T_true
    rf, rn = icp_mod.sampling_surface_normal(ref, 10, 1.0, 0)
    assert rf.shape[0] > 500000
    with icp_mod.IcpHandle() as h:
        h.set_reference(rf, rn)
        Tg, stg = h.align(rd, T_init)
        trg = h.trace()
    rc, To, sto, tro = oracle.icp_compute(oracle.config_yaml(accum_double=1, num_threads=8), rd, rf, rn,
                                          synth.colmajor(T_init), 40)
    assert rc == 0 and stg.iterations == sto.iterations
    assert [t["n_used"] for t in trg] == [t["n_used"] for t in tro]
    dt, dr = synth.pose_error(synth.from_colmajor(To), Tg.astype(np.float64))
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


def test_align_batch_matches_sequential(icp_mod, oracle):
    """BASELINE config 3 (many independent pairs, several streams on one GPU): the batch entry point gives,
    for every pair, exactly what {set_reference, align} on a single handle gives, whatever the pool
    size; pairs differ in size, one has an empty reading (rc 1, T_out = T_init), and the first pair is
    checked against the oracle."""
    from laser_slam_amd import synth
    pairs = []
    for i, n_az in enumerate([96, 160, 64, 128, 200, 80, 112]):
        ref, rd, _Tt, Ti = synth.scan_pair(n_az, noise_seeds=(1000 + i, 2000 + i), guess_seed=1000 + i)
        rf, rn = icp_mod.sampling_surface_normal(ref, 10, 1.0, 0)
        pairs.append((rf, rn, rd, Ti))
    pairs.insert(3, (pairs[0][0], pairs[0][1], np.zeros((0, 4), np.float32), pairs[0][3]))
    refs, nrms, rds, Tis = map(list, zip(*pairs))
    seq = []
    with icp_mod.IcpHandle() as h:
        for rf, rn, rd, Ti in pairs:
            if len(rd) == 0:
                seq.append(np.asarray(Ti, np.float32))
                continue
            h.set_reference(rf, rn)
            seq.append(h.align(rd, Ti)[0])
    for pool in (1, 3, 8):
        hs = [icp_mod.IcpHandle() for _ in range(pool)]
        T, st, rc = icp_mod.align_batch(hs, refs, nrms, rds, Tis)
        for h in hs:
            h.close()
        assert list(rc) == [0, 0, 0, 1, 0, 0, 0, 0]
        for i in range(len(pairs)):
            assert np.array_equal(T[i], seq[i]), (pool, i)
        assert st[0].iterations > 0 and st[3].iterations == 0
    rco, To, _sto, _tr = oracle.icp_compute(oracle.config_yaml(accum_double=1), rds[0], refs[0], nrms[0],
                                            synth.colmajor(Tis[0]))
    assert rco == 0
    dt, dr = synth.pose_error(T[0], synth.from_colmajor(To))
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


def test_align_batch_reuses_a_shared_reference(icp_mod):
    """SURVEY.md §8f N1, the reuse that is exact: several readings against ONE filtered reference (loop-closure candidates
    against one sub-map, several robots against one map).  Pairs that name the same reference buffers as the previous
    pair of their handle skip set_reference (stats.reference_reused) and give bit-identical transforms, iteration
    counts and limits to rebuilding the reference for every pair."""
    ref, rd0, _Tt, Ti0 = synth.scan_pair(512)
    rf, rn = icp_mod.sampling_surface_normal(ref, 10, 1.0, 0)
    rds, Tis = [rd0], [Ti0]
    for i in range(1, 6):                                  # five more readings of the same scene, other noise / guesses
        _r, rd, _t, Ti = synth.scan_pair(512, noise_seeds=(1, 50 + i), guess_seed=90 + i)
        rds.append(rd); Tis.append(Ti)
    seq = []
    with icp_mod.IcpHandle() as h:
        for rd, Ti in zip(rds, Tis):
            h.set_reference(rf, rn)                        # rebuilt every time
            T, st = h.align(rd, Ti)
            seq.append((T, st.iterations, st.final_limit, st.final_n_used))
    for pool in (1, 2):
        hs = [icp_mod.IcpHandle() for _ in range(pool)]
        T, st, rc = icp_mod.align_batch(hs, [rf] * 6, [rn] * 6, rds, Tis)
        for h in hs:
            h.close()
        assert list(rc) == [0] * 6
        assert [s.reference_reused for s in st] == [0] * pool + [1] * (6 - pool)
        for i in range(6):
            assert np.array_equal(T[i], seq[i][0]) and (st[i].iterations, st[i].final_limit, st[i].final_n_used) == seq[i][1:], (pool, i)
    # a different buffer with the same content is NOT assumed equal (the library compares pointers, never contents)
    hs = [icp_mod.IcpHandle()]
    T2, st2, _ = icp_mod.align_batch(hs, [rf, rf.copy()], [rn, rn], rds[:2], Tis[:2])
    hs[0].close()
    assert [s.reference_reused for s in st2] == [0, 0] and np.array_equal(T2[1], seq[1][0])


@pytest.mark.timeout(900)
def test_config2_batch_at_size(icp_mod):
    """BASELINE configs[2] at its real pair size on one GPU: 32 independent pairs of 200 k-point scans (64 x 3125 rays)
    through lsgpu_icp_align_batch on a pool of 8 handles == the same pairs one after the other on one handle, bit for
    bit (transform, iteration count, final limit and inlier count); every pair recovers the synthetic motion; and a
    sampled brute-force check of one pair's first correspondence search (no oracle at this size)."""
    import torch
    B, n_az = 32, 3125
    refs, nrms, rds, Tis, truths = [], [], [], [], []
    with icp_mod.IcpHandle() as hf:
        for i in range(B):
            ref, rd, Tt, Ti = synth.scan_pair(n_az, noise_seeds=(1000 + i, 2000 + i), guess_seed=1000 + i)
            rf, rn = hf.filter_reference(torch.from_numpy(ref).cuda(), 10, 1.0, 0)      # device filter == host filter == oracle
            refs.append(rf.contiguous().clone()); nrms.append(rn.contiguous().clone())
            rds.append(torch.from_numpy(rd).cuda()); Tis.append(Ti); truths.append(Tt)
    torch.cuda.synchronize()
    assert 150_000 < rds[0].shape[0] < 210_000
    seq = []
    with icp_mod.IcpHandle() as h:
        for i in range(B):
            h.set_reference(refs[i], nrms[i])
            T, st = h.align(rds[i], Tis[i])
            seq.append((T, st.iterations, st.final_limit, st.final_n_used))
        # sampled brute force on pair 0's search at the initial guess
        h.set_reference(refs[0], nrms[0])
        mean = h.reference_mean()
        Tm = synth.colmajor(Tis[0]).copy(); Tm[12:15] -= mean
        rd0 = rds[0].cpu().numpy()
        ids, d2 = h.knn(rd0, Tm)
        q = h.transform_points(Tm, rd0)
        ref_c = refs[0].cpu().numpy()[:, :3] - mean
        pick = np.random.default_rng(0).choice(rd0.shape[0], 128, replace=False)
        D = ((q[pick, None, :3].astype(np.float64) - ref_c[None, :, :]) ** 2).sum(-1)
        assert (D.min(1) >= d2[pick] * (1 - 1e-5)).all() and np.allclose(D[np.arange(128), ids[pick]], d2[pick], rtol=1e-5, atol=1e-12)
    hs = [icp_mod.IcpHandle() for _ in range(8)]
    T, st, rc = icp_mod.align_batch(hs, refs, nrms, rds, Tis)
    for h in hs:
        h.close()
    assert list(rc) == [0] * B
    errs = []
    for i in range(B):
        assert np.array_equal(T[i], seq[i][0]) and (st[i].iterations, st[i].final_limit, st[i].final_n_used) == seq[i][1:], i
        errs.append(synth.pose_error(T[i].astype(np.float64), truths[i]))
    # plausibility, not parity: the yaml checker (1e-2 m / 1e-3 rad) stops early and a few guesses end in a neighbouring
    # minimum of this street scene; most pairs must recover the motion to the scene's own accuracy (2 cm range noise)
    dts = np.array([e[0] for e in errs])
    assert np.median(dts) < 0.03 and (dts < 0.05).mean() >= 0.8 and dts.max() < 0.5, sorted(dts)[-5:]


# ---- the sampling filters and the whole of ICP::compute on the device

@pytest.mark.parametrize("n_az,knn,ratio,seed", [(64, 10, 0.5, 3), (256, 10, 1.0, 0), (1024, 7, 0.5, 11),
                                                 (64, 32, 0.3, 5)])
def test_device_reference_filter_is_bit_identical(icp_mod, oracle, n_az, knn, ratio, seed):
    """SamplingSurfaceNormal on the GPU == oracle == host filter: same points, same order, same normals."""
    ref = synth.scan_pair(n_az)[0]
    of, on = oracle.sampling_surface_normal(ref, knn, ratio, seed)
    hf, hn = icp_mod.sampling_surface_normal(ref, knn, ratio, seed)
    with icp_mod.IcpHandle() as h:
        gf, gn = h.filter_reference(ref, knn, ratio, seed)
        import torch
        tf, tn = h.filter_reference(torch.from_numpy(ref).cuda(), knn, ratio, seed)
    assert len(of) > 0.2 * ratio * len(ref)
    assert np.array_equal(gf, of) and np.array_equal(gn, on)
    assert np.array_equal(hf, of) and np.array_equal(hn, on)
    assert np.array_equal(tf.cpu().numpy(), of) and np.array_equal(tn.cpu().numpy(), on)


def test_device_reference_filter_edge_cases(icp_mod, oracle):
    from laser_slam_amd._lib import LsgpuError
    rng = np.random.default_rng(5)
    with icp_mod.IcpHandle() as h:
        # fewer points than one box; duplicate coordinates (ties at the split); a degenerate (collinear) cloud
        tiny = np.concatenate([rng.normal(size=(7, 3)), np.ones((7, 1))], 1).astype(np.float32)
        dup = np.repeat(np.concatenate([rng.integers(0, 6, size=(400, 3)), np.ones((400, 1))], 1), 3, 0).astype(np.float32)
        line = np.zeros((500, 4), np.float32); line[:, 0] = np.arange(500); line[:, 3] = 1
        # coordinates on a coarse grid, more points than one workgroup of k_ssn_tree holds: runs of equal coordinates at the
        # global levels AND inside the workgroups (the stable order of ties is what the presorted lists have to reproduce)
        grid = np.ones((40000, 4), np.float32)
        grid[:, :3] = (np.round(rng.normal(size=(40000, 3)) * np.array([40.0, 25.0, 6.0])) / 4).astype(np.float32)
        # ... and one constant axis on top (a sheet): every cut alternates between the two others
        sheet = grid[:20000].copy(); sheet[:, 2] = np.float32(1.5)
        # few distinct values along the two widest axes: thousands of equal coordinates around the upper levels' medians -- more
        # than k_gs_select holds in LDS (kGsCandCap): it selects among them on the global list, same output
        coarse = np.ones((40000, 4), np.float32)
        coarse[:, 0] = 10.0 * rng.integers(0, 8, size=40000); coarse[:, 1] = 8.0 * rng.integers(0, 4, size=40000)
        coarse[:, 2] = rng.normal(size=40000)
        for cloud in (tiny, dup, line, grid, sheet, coarse):
            of, on = oracle.sampling_surface_normal(cloud, 10, 1.0, 2)
            gf, gn = h.filter_reference(cloud, 10, 1.0, 2)
            assert np.array_equal(gf, of) and np.array_equal(gn, on), len(cloud)
        assert len(h.filter_reference(line, 10, 1.0, 2)[0]) == 0      # every box is rank deficient: all dropped
        assert len(h.filter_reference(np.zeros((0, 4), np.float32), 10, 1.0, 2)[0]) == 0
        with pytest.raises(LsgpuError):
            h.filter_reference(tiny, 2, 1.0, 0)


def test_device_reading_filter_and_stream_continuation(icp_mod, oracle):
    """RandomSampling on the GPU keeps the points the oracle keeps; seed -1 continues the stream the way
    consecutive rand() calls do (reference filter first, then reading filter: ICP::compute's order)."""
    ref, rd = synth.scan_pair(256)[:2]
    of, on = oracle.sampling_surface_normal(ref, 10, 0.5, 9)
    ok = oracle.random_sampling(len(rd), 0.5, -1)
    with icp_mod.IcpHandle() as h:
        gf, gn = h.filter_reference(ref, 10, 0.5, 9)
        gr = h.filter_reading(rd, 0.5, -1)
    assert np.array_equal(gf, of) and np.array_equal(gr, rd[ok])
    assert 0.4 * len(rd) < len(gr) < 0.6 * len(rd)


def test_device_compute_matches_oracle_full_chain(icp_mod, oracle, pair64k):
    """lsgpu_icp_compute (filters + ICP on the device) vs the oracle's whole ICP::compute with the yaml
    chain: identical filtered clouds, same iteration count, transform within tolerance."""
    rc, To, sto = oracle.icp_compute_full(oracle.config_yaml(accum_double=1), pair64k["rd"], pair64k["ref"],
                                          synth.colmajor(pair64k["T_init"]), seed=4)
    assert rc == 0
    with icp_mod.IcpHandle() as h:
        T, st = h.compute(pair64k["rd"], pair64k["ref"], pair64k["T_init"], 0.5, 10, 0.5, seed=4)
        info = h.info()
    assert st.iterations == sto.iterations and st.final_n_used == sto.final_n_used
    assert st.final_limit == sto.final_limit or abs(st.final_limit - sto.final_limit) < 1e-6 * sto.final_limit
    assert 0.3 * len(pair64k["ref"]) < info.n_reference < 0.6 * len(pair64k["ref"])
    dt, dr = synth.pose_error(T, synth.from_colmajor(To))
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    dt, dr = synth.pose_error(T, pair64k["T_true"])
    assert dt < 0.05 and dr < 0.005


def test_compute_clouds_assembles_the_submap_on_the_device(icp_mod):
    """lsgpu_icp_compute_clouds (scans resident in HBM, sub-map = concat(T_i * scan_i) built on the device)
    must equal lsgpu_icp_compute on the sub-map assembled by the caller with lsgpu_transform_points."""
    from laser_slam_amd._lib import LsgpuError
    scene = synth.Scene(1234)
    poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(4)]
    scans = [synth.hdl64_scan(scene, T, 256, 70 + i) for i, T in enumerate(poses)]
    # reference frame = scan 2 (the scan before the reading), plus scans 1 and 0 moved into it
    rel = [np.eye(4)] + [np.linalg.inv(poses[2]) @ poses[j] for j in (1, 0)]   # (an exact identity is copied, not applied)
    T_init = (np.linalg.inv(poses[2]) @ poses[3] @ synth.se3(0.2, -0.1, 0.0, yaw=np.deg2rad(1.0))).astype(np.float32)
    with icp_mod.IcpHandle() as h:
        for s, sc in enumerate(scans):
            h.cloud_upload(s, sc)
        assert h.cloud_size(3) == len(scans[3]) and h.cloud_size(9) == -1
        T_dev, st_dev = h.compute_clouds(3, [2, 1, 0], rel, T_init, 0.5, 10, 0.5, seed=6)
        sub = np.concatenate([scans[2], h.transform_points(rel[1], scans[1]), h.transform_points(rel[2], scans[0])])
        T_host, st_host = h.compute(scans[3], sub, T_init, 0.5, 10, 0.5, seed=6)
        assert np.array_equal(T_dev, T_host) and st_dev.iterations == st_host.iterations
        # identity transforms may be passed as NULL
        T_a, _ = h.compute_clouds(3, [2], None, T_init, 0.5, 10, 0.5, seed=6)
        T_b, _ = h.compute(scans[3], scans[2], T_init, 0.5, 10, 0.5, seed=6)
        assert np.array_equal(T_a, T_b)
        want = np.linalg.inv(poses[2]) @ poses[3]
        et, er = synth.pose_error(T_dev, want)
        assert et < 0.03 and er < 3e-3
        # upload + compute in one call (the new scan crosses PCIe while the sub-map is filtered): the same transform, the
        # same draws, and the slot holds the scan afterwards -- also when the registration itself is refused
        h.cloud_release(3)
        T_up, st_up = h.compute_clouds_upload(3, scans[3], [2, 1, 0], rel, T_init, 0.5, 10, 0.5, seed=6)
        assert np.array_equal(T_up, T_dev) and st_up.iterations == st_dev.iterations and h.cloud_size(3) == len(scans[3])
        T_again, _ = h.compute_clouds(3, [2, 1, 0], rel, T_init, 0.5, 10, 0.5, seed=6)     # (from the slot the fused call filled)
        assert np.array_equal(T_again, T_dev)
        h.cloud_release(3)
        with pytest.raises(LsgpuError):
            bad_guess = T_init.copy(); bad_guess[0, 0] = 1.5
            h.compute_clouds_upload(3, scans[3], [2], None, bad_guess)
        assert h.cloud_size(3) == len(scans[3])
        T_c, _ = h.compute_clouds(3, [2], None, T_init, 0.5, 10, 0.5, seed=6)
        assert np.array_equal(T_c, T_a)
        T_d, _ = h.compute_clouds_upload(2, scans[2], [2], None, T_init, 0.5, 10, 0.5, seed=6)   # reading slot among the references: the two calls in a row
        T_e, _ = h.compute(scans[2], scans[2], T_init, 0.5, 10, 0.5, seed=6)
        assert np.array_equal(T_d, T_e)
        h.cloud_release(1)
        with pytest.raises(LsgpuError):
            h.compute_clouds(3, [2, 1], None, T_init)
        with pytest.raises(LsgpuError):
            bad = np.eye(4); bad[0, 0] = 2.0
            h.compute_clouds(3, [2], [bad], T_init)


def test_align_degenerate_and_badly_initialised_cases_match_oracle(icp_mod, oracle, pair4k):
    """(1) a single plane: the 6x6 system is singular -> ConvergenceError on both sides;
    (2) a start 2 m / 10 deg off: whatever ICP does with it, the GPU does the same as the oracle."""
    from laser_slam_amd._lib import ConvergenceError
    rng = np.random.default_rng(8)
    plane = np.ones((3000, 4), np.float32)
    plane[:, :2] = rng.uniform(-5, 5, (3000, 2))
    plane[:, 2] = 0.0
    nrm = np.tile(np.float32([0, 0, 1]), (3000, 1))
    rd = plane[::2].copy()
    rd[:, 2] += 0.05
    rc, To, sto, _ = oracle.icp_compute(oracle.config_yaml(accum_double=1), rd, plane, nrm, synth.colmajor(np.eye(4)), 0)
    assert rc != 0
    with icp_mod.IcpHandle() as h:
        h.set_reference(plane, nrm)
        with pytest.raises(ConvergenceError):
            h.align(rd, np.eye(4))
        # (2)
        rf, rn = _filtered(icp_mod, pair4k)
        T_bad = pair4k["T_true"] @ synth.se3(1.5, -1.2, 0.3, yaw=np.deg2rad(10.0))
        rc, To, sto, _ = oracle.icp_compute(oracle.config_yaml(accum_double=1), pair4k["rd"], rf, rn,
                                            synth.colmajor(T_bad), 0)
        h.set_reference(rf, rn)
        if rc == 0:
            Tg, stg = h.align(pair4k["rd"], T_bad)
            assert stg.iterations == sto.iterations
            dt, dr = synth.pose_error(Tg, synth.from_colmajor(To))
            assert dt <= 10 * TOL_T and dr <= 10 * TOL_R, (dt, dr)   # (40 iterations of float round-off apart)
        else:
            with pytest.raises(ConvergenceError):
                h.align(pair4k["rd"], T_bad)


def test_direction_index_on_clouds_it_is_not_made_for(icp_mod, oracle):
    """k_knn_cone (settled launches) must stay exact where its index is of little use: (1) a uniform cube around the
    origin -- queries next to the origin, cones that reach the polar axis, no ring structure: lanes fall back to the voxel
    grid one by one, and the host gives the index up for the align once they are not rare; (2) a cloud whose points
    all share one elevation (a single row); (3) a reading that wraps around azimuth 0.  Per iteration the trim limit
    (an order statistic of ALL distances) and the inlier count equal the oracle's bit for bit, the system to 1e-9."""
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    rng = np.random.default_rng(21)
    cases = []
    cube = np.ones((30000, 4), np.float32)
    cube[:, :3] = rng.uniform(-4, 4, (30000, 3))
    cube[:, 2] = (0.3 * np.sin(cube[:, 0]) + 0.2 * cube[:, 1] + rng.normal(0, 0.6, 30000)).astype(np.float32)   # a thick wavy sheet through the origin
    cases.append(("cube", cube, synth.se3(0.05, -0.04, 0.03, yaw=np.deg2rad(1.0), pitch=np.deg2rad(0.5))))
    az = rng.uniform(0, 2 * np.pi, 20000); r = rng.uniform(3, 30, 20000)
    disc = np.ones((20000, 4), np.float32)
    disc[:, 0] = r * np.cos(az); disc[:, 1] = r * np.sin(az); disc[:, 2] = (-0.1 * r).astype(np.float32)                # a cone z = -0.1 rho: ONE elevation
    disc[:, 2] += (0.05 * np.sin(3 * az) * r / 30).astype(np.float32)
    cases.append(("one elevation", disc, synth.se3(0.1, 0.05, 0.02, yaw=np.deg2rad(2.0))))
    ring = synth.scan_pair(512)[0]
    cases.append(("lidar, large yaw", ring, synth.se3(0.2, 0.1, 0.0, yaw=np.deg2rad(3.0))))
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
    for name, cloud, T_off in cases:
        rf, rn = icp_mod.sampling_surface_normal(cloud, 10, 1.0, 3)
        rd = cloud[::2].copy()
        rd[:, :3] = (rd[:, :3].astype(np.float64) @ T_off[:3, :3].T + T_off[:3, 3] + rng.normal(0, 0.005, (rd.shape[0], 3))).astype(np.float32)
        T_init = np.linalg.inv(T_off) @ synth.se3(0.03, 0.02, -0.01, yaw=np.deg2rad(0.4))
        ocfg = oracle.config_yaml(accum_double=1, min_diff_rot=1e-5, min_diff_trans=1e-4)
        rc, To, sto, tro = oracle.icp_compute(ocfg, rd, rf, rn, synth.colmajor(T_init), 40)
        with icp_mod.IcpHandle(cfg) as h:
            h.set_reference(rf, rn)
            if rc != 0:
                from laser_slam_amd._lib import ConvergenceError
                with pytest.raises(ConvergenceError):
                    h.align(rd, T_init)
                continue
            Tg, stg = h.align(rd, T_init)
            trg = h.trace()
        assert stg.iterations == sto.iterations and sto.iterations > 4, (name, stg.iterations, sto.iterations)
        for k, (a, b) in enumerate(zip(trg, tro)):
            assert np.float32(a["limit"]) == np.float32(b["limit"]), (name, k)
            assert a["n_used"] == b["n_used"], (name, k)
            assert np.linalg.norm(a["A"] - b["A"]) / np.linalg.norm(b["A"]) < 1e-9, (name, k)
        dt, dr = synth.pose_error(synth.from_colmajor(To), Tg.astype(np.float64))
        assert dt <= TOL_T and dr <= TOL_R, (name, dt, dr)


def test_map_maintenance_filters_match_oracle(icp_mod, oracle):
    """the worker's local-map filters on the device -- cylindrical crop (order preserved) and
    pcl::VoxelGrid centroids (ascending voxel index) -- bit-identical with the oracle's restatement."""
    import torch
    from laser_slam_amd._lib import LsgpuError
    rng = np.random.default_rng(11)
    cloud = synth.scan_pair(512)[0]
    rnd = np.ones((20000, 4), np.float32)
    rnd[:, :3] = rng.uniform(-6, 6, (20000, 3))
    with icp_mod.IcpHandle() as h:
        for pts in (cloud, rnd):
            for inside in (False, True):
                want = oracle.cylinder_filter(pts, [1.0, -2.0, 0.5], 7.5, 3.0, inside)
                got = h.filter_cylinder(pts, [1.0, -2.0, 0.5], 7.5, 3.0, inside)
                assert np.array_equal(got, want) and 0 < len(got) < len(pts)
            n_in = len(h.filter_cylinder(pts, [1.0, -2.0, 0.5], 7.5, 3.0, False))
            n_out = len(h.filter_cylinder(pts, [1.0, -2.0, 0.5], 7.5, 3.0, True))
            assert n_in + n_out >= len(pts)   # (points exactly on the boundary are in both)
            for leaf, minpts in ((0.1, 1), (0.25, 2), ((0.5, 0.2, 1.0), 3)):
                lf = np.broadcast_to(np.asarray(leaf, np.float32), (3,))
                want = oracle.voxel_grid(pts, lf, minpts)
                got = h.filter_voxel_grid(pts, leaf, minpts)
                assert got.shape == want.shape and np.array_equal(got, want), (leaf, minpts)
                assert 0 < len(got) < len(pts)
        # device in, device out
        d = torch.from_numpy(cloud).cuda()
        assert np.array_equal(h.filter_voxel_grid(d, 0.1, 1).cpu().numpy(), oracle.voxel_grid(cloud, [0.1] * 3, 1))
        assert np.array_equal(h.filter_cylinder(d, [0, 0, 0], 10.0, 40.0).cpu().numpy(),
                              oracle.cylinder_filter(cloud, [0, 0, 0], 10.0, 40.0, False))
        # empty input, overflow
        assert len(h.filter_voxel_grid(np.zeros((0, 4), np.float32), 0.1)) == 0
        with pytest.raises(LsgpuError):
            h.filter_voxel_grid(cloud, 1e-4, 1)
        with pytest.raises(OverflowError):
            oracle.voxel_grid(cloud, [1e-4] * 3, 1)


def test_device_filters_reproduce_golden_vectors(icp_mod):
    """Committed fixture tests/golden/filters_4k.npz (oracle outputs): the DEVICE filters reproduce it bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "filters_4k.npz"))
    scan = g["scan"]
    with icp_mod.IcpHandle() as h:
        xyz, nrm = h.filter_reference(scan, 10, 0.5, 5)
        kept = h.filter_reading(scan, 0.5, -1)
        assert np.array_equal(xyz, g["ssn_xyz"]) and np.array_equal(nrm, g["ssn_nrm"])
        assert np.array_equal(kept, scan[g["keep_after_ssn"]])
        xyz, nrm = h.filter_reference(scan, 7, 1.0, 0)
        assert np.array_equal(xyz, g["ssn_full_xyz"]) and np.array_equal(nrm, g["ssn_full_nrm"])
        assert np.array_equal(h.filter_voxel_grid(scan, 0.5, 1), g["voxel_0p5"])
        assert np.array_equal(h.filter_voxel_grid(scan, 1.0, 3), g["voxel_1p0_min3"])
        assert np.array_equal(h.filter_cylinder(scan, [0.5, -0.5, 0.0], 10.0, 40.0, False), g["cyl_in"])
        assert np.array_equal(h.filter_cylinder(scan, [0.5, -0.5, 0.0], 10.0, 40.0, True), g["cyl_out"])


@pytest.mark.timeout(120)
def test_align_runs_to_the_iteration_cap_like_the_oracle(icp_mod, oracle, pair64k):
    """No differential stop (thresholds 0): the loop must end on CounterTransformationChecker after exactly
    max_iterations, however many launches the host had to enqueue around repeated iterations (a predicted select
    that misses repeats its iteration and voids the launches queued behind it)."""
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    rf, rn = _filtered(icp_mod, pair64k)
    ocfg = oracle.config_yaml(accum_double=1, min_diff_rot=0.0, min_diff_trans=0.0, max_iterations=25)
    rc, To, sto, _ = oracle.icp_compute(ocfg, pair64k["rd"], rf, rn, synth.colmajor(pair64k["T_init"]), 0)
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans, cfg.max_iterations = 0.0, 0.0, 25
    with icp_mod.IcpHandle(cfg) as h:
        h.set_reference(rf, rn)
        if rc != 0:  # the counter checker ends in "no convergence" upstream (it throws once the cap is exceeded)
            from laser_slam_amd._lib import ConvergenceError
            with pytest.raises(ConvergenceError):
                h.align(pair64k["rd"], pair64k["T_init"])
        else:
            Tg, stg = h.align(pair64k["rd"], pair64k["T_init"])
            assert stg.iterations == sto.iterations == 25
            dt, dr = synth.pose_error(Tg, synth.from_colmajor(To))
            assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


@pytest.mark.gpu
def test_split_scan_two_ranks_equals_unsplit():
    """BASELINE configs[3] with a real exchange: two ranks (one per GPU, RCCL over xGMI), reading sharded, reference
    replicated; every rank must reproduce the unsplit alignment: integer results of the first iteration bit for bit, the
    final transform to 1e-6 (RCCL adds the 29 double sums in its own order), same iteration count; the worker also
    reports whether the whole trace came out bitwise equal and how many iterations used the fused (committed) exchange.
    Skips on a box with fewer than 2 GPUs."""
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "split_worker.py"), "1024"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SPLIT_RESULT" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_config3_full_size_submap_vs_scan(icp_mod):
    """BASELINE configs[3] at full size on one GPU: an 8-scan aggregated local map (8 x 1 M rays -> 8.4 M points) against
    a 1 M-point scan.  The oracle needs minutes at this size, so: (a) kNN self-consistency + sampled brute force,
    (b) the alignment recovers the synthetic motion, (c) the one-rank communicator path (every collective of the split
    layout executed) reproduces the plain result bit for bit, (d) the device-resident sub-map assembly
    (lsgpu_icp_compute_clouds, laser_track.cpp:474-486) runs at this size."""
    import torch
    scene = synth.Scene(1234)
    poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(9)]
    scans = [synth.hdl64_scan(scene, poses[i], 16384, 20 + i) for i in range(9)]
    rel = [np.linalg.inv(poses[7]) @ poses[i] for i in range(8)]
    T_true = np.linalg.inv(poses[7]) @ poses[8]
    T_init = synth.se3(0.25, -0.1, 0.05, yaw=np.deg2rad(1.2)) @ T_true
    rd = scans[8]
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4     # "to 1e-4 m tolerance" (BASELINE configs[1]); the yaml
    with icp_mod.IcpHandle(cfg) as h:                      # thresholds (1e-3 rad / 1e-2 m) stop 20 cm short here
        parts = [torch.from_numpy(h.transform_points(synth.colmajor(rel[i]), scans[i])) for i in range(8)]
        ref = torch.cat(parts).cuda()
        assert ref.shape[0] > 8_000_000
        d_rf, d_rn = h.filter_reference(ref, 10, 1.0, 0)            # chain F: every point keeps its box normal
        d_rf, d_rn = d_rf.clone(), d_rn.clone()
        assert d_rf.shape[0] == ref.shape[0]
        h.set_reference(d_rf, d_rn)
        mean = h.reference_mean()
        T = synth.colmajor(T_init).copy()
        T[12:15] -= mean
        ids, d2 = h.knn(rd, T)
        q = h.transform_points(T, rd)
        rf = d_rf.cpu().numpy()
        ref_c = rf[:, :3] - mean
        dd = ((q[:, :3] - ref_c[ids]).astype(np.float64) ** 2).sum(1)
        assert np.allclose(dd, d2, rtol=1e-5, atol=1e-12)
        pick = np.random.default_rng(0).choice(rd.shape[0], 48, replace=False)
        for j in pick:                                             # brute force over all 8.4 M reference points
            best = ((ref_c.astype(np.float64) - q[j, :3].astype(np.float64)) ** 2).sum(1).min()
            assert best >= d2[j] * (1 - 1e-5), (j, best, d2[j])
        d_rd = torch.from_numpy(rd).cuda()
        Tg, st = h.align(d_rd, T_init)
        tr0 = [(t["limit"], t["n_used"]) for t in h.trace()]
        et, er = synth.pose_error(Tg.astype(np.float64), T_true)
        et0, er0 = synth.pose_error(T_init, T_true)
        # (the 8-scan street map constrains the driving direction weakly: trimmed point-to-plane ICP closes the
        # 27 cm / 1.2 deg offset of the guess only in part within the 40-iteration budget -- what matters here is that
        # it moves towards the truth, stays finite, and that every code path below reproduces it bit for bit)
        assert et < et0 and er < 0.1 * er0 and 2 <= st.iterations <= 40, (et, er, et0, er0, st.iterations)
        h.comm_init(0, 1, icp_mod.comm_unique_id())
        T1, st1 = h.align(d_rd, T_init)
        assert np.array_equal(Tg, T1) and st1.iterations == st.iterations
        assert tr0 == [(t["limit"], t["n_used"]) for t in h.trace()]
    with icp_mod.IcpHandle(cfg) as h2:                             # (d) sub-map assembled on the device from resident scans
        for i in range(9):
            h2.cloud_upload(i, scans[i])
        Tc, stc = h2.compute_clouds(8, list(range(8)), [synth.colmajor(r) for r in rel], T_init, 1.0, 10, 1.0, seed=0)
        assert int(h2.info().n_reference) == ref.shape[0]
        assert np.array_equal(Tc, Tg)                              # same clouds, same chain: the same transform


def test_a_refused_guess_leaves_nothing_behind_and_the_policy_info_reads(icp_mod, pair64k):
    """Round-5 advisor: lsgpu_icp_compute orders and moves the queries on its side stream with the guess it was given; a
    guess that is not rigid is refused only after both filters (as upstream).  A later direct align with the SAME device
    pointer and size and a good guess must not pick those queries up: it has to equal a fresh handle's result bit for bit.
    Also: lsgpu_icp_get_policy_info answers (index rest, filter fallbacks) and counts the filters run."""
    import torch
    d_ref = torch.from_numpy(pair64k["ref"]).cuda()
    d_rd = torch.from_numpy(pair64k["rd"]).cuda()
    T_good = pair64k["T_init"]
    T_bad = T_good.copy()
    T_bad[:3, :3] *= 1.2                       # |1 - det R| > 1e-3
    with icp_mod.IcpHandle() as h:
        with pytest.raises(Exception):
            h.compute(d_rd, d_ref, T_bad, -1.0, 10, 1.0, seed=0)    # no reading filter: align sees the caller's pointer
        T1, st1 = h.align(d_rd, T_good)
        pi = h.policy_info()
        assert pi.ssn_calls == 1 and pi.ssn_sort_fallbacks == 0 and pi.index_rest >= 0
        d_rf, d_rn = h.filter_reference(d_ref, 10, 1.0, 0)
        rf, rn = d_rf.clone(), d_rn.clone()
    with icp_mod.IcpHandle() as h2:
        h2.set_reference(rf, rn)
        T2, st2 = h2.align(d_rd, T_good)
    assert np.array_equal(T1, T2) and st1.iterations == st2.iterations


def _oracle_threads():
    return max(1, min(os.cpu_count() or 1, 128))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_config1_full_size_against_the_oracle(icp_mod, oracle):
    """BASELINE configs[1] at FULL size against the oracle itself (round-5 verdict: "exercised" -> "compared"): the 1 M-point
    pair, chain F, checker 1e-4 m / 1e-5 rad.  The oracle's query loop runs on the host's threads (libnabo's is an OpenMP
    loop too; the results do not depend on the thread count).  (a) the first search: every squared distance bit for bit
    (ids up to exact ties); (b) the whole alignment: the same number of iterations, per iteration the same trim limit (bits)
    and the same number of inliers; (c) the final transform within 1e-4 m / 1e-5 rad."""
    ref, rd, T_true, T_init = synth.scan_pair(16384)
    rf, rn = icp_mod.sampling_surface_normal(ref, 10, 1.0, 0)
    nt = _oracle_threads()
    ocfg = oracle.config_yaml(accum_double=1, min_diff_rot=1e-5, min_diff_trans=1e-4, num_threads=nt)
    rc, To, sto, tro = oracle.icp_compute(ocfg, rd, rf, rn, synth.colmajor(T_init), 40)
    assert rc == 0 and sto.iterations >= 10
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
    with icp_mod.IcpHandle(cfg) as h:
        h.set_reference(rf, rn)
        mean = h.reference_mean()
        ref_c = rf.copy()
        ref_c[:, :3] -= mean
        T = synth.colmajor(T_init).copy()
        T[12:15] -= mean
        ids, d2 = h.knn(rd, T)
        q = h.transform_points(T, rd)
        kd = oracle.KdTree(ref_c)
        oid, od2 = kd.nn(q, nt)
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32)), np.flatnonzero(d2 != od2)[:5]
        neq = np.flatnonzero(ids != oid)          # exact ties only
        assert neq.size < 1000
        if neq.size:
            diff = q[neq, :3] - ref_c[ids[neq], :3]
            dx, dy, dz = (diff[:, k].astype(np.float32) for k in range(3))
            dd = np.float32(dx * dx)
            dd = (dy.astype(np.float64) * dy + dd).astype(np.float32)
            dd = (dz.astype(np.float64) * dz + dd).astype(np.float32)
            assert np.array_equal(dd, od2[neq])
        Tg, stg = h.align(rd, T_init)
        trg = h.trace()
    assert stg.iterations == sto.iterations and stg.converged == sto.converged
    for k, (a_, b_) in enumerate(zip(trg, tro)):
        assert np.float32(a_["limit"]) == np.float32(b_["limit"]), (k, a_["limit"], b_["limit"])
        assert a_["n_used"] == b_["n_used"], (k, a_["n_used"], b_["n_used"])
    dt, dr = synth.pose_error(synth.from_colmajor(To), Tg.astype(np.float64))
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_config3_full_size_against_the_oracle(icp_mod, oracle):
    """BASELINE configs[3] at FULL size against the oracle: the 8-scan local map (8.4 M points, every point with its box
    normal from the DEVICE filter, which other tests hold bit-identical to the oracle's) against a 1 M-point scan, through
    the plain one-GPU path.  Per iteration the same number of inliers and the same trim limit, the same number of
    iterations, the final transform within 1e-4 m / 1e-5 rad."""
    import torch
    scene = synth.Scene(1234)
    poses = [synth.se3(0.8 * i, 0.05 * i, synth.SENSOR_HEIGHT, yaw=np.deg2rad(2.0 * i)) for i in range(9)]
    scans = [synth.hdl64_scan(scene, poses[i], 16384, 20 + i) for i in range(9)]
    rel = [np.linalg.inv(poses[7]) @ poses[i] for i in range(8)]
    T_true = np.linalg.inv(poses[7]) @ poses[8]
    T_init = synth.se3(0.25, -0.1, 0.05, yaw=np.deg2rad(1.2)) @ T_true
    rd = scans[8]
    from laser_slam_amd._lib import IcpConfig, lib
    import ctypes as C
    cfg = IcpConfig()
    lib().lsgpu_icp_config_yaml(C.byref(cfg))
    cfg.min_diff_rot, cfg.min_diff_trans = 1e-5, 1e-4
    with icp_mod.IcpHandle(cfg) as h:
        parts = [torch.from_numpy(h.transform_points(synth.colmajor(rel[i]), scans[i])) for i in range(8)]
        ref = torch.cat(parts).cuda()
        d_rf, d_rn = h.filter_reference(ref, 10, 1.0, 0)
        rf, rn = d_rf.cpu().numpy().copy(), d_rn.cpu().numpy().copy()
        assert rf.shape[0] > 8_000_000
        h.set_reference(d_rf.clone(), d_rn.clone())
        Tg, stg = h.align(torch.from_numpy(rd).cuda(), T_init)
        trg = h.trace()
    ocfg = oracle.config_yaml(accum_double=1, min_diff_rot=1e-5, min_diff_trans=1e-4, num_threads=_oracle_threads())
    rc, To, sto, tro = oracle.icp_compute(ocfg, rd, rf, rn, synth.colmajor(T_init), 40)
    assert rc == 0
    assert stg.iterations == sto.iterations and stg.converged == sto.converged
    for k, (a_, b_) in enumerate(zip(trg, tro)):
        assert a_["n_used"] == b_["n_used"], (k, a_["n_used"], b_["n_used"])
        assert np.float32(a_["limit"]) == np.float32(b_["limit"]), (k, a_["limit"], b_["limit"])
    dt, dr = synth.pose_error(synth.from_colmajor(To), Tg.astype(np.float64))
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)


def test_descriptor_rotation_matches_oracle(icp_mod, oracle):
    """RigidTransformation::compute rotates the `normals` / `observationDirections` descriptors of a cloud beside its
    features (laser_track.cpp:265, 485, 630, 643): lsgpu_rotate_descriptors against the oracle, bit for bit, host and
    device buffers; a non-rigid matrix is refused like upstream's TransformationError."""
    import torch
    rng = np.random.default_rng(5)
    nrm = rng.normal(size=(70001, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    T = synth.se3(1.0, -2.0, 0.5, yaw=0.7, pitch=-0.2, roll=0.1)
    with icp_mod.IcpHandle() as h:
        got = h.rotate_descriptors(T, nrm)
        want = oracle.rotate_normals(synth.colmajor(T), nrm)
        assert np.array_equal(got, want)
        d_out = torch.empty((nrm.shape[0], 3), dtype=torch.float32, device="cuda")
        from laser_slam_amd import _lib
        import ctypes as C
        rc = _lib.lib().lsgpu_rotate_descriptors(h._h, synth.colmajor(T).ctypes.data_as(C.POINTER(C.c_float)),
                                                 C.c_void_p(torch.from_numpy(nrm).cuda().data_ptr()), nrm.shape[0], C.c_void_p(d_out.data_ptr()))
        assert rc == 0 and np.array_equal(d_out.cpu().numpy(), want)
        bad = T.copy(); bad[:3, :3] *= 1.1
        with pytest.raises(_lib.LsgpuError):
            h.rotate_descriptors(bad, nrm)
        assert h.rotate_descriptors(T, nrm[:0]).shape == (0, 3)


def test_independent_known_answers_on_device(icp_mod):
    """tests/golden/independent_kat.npz (numpy / scipy only, tests/golden/make_golden_independent.py) replayed on the
    DEVICE path: box normals of the device surface-normal filter against numpy.linalg.eigh, the device radix select
    against numpy.partition, the 6x6 solve of the device-side update lane against numpy's Cholesky, and the differential
    checker's rotation metric (shared host / device source) against scipy.  The oracle is not involved."""
    import os
    import torch
    from test_oracle import _check_box_normals
    gold = os.path.join(os.path.dirname(__file__), "golden")
    kat, g = np.load(os.path.join(gold, "independent_kat.npz")), np.load(os.path.join(gold, "icp_pair4k.npz"))
    with icp_mod.IcpHandle() as h:
        pts, nrm = h.filter_reference(torch.from_numpy(kat["box_cloud"]).cuda(), 8, 1.0, 0)
        _check_box_normals(kat, pts.cpu().numpy(), nrm.cpu().numpy())
        for i in range(int(kat["trim_n"])):
            d2 = kat[f"trim{i}_d2"]
            # the kernel-level select takes the distances of the matched pairs (the loop never produces infinities: an
            # unmatched query does not exist with maxDist = inf, yaml:9-12)
            lim = h.trim_limit(d2[np.isfinite(d2)], float(kat[f"trim{i}_ratio"]))
            assert np.float32(lim) == kat[f"trim{i}_limit"], i
        h.set_reference(g["ref"], g["nrm"])
        T, st = h.align(g["rd"], g["T_init"])
        x0 = np.asarray(h.trace()[0]["x"], np.float64)          # first iteration: the matches of the fixture, device solve
        assert np.allclose(x0, kat["solve_x_f32"], rtol=2e-4, atol=1e-9) and np.allclose(x0, kat["solve_x_f64"], rtol=2e-3, atol=1e-8)
    for Ta, Tb, want in zip(kat["rot_Ta"], kat["rot_Tb"], kat["rot_angle"]):
        got = icp_mod.rotation_distance(Ta.reshape(4, 4).T, Tb.reshape(4, 4).T)
        assert abs(got - want) <= 5e-7 + 2e-6 * want, (got, want)


def _chain(mod, specs):
    arr = (mod.PointFilter * len(specs))()
    for a, (typ, dim, flag, v) in zip(arr, specs):
        a.type, a.dim, a.flag = typ, dim, flag
        for i, x in enumerate(v):
            a.v[i] = x
        a.state = 0.0
    return arr


def test_input_filter_chain_matches_oracle(icp_mod, oracle):
    """The input filter chain (laser_track.cpp:24-30, :146) on the device against its oracle restatement: same points,
    same order, bit for bit -- every filter alone, the whole chain, and the step of FixStepSampling carried from one
    scan to the next like the upstream filter object does."""
    from laser_slam_amd import _lib
    scene = synth.Scene(1234)
    scans = [synth.hdl64_scan(scene, synth.se3(0.8 * i, 0.0, synth.SENSOR_HEIGHT), 512, 70 + i) for i in range(3)]
    specs = [(_lib.FILTER_BOUNDING_BOX, 0, 1, [-6.0, 6.0, -4.0, 4.0, -2.5, 0.5]),
             (_lib.FILTER_MAX_DIST, -1, 0, [60.0]),
             (_lib.FILTER_MIN_DIST, -1, 0, [2.5]),
             (_lib.FILTER_FIX_STEP_SAMPLING, 0, 0, [3.0, 5.0, 1.3]),
             (_lib.FILTER_RANDOM_SAMPLING, 0, 0, [0.8])]
    extra = [(_lib.FILTER_MAX_DIST, 2, 0, [1.0]), (_lib.FILTER_MIN_DIST, 0, 0, [4.0]),
             (_lib.FILTER_BOUNDING_BOX, 0, 0, [-20.0, 20.0, -6.0, 6.0, -3.0, 3.0]),
             (_lib.FILTER_FIX_STEP_SAMPLING, 0, 0, [7.0, 2.0, 0.6])]
    with icp_mod.IcpHandle() as h:
        removed_something = False
        for sp in specs + extra:                                   # each filter on its own
            got = h.apply_point_filters(_chain(_lib, [sp]), scans[0], seed=5)
            want = oracle.apply_point_filters(_chain(oracle, [sp]), scans[0], seed=5)
            assert 0 < got.shape[0] <= scans[0].shape[0] and np.array_equal(got, want), sp
            removed_something = removed_something or got.shape[0] < scans[0].shape[0]
        assert removed_something
        dev, ora = _chain(_lib, specs), _chain(oracle, specs)     # the chain, state carried over three scans
        sizes = []
        for i, s in enumerate(scans):
            got = h.apply_point_filters(dev, s, seed=9 if i == 0 else -1)
            want = oracle.apply_point_filters(ora, s, seed=9 if i == 0 else -1)
            assert np.array_equal(got, want)
            assert dev[3].state == ora[3].state
            sizes.append(got.shape[0])
        assert [round(dev[3].state, 6)] == [5.0] and sizes[0] > sizes[2] > 0     # step 3 -> 3.9 -> 5 (clamped)
        import torch
        d = h.apply_point_filters(_chain(_lib, specs), torch.from_numpy(scans[0]).cuda(), seed=9)   # device in, device out
        assert d.is_cuda and np.array_equal(d.cpu().numpy(), oracle.apply_point_filters(_chain(oracle, specs), scans[0], seed=9))
        # a filter that is handed an empty cloud: ConvergenceError ("no points to filter") / None from the oracle
        empty_after = [(_lib.FILTER_MAX_DIST, -1, 0, [0.001]), (_lib.FILTER_MIN_DIST, -1, 0, [1.0])]
        with pytest.raises(_lib.ConvergenceError):
            h.apply_point_filters(_chain(_lib, empty_after), scans[0])
        assert oracle.apply_point_filters(_chain(oracle, empty_after), scans[0]) is None
        assert h.apply_point_filters(_chain(_lib, empty_after[:1]), scans[0]).shape[0] == 0     # ... the last one may empty it
        with pytest.raises(_lib.ConvergenceError):                                              # an empty cloud into a non-empty chain
            h.apply_point_filters(_chain(_lib, empty_after[:1]), scans[0][:0])
        assert h.apply_point_filters(_chain(_lib, []), scans[0][:0]).shape[0] == 0              # empty chain: no-op
        # MaxDist on one axis is SIGNED (keeps every negative coordinate), MinDist absolute; radial limits by magnitude
        for sp in ((_lib.FILTER_MAX_DIST, 0, 0, [3.0]), (_lib.FILTER_MIN_DIST, 1, 0, [3.0]), (_lib.FILTER_MAX_DIST, -1, 0, [-20.0])):
            got = h.apply_point_filters(_chain(_lib, [sp]), scans[0])
            assert np.array_equal(got, oracle.apply_point_filters(_chain(oracle, [sp]), scans[0]))
        assert (h.apply_point_filters(_chain(_lib, [(_lib.FILTER_MAX_DIST, 0, 0, [3.0])]), scans[0])[:, 0] < -3.0).any()
        with pytest.raises(_lib.LsgpuError):
            h.apply_point_filters(_chain(_lib, [(77, 0, 0, [1.0])]), scans[0])
        # RemoveNaN: a NaN in any of the four feature rows removes the point, an infinity does not; order preserved
        dirty = scans[0][:5000].copy()
        dirty[7, 0] = np.nan; dirty[8, 1] = np.nan; dirty[9, 2] = np.nan; dirty[10, 3] = np.nan; dirty[11, 0] = np.inf; dirty[4999, 2] = np.nan
        got = h.apply_point_filters(_chain(_lib, [(_lib.FILTER_REMOVE_NAN, 0, 0, [0.0])]), dirty)
        want = oracle.apply_point_filters(_chain(oracle, [(_lib.FILTER_REMOVE_NAN, 0, 0, [0.0])]), dirty)
        keep = ~np.isnan(dirty).any(1)
        assert got.shape[0] == 4995 and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got.view(np.uint32), dirty[keep].view(np.uint32))


def test_pointcloud2_conversion_round_trip(icp_mod):
    """sensor_msgs/PointCloud2 data block -> DataPoints.features on the device (rosMsgToPointMatcherCloud<float>,
    laser_slam_worker.cpp:125) against a numpy restatement of the record layout: Velodyne-style 22-byte records with
    x/y/z at unaligned offsets, big-endian variant, NaN records dropped when the message is not dense; and back to the
    16-byte PointXYZ records that lpmToPcl / pcl::toROSMsg produce (common.hpp:159-191)."""
    import torch
    rng = np.random.default_rng(5)
    n, step = 50_000, 22                                             # x@1 y@5 z@9 (unaligned), intensity@13, ring@17, pad
    xyz = rng.normal(0, 20, (n, 3)).astype(np.float32)
    xyz[rng.choice(n, 500, replace=False), rng.integers(0, 3, 500)] = np.nan
    xyz[7, 1] = np.inf
    rec = np.zeros((n, step), np.uint8)
    rec[:] = rng.integers(0, 255, (n, step), dtype=np.uint8)
    for k, off in enumerate((1, 5, 9)):
        rec[:, off:off + 4] = xyz[:, k:k + 1].view(np.uint8).reshape(n, 4)
    fin = np.isfinite(xyz).all(1)
    want_all = np.concatenate([xyz, np.ones((n, 1), np.float32)], 1)
    with icp_mod.IcpHandle() as h:
        got = h.cloud_from_pointcloud2(rec.tobytes(), n, step, 1, 5, 9, is_dense=True)
        assert np.array_equal(got.view(np.uint32), want_all.view(np.uint32))            # dense: records as they are
        got = h.cloud_from_pointcloud2(rec, n, step, 1, 5, 9, is_dense=False)
        assert np.array_equal(got, want_all[fin])                                       # NaN / Inf records dropped, in order
        big = rec.copy()
        for off in (1, 5, 9):
            big[:, off:off + 4] = rec[:, off:off + 4][:, ::-1]
        got_b = h.cloud_from_pointcloud2(big, n, step, 1, 5, 9, is_bigendian=True, is_dense=False)
        assert np.array_equal(got_b, want_all[fin])
        d = h.cloud_from_pointcloud2(torch.from_numpy(rec).cuda(), n, step, 1, 5, 9, is_dense=False, device_out=True)
        assert d.is_cuda and np.array_equal(d.cpu().numpy(), want_all[fin])             # device in, device out
        back = h.cloud_to_pointxyz(d)
        assert back.size == 16 * int(fin.sum())
        pts = back.view(np.float32).reshape(-1, 4)
        assert np.array_equal(pts[:, :3], xyz[fin]) and (pts[:, 3] == 1).all()
        with pytest.raises(icp_mod.LsgpuError):
            h.cloud_from_pointcloud2(rec, n, step, 1, 5, 20)                            # z would leave the record
        assert h.cloud_from_pointcloud2(b"", 0, step, 1, 5, 9).shape[0] == 0
