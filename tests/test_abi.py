"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the header
declares, the host-side filters agree with the oracle, the YAML loader mirrors PointMatcher's module
selection, and a GPU-less box fails loudly instead of falling back."""
import ctypes as C
import io
import os
import re

import numpy as np
import pytest

from laser_slam_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "lsgpu_icp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lsgpu_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/lsgpu_icp.h but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == declared
    assert L.lsgpu_abi_version() == 4


def test_config_presets_match_yaml_and_setdefault():
    c = _lib.IcpConfig()
    _lib.lib().lsgpu_icp_config_yaml(C.byref(c))
    assert (round(c.trim_ratio, 6), c.max_iterations, c.smooth_length) == (0.75, 40, 4)
    assert np.float32(c.min_diff_rot) == np.float32(0.001) and np.float32(c.min_diff_trans) == np.float32(0.01)
    _lib.lib().lsgpu_icp_config_default(C.byref(c))
    assert (round(c.trim_ratio, 6), c.max_iterations, c.smooth_length) == (0.85, 40, 3)
    assert np.float32(c.min_diff_trans) == np.float32(0.001)


def test_strerror_and_bad_config():
    assert "ok" in _lib.strerror(0)
    assert "Convergence" in _lib.strerror(_lib.NO_CONVERGENCE)
    c = _lib.IcpConfig()
    _lib.lib().lsgpu_icp_config_yaml(C.byref(c))
    c.trim_ratio = 0.0
    h = C.c_void_p()
    assert _lib.lib().lsgpu_icp_create(C.byref(c), 0, C.byref(h)) == _lib.BAD_CONFIG


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from laser_slam_amd import icp
    with pytest.raises(_lib.LsgpuError) as e:
        icp.IcpHandle()
    assert e.value.code == _lib.HIP_ERROR
    with pytest.raises(_lib.LsgpuError):
        icp.ICP().compute(np.ones((8, 4), np.float32), np.ones((64, 4), np.float32) * np.arange(64)[:, None], np.eye(4))


def test_random_sampling_matches_oracle(oracle):
    from laser_slam_amd import icp
    for seed in (0, 5):
        a = icp.random_sampling(5000, 0.5, seed)
        b = oracle.random_sampling(5000, 0.5, seed)
        assert np.array_equal(a, b)
        assert 0.45 < a.size / 5000 < 0.55
    assert icp.random_sampling(0, 0.5, 1).size == 0
    assert icp.random_sampling(100, 1.1, 1).size == 100


def test_surface_normal_filter_matches_oracle(oracle, pair4k):
    """Same boxes (median splits) and normals as the oracle; std::nth_element vs the oracle's
    quickselect permute points inside a box differently, and near-collinear boxes are ill
    conditioned, so compare per point and allow a small fraction of ambiguous boxes."""
    from laser_slam_amd import icp
    a, an = icp.sampling_surface_normal(pair4k["ref"], 10, 1.0, 3)
    b, bn = oracle.sampling_surface_normal(pair4k["ref"], 10, 1.0, 3)
    ka, kb = np.lexsort(a[:, :3].T), np.lexsort(b[:, :3].T)
    assert np.array_equal(a[ka], b[kb])
    dev = np.abs(np.abs((an[ka] * bn[kb]).sum(1)) - 1)
    assert (dev > 1e-4).mean() < 0.01
    assert np.allclose(np.linalg.norm(an, axis=1), 1, atol=1e-5)
    # ratio sub-samples, empty input is fine
    c, _ = icp.sampling_surface_normal(pair4k["ref"], 10, 0.5, 3)
    assert 0.4 < c.shape[0] / a.shape[0] < 0.6
    e, en = icp.sampling_surface_normal(np.zeros((0, 4), np.float32), 10, 0.5, 3)
    assert e.shape == (0, 4) and en.shape == (0, 3)


def test_rigid_check_and_correct_match_oracle(oracle):
    from laser_slam_amd import icp
    T = synth.se3(1, 2, 3, yaw=0.3, pitch=-0.1, roll=0.2)
    assert icp.check_rigid(T)
    bad = T.copy()
    bad[:3, 0] *= 1.01
    bad[:3, 1] *= 0.97
    assert icp.check_rigid(bad) == oracle.check_rigid(synth.colmajor(bad))
    got = icp.correct_rigid(bad)
    want = oracle.correct_rigid(synth.colmajor(bad)).reshape(4, 4).T
    assert np.array_equal(got, want)


def test_yaml_loader_accepts_the_reference_chain_and_rejects_others():
    from laser_slam_amd import icp
    o = icp.ICP()
    o.load_from_yaml(os.path.join(ROOT, "tests", "golden", "icp_chain.yaml"))
    ch = o.chain
    assert (ch.reading_sampling_prob, ch.surface_normal_knn, ch.trim_ratio) == (0.5, 10, 0.75)
    assert (ch.max_iterations, ch.min_diff_rot, ch.min_diff_trans, ch.smooth_length) == (40, 0.001, 0.01, 4)
    o.set_default()
    assert (o.chain.reading_sampling_prob, o.chain.surface_normal_knn, o.chain.trim_ratio,
            o.chain.min_diff_trans, o.chain.smooth_length) == (0.75, 7, 0.85, 0.001, 3)
    # module defaults apply when a parameter is absent; inspector/logger are accepted and ignored
    base = ("referenceDataPointsFilters:\n  - SamplingSurfaceNormalDataPointsFilter\nmatcher:\n  KDTreeMatcher: {}\n"
            "errorMinimizer: PointToPlaneErrorMinimizer\ntransformationCheckers:\n  - CounterTransformationChecker\n")
    o.load_from_yaml(io.StringIO(base + "outlierFilters:\n  - TrimmedDistOutlierFilter\n"
                                 "inspector:\n  NullInspector\nlogger:\n  NullLogger\n"))
    assert o.chain.trim_ratio == 0.85 and o.chain.surface_normal_knn == 7
    # an absent section is "no module" (libpointmatcher clears the chains first), not the default module
    o.load_from_yaml(io.StringIO(base))
    assert o.chain.reading_sampling_prob < 0 and o.chain.trim_ratio == 1.0 and o.chain.min_diff_rot < 0
    with pytest.raises(_lib.LsgpuError):
        o.load_from_yaml(io.StringIO("matcher:\n  KDTreeMatcher: {}\n"))      # no normals, no minimizer, no stop
    for bad in ("outlierFilters:\n  - MaxDistOutlierFilter: {maxDist: 1}\n",
                "errorMinimizer: PointToPointErrorMinimizer\n",
                "matcher:\n  KDTreeMatcher: {knn: 3}\n",
                "readingStepDataPointsFilters:\n  - RandomSamplingDataPointsFilter: {prob: 0.5}\n"):
        with pytest.raises(_lib.LsgpuError) as e:
            o.load_from_yaml(io.StringIO(bad))
        assert e.value.code == _lib.BAD_CONFIG


def test_synthetic_scan_generator_is_seeded_and_sane():
    a = synth.scan_pair(32)
    b = synth.scan_pair(32)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ref, rd, T_true, T_init = a
    assert ref.dtype == np.float32 and ref.shape[1] == 4 and (ref[:, 3] == 1).all()
    assert 0.9 * 64 * 32 < ref.shape[0] <= 64 * 32
    r = np.linalg.norm(ref[:, :3], axis=1)
    assert r.min() > 1.0 and r.max() < synth.MAX_RANGE + 1
    dt, dr = synth.pose_error(T_init, T_true)
    assert abs(dt - 0.3) < 0.05 and abs(np.rad2deg(dr) - 1.5) < 0.2


def test_draw_stream_is_the_glibc_rand_sequence():
    """The filters' draws come from the library's own generator (csrc/lsgpu_rand.h); it must reproduce
    std::srand(seed) + std::rand() of glibc, which is what the reference's filters consume, including the
    continuation of the stream across calls (seed < 0)."""
    import ctypes as C
    import numpy as np
    from laser_slam_amd import icp
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int

    def libc_keep(n, prob):
        r = np.array([libc.rand() for _ in range(n)], np.int64)
        draws = r.astype(np.float32) / np.float32(2147483648.0)   # (float)rand() / (float)RAND_MAX
        return np.nonzero(draws < np.float32(prob))[0]

    for seed in (0, 1, 7, 123456, 2 ** 32 - 1):
        libc.srand(C.c_uint(seed))
        assert np.array_equal(icp.random_sampling(4000, 0.37, seed), libc_keep(4000, 0.37)), seed
        assert np.array_equal(icp.random_sampling(1500, 0.5, -1), libc_keep(1500, 0.5)), seed   # continues
    # large requests (>= 1 M draws) are produced in parallel segments whose start states come from the recurrence's
    # jump-ahead matrix
    libc.srand(C.c_uint(99))
    assert np.array_equal(icp.random_sampling(1100001, 0.37, 99), libc_keep(1100001, 0.37))
    assert np.array_equal(icp.random_sampling(777, 0.5, -1), libc_keep(777, 0.5))             # continues after it


def test_draw_stream_speculative_use_and_parallel_path(tmp_path):
    """begin(kmax) / commit(k < kmax) / continue, as the device filters use the stream, for small requests and for the
    large ones that are produced in parallel segments -- against glibc's rand() (tests/cpp/draw_stream_check.cpp)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "draw_stream_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "laser_slam_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "draw_stream_check.cpp"), "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "DRAWS_OK" in r.stdout, r.stdout + r.stderr


def test_header_is_c99_and_the_c_example_links(tmp_path):
    """include/lsgpu_icp.h must stay a plain C header (the reference's maintainers would bind it from C++ or through
    an FFI), and examples/compute_pair.c must build against it and the shared library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(root, "include", "lsgpu_icp.h")])
    exe = str(tmp_path / "compute_pair")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "compute_pair.c"), "-o", exe,
                           "-L", os.path.join(root, "laser_slam_amd"), "-llsgpu_icp",
                           "-Wl,-rpath," + os.path.join(root, "laser_slam_amd")])
    assert os.path.exists(exe)
