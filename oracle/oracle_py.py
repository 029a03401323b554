"""ctypes binding of oracle/liblsoracle.so (CPU restatement; TEST INFRASTRUCTURE ONLY).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  PARITY UNPINNED:
see icp_oracle.h.  Never imported by laser_slam_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblsoracle.so")


class Config(C.Structure):
    _fields_ = [
        ("reading_sampling_prob", C.c_float),
        ("surface_normal_knn", C.c_int),
        ("surface_normal_ratio", C.c_float),
        ("trim_ratio", C.c_float),
        ("max_iterations", C.c_int),
        ("min_diff_rot", C.c_float),
        ("min_diff_trans", C.c_float),
        ("smooth_length", C.c_int),
        ("accum_double", C.c_int),
        ("num_threads", C.c_int),
    ]


class IterTrace(C.Structure):
    _fields_ = [
        ("T_iter", C.c_float * 16),
        ("limit", C.c_float),
        ("n_used", C.c_int64),
        ("A", C.c_double * 36),
        ("b", C.c_double * 6),
        ("x", C.c_double * 6),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("final_limit", C.c_float),
        ("final_n_used", C.c_int64),
        ("t_filter_ms", C.c_double),
        ("t_build_ms", C.c_double),
        ("t_loop_ms", C.c_double),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "icp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, ip, i64 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int64
        dp = C.POINTER(C.c_double)
        L.lso_config_yaml.argtypes = [C.POINTER(Config)]
        L.lso_config_default.argtypes = [C.POINTER(Config)]
        L.lso_transform_points.argtypes = [fp, fp, i64, fp]
        L.lso_rotate_normals.argtypes = [fp, fp, i64, fp]
        L.lso_check_rigid.argtypes = [fp]
        L.lso_check_rigid.restype = C.c_int
        L.lso_correct_rigid.argtypes = [fp, fp]
        L.lso_random_sampling.argtypes = [i64, C.c_float, i64, C.POINTER(C.c_int64)]
        L.lso_random_sampling.restype = i64
        L.lso_sampling_surface_normal.argtypes = [fp, i64, C.c_int, C.c_float, i64, fp, fp]
        L.lso_sampling_surface_normal.restype = i64
        L.lso_kdtree_build.argtypes = [fp, i64]
        L.lso_kdtree_build.restype = C.c_void_p
        L.lso_kdtree_free.argtypes = [C.c_void_p]
        L.lso_kdtree_nn.argtypes = [C.c_void_p, fp, i64, ip, fp, C.c_int]
        L.lso_brute_nn.argtypes = [fp, i64, fp, i64, ip, fp]
        L.lso_trim_limit.argtypes = [fp, i64, C.c_float, fp]
        L.lso_trim_limit.restype = C.c_int
        L.lso_point_to_plane.argtypes = [fp, fp, fp, ip, fp, C.c_float, i64, C.c_int, dp, dp, dp,
                                         fp, C.POINTER(C.c_int64)]
        L.lso_point_to_plane.restype = C.c_int
        L.lso_icp_compute.argtypes = [C.POINTER(Config), fp, i64, fp, fp, i64, fp, fp,
                                      C.POINTER(Stats), C.POINTER(IterTrace), C.c_int]
        L.lso_icp_compute.restype = C.c_int
        L.lso_icp_compute_full.argtypes = [C.POINTER(Config), fp, i64, fp, i64, fp, i64, fp,
                                           C.POINTER(Stats)]
        L.lso_icp_compute_full.restype = C.c_int
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def config_yaml(**over) -> Config:
    c = Config()
    lib().lso_config_yaml(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def config_default(**over) -> Config:
    c = Config()
    lib().lso_config_default(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def transform_points(T16, xyz1):
    T, Tp = _f(T16)
    x, xp = _f(xyz1)
    out = np.empty_like(x)
    lib().lso_transform_points(Tp, xp, x.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def rotate_normals(T16, nrm):
    T, Tp = _f(T16)
    x, xp = _f(nrm)
    out = np.empty_like(x)
    lib().lso_rotate_normals(Tp, xp, x.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def check_rigid(T16) -> bool:
    T, Tp = _f(T16)
    return bool(lib().lso_check_rigid(Tp))


def rotation_distance(Ta16, Tb16) -> float:
    L = lib()
    L.lso_rotation_distance.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lso_rotation_distance.restype = C.c_float
    a, ap = _f(Ta16)
    b, bp = _f(Tb16)
    return float(L.lso_rotation_distance(ap, bp))


def correct_rigid(T16):
    T, Tp = _f(T16)
    out = np.empty(16, np.float32)
    lib().lso_correct_rigid(Tp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def random_sampling(n, prob, seed):
    keep = np.empty(n, np.int64)
    m = lib().lso_random_sampling(n, prob, seed, keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return keep[:m].copy()


def sampling_surface_normal(xyz1, knn=10, ratio=0.5, seed=0):
    x, xp = _f(xyz1)
    n = x.shape[0]
    o = np.empty((n, 4), np.float32)
    nr = np.empty((n, 3), np.float32)
    m = lib().lso_sampling_surface_normal(xp, n, knn, ratio, seed,
                                          o.ctypes.data_as(C.POINTER(C.c_float)),
                                          nr.ctypes.data_as(C.POINTER(C.c_float)))
    return o[:m].copy(), nr[:m].copy()


class KdTree:
    def __init__(self, ref_xyz1):
        self.ref, rp = _f(ref_xyz1)
        self.h = lib().lso_kdtree_build(rp, self.ref.shape[0])

    def nn(self, q_xyz1, threads=1):
        q, qp = _f(q_xyz1)
        n = q.shape[0]
        ids = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        lib().lso_kdtree_nn(self.h, qp, n, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                            d2.ctypes.data_as(C.POINTER(C.c_float)), threads)
        return ids, d2

    def __del__(self):
        if getattr(self, "h", None):
            lib().lso_kdtree_free(self.h)
            self.h = None


def brute_nn(ref_xyz1, q_xyz1):
    r, rp = _f(ref_xyz1)
    q, qp = _f(q_xyz1)
    ids = np.empty(q.shape[0], np.int32)
    d2 = np.empty(q.shape[0], np.float32)
    lib().lso_brute_nn(rp, r.shape[0], qp, q.shape[0], ids.ctypes.data_as(C.POINTER(C.c_int32)),
                       d2.ctypes.data_as(C.POINTER(C.c_float)))
    return ids, d2


def trim_limit(d2, ratio):
    d, dp = _f(d2)
    lim = C.c_float()
    rc = lib().lso_trim_limit(dp, d.shape[0], ratio, C.byref(lim))
    return rc, lim.value


def point_to_plane(p_xyz1, ref_xyz1, ref_nrm, ids, d2, limit, accum_double=1):
    p, pp = _f(p_xyz1)
    r, rp = _f(ref_xyz1)
    n, np_ = _f(ref_nrm)
    ids = np.ascontiguousarray(ids, np.int32)
    d, dp = _f(d2)
    A = np.empty(36)
    b = np.empty(6)
    x = np.zeros(6)
    dT = np.empty(16, np.float32)
    used = C.c_int64()
    D = C.POINTER(C.c_double)
    rc = lib().lso_point_to_plane(pp, rp, np_, ids.ctypes.data_as(C.POINTER(C.c_int32)), dp, limit,
                                  p.shape[0], accum_double, A.ctypes.data_as(D), b.ctypes.data_as(D),
                                  x.ctypes.data_as(D), dT.ctypes.data_as(C.POINTER(C.c_float)),
                                  C.byref(used))
    return rc, A.reshape(6, 6), b, x, dT, used.value


def icp_compute(cfg: Config, reading_xyz1, ref_xyz1, ref_nrm, T_init16, trace_cap=0):
    q, qp = _f(reading_xyz1)
    r, rp = _f(ref_xyz1)
    n, np_ = _f(ref_nrm)
    T, Tp = _f(T_init16)
    out = np.empty(16, np.float32)
    st = Stats()
    tr = (IterTrace * max(trace_cap, 1))()
    rc = lib().lso_icp_compute(C.byref(cfg), qp, q.shape[0], rp, np_, r.shape[0], Tp,
                               out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st), tr, trace_cap)
    traces = []
    for i in range(min(st.iterations, trace_cap)):
        t = tr[i]
        traces.append(dict(T_iter=np.array(t.T_iter[:], np.float32), limit=t.limit, n_used=t.n_used,
                           A=np.array(t.A[:]).reshape(6, 6), b=np.array(t.b[:]),
                           x=np.array(t.x[:])))
    return rc, out, st, traces


def icp_compute_full(cfg: Config, reading_xyz1, ref_xyz1, T_init16, seed=0):
    q, qp = _f(reading_xyz1)
    r, rp = _f(ref_xyz1)
    T, Tp = _f(T_init16)
    out = np.empty(16, np.float32)
    st = Stats()
    rc = lib().lso_icp_compute_full(C.byref(cfg), qp, q.shape[0], rp, r.shape[0], Tp, seed,
                                    out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
    return rc, out, st


def cylinder_filter(xyz1, center, radius_m, height_m, remove_point_inside):
    a, ap = _f(xyz1)
    c = np.ascontiguousarray(center, np.float32)
    out = np.empty((max(len(a), 1), 4), np.float32)
    L = lib()
    L.lso_cylinder_filter.restype = C.c_int64
    L.lso_cylinder_filter.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]
    m = L.lso_cylinder_filter(a.ctypes.data, len(a), c.ctypes.data, radius_m, height_m, int(remove_point_inside),
                              out.ctypes.data)
    return out[:m].copy()


def voxel_grid(xyz1, leaf, min_points=1):
    a, ap = _f(xyz1)
    lf = np.ascontiguousarray(leaf, np.float32)
    out = np.empty((max(len(a), 1), 4), np.float32)
    L = lib()
    L.lso_voxel_grid.restype = C.c_int64
    L.lso_voxel_grid.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    m = L.lso_voxel_grid(a.ctypes.data, len(a), lf.ctypes.data, int(min_points), out.ctypes.data)
    if m < 0:
        raise OverflowError("voxel index overflows an int")
    return out[:m].copy()


class PointFilter(C.Structure):
    """lso_point_filter: same layout as lsgpu_point_filter."""
    _fields_ = [("type", C.c_int), ("dim", C.c_int), ("flag", C.c_int), ("pad_", C.c_int), ("v", C.c_float * 6),
                ("state", C.c_double)]


def apply_point_filters(filters, xyz1, seed=-1):
    """Input filter chain (laser_track.cpp:146).  Returns the filtered cloud, or None if a filter got an empty cloud."""
    L = lib()
    L.lso_apply_point_filters.argtypes = [C.POINTER(PointFilter), C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.lso_apply_point_filters.restype = C.c_int64
    a = np.ascontiguousarray(xyz1, np.float32)
    out = np.empty((max(a.shape[0], 1), 4), np.float32)
    m = L.lso_apply_point_filters(filters, len(filters), a.ctypes.data, a.shape[0], seed, out.ctypes.data)
    return None if m < 0 else out[:m]
