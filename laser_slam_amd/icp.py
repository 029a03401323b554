"""Host-side mirror of the ICP object seam the reference calls (seam B2).

The reference owns a ``PointMatcher::ICP icp_`` (laser_slam/include/laser_slam/laser_track.hpp:217,
incremental_estimator.hpp:70) and uses exactly three members of it:

    icp_.loadFromYaml(std::istream&)      laser_slam/src/laser_track.cpp:17
    icp_.setDefault()                     laser_slam/src/laser_track.cpp:20
    icp_.compute(reading, reference, T)   laser_slam/src/laser_track.cpp:496,
                                          laser_slam/src/incremental_estimator.cpp:108

``ICP`` below keeps those names/semantics (snake_case) over the C ABI in include/lsgpu_icp.h;
``IcpHandle`` is the thin 1:1 wrapper of that ABI.  Clouds are (N,4) float32 x,y,z,1 arrays
(DataPoints.features transposed: memory is identical to Eigen's column-major 4xN); they may be numpy
arrays (host) or torch CUDA tensors (HBM-resident, zero copy).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib
from ._lib import ConvergenceError, IcpConfig, IcpStats, IterTrace, LsgpuError

try:  # torch is plumbing only (device memory); the package works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _as_f32(x, cols: int):
    """-> (address, keepalive, n_rows).  numpy: C-contiguous float32 copy if needed."""
    if _is_torch(x):
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float32).contiguous()
        if x.dim() != 2 or x.shape[1] != cols:
            raise ValueError(f"expected (N,{cols}) tensor, got {tuple(x.shape)}")
        if x.is_cuda:
            # stream contract of include/lsgpu_icp.h: the library works on the handle's own (non-blocking) stream, so
            # whatever torch still has in flight for this tensor -- including the copy made just above -- must be done
            torch.cuda.current_stream(x.device).synchronize()
        return (x.data_ptr() if x.numel() else None), x, x.shape[0]
    a = np.ascontiguousarray(x, np.float32)
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError(f"expected (N,{cols}) array, got {a.shape}")
    return (a.ctypes.data if a.size else None), a, a.shape[0]


def _t16(T) -> np.ndarray:
    """4x4 (row-major numpy) or 16 column-major floats -> 16 float32 column-major."""
    T = np.asarray(T)
    if T.shape == (4, 4):
        return np.ascontiguousarray(T.astype(np.float32).T).reshape(16)
    if T.size == 16:
        return np.ascontiguousarray(T, np.float32).reshape(16)
    raise ValueError("transform must be 4x4 or 16 floats")


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _raise(code: int, what: str, h=None):
    detail = ""
    if h is not None:
        detail = _lib.lib().lsgpu_last_error(h).decode()
    if code == _lib.NO_CONVERGENCE:
        raise ConvergenceError(code, what, detail)
    raise LsgpuError(code, what, detail)


class IcpHandle:
    """One lsgpu_icp handle == one reference ``icp_`` member: one device, one HIP stream."""

    def __init__(self, cfg: Optional[IcpConfig] = None, device: int = 0):
        L = _lib.lib()
        if cfg is None:
            cfg = IcpConfig()
            L.lsgpu_icp_config_yaml(C.byref(cfg))
        self.cfg = cfg
        self.device = device
        self._h = C.c_void_p()
        rc = L.lsgpu_icp_create(C.byref(cfg), device, C.byref(self._h))
        if rc != _lib.OK:
            self._h = None
            _raise(rc, "lsgpu_icp_create (is a ROCm GPU visible?)")

    def close(self):
        if getattr(self, "_h", None) and _lib is not None and _lib._lib is not None:
            _lib._lib.lsgpu_icp_destroy(self._h)  # (module globals may be gone at interpreter exit)
        self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- ICP::compute steps 2-3
    def set_reference(self, ref_xyz1, ref_normals):
        p, _k1, n = _as_f32(ref_xyz1, 4)
        q, _k2, m = _as_f32(ref_normals, 3)
        if m != n:
            raise ValueError("normals must have one row per reference point")
        rc = _lib.lib().lsgpu_icp_set_reference(self._h, p, q, n)
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_set_reference", self._h)

    def reference_mean(self) -> np.ndarray:
        m = np.zeros(3, np.float32)
        rc = _lib.lib().lsgpu_icp_get_reference_mean(self._h, _fp(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_get_reference_mean", self._h)
        return m

    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        """Split-scan mode: this handle's align() now takes the LOCAL shard of the reading and
        all-reduces the select histograms and normal-equation sums over RCCL (include/lsgpu_icp.h)."""
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        rc = _lib.lib().lsgpu_icp_comm_init(self._h, rank, nranks, unique_id)
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_comm_init", self._h)

    def info(self) -> "_lib.IcpInfo":
        out = _lib.IcpInfo()
        rc = _lib.lib().lsgpu_icp_get_info(self._h, C.byref(out))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_get_info", self._h)
        return out

    def policy_info(self):
        """lsgpu_icp_get_policy_info: what the handle's launch policy remembers across calls (index rest, fallbacks)."""
        out = _lib.PolicyInfo()
        rc = _lib.lib().lsgpu_icp_get_policy_info(self._h, C.byref(out))
        if rc:
            _raise(rc, "lsgpu_icp_get_policy_info", self._h)
        return out

    # ---- ICP::compute steps 5-7
    def align(self, reading_xyz1, T_init):
        """-> (T 4x4 float32, IcpStats).  Raises ConvergenceError like PointMatcher."""
        p, _k, n = _as_f32(reading_xyz1, 4)
        ti = _t16(T_init)
        to = np.empty(16, np.float32)
        st = IcpStats()
        rc = _lib.lib().lsgpu_icp_align(self._h, p, n, _fp(ti), _fp(to), C.byref(st))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_align", self._h)
        return to.reshape(4, 4).T.copy(), st

    # ---- the sampling filters on the device, and the whole of ICP::compute
    def filter_reference(self, xyz1, knn: int = 10, ratio: float = 0.5, seed: int = -1):
        """SamplingSurfaceNormalDataPointsFilter (icp_default.yaml:5-7) on the GPU -> (xyz1', normals);
        torch CUDA input gives torch CUDA output, anything else numpy."""
        p, _k, n = _as_f32(xyz1, 4)
        m = C.c_int64(0)
        if _is_torch(xyz1) and xyz1.is_cuda:
            o = torch.empty((max(n, 1), 4), dtype=torch.float32, device=xyz1.device)
            nr = torch.empty((max(n, 1), 3), dtype=torch.float32, device=xyz1.device)
            torch.cuda.synchronize()
            po, pn = o.data_ptr(), nr.data_ptr()
        else:
            o = np.empty((max(n, 1), 4), np.float32)
            nr = np.empty((max(n, 1), 3), np.float32)
            po, pn = o.ctypes.data, nr.ctypes.data
        rc = _lib.lib().lsgpu_icp_filter_reference(self._h, p, n, knn, ratio, seed, po, pn, C.byref(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_filter_reference", self._h)
        return o[:m.value], nr[:m.value]

    def filter_reading(self, xyz1, prob: float = 0.5, seed: int = -1):
        """RandomSamplingDataPointsFilter (icp_default.yaml:1-3) on the GPU -> xyz1'."""
        p, _k, n = _as_f32(xyz1, 4)
        m = C.c_int64(0)
        if _is_torch(xyz1) and xyz1.is_cuda:
            o = torch.empty((max(n, 1), 4), dtype=torch.float32, device=xyz1.device)
            torch.cuda.synchronize()
            po = o.data_ptr()
        else:
            o = np.empty((max(n, 1), 4), np.float32)
            po = o.ctypes.data
        rc = _lib.lib().lsgpu_icp_filter_reading(self._h, p, n, prob, seed, po, C.byref(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_filter_reading", self._h)
        return o[:m.value]

    def compute(self, reading_xyz1, reference_xyz1, T_init, reading_prob: float = 0.5, ssn_knn: int = 10,
                ssn_ratio: float = 0.5, seed: int = -1):
        """``icp_.compute(reading, reference, T_init)`` entirely on the device: reference filter,
        set_reference, reading filter, align.  -> (T 4x4, IcpStats); raises ConvergenceError."""
        q, _k1, nq = _as_f32(reading_xyz1, 4)
        r, _k2, nr = _as_f32(reference_xyz1, 4)
        ch = _lib.ChainCfg(reading_prob, ssn_knn, ssn_ratio, 0, seed)
        ti = _t16(T_init)
        to = np.empty(16, np.float32)
        st = IcpStats()
        rc = _lib.lib().lsgpu_icp_compute(self._h, q, nq, r, nr, _fp(ti), C.byref(ch), _fp(to), C.byref(st))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_compute", self._h)
        return to.reshape(4, 4).T.copy(), st

    # ---- local-map maintenance (laser_slam_ros worker): cylinder crop, voxel grid
    def _filter_out(self, xyz1, n):
        if _is_torch(xyz1) and xyz1.is_cuda:
            o = torch.empty((max(n, 1), 4), dtype=torch.float32, device=xyz1.device)
            torch.cuda.synchronize()
            return o, o.data_ptr()
        o = np.empty((max(n, 1), 4), np.float32)
        return o, o.ctypes.data

    def filter_cylinder(self, xyz1, center, radius_m: float, height_m: float, remove_point_inside: bool = False):
        """applyCylindricalFilter (laser_slam_ros common.hpp:194-223) on the GPU, order preserved."""
        p, _k, n = _as_f32(xyz1, 4)
        c = np.ascontiguousarray(center, np.float32).reshape(3)
        o, po = self._filter_out(xyz1, n)
        m = C.c_int64(0)
        rc = _lib.lib().lsgpu_filter_cylinder(self._h, p, n, _fp(c), radius_m, height_m, int(remove_point_inside),
                                              po, C.byref(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_filter_cylinder", self._h)
        return o[:m.value]

    def filter_voxel_grid(self, xyz1, leaf, min_points: int = 1):
        """pcl::VoxelGrid (laser_slam_worker.cpp:70-72, 439-440) on the GPU: one centroid per occupied voxel."""
        p, _k, n = _as_f32(xyz1, 4)
        lf = np.ascontiguousarray(np.broadcast_to(np.asarray(leaf, np.float32), (3,)), np.float32)
        o, po = self._filter_out(xyz1, n)
        m = C.c_int64(0)
        rc = _lib.lib().lsgpu_filter_voxel_grid(self._h, p, n, _fp(lf), int(min_points), po, C.byref(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_filter_voxel_grid", self._h)
        return o[:m.value]

    # ---- the input filter chain (laser_track.cpp:24-30, :146)
    def apply_point_filters(self, filters, xyz1, seed: int = -1):
        """`filters`: ctypes array of _lib.PointFilter (FixStepSampling's `state` is updated in place)."""
        p, _k, n = _as_f32(xyz1, 4)
        o, po = self._filter_out(xyz1, n)
        m = C.c_int64(0)
        rc = _lib.lib().lsgpu_apply_point_filters(self._h, filters, len(filters), p, n, seed, po, C.byref(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_apply_point_filters", self._h)
        return o[:m.value]

    # ---- ROS message surface: sensor_msgs/PointCloud2 data block <-> x,y,z,1 (laser_slam_worker.cpp:125, common.hpp:159-191)
    def cloud_from_pointcloud2(self, data, n_points: int, point_step: int, off_x: int, off_y: int, off_z: int,
                               is_bigendian: bool = False, is_dense: bool = True, device_out: bool = False):
        """`data`: bytes / uint8 numpy array / uint8 torch tensor holding n_points records of point_step bytes."""
        if _is_torch(data):
            if data.is_cuda:
                torch.cuda.current_stream(data.device).synchronize()
            keep, p = data, data.data_ptr()
        else:
            keep = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data, np.uint8)
            p = keep.ctypes.data
        if device_out:
            o = torch.empty((max(n_points, 1), 4), dtype=torch.float32, device=f"cuda:{self.device}")
            torch.cuda.synchronize()
            po = o.data_ptr()
        else:
            o = np.empty((max(n_points, 1), 4), np.float32)
            po = o.ctypes.data
        m = C.c_int64(0)
        rc = _lib.lib().lsgpu_cloud_from_pointcloud2(self._h, p, n_points, point_step, off_x, off_y, off_z,
                                                     int(is_bigendian), int(not is_dense), po, C.byref(m))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_cloud_from_pointcloud2", self._h)
        return o[:m.value]

    def cloud_to_pointxyz(self, xyz1) -> np.ndarray:
        """x,y,z,1 -> the data block (uint8, 16 bytes per point) of a PointCloud2 / pcl::PointCloud<PointXYZ>."""
        p, _k, n = _as_f32(xyz1, 4)
        o = np.empty(16 * max(n, 1), np.uint8)
        rc = _lib.lib().lsgpu_cloud_to_pointxyz(self._h, p, n, o.ctypes.data)
        if rc != _lib.OK:
            _raise(rc, "lsgpu_cloud_to_pointxyz", self._h)
        return o[:16 * n]

    # ---- clouds kept in HBM between calls; sub-map assembly on the device (laser_track.cpp:474-486)
    def cloud_upload(self, slot: int, xyz1):
        p, _k, n = _as_f32(xyz1, 4)
        rc = _lib.lib().lsgpu_cloud_upload(self._h, slot, p, n)
        if rc != _lib.OK:
            _raise(rc, "lsgpu_cloud_upload", self._h)

    def cloud_release(self, slot: int):
        _lib.lib().lsgpu_cloud_release(self._h, slot)

    def cloud_size(self, slot: int) -> int:
        n = C.c_int64(-1)
        _lib.lib().lsgpu_cloud_size(self._h, slot, C.byref(n))
        return n.value

    def compute_clouds(self, reading_slot: int, ref_slots, ref_T, T_init, reading_prob: float = 0.5,
                       ssn_knn: int = 10, ssn_ratio: float = 0.5, seed: int = -1):
        """ICP::compute with reading = cloud `reading_slot` and reference = concat(T_i * cloud ref_slots[i])."""
        k = len(ref_slots)
        slots = (C.c_int * max(k, 1))(*ref_slots)
        Ts = None
        if ref_T is not None:
            Ts = np.concatenate([_t16(T) for T in ref_T]) if k else np.zeros(0, np.float32)
        ch = _lib.ChainCfg(reading_prob, ssn_knn, ssn_ratio, 0, seed)
        ti = _t16(T_init)
        to = np.empty(16, np.float32)
        st = IcpStats()
        rc = _lib.lib().lsgpu_icp_compute_clouds(self._h, reading_slot, slots, _fp(Ts) if Ts is not None else None,
                                                 k, _fp(ti), C.byref(ch), _fp(to), C.byref(st))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_compute_clouds", self._h)
        return to.reshape(4, 4).T.copy(), st

    def compute_clouds_upload(self, reading_slot: int, reading_xyz1, ref_slots, ref_T, T_init, reading_prob: float = 0.5,
                              ssn_knn: int = 10, ssn_ratio: float = 0.5, seed: int = -1):
        """upload_cloud(reading_slot, reading) + compute_clouds(reading_slot, ...) in one call: the host reading crosses
        PCIe while the sub-map is assembled and filtered (the call shape of LaserTrack::localScanToSubMap)."""
        rd = np.ascontiguousarray(reading_xyz1, np.float32)
        k = len(ref_slots)
        slots = (C.c_int * max(k, 1))(*ref_slots)
        Ts = None
        if ref_T is not None:
            Ts = np.concatenate([_t16(T) for T in ref_T]) if k else np.zeros(0, np.float32)
        ch = _lib.ChainCfg(reading_prob, ssn_knn, ssn_ratio, 0, seed)
        ti = _t16(T_init)
        to = np.empty(16, np.float32)
        st = IcpStats()
        rc = _lib.lib().lsgpu_icp_compute_clouds_upload(self._h, reading_slot, _fp(rd), rd.shape[0], slots,
                                                        _fp(Ts) if Ts is not None else None, k, _fp(ti), C.byref(ch),
                                                        _fp(to), C.byref(st))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_icp_compute_clouds_upload", self._h)
        return to.reshape(4, 4).T.copy(), st

    def trace(self, cap: int = 64):
        buf = (IterTrace * cap)()
        n = _lib.lib().lsgpu_icp_get_trace(self._h, buf, cap)
        out = []
        for i in range(n):
            t = buf[i]
            out.append(dict(T_iter=np.array(t.T_iter[:], np.float32), limit=t.limit,
                            n_used=t.n_used, A=np.array(t.A[:]).reshape(6, 6),
                            b=np.array(t.b[:]), x=np.array(t.x[:]),
                            knn_main_us=t.knn_main_us, knn_fallback_us=t.knn_fallback_us,
                            stragglers=t.stragglers, searching=t.reserved))
        return out

    # ---- kernel-level entry points (reference-mean frame)
    def knn(self, query_xyz1, T=None):
        p, _k, n = _as_f32(query_xyz1, 4)
        ids = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        tp = _fp(_t16(T)) if T is not None else None
        rc = _lib.lib().lsgpu_knn(self._h, p, n, tp, ids.ctypes.data if n else None,
                                  d2.ctypes.data if n else None)
        if rc != _lib.OK:
            _raise(rc, "lsgpu_knn", self._h)
        return ids, d2

    def trim_limit(self, d2, ratio: float) -> float:
        a = np.ascontiguousarray(d2, np.float32)
        lim = C.c_float()
        rc = _lib.lib().lsgpu_trim_limit(self._h, a.ctypes.data if a.size else None, a.size,
                                         ratio, C.byref(lim))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_trim_limit", self._h)
        return lim.value

    def normal_eq(self, query_xyz1, T, ids, d2, limit: float):
        """-> (A 6x6, b 6, n_used, sum r^2), double."""
        p, _k, n = _as_f32(query_xyz1, 4)
        ids = np.ascontiguousarray(ids, np.int32)
        d2 = np.ascontiguousarray(d2, np.float32)
        out = np.zeros(29)
        tp = _fp(_t16(T)) if T is not None else None
        rc = _lib.lib().lsgpu_normal_eq(self._h, p, n, tp, ids.ctypes.data, d2.ctypes.data, limit,
                                        out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc != _lib.OK:
            _raise(rc, "lsgpu_normal_eq", self._h)
        A = np.zeros((6, 6))
        k = 0
        for a in range(6):
            for c in range(a, 6):
                A[a, c] = A[c, a] = out[k]
                k += 1
        return A, out[21:27].copy(), int(out[27]), float(out[28])

    def transform_points(self, T, xyz1):
        p, _k, n = _as_f32(xyz1, 4)
        out = np.empty((n, 4), np.float32)
        rc = _lib.lib().lsgpu_transform_points(self._h, _fp(_t16(T)), p, n,
                                               out.ctypes.data if n else None)
        if rc != _lib.OK:
            _raise(rc, "lsgpu_transform_points", self._h)
        return out


def _rotate_descriptors(self, T, desc3):
    """RigidTransformation::compute on a 3-row descriptor (normals / observationDirections): R * d on the device."""
    p, _k, n = _as_f32(desc3, 3)
    out = np.empty((n, 3), np.float32)
    rc = _lib.lib().lsgpu_rotate_descriptors(self._h, _fp(_t16(T)), p, n, out.ctypes.data if n else None)
    if rc != _lib.OK:
        _raise(rc, "lsgpu_rotate_descriptors", self._h)
    return out


IcpHandle.rotate_descriptors = _rotate_descriptors


def align_batch(handles, references, normals, readings, T_inits):
    """BASELINE config 3: many independent pairs on one GPU (``lsgpu_icp_align_batch``).

    ``handles``: IcpHandle pool on one device (pair i -> handles[i % len]); every cloud may be a host
    array or a CUDA tensor.  -> (T [B,4,4] float32, [IcpStats], rc int array); non-converged pairs keep
    T_init and have rc == 1, any other failure raises."""
    B = len(readings)
    if not (len(references) == len(normals) == len(T_inits) == B):
        raise ValueError("one reference, normals, reading and T_init per pair")
    keep, rp, npn, qp = [], (C.c_void_p * B)(), (C.c_void_p * B)(), (C.c_void_p * B)()
    nr, nq = (C.c_int64 * B)(), (C.c_int64 * B)()
    for i in range(B):
        p, k1, n = _as_f32(references[i], 4)
        q, k2, m = _as_f32(normals[i], 3)
        r, k3, l = _as_f32(readings[i], 4)
        if m != n:
            raise ValueError("normals must have one row per reference point")
        keep += [k1, k2, k3]
        rp[i], npn[i], qp[i], nr[i], nq[i] = p, q, r, n, l
    ti = np.concatenate([_t16(T) for T in T_inits]) if B else np.zeros(0, np.float32)
    to = np.empty(16 * B, np.float32)
    st = (IcpStats * max(B, 1))()
    rc = (C.c_int * max(B, 1))()
    hs = (C.c_void_p * len(handles))(*[h._h for h in handles])
    code = _lib.lib().lsgpu_icp_align_batch(hs, len(handles), B, rp, npn, nr, qp, nq, _fp(ti), _fp(to), st, rc)
    if code not in (_lib.OK, _lib.NO_CONVERGENCE):
        bad = next((i for i in range(B) if rc[i] not in (_lib.OK, _lib.NO_CONVERGENCE)), None)
        _raise(code, "lsgpu_icp_align_batch" + (f" (pair {bad})" if bad is not None else ""),
               handles[bad % len(handles)]._h if bad is not None else None)
    T = to.reshape(B, 4, 4).transpose(0, 2, 1).copy()
    return T, [st[i] for i in range(B)], np.array(rc[:B], np.int32)


def comm_unique_id() -> bytes:
    """RCCL unique id (call on rank 0, ship to the other ranks)."""
    buf = C.create_string_buffer(128)
    rc = _lib.lib().lsgpu_comm_get_unique_id(buf)
    if rc != _lib.OK:
        _raise(rc, "lsgpu_comm_get_unique_id")
    return buf.raw


# ---------------------------------------------------------------------------------------------
# host-side modules (CPU in the reference too)

def random_sampling(n: int, prob: float, seed: int = -1) -> np.ndarray:
    """RandomSamplingDataPointsFilter (icp_default.yaml:1-3) -> kept indices."""
    keep = np.empty(max(n, 1), np.int64)
    m = _lib.lib().lsgpu_filter_random_sampling(n, prob, seed,
                                                keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return keep[:m].copy()


def sampling_surface_normal(xyz1, knn: int = 10, ratio: float = 0.5, seed: int = -1):
    """SamplingSurfaceNormalDataPointsFilter (icp_default.yaml:5-7) -> (xyz1', normals)."""
    a = np.ascontiguousarray(xyz1, np.float32)
    n = a.shape[0]
    o = np.empty((max(n, 1), 4), np.float32)
    nr = np.empty((max(n, 1), 3), np.float32)
    m = _lib.lib().lsgpu_filter_sampling_surface_normal(a.ctypes.data if n else None, n, knn, ratio,
                                                        seed, o.ctypes.data, nr.ctypes.data)
    return o[:m].copy(), nr[:m].copy()


def check_rigid(T) -> bool:
    return bool(_lib.lib().lsgpu_check_rigid(_fp(_t16(T))))


def rotation_distance(Ta, Tb) -> float:
    """DifferentialTransformationChecker's rotation metric between two transforms (Eigen angularDistance, float)."""
    return float(_lib.lib().lsgpu_rotation_distance(_fp(_t16(Ta)), _fp(_t16(Tb))))


def correct_rigid(T) -> np.ndarray:
    out = np.empty(16, np.float32)
    _lib.lib().lsgpu_correct_rigid(_fp(_t16(T)), _fp(out))
    return out.reshape(4, 4).T.copy()


# ---------------------------------------------------------------------------------------------

_SUPPORTED = {
    "readingDataPointsFilters": {"RandomSamplingDataPointsFilter"},
    "referenceDataPointsFilters": {"SamplingSurfaceNormalDataPointsFilter"},
    "matcher": {"KDTreeMatcher"},
    "outlierFilters": {"TrimmedDistOutlierFilter"},
    "errorMinimizer": {"PointToPlaneErrorMinimizer"},
    "transformationCheckers": {"CounterTransformationChecker", "DifferentialTransformationChecker"},
}


@dataclass
class ChainConfig:
    """The module chain of laser_slam/configurations/icp_default.yaml, as parameters."""
    reading_sampling_prob: float = 0.5      # yaml:3   (module default 0.75)
    surface_normal_knn: int = 10            # yaml:7   (module default 7)
    surface_normal_ratio: float = 0.5       # module default
    trim_ratio: float = 0.75                # yaml:16  (module default 0.85)
    max_iterations: int = 40                # yaml:23
    min_diff_rot: float = 0.001             # yaml:25
    min_diff_trans: float = 0.01            # yaml:26  (module default 0.001)
    smooth_length: int = 4                  # yaml:27  (module default 3)
    seed: int = -1                          # >= 0: srand(seed) before the filters
    extra: dict = field(default_factory=dict)


class ICP:
    """Drop-in for the reference's ``PointMatcher::ICP icp_`` member."""

    def __init__(self, device: int = 0):
        self.device = device
        self.chain = ChainConfig()
        self._handle: Optional[IcpHandle] = None
        self.last_stats: Optional[IcpStats] = None

    # -- laser_track.cpp:20
    def set_default(self):
        self.chain = ChainConfig(reading_sampling_prob=0.75, surface_normal_knn=7,
                                 surface_normal_ratio=0.5, trim_ratio=0.85, max_iterations=40,
                                 min_diff_rot=0.001, min_diff_trans=0.001, smooth_length=3)
        self._handle = None

    # -- laser_track.cpp:17
    def load_from_yaml(self, stream):
        """stream: file object, path or YAML text.  Unsupported modules raise (bad config), as
        PointMatcher's registrar does for unknown module names."""
        import yaml
        if hasattr(stream, "read"):
            doc = yaml.safe_load(stream.read())
        else:
            try:
                with open(stream) as f:
                    doc = yaml.safe_load(f.read())
            except (OSError, ValueError):
                doc = yaml.safe_load(stream)
        if not isinstance(doc, dict):
            raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml", "not a YAML mapping")
        # libpointmatcher's loadFromYaml starts from EMPTY chains: a section the file does not mention means "no such
        # module".  No reading filter = every point and no draw (reading_prob < 0), no outlier filter = every pair (ratio 1), no differential
        # checker = only the counter stops the loop; the modules the device loop cannot run without are required.
        ch = ChainConfig(reading_sampling_prob=-1.0, surface_normal_knn=7, trim_ratio=1.0,
                         min_diff_rot=-1.0, min_diff_trans=-1.0, smooth_length=1)
        seen = set()

        def modules(section):
            v = doc.get(section)
            if v is None:
                return []
            items = v if isinstance(v, list) else [v]
            out = []
            for it in items:
                if isinstance(it, str):
                    out.append((it, {}))
                elif isinstance(it, dict):
                    for k, p in it.items():
                        out.append((k, p or {}))
            return out

        for section, allowed in _SUPPORTED.items():
            for name, params in modules(section):
                if name not in allowed:
                    raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml",
                                     f"{section}: module {name} is not implemented on the HIP path")
                if name in seen and name != "KDTreeMatcher":
                    raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml", f"{section}: {name} given twice")
                seen.add(name)
                if name == "RandomSamplingDataPointsFilter":
                    ch.reading_sampling_prob = float(params.get("prob", 0.75))
                elif name == "SamplingSurfaceNormalDataPointsFilter":
                    ch.surface_normal_knn = int(params.get("knn", 7))
                    ch.surface_normal_ratio = float(params.get("ratio", 0.5))
                    if int(params.get("samplingMethod", 0)) != 0:
                        raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml", "samplingMethod != 0")
                elif name == "KDTreeMatcher":
                    if int(params.get("knn", 1)) != 1 or float(params.get("epsilon", 0)) != 0.0:
                        raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml",
                                         "only knn 1 / epsilon 0 is implemented")
                elif name == "TrimmedDistOutlierFilter":
                    ch.trim_ratio = float(params.get("ratio", 0.85))
                elif name == "CounterTransformationChecker":
                    ch.max_iterations = int(params.get("maxIterationCount", 40))
                elif name == "DifferentialTransformationChecker":
                    ch.min_diff_rot = float(params.get("minDiffRotErr", 0.001))
                    ch.min_diff_trans = float(params.get("minDiffTransErr", 0.001))
                    ch.smooth_length = int(params.get("smoothLength", 3))
        if modules("readingStepDataPointsFilters"):
            raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml", "readingStepDataPointsFilters")
        for need, why in (("SamplingSurfaceNormalDataPointsFilter", "it provides the normals"),
                          ("KDTreeMatcher", "the matcher"), ("PointToPlaneErrorMinimizer", "the error minimizer"),
                          ("CounterTransformationChecker", "the loop would not stop")):
            if need not in seen:
                raise LsgpuError(_lib.BAD_CONFIG, "load_from_yaml", f"{need} is required ({why})")
        # inspector / logger (yaml:32-44) only produce debug dumps: accepted and ignored
        self.chain = ch
        self._handle = None

    def _ensure_handle(self) -> IcpHandle:
        if self._handle is None:
            cfg = IcpConfig()
            _lib.lib().lsgpu_icp_config_yaml(C.byref(cfg))
            cfg.trim_ratio = self.chain.trim_ratio
            cfg.max_iterations = self.chain.max_iterations
            cfg.min_diff_rot = self.chain.min_diff_rot
            cfg.min_diff_trans = self.chain.min_diff_trans
            cfg.smooth_length = self.chain.smooth_length
            self._handle = IcpHandle(cfg, self.device)
        return self._handle

    # -- laser_track.cpp:496 / incremental_estimator.cpp:108
    def compute(self, reading_xyz1, reference_xyz1, T_init) -> np.ndarray:
        """T (4x4 float32) with p_reference = T p_reading.  Raises ConvergenceError."""
        h = self._ensure_handle()
        ch = self.chain
        T, st = h.compute(reading_xyz1, reference_xyz1, T_init, ch.reading_sampling_prob,
                          ch.surface_normal_knn, ch.surface_normal_ratio, ch.seed)
        self.last_stats = st
        return T
