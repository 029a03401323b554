"""Cloud files a libpointmatcher user can load: the host-side twin of ``DataPoints::save`` as LaserTrack uses it for
its debug dumps (laser_slam/src/laser_track.cpp:504-513: ``last_scan.scan.save("/tmp/last_scan.vtk")`` ...).

Legacy ASCII VTK POLYDATA in the layout libpointmatcher's VTK inspector writes (POINTS / VERTICES / POINT_DATA, the
``normals`` descriptor as NORMALS) and CSV with an ``x,y,z[,nx,ny,nz]`` header, both of which
``PointMatcher<float>::DataPoints::load`` reads.  Numbers are printed with nine significant digits ("%.9g"): a float32
survives the round trip bit for bit (upstream's own writer prints Eigen's default six and is lossy).  The C++ mirror
writes the same bytes (laser_slam_amd/cpp/include/laser_slam_amd/cloud_io.hpp; tests/test_cpp_mirror.py compares).
"""
from __future__ import annotations

import numpy as np


def _g(v) -> str:
    return "%.9g" % float(v)


def vtk_text(xyz1: np.ndarray, normals: np.ndarray | None = None) -> str:
    xyz1 = np.asarray(xyz1, np.float32)
    n = xyz1.shape[0]
    out = ["# vtk DataFile Version 3.0", "File created by libpointmatcher", "ASCII", "DATASET POLYDATA",
           "POINTS %d float" % n]
    out += ["%s %s %s" % (_g(p[0]), _g(p[1]), _g(p[2])) for p in xyz1]
    out.append("VERTICES %d %d" % (n, 2 * n))
    out += ["1 %d" % i for i in range(n)]
    out.append("POINT_DATA %d" % n)
    if normals is not None and len(normals):
        nrm = np.asarray(normals, np.float32).reshape(n, 3)
        out.append("NORMALS normals float")
        out += ["%s %s %s" % (_g(v[0]), _g(v[1]), _g(v[2])) for v in nrm]
    return "\n".join(out) + "\n"


def csv_text(xyz1: np.ndarray, normals: np.ndarray | None = None) -> str:
    xyz1 = np.asarray(xyz1, np.float32)
    n = xyz1.shape[0]
    if normals is not None and len(normals):
        nrm = np.asarray(normals, np.float32).reshape(n, 3)
        rows = ["x,y,z,nx,ny,nz"] + [",".join(_g(v) for v in (*p[:3], *q)) for p, q in zip(xyz1, nrm)]
    else:
        rows = ["x,y,z"] + [",".join(_g(v) for v in p[:3]) for p in xyz1]
    return "\n".join(rows) + "\n"


def save_vtk(path: str, xyz1, normals=None) -> None:
    with open(path, "w") as f:
        f.write(vtk_text(xyz1, normals))


def save_csv(path: str, xyz1, normals=None) -> None:
    with open(path, "w") as f:
        f.write(csv_text(xyz1, normals))


def load_vtk(path: str):
    """Reads back what save_vtk wrote (tests): (xyz1 float32 (n,4), normals float32 (n,3) or None)."""
    with open(path) as f:
        lines = f.read().split("\n")
    i = next(k for k, l in enumerate(lines) if l.startswith("POINTS "))
    n = int(lines[i].split()[1])
    pts = np.array([[np.float32(t) for t in l.split()] for l in lines[i + 1:i + 1 + n]], np.float32).reshape(n, 3)
    xyz1 = np.ones((n, 4), np.float32)
    xyz1[:, :3] = pts
    nrm = None
    for k, l in enumerate(lines):
        if l.startswith("NORMALS "):
            nrm = np.array([[np.float32(t) for t in r.split()] for r in lines[k + 1:k + 1 + n]], np.float32).reshape(n, 3)
    return xyz1, nrm
