"""Synthetic Velodyne HDL-64E scan generator.

The reference ships no data (rosbags/pcd are git-ignored, /root/reference/.gitignore), so the
benchmark and the parity tests ray-cast an HDL-64E beam pattern into an analytic street scene:
ground plane z=0, two walls y=+-8 m (6 m high), 24 boxes, 16 vertical cylinders, max range 120 m,
range noise N(0, 0.02 m).  Clouds come out in the layout the reference's ICP consumes:
``PointMatcher<float>::DataPoints.features`` = (dim+1) x N column-major == AoS x,y,z,1 float32
(laser_slam/include/laser_slam/common.hpp:14-17).

numpy only; no device code, no oracle dependency.
"""
from __future__ import annotations

import numpy as np

SENSOR_HEIGHT = 1.73
MAX_RANGE = 120.0
N_BEAMS = 64


def se3(tx=0.0, ty=0.0, tz=0.0, yaw=0.0, pitch=0.0, roll=0.0) -> np.ndarray:
    """4x4 float64 pose, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = (tx, ty, tz)
    return T


class Scene:
    """Analytic scene, seeded."""

    def __init__(self, seed: int = 1234, n_boxes: int = 24, n_cyl: int = 16, extent: float = 60.0):
        rng = np.random.default_rng(seed)
        self.wall_y = 8.0
        self.wall_h = 6.0
        # boxes: centre xy within +-extent in x, inside the street in y; size 0.5..4 m
        cx = rng.uniform(-extent, extent, n_boxes)
        cy = rng.uniform(-7.0, 7.0, n_boxes)
        sx = rng.uniform(0.5, 4.0, n_boxes)
        sy = rng.uniform(0.5, 2.0, n_boxes)
        sz = rng.uniform(0.5, 3.0, n_boxes)
        # keep the sensor's immediate surroundings free
        near = (np.abs(cx) < 4.0) & (np.abs(cy) < 3.0)
        cx[near] += 8.0
        self.box_lo = np.stack([cx - sx / 2, cy - sy / 2, np.zeros(n_boxes)], 1)
        self.box_hi = np.stack([cx + sx / 2, cy + sy / 2, sz], 1)
        kx = rng.uniform(-extent, extent, n_cyl)
        ky = rng.uniform(-7.5, 7.5, n_cyl)
        near = (np.abs(kx) < 4.0) & (np.abs(ky) < 3.0)
        kx[near] -= 9.0
        self.cyl_c = np.stack([kx, ky], 1)
        self.cyl_r = rng.uniform(0.15, 0.6, n_cyl)
        self.cyl_h = rng.uniform(2.0, 8.0, n_cyl)

    def raycast(self, o: np.ndarray, d: np.ndarray) -> np.ndarray:
        """o (3,), d (N,3) unit vectors in world frame -> range (N,), inf where nothing is hit."""
        n = d.shape[0]
        t = np.full(n, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            # ground
            tg = -o[2] / d[:, 2]
            tg[~(tg > 0)] = np.inf
            t = np.minimum(t, tg)
            # walls
            for wy in (self.wall_y, -self.wall_y):
                tw = (wy - o[1]) / d[:, 1]
                z = o[2] + tw * d[:, 2]
                tw[~((tw > 0) & (z >= 0) & (z <= self.wall_h))] = np.inf
                t = np.minimum(t, tw)
            # boxes (slab test)
            inv = 1.0 / d
            for lo, hi in zip(self.box_lo, self.box_hi):
                t1 = (lo - o) * inv
                t2 = (hi - o) * inv
                tn = np.nanmax(np.minimum(t1, t2), axis=1)
                tf = np.nanmin(np.maximum(t1, t2), axis=1)
                hit = (tf >= tn) & (tf > 0)
                tb = np.where(tn > 0, tn, tf)
                tb[~hit] = np.inf
                t = np.minimum(t, tb)
            # vertical cylinders (side surface only)
            for c, r, h in zip(self.cyl_c, self.cyl_r, self.cyl_h):
                ox, oy = o[0] - c[0], o[1] - c[1]
                a = d[:, 0] ** 2 + d[:, 1] ** 2
                b = 2 * (ox * d[:, 0] + oy * d[:, 1])
                cc = ox * ox + oy * oy - r * r
                disc = b * b - 4 * a * cc
                sq = np.sqrt(np.where(disc > 0, disc, np.nan))
                tc = (-b - sq) / (2 * a)
                z = o[2] + tc * d[:, 2]
                tc[~((tc > 0) & (z >= 0) & (z <= h))] = np.inf
                t = np.minimum(t, np.nan_to_num(tc, nan=np.inf))
        return t


def hdl64_directions(n_az: int) -> np.ndarray:
    """(64*n_az, 3) unit ray directions in the sensor frame, azimuth-major (one firing = 64 beams)."""
    elev = np.deg2rad(np.linspace(2.0, -24.8, N_BEAMS))
    az = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    ce, se_ = np.cos(elev), np.sin(elev)
    d = np.empty((n_az, N_BEAMS, 3))
    d[:, :, 0] = np.cos(az)[:, None] * ce[None, :]
    d[:, :, 1] = np.sin(az)[:, None] * ce[None, :]
    d[:, :, 2] = se_[None, :]
    return d.reshape(-1, 3)


def hdl64_scan(scene: Scene, T_w_s: np.ndarray, n_az: int, noise_seed: int,
               sigma: float = 0.02) -> np.ndarray:
    """One scan from sensor pose T_w_s (4x4, world<-sensor).  Returns (N,4) float32 x,y,z,1 in the
    sensor frame; rays with no return inside MAX_RANGE are dropped."""
    ds = hdl64_directions(n_az)
    R = T_w_s[:3, :3]
    o = T_w_s[:3, 3]
    dw = ds @ R.T
    rng_ = scene.raycast(o, dw)
    ok = rng_ < MAX_RANGE
    rng_ = rng_[ok]
    noise = np.random.default_rng(noise_seed).normal(0.0, sigma, rng_.shape[0])
    r = rng_ + noise
    pts = ds[ok] * r[:, None]
    out = np.ones((pts.shape[0], 4), np.float32)
    out[:, :3] = pts.astype(np.float32)
    return out


def scan_job(job):
    """(scene_seed, pose 4x4, n_az, noise_seed) -> scan; a top-level function so that a process pool can run it."""
    scene_seed, T, n_az, noise_seed = job
    return hdl64_scan(Scene(scene_seed), np.asarray(T, np.float64), n_az, noise_seed)


def scan_pair(n_az: int, scene_seed: int = 1234, noise_seeds=(1, 2), guess_seed: int = 7,
              step=(0.8, 0.05, 0.0), yaw_deg: float = 2.0, pitch_deg: float = 0.2,
              guess_err=(0.3, 1.5)):
    """(reference_scan, reading_scan, T_true, T_init): T maps reading(sensor 2) -> reference(sensor 1).

    pose2 = pose1 o (step, yaw, pitch); T_init = truth perturbed by (guess_err[0] m, guess_err[1] deg)
    in a seeded random direction."""
    scene = Scene(scene_seed)
    T1 = se3(0.0, 0.0, SENSOR_HEIGHT)
    T12 = se3(step[0], step[1], step[2], yaw=np.deg2rad(yaw_deg), pitch=np.deg2rad(pitch_deg))
    T2 = T1 @ T12
    ref = hdl64_scan(scene, T1, n_az, noise_seeds[0])
    rd = hdl64_scan(scene, T2, n_az, noise_seeds[1])
    rng = np.random.default_rng(guess_seed)
    dt = rng.normal(size=3)
    dt *= guess_err[0] / np.linalg.norm(dt)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = np.deg2rad(guess_err[1])
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    dR = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    P = np.eye(4)
    P[:3, :3] = dR
    P[:3, 3] = dt
    T_init = P @ T12
    return ref, rd, T12.astype(np.float64), T_init.astype(np.float64)


def colmajor(T: np.ndarray) -> np.ndarray:
    """4x4 -> 16 float32 column-major (Eigen default, what the C-ABI takes)."""
    return np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(16)


def from_colmajor(t16) -> np.ndarray:
    return np.asarray(t16, np.float64).reshape(4, 4).T.copy()


def pose_error(Ta: np.ndarray, Tb: np.ndarray):
    """(translation error m, rotation angle rad) between two 4x4 transforms."""
    dt = float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    Rd = Ta[:3, :3].T @ Tb[:3, :3]
    # robust small-angle: |log(R)| from the skew part
    s = np.array([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]]) * 0.5
    sn = float(np.linalg.norm(s))
    c = (np.trace(Rd) - 1.0) * 0.5
    return dt, float(np.arctan2(sn, c))


# ---------------------------------------------------------------- BASELINE configs[4]: a long sequence
class FieldScene(Scene):
    """Open field for the figure-eight sequence (SURVEY.md §8d, config 5 of the survey = configs[4]): ground plane,
    no street walls, boxes and cylinders scattered over [-ext_x, ext_x] x [-ext_y, ext_y] at about the density of the
    street scene; nothing closer than `clear` metres to the driven path.  Ray casting only looks at the primitives
    within `cull` metres of the sensor (fixed per scan, identical for every consumer of the scan)."""

    def __init__(self, seed: int, path_xy: np.ndarray, ext_x: float, ext_y: float, n_boxes: int = 700, n_cyl: int = 500,
                 clear: float = 2.5, cull: float = 70.0):
        rng = np.random.default_rng(seed)
        self.wall_y = None
        self.wall_h = 0.0
        self.cull = cull
        step = max(1, len(path_xy) // 4000)
        path = path_xy[::step]

        def free(xy, r):
            d = np.sqrt(((xy[:, None, :] - path[None, :, :]) ** 2).sum(-1)).min(1)
            return d > clear + r

        cx, cy = rng.uniform(-ext_x, ext_x, n_boxes), rng.uniform(-ext_y, ext_y, n_boxes)
        sx, sy, sz = rng.uniform(0.5, 4.0, n_boxes), rng.uniform(0.5, 4.0, n_boxes), rng.uniform(0.5, 3.0, n_boxes)
        ok = free(np.stack([cx, cy], 1), np.hypot(sx, sy) / 2)
        self.box_lo = np.stack([cx - sx / 2, cy - sy / 2, np.zeros(n_boxes)], 1)[ok]
        self.box_hi = np.stack([cx + sx / 2, cy + sy / 2, sz], 1)[ok]
        kx, ky = rng.uniform(-ext_x, ext_x, n_cyl), rng.uniform(-ext_y, ext_y, n_cyl)
        kr, kh = rng.uniform(0.15, 0.6, n_cyl), rng.uniform(2.0, 8.0, n_cyl)
        ok = free(np.stack([kx, ky], 1), kr)
        self.cyl_c, self.cyl_r, self.cyl_h = np.stack([kx, ky], 1)[ok], kr[ok], kh[ok]

    def raycast(self, o: np.ndarray, d: np.ndarray) -> np.ndarray:
        t = np.full(d.shape[0], np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = -o[2] / d[:, 2]
            tg[~(tg > 0)] = np.inf
            t = np.minimum(t, tg)
            inv = 1.0 / d
            bc = (self.box_lo[:, :2] + self.box_hi[:, :2]) / 2
            for k in np.nonzero(np.hypot(bc[:, 0] - o[0], bc[:, 1] - o[1]) < self.cull)[0]:
                t1 = (self.box_lo[k] - o) * inv
                t2 = (self.box_hi[k] - o) * inv
                lo_, hi_ = np.minimum(t1, t2), np.maximum(t1, t2)
                tn = np.fmax(np.fmax(lo_[:, 0], lo_[:, 1]), lo_[:, 2])   # (fmax / fmin skip NaNs like nanmax / nanmin)
                tf = np.fmin(np.fmin(hi_[:, 0], hi_[:, 1]), hi_[:, 2])
                hit = (tf >= tn) & (tf > 0)
                tb = np.where(tn > 0, tn, tf)
                tb[~hit] = np.inf
                t = np.minimum(t, tb)
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            for k in np.nonzero(np.hypot(self.cyl_c[:, 0] - o[0], self.cyl_c[:, 1] - o[1]) < self.cull)[0]:
                c, r, h = self.cyl_c[k], self.cyl_r[k], self.cyl_h[k]
                ox, oy = o[0] - c[0], o[1] - c[1]
                b = 2 * (ox * d[:, 0] + oy * d[:, 1])
                disc = b * b - 4 * a * (ox * ox + oy * oy - r * r)
                sq = np.sqrt(np.where(disc > 0, disc, np.nan))
                tc = (-b - sq) / (2 * a)
                z = o[2] + tc * d[:, 2]
                tc[~((tc > 0) & (z >= 0) & (z <= h))] = np.inf
                t = np.minimum(t, np.nan_to_num(tc, nan=np.inf))
        return t


def figure_eight(n_poses: int, step_m: float = 0.8, laps: int = 2):
    """Ground-truth poses (4x4, world <- sensor) on a figure-eight (lemniscate of Gerono) driven `laps` times,
    `step_m` apart on average, heading along the path.  Returns (poses, half_extent_x, half_extent_y)."""
    A = n_poses * step_m / (6.0973 * laps)          # curve length of one lap = 6.0973 A
    # equal arc-length parametrisation (numerically), so that consecutive poses are step_m apart everywhere
    tt = np.linspace(0.0, 2 * np.pi * laps, 200 * n_poses + 1)
    x, y = A * np.sin(tt), A * np.sin(tt) * np.cos(tt)
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    ti = np.interp(np.arange(n_poses) * (s[-1] / n_poses), s, tt)
    poses = []
    for t in ti:
        dx, dy = A * np.cos(t), A * np.cos(2 * t)
        poses.append(se3(A * np.sin(t), A * np.sin(t) * np.cos(t), SENSOR_HEIGHT, yaw=np.arctan2(dy, dx)))
    return poses, float(A), float(A / 2)


def drifting_odometry(truth, seed: int, sigma_t: float = 0.02, sigma_r_deg: float = 0.1, bias=(0.004, 0.0, 0.0, 0.02)):
    """Odometry pose measurements: truth increments perturbed by N(0, [sigma_t m, sigma_r deg]) per step
    plus a small constant bias (bias = dx, dy, dz [m], dyaw [deg]) -- dead reckoning drifts by metres over a lap."""
    rng = np.random.default_rng(seed)
    odom = [truth[0].copy()]
    for i in range(1, len(truth)):
        rel = np.linalg.inv(truth[i - 1]) @ truth[i]
        e = rng.normal(0.0, 1.0, 6)
        err = se3(bias[0] + sigma_t * e[0], bias[1] + sigma_t * e[1], bias[2] + 0.25 * sigma_t * e[2],
                  yaw=np.deg2rad(bias[3] + sigma_r_deg * e[3]), pitch=np.deg2rad(0.25 * sigma_r_deg * e[4]),
                  roll=np.deg2rad(0.25 * sigma_r_deg * e[5]))
        odom.append(odom[-1] @ rel @ err)
    return odom


def quat_wxyz(T: np.ndarray):
    """Unit quaternion (w, x, y, z) of a 4x4 pose (Shepperd's method)."""
    R = T[:3, :3]
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.asarray(q, np.float64)
    return q / np.linalg.norm(q)
