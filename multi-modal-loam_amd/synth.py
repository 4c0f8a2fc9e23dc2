"""Synthetic LiDAR scans for parity tests and bench.py.

Scene: axis-aligned box room 20 x 15 x 3 m with box pillars, sensor 1 m above the floor, constant-twist
motion (0.5 m/s, 0.2 rad/s yaw, 0.1 s per scan), range noise N(0, 0.01 m), numpy default_rng(1234 + scan).
VLP-16: n_rings x n_az points, azimuth-major (ring inner), pitch -15..+15 deg; Livox Horizon: 24 000 points on
6 lines, Lissajous sweep inside 81.7 x 25.1 deg, offset_time linear over 0.1 s.

Pure numpy, no reference code, no oracle: this only produces *inputs*.
"""
import numpy as np

ROOM_MIN = np.array([-10.0, -7.5, 0.0])
ROOM_MAX = np.array([10.0, 7.5, 3.0])
# (cx, cy, half_x, half_y) pillars, full height
PILLARS = [
    (4.0, 3.0, 0.2, 0.2), (-5.0, 4.5, 0.25, 0.25), (6.5, -4.0, 0.3, 0.2), (-3.0, -5.0, 0.2, 0.3),
    (8.0, 1.0, 0.15, 0.4), (2.0, -2.5, 0.2, 0.2), (-7.5, -1.0, 0.3, 0.3), (5.0, 6.0, 0.5, 0.15),
    (7.0, 0.0, 0.2, 0.2), (6.0, 2.0, 0.15, 0.15), (9.0, -2.0, 0.25, 0.25),
]

LIVOX_DTYPE = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
                        ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1"), ("_pad", "u1")])
assert LIVOX_DTYPE.itemsize == 20


def pose_at(k, dt=0.1, v=0.5, w=0.2):
    """Sensor pose (R, t) of scan k under a constant twist (forward v, yaw rate w), start (0,0,1)."""
    th = w * dt * k
    if abs(w) < 1e-12:
        x, y = v * dt * k, 0.0
    else:
        x = v / w * np.sin(th)
        y = v / w * (1.0 - np.cos(th))
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    return R, np.array([x, y, 1.0])


GRAVITY = np.array([0.0, 0.0, -9.805])  # the vector of Cost_NavState_PRV_Bias: P_j = P_i + V_i dt + g dt^2 / 2 + R_i dP


def velocity_at(k, dt=0.1, v=0.5, w=0.2):
    """World-frame velocity of the sensor at scan time k (constant-twist motion of pose_at)."""
    R, _ = pose_at(k, dt, v, w)
    return R @ np.array([v, 0.0, 0.0])


def imu_samples(k0, k1, rate=200, dt=0.1, v=0.5, w=0.2, gnorm=9.805):
    """IMU messages between scan times k0 and k1 for the motion of pose_at: rows of (angular_velocity xyz,
    linear_acceleration xyz in units of g as the Livox IMU reports it, dt to the previous message).  Each sample
    holds over the following dt (the forward-Euler convention of IMUIntegrator::PreIntegration)."""
    n = int(round((k1 - k0) * dt * rate))
    h = (k1 - k0) * dt / n
    out = np.zeros((n, 7))
    for i in range(n):
        out[i, 0:3] = [0.0, 0.0, w]
        # body-frame specific force: centripetal acceleration (0, v w, 0) minus gravity, both constant in the body frame
        out[i, 3:6] = (np.array([0.0, v * w, 0.0]) - GRAVITY) / gnorm
        out[i, 6] = h
    return out


def _raycast(origin, dirs):
    """Nearest hit range for unit directions `dirs` (n,3) from `origin` ((3,) or (n,3)) inside the room."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs
        # room: we are inside, so the exit distance is the min over axes of the positive slab distance
        t1 = (ROOM_MIN - origin) * inv
        t2 = (ROOM_MAX - origin) * inv
        tfar = np.maximum(t1, t2)
        rng = np.min(tfar, axis=1)
        for (cx, cy, hx, hy) in PILLARS:
            bmin = np.array([cx - hx, cy - hy, 0.0])
            bmax = np.array([cx + hx, cy + hy, 3.0])
            a = (bmin - origin) * inv
            b = (bmax - origin) * inv
            tn = np.max(np.minimum(a, b), axis=1)
            tf = np.min(np.maximum(a, b), axis=1)
            hit = (tn <= tf) & (tn > 0.0)
            rng = np.where(hit & (tn < rng), tn, rng)
    return rng


def sweep_motion(k):
    """(dR, dt): sensor motion over sweep k, start frame -> end frame (the arguments of RemoveLidarDistortion)."""
    rel = np.linalg.inv(pose_matrix(k - 1)) @ pose_matrix(k)
    return rel[:3, :3].copy(), rel[:3, 3].copy()


def _sweep_rays(k, s, d, motion):
    """World-frame origins and directions for sensor-frame unit rays d fired at in-sweep time s in [0,1]."""
    if not motion:
        R, t = pose_at(k)
        return np.broadcast_to(t, d.shape), d @ R.T
    from scipy.spatial.transform import Rotation as Rsc
    T0 = pose_matrix(k - 1)
    dR, dt = sweep_motion(k)
    rv = Rsc.from_matrix(dR).as_rotvec()
    Rs = Rsc.from_rotvec(s[:, None] * rv[None, :]).as_matrix()          # slerp(identity, dR, s)
    Rw = np.einsum("ij,njk->nik", T0[:3, :3], Rs)
    org = (s[:, None] * dt[None, :]) @ T0[:3, :3].T + T0[:3, 3]
    return org, np.einsum("nij,nj->ni", Rw, d)


def velo_scan(k, n_rings=16, n_az=1800, pitch0=-15.0, pitch_step=2.0, noise=0.01, motion=False):
    """VLP-16 style scan k in the sensor frame: float32 (n_rings*n_az, 4) = x,y,z,intensity.
    motion=True: every azimuth column is fired from the pose interpolated over the sweep (pose k-1 -> k)."""
    rng = np.random.default_rng(1234 + k)
    az = -np.linspace(0.0, 2.0 * np.pi, n_az, endpoint=False) - 0.01  # clockwise like a Velodyne
    pitch = np.deg2rad(pitch0 + pitch_step * np.arange(n_rings))
    azg, pg = np.meshgrid(az, pitch, indexing="ij")  # azimuth-major
    d = np.stack([np.cos(pg) * np.cos(azg), np.cos(pg) * np.sin(azg), np.sin(pg)], axis=-1).reshape(-1, 3)
    s = np.repeat(np.arange(n_az) / float(n_az), n_rings)
    org, dw = _sweep_rays(k, s, d, motion)
    r = _raycast(org, dw)
    r = r + rng.normal(0.0, noise, size=r.shape)
    pts = d * r[:, None]
    inten = rng.uniform(0.0, 100.0, size=r.shape)
    return np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)


def livox_scan(k, n=24000, n_lines=6, noise=0.01, motion=False):
    """Livox Horizon style scan k (structured array LIVOX_DTYPE, 20 B records) in the sensor frame."""
    rng = np.random.default_rng(91234 + k)
    j = np.arange(n)
    line = (j % n_lines).astype(np.uint8)
    tt = j / float(n)
    half_h = np.deg2rad(81.7 / 2.0) * 0.98
    half_v = np.deg2rad(25.1 / 2.0) * 0.80
    az = half_h * np.sin(2.0 * np.pi * 5.0 * tt + 0.3)
    el = half_v * np.sin(2.0 * np.pi * 3.7 * tt + 1.1) + np.deg2rad(0.28) * (line.astype(np.float64) - 2.5)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1)
    org, dw = _sweep_rays(k, np.linspace(0.0, 1.0, n), d, motion)
    r = _raycast(org, dw)
    r = r + rng.normal(0.0, noise, size=r.shape)
    pts = d * r[:, None]
    out = np.zeros(n, dtype=LIVOX_DTYPE)
    out["offset_time"] = np.round(np.linspace(0.0, 0.1, n) * 1e9).astype(np.uint32)
    out["x"], out["y"], out["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    out["reflectivity"] = rng.integers(0, 256, size=n).astype(np.uint8)
    out["line"] = line
    return out


# ---- sensor-faithful streams -------------------------------------------------------------------------------------------
# The scans above are an idealised grid.  The two generators below reproduce what the drivers of the two sensors put on the
# wire, i.e. the properties of real bags that bit-parity is sensitive to (unionFeatureExtract.cpp:369-388 non-finite points,
# :453-479 ties in the partition sorts, :985-998 Livox record filter, :1133-1195 start / end azimuth and the half-turn flag):
#   VLP-16 (velodyne_pointcloud): points in FIRING order (16 lasers per firing in the interleaved elevation order
#     -15, +1, -13, +3 ... -1, +15 deg, 55.296 us per firing, 2.304 us between lasers), the azimuth of every point interpolated
#     from encoder readings quantised to 0.01 deg with a slowly wobbling rotor speed; a scan is cut at a packet boundary
#     (76 packets x 24 firings = 1824 firings: a little MORE than one revolution, so the end azimuth overlaps the start);
#     ranges in units of 2 mm, calibrated reflectivity as an integer 0..255; no-returns either absent (the driver's default),
#     NaN (organised clouds) or (0,0,0).
#   Livox Horizon (livox_ros_driver CustomMsg): 6 lines fired together along a Risley-prism rosette, coordinates in whole
#     millimetres, integer reflectivity with retro-reflector saturation, `tag` noise / return-number bits (ignored by the
#     reference -- and so here), no-returns as (0,0,0) records, stray `line` ids above 5, offset_time in whole nanoseconds of
#     the 240 kHz point clock.
VLP16_LASER_DEG = np.array([-15, 1, -13, 3, -11, 5, -9, 7, -7, 9, -5, 11, -3, 13, -1, 15], dtype=np.float64)


def _reflectivity(rng, world_pts, n):
    """Integer reflectivity 0..255: a few large-scale surface patches (plateaus -> ties), speckle, retro-reflective stripes."""
    base = 18.0 + 10.0 * np.sin(0.9 * world_pts[:, 0]) * np.cos(0.7 * world_pts[:, 1]) + 6.0 * np.sin(2.3 * world_pts[:, 2])
    refl = np.round(base + rng.normal(0.0, 1.2, n))
    retro = (np.abs(np.mod(world_pts[:, 0] + 0.37 * world_pts[:, 1], 3.1)) < 0.12) | (np.abs(np.mod(world_pts[:, 2], 1.4) - 0.7) < 0.02)
    refl = np.where(retro, 255.0, refl)
    return np.clip(refl, 0.0, 255.0)


def velo_scan_vlp16(k, n_firings=1824, noise=0.01, motion=False, dropout="skip", drop_rate=0.015, seed=None, firing_step_scale=1.0):
    """One VLP-16 revolution as velodyne_pointcloud publishes it (see the block comment above): float32 (n, 4) x, y, z,
    intensity in firing order.  dropout: "skip" (no-returns absent), "nan" or "zero".  firing_step_scale > 1 turns the rotor
    faster (fewer firings per revolution: reduced scans for fixtures)."""
    rng = np.random.default_rng((4321 + k) if seed is None else seed)
    f = np.arange(n_firings)
    # rotor: 600 rpm nominal with a slow +-0.3 % wobble; encoder readings in 0.01 deg; 0.19906 deg per firing
    rate = 3600.0 * firing_step_scale * (1.0 + 0.003 * np.sin(2.0 * np.pi * f / n_firings * 1.7 + 0.4 * k))
    enc = np.cumsum(np.concatenate([[rng.uniform(0.0, 360.0)], rate[:-1] * 55.296e-6]))
    enc = np.round(enc * 100.0) / 100.0
    dstep = np.diff(enc, append=enc[-1] + rate[-1] * 55.296e-6)
    lasers = np.arange(16)
    az_deg = enc[:, None] + dstep[:, None] * (lasers[None, :] * 2.304 / 55.296)            # (firings, 16)
    az = -np.deg2rad(az_deg)                                                                 # clockwise seen from above
    pitch = np.deg2rad(VLP16_LASER_DEG)[None, :] * np.ones((n_firings, 1))
    d = np.stack([np.cos(pitch) * np.cos(az), np.cos(pitch) * np.sin(az), np.sin(pitch)], axis=-1).reshape(-1, 3)
    s = ((f[:, None] * 55.296e-6 + lasers[None, :] * 2.304e-6) / (n_firings * 55.296e-6)).reshape(-1)
    org, dw = _sweep_rays(k, s, d, motion)
    r = _raycast(org, dw) + rng.normal(0.0, noise, size=len(d))
    r = np.round(r / 0.002) * 0.002                                                          # 2 mm range units
    world = org + dw * r[:, None]
    inten = _reflectivity(rng, world, len(d))
    # no-returns: random misses, a glass-like azimuth sector, everything beyond the 130 m / below the 0.4 m driver limits
    miss = rng.random(len(d)) < drop_rate
    sector = (np.mod(az_deg.reshape(-1) - 200.0, 360.0) < 6.0) & (pitch.reshape(-1) > np.deg2rad(2.0))
    miss |= sector | (r > 130.0) | (r < 0.4)
    pts = (d * r[:, None]).astype(np.float32)
    out = np.concatenate([pts, inten[:, None].astype(np.float32)], axis=1)
    if dropout == "skip":
        return np.ascontiguousarray(out[~miss])
    out[miss, :3] = np.nan if dropout == "nan" else 0.0
    if dropout == "zero":
        out[miss, 3] = 0.0
    return out


def livox_scan_horizon(k, n=24000, noise=0.01, motion=False, drop_rate=0.04, stray_lines=True, seed=None):
    """One 100 ms Horizon frame as livox_ros_driver publishes it (CustomMsg records, LIVOX_DTYPE)."""
    rng = np.random.default_rng((94321 + k) if seed is None else seed)
    n_lines = 6
    j = np.arange(n)
    col = j // n_lines                               # the six beams of a column fire together
    line = (j % n_lines).astype(np.uint8)
    tt = col / float((n + n_lines - 1) // n_lines)
    # Risley rosette: two prisms turning at incommensurate rates, stretched to the 81.7 x 25.1 deg field of view
    w1, w2 = 2.0 * np.pi * 7.0, -2.0 * np.pi * 4.27
    ph = 0.31 * k
    half_h = np.deg2rad(81.7 / 2.0) * 0.97
    half_v = np.deg2rad(25.1 / 2.0) * 0.78
    az = half_h * 0.5 * (np.cos(w1 * tt + ph) + np.cos(w2 * tt + 1.3 * ph))
    el = half_v * 0.5 * (np.sin(w1 * tt + ph) + np.sin(w2 * tt + 1.3 * ph)) + np.deg2rad(0.28) * (line.astype(np.float64) - 2.5)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1)
    off_ns = np.round(j * (1e9 / 240000.0)).astype(np.uint32)   # 240 kHz point clock, whole nanoseconds
    srel = off_ns.astype(np.float64) / float(off_ns[-1])
    org, dw = _sweep_rays(k, srel, d, motion)
    r = _raycast(org, dw) + rng.normal(0.0, noise, size=n)
    world = org + dw * r[:, None]
    pts = np.round(d * r[:, None] * 1000.0) / 1000.0                                          # whole millimetres
    out = np.zeros(n, dtype=LIVOX_DTYPE)
    out["offset_time"] = off_ns
    out["x"], out["y"], out["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    out["reflectivity"] = _reflectivity(rng, world, n).astype(np.uint8)
    out["line"] = line
    # tag: bits 0-1 spatial-position noise confidence, 2-3 intensity noise confidence, 4-5 return number
    tag = (rng.random(n) < 0.03) * rng.integers(1, 4, n) + ((rng.random(n) < 0.02) * rng.integers(1, 4, n) << 2) + ((rng.random(n) < 0.1) * 1 << 4)
    out["tag"] = tag.astype(np.uint8)
    miss = rng.random(n) < drop_rate
    # bursts of no-returns (a dark surface): whole stretches of a column sequence
    for _ in range(6):
        a = int(rng.integers(0, n - 400))
        miss[a:a + int(rng.integers(30, 400))] = True
    out["x"][miss] = 0.0
    out["y"][miss] = 0.0
    out["z"][miss] = 0.0
    out["reflectivity"][miss] = 0
    if stray_lines:
        stray = rng.random(n) < 0.002
        out["line"][stray] = rng.integers(6, 9, int(stray.sum())).astype(np.uint8)
    return out


def transform(T, xyz):
    """Apply a 4x4 to (n,3) float64."""
    return xyz @ T[:3, :3].T + T[:3, 3]


def pose_matrix(k):
    R, t = pose_at(k)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def voxel_filter(xyz, leaf):
    """Centroid per occupied voxel (numpy): the role of downSizeFilter on the local map (Estimator.cpp:1630-1637).
    Only used to prepare INPUT maps for bench / tests; the product's own voxel kernel is k_voxel."""
    xyz = np.asarray(xyz, dtype=np.float32)
    if len(xyz) == 0:
        return xyz
    ijk = np.floor(xyz / np.float32(leaf)).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    order = np.argsort(key, kind="stable")
    key = key[order]
    starts = np.concatenate([[0], np.nonzero(np.diff(key))[0] + 1])
    sums = np.add.reduceat(xyz[order].astype(np.float64), starts, axis=0)
    cnt = np.diff(np.concatenate([starts, [len(key)]]))
    return (sums / cnt[:, None]).astype(np.float32)


def tile_offsets(count, tile=(40.0, 30.0)):
    """Offsets of the first `count` lattice tiles grow_map lays copies on (tile 0 = the original, at the origin)."""
    side = int(np.ceil(np.sqrt(max(count, 1))))
    cells = [(i, j) for i in range(-side, side + 1) for j in range(-side, side + 1) if (i, j) != (0, 0)]
    cells.sort(key=lambda c: (max(abs(c[0]), abs(c[1])), c))
    out = [np.zeros(3)]
    for r in range(count - 1):
        i, j = cells[r]
        out.append(np.array([i * tile[0], j * tile[1], 0.0]))
    return out


def grow_map(xyz, target, seed=7, jitter=0.02, tile=(40.0, 30.0)):
    """Grow a (m,3) float32 feature cloud to `target` points ("replicated / jittered").

    The reference voxel-filters its local map on every update (Estimator.cpp:1630-1637), so a map of 200 k points
    is spatially LARGE, never dense: copies are therefore laid out on a lattice of room-sized tiles around the
    original (which stays in place, so the scans still register against it) and jittered by a few centimetres,
    keeping the per-voxel density of a real map."""
    xyz = np.asarray(xyz, dtype=np.float32)
    if len(xyz) >= target:
        return xyz[:target].copy()
    rng = np.random.default_rng(seed)
    reps = int(np.ceil(target / len(xyz)))
    out = [xyz]
    side = int(np.ceil(np.sqrt(reps)))
    cells = [(i, j) for i in range(-side, side + 1) for j in range(-side, side + 1) if (i, j) != (0, 0)]
    cells.sort(key=lambda c: (max(abs(c[0]), abs(c[1])), c))
    for r in range(reps - 1):
        i, j = cells[r]
        off = np.array([i * tile[0], j * tile[1], 0.0], dtype=np.float32)
        out.append(xyz + off + rng.normal(0.0, jitter, size=xyz.shape).astype(np.float32))
    return np.concatenate(out, axis=0)[:target].astype(np.float32)
