"""Known answers of the libm the reference links: atanf / atan2f of THIS image's glibc (2.35; the same fdlibm float routines as
melodic's 2.27, see oracle/libm_f32.h), called through ctypes on inputs that exercise every branch of the two routines -- the
reduction thresholds of atanf +- a few ulps, ratios y / x on those thresholds, octant edges, axes, signed zeros, denormals,
infinities, exponent gaps beyond 2^60 -- plus lidar-like coordinates.  Data only: tests/golden/libm_f32_kat.npz.

    python tests/golden/make_libm_kat.py
"""
import ctypes
import os

import numpy as np

libm = ctypes.CDLL("libm.so.6")
libm.atanf.restype = ctypes.c_float
libm.atanf.argtypes = [ctypes.c_float]
libm.atan2f.restype = ctypes.c_float
libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


def ulps(v, k):
    return (np.float32(v).view(np.int32) + np.int32(k)).view(np.float32)


def main():
    rng = np.random.default_rng(20260929)
    thr = [0.4375, 0.6875, 1.1875, 2.4375, 2.0 ** -29, 2.0 ** 25, 1.0, 0.5, 1.5]
    a = [ulps(t, k) for t in thr for k in range(-4, 5)]
    a += [np.float32(v) for v in (0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 1.1754944e-38, 3.4e38, -3.4e38, 1e-10, 1e10)]
    a += list((rng.normal(size=4000) * 10.0 ** rng.uniform(-3, 3, 4000)).astype(np.float32))
    a = np.array(a, np.float32)
    a = np.concatenate([a, -a])
    # pairs
    x = rng.uniform(-120, 120, 6000).astype(np.float32)
    y = rng.uniform(-120, 120, 6000).astype(np.float32)
    xs, ys = [x], [y]
    xe = (rng.uniform(0.5, 100, 600)).astype(np.float32)
    for t in thr[:4] + [1.0, 0.41421357, 2.4142137]:
        for k in (-2, -1, 0, 1, 2):
            ye = ulps(xe * np.float32(t), k)
            for sx in (1, -1):
                for sy in (1, -1):
                    xs.append(sx * xe[:60])
                    ys.append(sy * ye[:60])
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-45, -1e-45, 1e-38, 3e38, -3e38, 1e-20, 1e20, 2.0 ** 61, 2.0 ** -61], np.float32)
    gx, gy = np.meshgrid(sp, sp)
    xs.append(gx.ravel())
    ys.append(gy.ravel())
    # millimetre-quantised coordinates (Livox / VLP-16 ranges in units of 2 mm)
    q = np.round(rng.uniform(-60, 60, (3000, 2)) * 500) / 500
    xs.append(q[:, 0].astype(np.float32))
    ys.append(q[:, 1].astype(np.float32))
    x = np.concatenate(xs).astype(np.float32)
    y = np.concatenate(ys).astype(np.float32)
    atan_out = np.array([libm.atanf(float(v)) for v in a], np.float32)
    atan2_out = np.array([libm.atan2f(float(yy), float(xx)) for yy, xx in zip(y, x)], np.float32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libm_f32_kat.npz")
    np.savez_compressed(out, atan_in=a, atan_out=atan_out, atan2_y=y, atan2_x=x, atan2_out=atan2_out,
                        glibc=np.array(os.confstr("CS_GNU_LIBC_VERSION")))
    d64 = (-np.arctan2(y.astype(np.float64), x.astype(np.float64))).astype(np.float32)
    fin = np.isfinite(atan2_out)
    print(out, len(a), "atanf,", len(x), "atan2f values;", int((d64[fin] != -atan2_out[fin]).sum()), "azimuths differ from float(-atan2(double))")


if __name__ == "__main__":
    main()
