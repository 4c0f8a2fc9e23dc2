"""(x, y) float pairs whose azimuth -atan2(y, x), as a double, lies within 2e-15 of the middle between two floats: the points the
bucketing kernels queue for k_azimuth_exact (csrc/atan2_cr.h).  Screened with numpy's atan2, confirmed with mpmath (the distance of
the TRUE angle from the middle); the first record is the point the randomised campaign found (seed 836, dense-batch round 51).
    python tests/golden/make_azimuth_edge.py   ->  tests/golden/azimuth_edge_xy.npy  (float32 [n, 2]: x, y)"""
import numpy as np
import mpmath as mp

mp.mp.prec = 200
rng = np.random.default_rng(20260929)
found = [(np.float32(4.3173523), np.float32(-6.7836986))]
while len(found) < 40:
    ang = rng.uniform(-np.pi, np.pi, 4_000_000)
    rr = 10 ** rng.uniform(0.0, 1.6, ang.size)
    x = (rr * np.cos(ang)).astype(np.float32)
    y = (rr * np.sin(ang)).astype(np.float32)
    th = -np.arctan2(y.astype(np.float64), x.astype(np.float64))
    f = th.astype(np.float32)
    lo, hi = np.nextafter(f, np.float32(-10)), np.nextafter(f, np.float32(10))
    dist = np.minimum(np.abs(th - (lo.astype(np.float64) + f) / 2), np.abs((hi.astype(np.float64) + f) / 2 - th))
    for k in np.flatnonzero(dist < 2e-15):
        t = -mp.atan2(mp.mpf(float(y[k])), mp.mpf(float(x[k])))
        fk = np.float32(float(t))
        mids = [(float(np.nextafter(fk, np.float32(-10))) + float(fk)) / 2, (float(np.nextafter(fk, np.float32(10))) + float(fk)) / 2]
        if min(abs(t - mp.mpf(m)) for m in mids) < mp.mpf(2e-15):
            found.append((x[k], y[k]))
out = np.array(found[:40], np.float32)
np.save(__file__.replace("make_azimuth_edge.py", "azimuth_edge_xy.npy"), out)
print(out.shape, out[:3])
