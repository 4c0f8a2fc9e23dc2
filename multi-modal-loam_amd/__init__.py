"""multi-modal-loam_amd -- MI355X-native scan-registration hot path of TIERS/multi-modal-loam.

This package is a thin ctypes view of the C-ABI in include/mmloam_hip.h (libmmloam_hip.so, built from
csrc/*.hip for gfx950).  It exists for the pytest harness, bench.py and Python users; the C++ adapter that
mirrors the reference classes (feature_extraction / Estimator) is host/mmloam_adapter.hpp.

There is no CPU fallback: importing works without a GPU (so the symbol-export test can run), but creating a
Context without a HIP device raises MmlError(MML_ERR_NO_DEVICE).

Import with importlib.import_module("multi-modal-loam_amd") (the directory name is not a Python identifier).
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MML_LIB_PATH") or os.path.join(_HERE, "libmmloam_hip.so")   # ($MML_LIB_PATH: an A/B build of the library)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mmloam_hip.h")

MML_OK, MML_ERR_INVALID, MML_ERR_NO_DEVICE, MML_ERR_HIP, MML_ERR_CAPACITY, MML_ERR_STATE = 0, -1, -2, -3, -4, -5
NEQ_RECORD_DOUBLES = 32
MAX_STAGES = 32
DIGEST_WORDS = 10

LIVOX_DTYPE = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
                        ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1"), ("_pad", "u1")])


class MmlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mmloam_hip error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("max_scans", C.c_int), ("max_velo_points", C.c_int), ("max_livox_points", C.c_int),
                ("n_rings", C.c_int), ("pitch0_deg", C.c_float), ("pitch_step_deg", C.c_float),
                ("n_livox_lines", C.c_int), ("near_th", C.c_float), ("far_th", C.c_float),
                ("leaf_corner", C.c_float), ("leaf_surf", C.c_float), ("cell_corner", C.c_float),
                ("cell_surf", C.c_float), ("max_features", C.c_int), ("max_map_points", C.c_int)]


class ScanInfo(C.Structure):
    _fields_ = [("n_points", C.c_int), ("n_velo", C.c_int), ("velo_corner_num", C.c_int), ("velo_surf_num", C.c_int),
                ("livox_corner_num", C.c_int), ("livox_surf_num", C.c_int), ("fused_corner_num", C.c_int),
                ("fused_surf_num", C.c_int)]


class AssocStats(C.Structure):
    _fields_ = [("n_line", C.c_int), ("n_plane", C.c_int), ("n_line_used", C.c_int), ("n_plane_used", C.c_int),
                ("normal_gram", C.c_double * 9), ("min_singular", C.c_double), ("is_degenerate", C.c_int)]


class SolveOpts(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("fixed_iterations", C.c_int), ("huber_delta", C.c_double),
                ("plan_weight_tan", C.c_double)]


class SolveSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful", C.c_int), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("termination", C.c_int)]


class EstimateInfo(C.Structure):
    _fields_ = [("outer_iterations", C.c_int), ("is_degenerate", C.c_int), ("n_corner_feat", C.c_int),
                ("n_surf_feat", C.c_int)]


class WindowTiming(C.Structure):
    _fields_ = [("evaluations", C.c_int), ("rounds", C.c_int), ("exchanges", C.c_int), ("device_ms", C.c_double)]


COMM_ID_BYTES = 128


class LoopbackGroup:
    """N contexts of this process on one device as the ranks 0 .. N-1 of a communicator whose collectives are device copies
    (mml_comm_*_loopback): the N-rank path of the C-ABI on a single GPU.  Every call drives all ranks."""

    def __init__(self, contexts):
        self.ctxs = list(contexts)
        self.n = len(self.ctxs)
        self._arr = (C.c_void_p * self.n)(*[c._h for c in self.ctxs])
        self._ck(lib().mml_comm_init_loopback(self._arr, C.c_int(self.n)), "mml_comm_init_loopback")

    def _ck(self, rc, what):
        if rc != 0:
            errs = [lib().mml_last_error(c._h).decode() for c in self.ctxs]
            raise MmlError(rc, what + ": " + " | ".join(e for e in errs if e))

    def window_solve(self, first_slots, n_local, x_window, T_bl, max_iters=10, fixed=False, huber=0.0, w_tan=3e-4):
        """Returns (n_ranks x W x 6 poses as every rank holds them, per-rank summaries)."""
        W = self.n * n_local
        x = _f64(x_window).reshape(W, 6).copy()
        out = np.zeros((self.n, W, 6))
        fs = np.ascontiguousarray(first_slots, dtype=np.int32)
        opts = SolveOpts(max_iters, 1 if fixed else 0, huber, w_tan)
        summ = (SolveSummary * self.n)()
        self._ck(lib().mml_window_solve_allgather_loopback(self._arr, C.c_int(self.n), _p(fs), C.c_int(n_local), _p(_f64(T_bl).reshape(16)),
                                                           C.byref(opts), _p(x), _p(out), summ), "mml_window_solve_allgather_loopback")
        return out, list(summ)

    def broadcast_features(self, slot, root):
        self._ck(lib().mml_comm_broadcast_features_loopback(self._arr, C.c_int(self.n), C.c_int(slot), C.c_int(root)),
                 "mml_comm_broadcast_features_loopback")

    def broadcast_local_map(self, root):
        self._ck(lib().mml_comm_broadcast_local_map_loopback(self._arr, C.c_int(self.n), C.c_int(root)),
                 "mml_comm_broadcast_local_map_loopback")


def comm_unique_id():
    """ncclGetUniqueId through the C-ABI: called by one rank, handed to mml_comm_init of every rank."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib().mml_comm_unique_id(buf)
    if rc != MML_OK:
        raise MmlError(rc, "mml_comm_unique_id")
    return bytes(buf)


class GicpInfo(C.Structure):
    _fields_ = [("outer_iterations", C.c_int), ("objective_evaluations", C.c_int), ("objective", C.c_double),
                ("n_source", C.c_int), ("n_target", C.c_int)]


class Profile(C.Structure):
    _fields_ = [("n_stages", C.c_int), ("name", C.c_char_p * MAX_STAGES), ("total_ms", C.c_double * MAX_STAGES),
                ("launches", C.c_long * MAX_STAGES)]


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libmmloam_hip.so (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", "-j8"])
    return LIB_PATH


_lib = None


def rccl_libraries():
    """ONE RCCL per process is what a multi-rank run wants.  libmmloam_hip.so needs `librccl.so.1` / `libamdhip64.so.7` (the
    ROCm copies, by its RUNPATH); a PyTorch wheel ships its own `torch/lib/librccl.so` and `libamdhip64.so` with the same
    SONAMEs.  The dynamic loader identifies a library by SONAME and by file: when torch is imported FIRST, our NEEDED entries
    resolve to its copies and the process holds one HIP runtime and one RCCL; the other way round torch's `librccl.so` request
    matches neither the name nor the file of the ROCm copy and a second runtime and a second RCCL would be mapped.  lib()
    therefore maps torch's copies itself before loading the library (_share_torch_runtime), so the order of the imports no
    longer matters; Context.comm_init still refuses to build a communicator of more than one rank when two copies are mapped.
    Returns the paths of every librccl in /proc/self/maps and the version behind the C-ABI's collectives."""
    v = C.c_int(0)
    lib().mml_rccl_version(C.byref(v))
    paths = []
    try:
        for line in open("/proc/self/maps"):
            f = line.split()
            # the library itself, not what it loads: RCCL dlopens net plugins (librccl-net.so, librccl-net-ofi.so ...) during init
            if len(f) >= 6 and f[5] not in paths and re.fullmatch(r"librccl\.so(\.\d+)*", os.path.basename(f[5])):
                paths.append(f[5])
    except OSError:
        pass
    return dict(loaded=paths, version=v.value)


def _share_torch_runtime():
    """Import-order independence of "one HIP runtime, one RCCL per process" (see rccl_libraries): when a PyTorch wheel with its
    own `libamdhip64.so` / `librccl.so` is installed, map THOSE files (by path, without importing torch) before
    libmmloam_hip.so is loaded.  Their SONAMEs are `libamdhip64.so.7` / `librccl.so.1`, so our NEEDED entries then bind to them,
    and a later `import torch` finds its own files already mapped -- the state "torch imported first" used to give, whichever
    comes first.  Without torch (a C++ host, a torch-free Python) nothing is preloaded and the ROCm copies are used.
    $MML_NO_TORCH_RUNTIME=1 switches the preload off."""
    if os.environ.get("MML_NO_TORCH_RUNTIME") == "1":
        return []
    done = []
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return done
        tl = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libamdhip64.so", "librccl.so"):
            path = os.path.join(tl, name)
            if os.path.exists(path):
                # RTLD_LOCAL: the loader's SONAME / file matching does not depend on symbol visibility, and RCCL's symbols put
                # into the global scope ahead of torch's own libraries end in a double free at process exit (measured)
                C.CDLL(path, mode=C.RTLD_LOCAL)
                done.append(path)
    except OSError:
        pass   # an unloadable wheel copy: fall back to the ROCm copies (rccl_libraries() still reports what is mapped)
    return done


def lib():
    """Load libmmloam_hip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MmlError(MML_ERR_STATE, "libmmloam_hip.so is missing: run __graft_entry__.build() "
                                          "(or make -C multi-modal-loam_amd/csrc); there is no CPU fallback")
        _share_torch_runtime()
        L = C.CDLL(LIB_PATH)
        L.mml_last_error.restype = C.c_char_p
        L.mml_last_error.argtypes = [C.c_void_p]
        L.mml_window_solver_create.restype = C.c_void_p
        L.mml_window_solver_create.argtypes = [C.c_int, C.POINTER(SolveOpts)]
        L.mml_window_solver_destroy.argtypes = [C.c_void_p]
        L.mml_window_solver_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mml_window_solver_summary.argtypes = [C.c_void_p, C.POINTER(SolveSummary)]
        L.mml_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def cube_index(xyz, cen=(10, 5, 10)):
    """Cube of every map-frame point, MAP_MANAGER::FindUsedCornerMap / FindUsedSurfMap (Map_Manager.cpp:583-629):
    ToIndex(i, j, k) = i + 21 j + 441 k over the 21 x 21 x 11 grid of 50 m cubes, 5000 for a point outside it.
    cen = laserCloudCen{Width,Height,Depth}_last.  Used to tag a global map for Context.map_set_global."""
    p = _f32(xyz).reshape(-1, 3).astype(np.float64)
    q = (p + 25.0) / 50.0
    with np.errstate(invalid="ignore"):
        c = np.where(np.isfinite(q), np.trunc(q), -1.0e6).astype(np.int64)
    c -= (p + 25.0 < 0)
    ci, cj, ck = c[:, 0] + cen[2], c[:, 1] + cen[0], c[:, 2] + cen[1]
    ok = (ci >= 0) & (ci < 21) & (cj >= 0) & (cj < 21) & (ck >= 0) & (ck < 11)
    return np.where(ok, ci + 21 * cj + 441 * ck, 5000).astype(np.int32)


def default_config(max_scans=1, **over):
    cfg = Config()
    lib().mml_config_default(C.byref(cfg), C.c_int(max_scans))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One mml_ctx: owns device buffers, maps and a HIP stream for `cfg.max_scans` scan slots."""

    def __init__(self, cfg=None, device=0, **over):
        self.cfg = cfg if cfg is not None else default_config(**over)
        self._h = C.c_void_p()
        rc = lib().mml_create(C.byref(self.cfg), C.c_int(device), C.byref(self._h))
        if rc != MML_OK:
            raise MmlError(rc, "mml_create failed (no HIP device?)" if rc == MML_ERR_NO_DEVICE else "mml_create failed")
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().mml_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != MML_OK:
            raise MmlError(rc, lib().mml_last_error(self._h).decode())

    def synchronize(self):
        self._ck(lib().mml_synchronize(self._h))

    # ---- input / feature extraction (feature_extraction::unionCloudHandler) ----
    def scan_upload(self, slot, velo_xyzi, livox):
        v = _f32(velo_xyzi).reshape(-1, 4) if velo_xyzi is not None else np.zeros((0, 4), np.float32)
        l = np.ascontiguousarray(livox) if livox is not None else np.zeros(0, LIVOX_DTYPE)
        assert l.dtype.itemsize == 20
        self._keep = getattr(self, "_keep", [])
        self._keep.append((v, l))  # host buffers must outlive the async copy
        self._ck(lib().mml_scan_upload(self._h, C.c_int(slot), _p(v), C.c_int(len(v)), _p(l), C.c_int(len(l))))
        if len(self._keep) > 4 * self.cfg.max_scans:
            self.synchronize()
            self._keep = self._keep[-self.cfg.max_scans:]

    def scan_upload_batch(self, first, velo_base, n_velo, livox_base, n_livox):
        """count scans in two copies: velo_base (count, max_velo_points, 4) float32, livox_base (count, max_livox_points)
        LIVOX_DTYPE, laid out with the slot stride; n_velo / n_livox: valid points per scan.  The host arrays must stay
        valid until the next synchronising call."""
        nv = np.ascontiguousarray(n_velo, dtype=np.int32)
        nl = np.ascontiguousarray(n_livox, dtype=np.int32)
        count = len(nv)
        v = np.ascontiguousarray(velo_base, dtype=np.float32)
        l = np.ascontiguousarray(livox_base)
        if v.size != count * self.cfg.max_velo_points * 4 or l.size * l.dtype.itemsize != count * self.cfg.max_livox_points * 20:
            raise ValueError("staging arrays must hold count x max points per sensor")
        self._keep = getattr(self, "_keep", [])
        self._keep.append((v, l, nv, nl))
        if len(self._keep) > 4 * self.cfg.max_scans + 8:
            self.synchronize()  # the copies are asynchronous: nothing may be released while one is still in flight
            self._keep = self._keep[-8:]
        self._ck(lib().mml_scan_upload_batch(self._h, C.c_int(first), C.c_int(count), _p(v), _p(nv), _p(l), _p(nl)))

    def scan_upload_pointcloud2(self, slot, data, n_points, point_step, off_x, off_y, off_z, off_intensity, livox):
        """Velodyne part as a sensor_msgs/PointCloud2 payload (bytes / uint8 array), decoded on the device."""
        raw = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        lv = np.ascontiguousarray(livox) if livox is not None else None
        nl = 0 if lv is None else len(lv)
        if len(raw) < n_points * point_step:
            raise ValueError("PointCloud2 payload shorter than n_points * point_step bytes")
        self._ck(lib().mml_scan_upload_pointcloud2(self._h, C.c_int(slot), _p(raw), C.c_int(n_points), C.c_int(point_step),
                                                   C.c_int(off_x), C.c_int(off_y), C.c_int(off_z), C.c_int(off_intensity),
                                                   _p(lv) if nl else None, C.c_int(nl)))
        self.synchronize()

    def scan_upload_wire(self, slot, data, n_points, point_step, off_x, off_y, off_z, off_intensity, livox_wire, n_livox):
        """Both parts in wire form: PointCloud2 payload + the serialised CustomPoint array (19 bytes per point)."""
        raw = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        lw = np.frombuffer(livox_wire, dtype=np.uint8) if not isinstance(livox_wire, np.ndarray) else np.ascontiguousarray(livox_wire, dtype=np.uint8)
        if len(lw) < 19 * n_livox:
            raise ValueError("livox_wire shorter than 19 * n_livox bytes")
        if len(raw) < n_points * point_step:
            raise ValueError("PointCloud2 payload shorter than n_points * point_step bytes")
        self._ck(lib().mml_scan_upload_wire(self._h, C.c_int(slot), _p(raw), C.c_int(n_points), C.c_int(point_step),
                                            C.c_int(off_x), C.c_int(off_y), C.c_int(off_z), C.c_int(off_intensity),
                                            _p(lw) if n_livox else None, C.c_int(n_livox)))
        self.synchronize()

    def time_offset_search(self, velo_xyz, livox_xyz, search_resolution=30, sliced_points=12000, tf=None):
        """estimate_timeoffset's numeric core (unionLidarsAligner.cpp:1077-1153): per-point 1-NN squared distances and
        the sliding-window error; defaults are the reference's (:111-112)."""
        v = np.ascontiguousarray(np.asarray(velo_xyz, np.float32).reshape(-1, 3))
        l = np.ascontiguousarray(np.asarray(livox_xyz, np.float32).reshape(-1, 3))
        t = np.ascontiguousarray(np.asarray(tf, np.float32).reshape(16)) if tf is not None else None
        nn = np.zeros(max(len(l), 1), np.float32)
        cap = max((len(l) - sliced_points) // max(search_resolution, 1) + 2, 1)
        err = np.zeros(cap, np.float64)
        nwin, best, lowest = C.c_int(0), C.c_int(-1), C.c_double(0)
        self._ck(lib().mml_time_offset_search(self._h, _p(v) if len(v) else None, C.c_int(len(v)), _p(t) if t is not None else None,
                                              _p(l) if len(l) else None, C.c_int(len(l)), C.c_int(search_resolution),
                                              C.c_int(sliced_points), _p(nn), _p(err), C.c_int(cap), C.byref(nwin),
                                              C.byref(best), C.byref(lowest)))
        return {"nn_d2": nn[:len(l)], "window_error": err[:nwin.value], "best_window": best.value, "lowest_error": lowest.value}

    def scan_download_pointxyzinormal(self, slot):
        """The fused labelled cloud as 48-byte PointXYZINormal records (the velo_combine / livox_combine payload)."""
        n = C.c_int(0)
        self._ck(lib().mml_scan_download_pointxyzinormal(self._h, C.c_int(slot), None, C.c_int(0), C.byref(n)))
        out = np.zeros((max(n.value, 1), 12), np.float32)
        self._ck(lib().mml_scan_download_pointxyzinormal(self._h, C.c_int(slot), _p(out), C.c_int(n.value), C.byref(n)))
        return out[:n.value]

    def cloud_upload(self, slot, records, n_velo=None):
        """A labelled fused cloud (n x 12 float32 = 48-byte PointXYZINormal records, velo_combine then livox_combine)
        into a slot: the PoseEstimation side of /union_feature_cloud."""
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 12)
        nv = len(rec) if n_velo is None else int(n_velo)
        self._ck(lib().mml_cloud_upload(self._h, C.c_int(slot), _p(rec) if len(rec) else None, C.c_int(len(rec)), C.c_int(nv)))

    def gicp_align(self, src, tgt, T0=None):
        """icp_ext_matching on plain clouds: returns (converged, T 4x4 float32, GicpInfo)."""
        src = _f32(src).reshape(-1, 3)
        tgt = _f32(tgt).reshape(-1, 3)
        T = np.ascontiguousarray(np.eye(4, dtype=np.float32) if T0 is None else np.asarray(T0, np.float32).reshape(4, 4).copy())
        conv, info = C.c_int(0), GicpInfo()
        self._ck(lib().mml_gicp_align(self._h, _p(src) if len(src) else None, C.c_int(len(src)), _p(tgt) if len(tgt) else None,
                                      C.c_int(len(tgt)), _p(T), C.byref(conv), C.byref(info)))
        return bool(conv.value), T, info

    def gicp_refresh(self, slot, extrinsic, apply=True):
        """unionCloudHandler's extrinsic refresh (:302-318) on a slot extracted without an extrinsic."""
        T = np.ascontiguousarray(np.asarray(extrinsic, np.float32).reshape(4, 4).copy())
        ref, info = C.c_int(0), GicpInfo()
        self._ck(lib().mml_gicp_refresh(self._h, C.c_int(slot), _p(T), C.c_int(1 if apply else 0), C.byref(ref), C.byref(info)))
        return bool(ref.value), T, info

    def extract(self, first=0, count=1, livox_extrinsic=None):
        e = _f32(livox_extrinsic).reshape(16) if livox_extrinsic is not None else None
        self._ck(lib().mml_extract(self._h, C.c_int(first), C.c_int(count), _p(e)))

    def scan_info(self, slot):
        info = ScanInfo()
        self._ck(lib().mml_scan_info_get(self._h, C.c_int(slot), C.byref(info)))
        return info

    def scan_download(self, slot):
        info = self.scan_info(slot)
        n = info.n_points
        xyzi = np.zeros((max(n, 1), 4), np.float32)
        rel = np.zeros(max(n, 1), np.float32)
        line = np.zeros(max(n, 1), np.uint8)
        label = np.zeros(max(n, 1), np.uint8)
        self._ck(lib().mml_scan_download(self._h, C.c_int(slot), _p(xyzi), _p(rel), _p(line), _p(label), C.c_int(max(n, 1))))
        return dict(xyzi=xyzi[:n], reltime=rel[:n], ring=line[:n].astype(np.int32), label=label[:n].astype(np.int32),
                    info=info)

    def detect_line(self, pts):
        """Twin of feature_extraction::detectFeaturePoints: returns (sharp idx, flat idx, CloudFeatureFlag)."""
        pts = _f32(pts).reshape(-1, 4)
        n = len(pts)
        sharp = np.zeros(max(n, 1), np.int32)
        flat = np.zeros(max(n, 1), np.int32)
        flags = np.zeros(max(n, 1), np.int32)
        ns, nf = C.c_int(0), C.c_int(0)
        self._ck(lib().mml_detect_line(self._h, _p(pts), C.c_int(n), _p(sharp), C.byref(ns), _p(flat), C.byref(nf),
                                       _p(flags)))
        return sharp[:ns.value].copy(), flat[:nf.value].copy(), flags[:n].copy()

    # ---- RemoveLidarDistortion ----
    def undistort(self, first, count, dR, dt):
        dR = _f64(dR).reshape(count, 9)
        dt = _f64(dt).reshape(count, 3)
        self._ck(lib().mml_undistort(self._h, C.c_int(first), C.c_int(count), _p(dR), _p(dt)))

    # ---- Estimator::EstimateLidarPose pieces ----
    def downsample(self, first=0, count=1):
        self._ck(lib().mml_downsample(self._h, C.c_int(first), C.c_int(count)))

    def features_download(self, slot, kind):
        n = C.c_int(0)
        self._ck(lib().mml_features_download(self._h, C.c_int(slot), C.c_int(kind), None, C.c_int(0), C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        self._ck(lib().mml_features_download(self._h, C.c_int(slot), C.c_int(kind), _p(out), C.c_int(max(n.value, 1)),
                                             C.byref(n)))
        return out[:n.value]

    def features_upload(self, slot, kind, xyz):
        xyz = _f32(xyz).reshape(-1, 3)
        self._ck(lib().mml_features_upload(self._h, C.c_int(slot), C.c_int(kind), _p(xyz), C.c_int(len(xyz))))

    def map_set_local(self, kind, xyz):
        xyz = _f32(xyz).reshape(-1, 3)
        self._ck(lib().mml_map_set_local(self._h, C.c_int(kind), _p(xyz), C.c_int(len(xyz))))

    def map_increment_local(self, slot, T_wl):
        """Estimator::MapIncrementLocal on the device; returns the new (corner, surf) local map sizes."""
        nc, ns = C.c_int(0), C.c_int(0)
        T = _f64(T_wl).reshape(16)
        self._ck(lib().mml_map_increment_local(self._h, C.c_int(slot), _p(T), C.byref(nc), C.byref(ns)))
        return nc.value, ns.value

    def map_local_reset(self):
        self._ck(lib().mml_map_local_reset(self._h))

    def map_local_download(self, kind):
        n = C.c_int(0)
        self._ck(lib().mml_map_local_download(self._h, C.c_int(kind), None, C.c_int(0), C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        self._ck(lib().mml_map_local_download(self._h, C.c_int(kind), _p(out), C.c_int(n.value), C.byref(n)))
        return out[:n.value].copy()

    def map_global_append(self, slot, T_wl):
        self._ck(lib().mml_map_global_append(self._h, C.c_int(slot), _p(_f64(T_wl).reshape(16))))

    def map_global_increment(self, T_wl):
        """MAP_MANAGER::MapIncrement on the device; returns the live (corner, surf) store sizes."""
        nc, ns = C.c_int(0), C.c_int(0)
        self._ck(lib().mml_map_global_increment(self._h, _p(_f64(T_wl).reshape(16)), C.byref(nc), C.byref(ns)))
        return nc.value, ns.value

    def map_global_download(self, kind):
        n = C.c_int(0)
        cen = np.zeros(3, np.int32)
        self._ck(lib().mml_map_global_download(self._h, C.c_int(kind), None, None, C.c_int(0), C.byref(n), _p(cen)))
        xyz = np.zeros((max(n.value, 1), 3), np.float32)
        cube = np.zeros(max(n.value, 1), np.int32)
        self._ck(lib().mml_map_global_download(self._h, C.c_int(kind), _p(xyz), _p(cube), C.c_int(n.value), C.byref(n), _p(cen)))
        return xyz[:n.value].copy(), cube[:n.value].copy(), cen

    def map_global_reset(self):
        self._ck(lib().mml_map_global_reset(self._h))

    def map_set_global(self, kind, xyz, cube, cen=None):
        """Cube store of the global map (a12): xyz (m, 3) and the ToIndex cube of every point."""
        xyz = _f32(xyz).reshape(-1, 3)
        cube = np.ascontiguousarray(cube, dtype=np.int32).reshape(-1)
        if len(cube) != len(xyz):
            raise ValueError("one cube index per point")
        cen_p = None if cen is None else _p(np.ascontiguousarray(cen, dtype=np.int32))
        self._ck(lib().mml_map_set_global(self._h, C.c_int(kind), _p(xyz), _p(cube), C.c_int(len(xyz)), cen_p))

    def knn5(self, kind, q, max_d2=np.inf):
        q = _f32(q).reshape(-1, 3)
        idx = np.zeros((len(q), 5), np.int32)
        d2 = np.zeros((len(q), 5), np.float32)
        md = np.float32(min(max_d2, 3.0e38))
        self._ck(lib().mml_knn5(self._h, C.c_int(kind), _p(q), C.c_int(len(q)), C.c_float(md), _p(idx), _p(d2)))
        return idx, d2

    def associate(self, first, count, T_wl, thres_dist, stats=True):
        """stats=False: enqueue only (no read-back, no synchronisation) -- what a caller does that goes straight on to
        mml_solve on the same slots."""
        T = _f64(T_wl).reshape(count, 16)
        if not stats:
            self._keep_T = T                              # the poses are staged before the call returns; kept anyway
            self._ck(lib().mml_associate(self._h, C.c_int(first), C.c_int(count), _p(T), C.c_double(thres_dist), None))
            return None
        st = (AssocStats * count)()
        self._ck(lib().mml_associate(self._h, C.c_int(first), C.c_int(count), _p(T), C.c_double(thres_dist), st))
        return list(st)

    def factors_download(self, slot, kind):
        n = C.c_int(0)
        self._ck(lib().mml_factors_download(self._h, C.c_int(slot), C.c_int(kind), None, None, C.c_int(0), C.byref(n)))
        out = np.zeros((max(n.value, 1), 10))
        src = np.zeros(max(n.value, 1), np.int32)
        self._ck(lib().mml_factors_download(self._h, C.c_int(slot), C.c_int(kind), _p(out), _p(src),
                                            C.c_int(max(n.value, 1)), C.byref(n)))
        return out[:n.value], src[:n.value]

    def factors_upload(self, slot, kind, rec):
        rec = _f64(rec).reshape(-1, 10)
        self._ck(lib().mml_factors_upload(self._h, C.c_int(slot), C.c_int(kind), _p(rec), C.c_int(len(rec))))

    def linearize(self, slot, x, T_bl, w_tan=0.0, huber=0.1 / 1.5e-3):
        H = np.zeros((6, 6))
        g = np.zeros(6)
        c = C.c_double(0)
        self._ck(lib().mml_linearize(self._h, C.c_int(slot), _p(_f64(x)), _p(_f64(T_bl).reshape(16)), C.c_double(w_tan),
                                     C.c_double(huber), _p(H), _p(g), C.byref(c)))
        return H, g, c.value

    def linearize_window(self, first, frames, x, T_bl, w_tan=0.0, huber=0.1 / 1.5e-3):
        """Records (frames, 32) of the consecutive slots first .. first + frames - 1 at x (frames, >= 6): one launch."""
        x = _f64(x).reshape(frames, -1)
        rec = np.zeros((frames, NEQ_RECORD_DOUBLES))
        self._ck(lib().mml_linearize_window(self._h, C.c_int(first), C.c_int(frames), C.c_int(x.shape[1]), _p(x),
                                            _p(_f64(T_bl).reshape(16)), C.c_double(w_tan), C.c_double(huber), _p(rec)))
        return rec

    def linearize_record(self, slot, x, T_bl, d_record_ptr, w_tan=0.0, huber=0.1 / 1.5e-3):
        self._ck(lib().mml_linearize_record(self._h, C.c_int(slot), _p(_f64(x)), _p(_f64(T_bl).reshape(16)),
                                            C.c_double(w_tan), C.c_double(huber), C.c_void_p(d_record_ptr)))

    def solve(self, first, count, x, T_bl, window=1, max_iters=10, fixed=False, huber=0.1 / 1.5e-3, w_tan=0.0,
              trace=False):
        x = _f64(x).reshape(count, 6).copy()
        opts = SolveOpts(max_iters, 1 if fixed else 0, huber, w_tan)
        nprob = count // window
        summ = (SolveSummary * nprob)()
        tr = np.zeros((nprob, max_iters, 6 * window)) if trace else None
        self._ck(lib().mml_solve(self._h, C.c_int(first), C.c_int(count), C.c_int(window), _p(_f64(T_bl).reshape(16)),
                                 C.byref(opts), _p(x), summ, _p(tr)))
        return x, list(summ), tr

    def estimate(self, first, count, exTlb, P, Q, max_outer=5, inner_iters=10):
        P = _f64(P).reshape(count, 3).copy()
        Q = _f64(Q).reshape(count, 4).copy()
        info = (EstimateInfo * count)()
        self._ck(lib().mml_estimate(self._h, C.c_int(first), C.c_int(count), _p(_f64(exTlb).reshape(16)), _p(P), _p(Q),
                                    C.c_int(max_outer), C.c_int(inner_iters), info))
        return P, Q, list(info)

    def step(self, first, count, dR, dt, exTlb, thres_dist, gn_iters, x):
        x = _f64(x).reshape(count, 6).copy()
        self._ck(lib().mml_step(self._h, C.c_int(first), C.c_int(count), _p(_f64(dR).reshape(count, 9)),
                                _p(_f64(dt).reshape(count, 3)), _p(_f64(exTlb).reshape(16)), C.c_double(thres_dist),
                                C.c_int(gn_iters), _p(x)))
        return x

    # ---- multi-GPU (RCCL inside the C-ABI) ----
    def comm_init(self, n_ranks, rank, comm_id):
        r = rccl_libraries()
        if n_ranks > 1 and len(r["loaded"]) > 1:
            raise MmlError(MML_ERR_STATE, "two RCCL copies are mapped into this process (%s): the library and torch.distributed "
                                          "would each drive their own" % ", ".join(r["loaded"]))
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("the communicator id is %d bytes" % COMM_ID_BYTES)
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self._ck(lib().mml_comm_init(self._h, C.c_int(n_ranks), C.c_int(rank), buf))

    def comm_destroy(self):
        self._ck(lib().mml_comm_destroy(self._h))

    def comm_info(self):
        n, r = C.c_int(0), C.c_int(0)
        self._ck(lib().mml_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def window_solve_allgather(self, first, n_local, x_local, T_bl, max_iters=10, fixed=False, huber=0.0, w_tan=3e-4):
        """Joint window solve over n_ranks * n_local frames; returns (own poses, window poses, summary, timing)."""
        n_ranks, _ = self.comm_info()
        x = _f64(x_local).reshape(n_local, 6).copy()
        xw = np.zeros((n_ranks * n_local, 6))
        opts = SolveOpts(max_iters, 1 if fixed else 0, huber, w_tan)
        summ, tim = SolveSummary(), WindowTiming()
        self._ck(lib().mml_window_solve_allgather(self._h, C.c_int(first), C.c_int(n_local), _p(_f64(T_bl).reshape(16)),
                                                  C.byref(opts), _p(x), _p(xw), C.byref(summ), C.byref(tim)))
        return x, xw, summ, tim

    def comm_broadcast_features(self, slot, root):
        self._ck(lib().mml_comm_broadcast_features(self._h, C.c_int(slot), C.c_int(root)))

    def comm_broadcast_local_map(self, root):
        self._ck(lib().mml_comm_broadcast_local_map(self._h, C.c_int(root)))

    # ---- measurement ----
    def set_lanes(self, lanes):
        self._ck(lib().mml_set_lanes(self._h, C.c_int(lanes)))

    def profile_enable(self, on=True):
        self._ck(lib().mml_profile_enable(self._h, C.c_int(1 if on else 0)))

    def profile_reset(self):
        self._ck(lib().mml_profile_reset(self._h))

    def profile_get(self):
        pr = Profile()
        self._ck(lib().mml_profile_get(self._h, C.byref(pr)))
        return {pr.name[i].decode(): (pr.total_ms[i], pr.launches[i]) for i in range(pr.n_stages)}

    def extract_queue_counts(self, slot):
        """(redo, brk): points of the slot's last extraction recomputed with the full decision chain / finished as break-point
        candidates."""
        r, b = C.c_int(0), C.c_int(0)
        self._ck(lib().mml_extract_queue_counts(self._h, C.c_int(slot), C.byref(r), C.byref(b)))
        return r.value, b.value

    def issue_rate(self, kind=0, reps=3):
        """wave64 instructions per second of the device on v_fma_f32 (kind 0) / v_add_u32 (kind 1) chains (mml_issue_rate)."""
        r = C.c_double(0)
        self._ck(lib().mml_issue_rate(self._h, C.c_int(kind), C.c_int(reps), C.byref(r)))
        return r.value

    def libm_f32(self, y, x):
        """(atan2f(y, x), atanf(y)) evaluated by the device's copies of the two libm routines (mml_libm_f32, a test hook)."""
        y, x = np.ascontiguousarray(y, np.float32).ravel(), np.ascontiguousarray(x, np.float32).ravel()
        assert len(x) == len(y)
        o2, o1 = np.empty_like(x), np.empty_like(x)
        self._ck(lib().mml_libm_f32(self._h, _p(y), _p(x), C.c_long(len(x)), _p(o2), _p(o1)))
        return o2, o1

    def associate_far_count(self):
        """5-NN queries of the last association that went to the far-query kernels (summed over the stream lanes)."""
        n = C.c_int(0)
        self._ck(lib().mml_associate_far_count(self._h, C.byref(n)))
        return n.value

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        mem = C.c_size_t(0)
        self._ck(lib().mml_device_info(self._h, name, C.c_int(256), C.byref(cus), C.byref(mem)))
        return name.value.decode(), cus.value, mem.value

    def slot_digest(self, first, count):
        """(count, DIGEST_WORDS) uint64: one digest per slot and piece of state (mml_slot_digest; recomputable on the host from
        the download entry points, see `host_digest`)."""
        out = np.zeros((count, DIGEST_WORDS), np.uint64)
        self._ck(lib().mml_slot_digest(self._h, C.c_int(first), C.c_int(count), _p(out)))
        return out

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        g = C.c_double(0)
        self._ck(lib().mml_copy_bandwidth(self._h, C.c_size_t(nbytes), C.c_int(reps), C.byref(g)))
        return g.value


class ImuPreint(C.Structure):
    _fields_ = [("dp", C.c_double * 3), ("dv", C.c_double * 3), ("dq", C.c_double * 4), ("dtime", C.c_double),
                ("bg", C.c_double * 3), ("ba", C.c_double * 3), ("jacobian", C.c_double * 225),
                ("covariance", C.c_double * 225)]


class Prior(C.Structure):
    _fields_ = [("J", C.c_double * 225), ("r0", C.c_double * 15), ("x0", C.c_double * 15)]


def imu_preintegrate(samples, bg, ba):
    """IMUIntegrator::PreIntegration: samples (n, 7) = gyro xyz, accel xyz (message units), dt."""
    smp = _f64(samples).reshape(-1, 7)
    out = ImuPreint()
    rc = lib().mml_imu_preintegrate(_p(smp), C.c_int(len(smp)), _p(_f64(bg)), _p(_f64(ba)), C.byref(out))
    if rc != MML_OK:
        raise MmlError(rc, "mml_imu_preintegrate")
    return out


def imu_factor(pre, gravity, pr_i, vb_i, pr_j, vb_j, jac=True):
    """Cost_NavState_PRV_Bias: weighted residual (15) and Jacobian (15, 30) [pr_i | vb_i | pr_j | vb_j]."""
    r = np.zeros(15)
    J = np.zeros((15, 30)) if jac else None
    rc = lib().mml_imu_factor(C.byref(pre), _p(_f64(gravity)), _p(_f64(pr_i)), _p(_f64(vb_i)), _p(_f64(pr_j)),
                              _p(_f64(vb_j)), _p(r), _p(J))
    if rc != MML_OK:
        raise MmlError(rc, "mml_imu_factor")
    return r, J


class FullWindowSolver:
    """Estimator::Estimate in full-window mode on the host: W x [PR 6 | VBias 9], lidar records from the device,
    IMU factors between consecutive frames, optional marginalization prior on frame 0."""

    def __init__(self, W, max_iters=10, fixed=False, huber=0.0, w_tan=3e-4):
        self.W = W
        self._opts = SolveOpts(max_iters, 1 if fixed else 0, huber, w_tan)
        lib().mml_fullwindow_create.restype = C.c_void_p
        self._h = C.c_void_p(lib().mml_fullwindow_create(C.c_int(W), C.byref(self._opts)))
        if not self._h:
            raise MmlError(MML_ERR_INVALID, "mml_fullwindow_create failed")

    def _ck(self, rc, what):
        if rc < 0:
            raise MmlError(rc, what)
        return rc

    def set_imu(self, f, pre, gravity):
        self._ck(lib().mml_fullwindow_set_imu(self._h, C.c_int(f), C.byref(pre), _p(_f64(gravity))), "set_imu")

    def set_prior(self, prior):
        self._ck(lib().mml_fullwindow_set_prior(self._h, C.byref(prior) if prior is not None else None), "set_prior")

    def step(self, records, x_eval):
        rec = _f64(records).reshape(self.W, NEQ_RECORD_DOUBLES)
        x = _f64(x_eval).reshape(self.W, 15).copy()
        rc = self._ck(lib().mml_fullwindow_step(self._h, _p(rec), _p(x)), "mml_fullwindow_step")
        return rc == 1, x

    def solve_device(self, ctx, first_slot, T_bl, x):
        """mml_fullwindow_solve: the whole trust-region loop on the device (frames in consecutive, associated slots).
        Returns (x, summary, evaluations)."""
        x = _f64(x).reshape(self.W, 15).copy()
        T = _f64(T_bl).reshape(16)
        s = SolveSummary()
        ev = C.c_int(0)
        ctx._ck(lib().mml_fullwindow_solve(ctx._h, self._h, C.c_int(first_slot), _p(T), _p(x), C.byref(s), C.byref(ev)))
        return x, s, ev.value

    def summary(self):
        s = SolveSummary()
        lib().mml_fullwindow_summary(self._h, C.byref(s))
        return s

    def normal_equations(self, records, x):
        n = 15 * self.W
        H = np.zeros((n, n))
        g = np.zeros(n)
        c = C.c_double(0)
        self._ck(lib().mml_fullwindow_normal_equations(self._h, _p(_f64(records).reshape(self.W, NEQ_RECORD_DOUBLES)),
                                                       _p(_f64(x).reshape(self.W, 15)), _p(H), _p(g), C.byref(c)), "normal_equations")
        return H, g, c.value

    def marginalize(self, lidar_record0, x):
        out = Prior()
        self._ck(lib().mml_fullwindow_marginalize(self._h, _p(_f64(lidar_record0).reshape(NEQ_RECORD_DOUBLES)),
                                                  _p(_f64(x).reshape(self.W, 15)), C.byref(out)), "marginalize")
        return out

    def __del__(self):
        try:
            if self._h:
                lib().mml_fullwindow_destroy(self._h)
                self._h = None
        except Exception:
            pass


class WindowSolver:
    """Host-side joint dogleg over W all-gathered 32-double records; needs no device."""

    def __init__(self, W, max_iters=10, fixed=False, huber=0.0, w_tan=3e-4):
        self.W = W
        self._opts = SolveOpts(max_iters, 1 if fixed else 0, huber, w_tan)
        self._h = lib().mml_window_solver_create(C.c_int(W), C.byref(self._opts))
        if not self._h:
            raise MmlError(MML_ERR_INVALID, "mml_window_solver_create failed")

    def step(self, records, x_eval):
        """records: (W,32) at x_eval (W,6).  Returns (done, next x_eval)."""
        rec = _f64(records).reshape(self.W, NEQ_RECORD_DOUBLES)
        x = _f64(x_eval).reshape(self.W, 6).copy()
        rc = lib().mml_window_solver_step(self._h, _p(rec), _p(x))
        if rc < 0:
            raise MmlError(rc, "mml_window_solver_step")
        return rc == 1, x

    def summary(self):
        s = SolveSummary()
        lib().mml_window_solver_summary(self._h, C.byref(s))
        return s

    def __del__(self):
        try:
            if self._h:
                lib().mml_window_solver_destroy(self._h)
                self._h = None
        except Exception:
            pass


def pack_record(H, g, cost, n_line_used=0, n_plane_used=0):
    """(6x6 H, g, cost) -> the 32-double all-gather record of include/mmloam_hip.h."""
    rec = np.zeros(NEQ_RECORD_DOUBLES)
    k = 0
    for a in range(6):
        for b in range(a, 6):
            rec[k] = H[a, b]
            k += 1
    rec[21:27] = g
    rec[27] = cost
    rec[28] = n_line_used
    rec[29] = n_plane_used
    return rec
