"""bench.py's harness on the CPU: the helpers directly, and both run_* functions end to end against a STUB context (no device, no
library call) -- every section of the JSON line is built, none may raise or report an error.  Round 4 shipped a `--config 2` whose
full-window section died on a TypeError that an `except Exception` turned into a string; the sections now collect what they catch
in "errors", `--strict` turns a non-empty list into a non-zero exit, and this test runs the same code paths without a GPU."""
import importlib
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    return importlib.import_module("bench")


def test_pose_helpers_and_perturbed(bench):
    bench.pose_helpers_selfcheck()
    T = np.eye(4)
    T[:3, 3] = [1.0, 2.0, 3.0]
    a = bench.perturbed(T)
    assert np.allclose(a[:3, 3], [1.02, 1.985, 3.01]) and np.allclose(a[:3, :3], bench.R_PERT)
    b = bench.perturbed(T, dt_=(0.0, 0.0, 0.5), rv=(0.0, 0.0, 0.1))           # the full-window section's call shape
    assert np.allclose(b[:3, 3], [1.0, 2.0, 3.5]) and np.isclose(b[0, 0], np.cos(0.1)) and np.isclose(b[1, 0], np.sin(0.1))


def test_replica_check(bench):
    dg = np.zeros((8, 10), np.uint64)
    x = np.zeros((8, 6))
    keys = [(s % 2, 0) for s in range(8)]
    for s in range(8):
        dg[s] = 100 + s % 2
        x[s] = s % 2
    r = bench.replica_check(dg, x, keys)
    assert r == {"slots": 8, "groups": 2, "replica_slots_compared": 6, "digest_words": 10, "mismatches": 0}
    dg[5, 3] += np.uint64(1)
    x[6, 0] += 1e-12
    r = bench.replica_check(dg, x, keys)
    assert r["mismatches"] == 2
    assert {"slot": 5, "first_slot_of_group": 1, "digest_words": [3]} in r["examples"]
    assert {"slot": 6, "first_slot_of_group": 0, "digest_words": []} in r["examples"]


# ---- the stub: what bench.py calls on a context, with results of the right type and shape ------------------------------------
class StubContext:
    def __init__(self, cfg, device=0):
        self.cfg = cfg
        self.rng = np.random.default_rng(1)
        self.calls = {}
        self.lanes = None

    def __getattr__(self, name):   # every entry point without a result
        if name.startswith("_"):
            raise AttributeError(name)

        def call(*a, **k):
            self.calls[name] = self.calls.get(name, 0) + 1
        return call

    def device_info(self):
        return "stub device", 256, 288 << 30

    def set_lanes(self, n):
        self.lanes = n

    def step(self, first, count, dR, dt, exTlb, thres, gn_iters, x):
        assert np.shape(dR) == (count, 9) and np.shape(dt) == (count, 3) and np.shape(x) == (count, 6)
        x = np.array(x, dtype=np.float64).copy()
        x[:, :3] -= np.array([0.03, -0.02, 0.01])      # "registers" the scans: removes the bench's initial translation error
        return x

    def slot_digest(self, first, count):
        return np.zeros((count, 10), np.uint64)

    def scan_info(self, slot):
        return types.SimpleNamespace(n_points=52800, n_velo=28800, fused_corner_num=400, fused_surf_num=3000)

    def features_download(self, slot, kind):
        return self.rng.uniform(-5, 5, (300 if kind == 0 else 900, 3)).astype(np.float32)

    def profile_get(self):
        return {"stencil": (2.4, 4), "assign_onepass": (2.0, 4), "solve": (1.2, 4), "fullwindow": (0.5, 3)}

    def issue_rate(self, kind=0, reps=3):
        return 6.0e11 if kind == 0 else 1.1e12

    def copy_bandwidth(self, nbytes, reps):
        return 5000.0

    def solve(self, first, count, x, T_bl, window=1, **k):
        return np.array(x, dtype=np.float64).copy(), None, None

    def window_solve_allgather(self, first, n_local, x_local, T_bl, **k):
        x = np.array(x_local, dtype=np.float64)
        return x, x.copy(), types.SimpleNamespace(iterations=3, termination=0), types.SimpleNamespace(device_ms=0.1, evaluations=4, rounds=12)


class StubOdometry:
    def __init__(self, ctx, lidar_mode=2):
        self.key_scans, self.n_corner_local, self.n_surf_local = 0, 0, 0

    def estimate_lidar_pose(self, slot, P, Q):
        self.key_scans += 1
        self.n_corner_local, self.n_surf_local = 500, 5000
        return np.asarray(P), np.asarray(Q), True


class StubWindowEstimator:
    def __init__(self, ctx, gravity=None, solver="device"):
        self.solver = solver

    def estimate(self, slots, frames, pres):
        assert len(frames) == len(slots) == len(pres)
        for fr in frames:
            assert fr["P"].shape == (3,) and fr["Q"].shape == (4,) and fr["V"].shape == (3,)
        return {"evaluations": 7, "outer": 2}


@pytest.fixture()
def stubbed(bench, monkeypatch):
    real_M = importlib.import_module("multi-modal-loam_amd")
    real_synth = importlib.import_module("multi-modal-loam_amd.synth")
    M = types.SimpleNamespace(Context=StubContext, default_config=lambda **kw: types.SimpleNamespace(
        leaf_corner=0.4, leaf_surf=0.2, max_velo_points=kw.get("max_velo_points", 0) + 1, max_livox_points=kw.get("max_livox_points", 0) + 1, **{
            k: v for k, v in kw.items() if k not in ("max_velo_points", "max_livox_points")}),
        imu_preintegrate=lambda *a: object(), rccl_libraries=lambda: {"loaded": ["librccl.so.1"], "version": 22606},
        comm_unique_id=lambda: bytes(128), COMM_ID_BYTES=128, __file__=real_M.__file__)
    odo = types.SimpleNamespace(LidarOdometry=StubOdometry, WindowEstimator=StubWindowEstimator)
    mods = {"multi-modal-loam_amd": M, "multi-modal-loam_amd.synth": real_synth, "multi-modal-loam_amd.odometry": odo}
    monkeypatch.setattr(bench.importlib, "import_module", lambda name: mods[name])
    monkeypatch.setattr(bench, "pinned", lambda a: np.ascontiguousarray(a))
    import torch
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    # (the C++ adapter leg of --config 2 builds and runs a real binary against the library: stubbed like the context)
    monkeypatch.setattr(bench, "cpp_live_loop", lambda scans, motions, predicted, reps=3: {"host": "stub", "per_scan_p50_ms": 0.5, "scans": len(scans)})
    return bench


def _args(bench, monkeypatch, argv):
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    return bench.parse()


def test_run_throughput_on_a_stub_context(stubbed, monkeypatch):
    bench = stubbed
    args = _args(bench, monkeypatch, ["--steps", "2", "--warmup", "1", "--slots", "8", "--batch", "16", "--distinct", "2", "--map-points", "20000",
                                      "--cpu-seconds", "0", "--kernel-steps", "1", "--skip-upload", "--window-demo"])
    line, stuck, errors = bench.run_throughput(args, 0, 0, 1, None)
    assert errors == [] and not stuck
    r = json.loads(line)
    assert r["errors"] == [] and r["steps"] == 2 and r["config"]["resident_slots"] == 8 and r["config"]["passes_per_step"] == 2
    assert r["value"] > 0 and r["unit"] == "scans/s" and r["n_gpus"] == 1
    assert r["replica_check"]["mismatches"] == 0 and r["replica_check"]["slots"] == 8
    assert r["replica_check"]["one_lane_vs_timed_region_mismatches"] == 0
    assert r["roofline"]["kernel"] == "stencil" and r["roofline"]["frac"] > 0 and "traffic_ratio" in r["roofline"]
    assert r["window_solve"]["frames"] == 1 and "error" not in r["window_solve"]


def test_run_replay_on_a_stub_context(stubbed, monkeypatch):
    bench = stubbed
    args = _args(bench, monkeypatch, ["--config", "2", "--steps", "3", "--replay-scans", "10", "--cpu-seconds", "0"])
    line, errors = bench.run_replay(args, 0, 0, 1, None)
    assert errors == []
    r = json.loads(line)
    assert r["errors"] == []
    fw = r["full_window_imu"]
    assert sorted(fw) == ["device_w5", "device_w8", "host_w5", "host_w8"]
    for k, v in fw.items():
        assert "error" not in v and v["evaluations"] == 7 and isinstance(v["estimate_ms"], float), (k, v)
    assert r["latency_ms"]["per_scan_p50"] is not None and r["config"]["key_scans"] == 10
    assert r["latency_ms_cpp_adapter"]["scans"] == 10


def test_strict_turns_errors_into_a_nonzero_exit(tmp_path):
    """A section that fails is reported in "errors", and `--strict` makes the process exit 3 after printing the line (the stub
    step has no failing section: exit 0 with and without --strict)."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-step", "--strict", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    src = open(os.path.join(ROOT, "bench.py")).read()
    # no section may swallow an exception without recording it: every `except Exception` of the two run_* functions either
    # appends to `errors` within the next lines or belongs to the short allow-list below
    lines = src.split("\n")
    allowed = ("traffic = ", "issue = None", "return None", "r[\"torch_version\"] = None", "return {\"error\"", "pass",
               "box[\"window\"] = {\"error\"", "with_upload = {\"error\"", "cpu = {\"error\"")
    for i, ln in enumerate(lines):
        if ln.strip().startswith("except Exception"):
            body = "\n".join(lines[i + 1:i + 6])
            assert "errors.append" in body or any(a in body for a in allowed), "bench.py:%d swallows an exception" % (i + 1)
