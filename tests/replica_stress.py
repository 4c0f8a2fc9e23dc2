"""Replica stress: 4096 slots holding 20 distinct scans, mml_step repeated; every slot must equal the first slot of its scan in all ten
digest words (mml_slot_digest: counts, labels, lines, cloud, times, both stacks, both factor lists, pose), every round, and every
round must equal the first.  Collected as tests/test_gpu_shapes.py::test_replica_stress (30 + 10 rounds); as a script:

    python tests/replica_stress.py [rounds] [lanes]      # MML_LANES / AMD_SERIALIZE_KERNEL from the environment
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIECES = ("counts", "label", "line", "xyzi", "reltime", "corner stack", "surf stack", "line factors", "plane factors", "pose")


def run(M, O, synth, rounds, lanes=None, B=4096, log=print):
    """-> (bad rounds, seconds per round).  Test infrastructure: the maps are built with the oracle."""
    import conftest
    scene = conftest.build_scene(O, synth)
    cases = conftest.batch_cases(synth)
    nd = len(cases)
    cm = synth.grow_map(scene["corner_map"], 40000, seed=7)
    sm = synth.grow_map(scene["surf_map"], 160000, seed=8)
    rng = np.random.default_rng(2024)
    assign = rng.integers(0, nd, B)
    assign[:nd] = np.arange(nd)
    c = M.Context(max_scans=B, max_velo_points=max(len(cs["velo"]) for cs in cases if cs["velo"] is not None), max_livox_points=24000,
                  max_map_points=200000)
    try:
        c.map_set_local(0, cm)
        c.map_set_local(1, sm)
        for s in range(B):
            c.scan_upload(s, cases[assign[s]]["velo"], cases[assign[s]]["livox"])
        dR = np.stack([cases[k]["dR"].reshape(9) for k in assign])
        dt = np.stack([cases[k]["dt"] for k in assign])
        x0 = np.stack([cases[k]["x0"] for k in assign])
        first = np.array([int(np.argmax(assign == k)) for k in range(nd)])
        want = first[assign]
        bad, ref, t0 = 0, None, time.time()
        for r in range(rounds):
            if lanes is not None and lanes[r % len(lanes)] is not None:   # (None: the lanes the context has)
                c.set_lanes(lanes[r % len(lanes)])
            x = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
            dg = c.slot_digest(0, B)
            diff = (dg != dg[want]).any(axis=1) | (x != x[want]).any(axis=1)
            if ref is None:
                ref = dg.copy()
            drift = int((dg != ref).any(axis=1).sum())
            if diff.any() or drift:
                bad += 1
                ex = [(int(s), int(assign[s]), [PIECES[w] for w in range(10) if dg[s][w] != dg[want[s]][w]]) for s in np.nonzero(diff)[0][:4]]
                log("round %d: %d replicas differ %s; %d slots differ from round 0" % (r, int(diff.sum()), ex, drift))
        return bad, (time.time() - t0) / max(rounds, 1)
    finally:
        c.close()


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mml_oracle as O_
    M_ = importlib.import_module("multi-modal-loam_amd")
    synth_ = importlib.import_module("multi-modal-loam_amd.synth")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ln = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
    bad_, per = run(M_, O_, synth_, n, ln)
    print("lanes=%s (env MML_LANES=%s) serialize=%s: %d rounds, %d bad, %.3f s per round" % (
        ln, os.environ.get("MML_LANES", "default"), os.environ.get("AMD_SERIALIZE_KERNEL", "0"), n, bad_, per))
    sys.exit(1 if bad_ else 0)
