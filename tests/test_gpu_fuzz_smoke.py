"""A short pass of the randomised campaign (tests/gpu_fuzz.py) inside the GPU suite: every section once, a few trials each,
fixed seed (the suite stays deterministic), including two rounds of the large-batch section (96 slots of random ideal / sensor-faithful
scans through the kernels bench.py runs) and one round of the dense-layout batch section (24 slots of 64 / 128-ring scans); the full-window
section runs 320 windows with equality asserted.  The long campaigns over fresh seeds are run by hand
(profiles/r02d_fuzz_campaign.txt)."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_campaign_short_pass():
    seed = 424242
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "gpu_fuzz.py"), "--seed", str(seed), "--lines", "60", "--scans", "4",
                        "--poses", "3", "--cubes", "2", "--dense", "2", "--solves", "8", "--maps", "2", "--windows", "320", "--batch", "2", "--dense-batch", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "seed %d\n%s\n%s" % (seed, r.stdout[-3000:], r.stderr[-3000:])
    for section in ("batch: 2 rounds x 96 slots ok", "dense-batch: 1 rounds x 24 slots ok", "lines:", "scans:", "poses:", "cubes:", "dense:", "solves:", "maps:", "windows:"):
        assert section in r.stdout, (section, r.stdout[-2000:])
    # 320 random full-window problems (IMU factors, priors, missing factors, restarts, windows off the map): the device-resident
    # trust-region loop against the host loop -- iterates, costs, iteration / evaluation counts and termination codes equal
    assert "windows: 320 ok, device loop EQUAL to the host loop (worst |dx| 0.00e+00" in r.stdout, r.stdout[-1500:]
