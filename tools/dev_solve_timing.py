"""Dev aid (needs a library built with the SMARK stamps in k_solve): accumulated shader clocks per section of one k_solve workgroup."""
import ctypes as C
import importlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("multi-modal-loam_amd")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-scans", "0", "--steps", "2", "--warmup", "1", "--kernel-steps", "1"],
                   capture_output=True, text=True, env=dict(os.environ, MML_SOLVE_DUMP="1"))
print(r.stdout[-300:])
