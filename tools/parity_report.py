"""Measured maxima of every float parity bar (run on the GPU box): drives `pytest -m gpu` with the ledger of tests/parity.py
switched on and writes the table to profiles/parity_<tag>.txt (default tag: r02).

    python tools/parity_report.py [tag] [extra pytest args]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r02"
extra = [a for a in sys.argv[1:] if a != tag]
out = os.path.join(ROOT, "profiles", f"parity_{tag}.txt")
env = dict(os.environ, OMX_PARITY_REPORT=out)
r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "gpu", "-p", "no:cacheprovider"] + extra,
                   env=env, cwd=ROOT)
print(open(out).read() if os.path.exists(out) else "no ledger written")
sys.exit(r.returncode)
