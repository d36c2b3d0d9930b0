"""Race check of the fused kernels: tools/determinism_stress.py feeds every kernel family (the 4096 pair kernel, the paired 1024 / 2048
kernels, the 8192 and 16384 kernels, the zero-padded and Bluestein paths, the classic kernel, the spectrum kernels) the same input
from fresh banks several times; every output must be bit-identical run to run — a missing barrier or an unordered LDS exchange shows
up as a run that differs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_kernel_family_is_bit_identical_run_to_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_stress.py"), "6", "96"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "all deterministic" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
