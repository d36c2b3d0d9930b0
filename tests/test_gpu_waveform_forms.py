"""The two waveform kernels — role per wavefront (waveform_roles_kernels.hip, the default with band analysis) and one wavefront per
four streams (waveform_kernels.hip; OMX_WAVEFORM_SINGLE=1 pins it) — run the same operations on every value in the same order:
their columns and previews must be BIT-identical.  Each form runs in its own process (the pin is read once per process) over the same
seeded call sequences: odd frame counts (short last batch and round), several calls (ring wrap, refresh of the Kahan pairs,
carried filter / min-max state), 2 and 6 channels, NaN / inf samples, RMS history on and off, a bank that is not a multiple of the
four streams of a workgroup."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import openmeters_amd
from openmeters_amd import banks, capi
api = openmeters_amd.api()
out = {}
cases = [(3, 2, False, 48000.0, [4800, 257, 1, 4095, 33]), (7, 2, True, 48000.0, [1023, 4800, 15, 2048]), (2, 6, True, 44100.0, [3000, 3001, 77]),
         (5, 2, True, 8000.0, [999, 2500, 64])]
for ci, (S, C, history, rate, calls) in enumerate(cases):
    rng = np.random.default_rng(100 + ci)
    bank = banks.WaveformBank(api, capi.WaveformConfig(sample_rate=rate, scroll_speed=420.0, max_columns=64, analyze_bands=True, track_history=history), S)
    for k, frames in enumerate(calls):
        pcm = (rng.standard_normal((S, frames, C)) * 0.3).astype(np.float32)
        if frames > 100:
            pcm[0, 50, 0] = np.nan
            pcm[S - 1, 77, C - 1] = np.inf
        up = bank.process_host(pcm, C, rate)
        n = int(up.n_columns) if up is not None else 0
        for s in range(S):
            cols, prev = bank.fetch(s, n, with_preview=True)
            out[f"c{ci}_k{k}_s{s}_cols"] = cols.view(np.uint32)
            out[f"c{ci}_k{k}_s{s}_prev"] = prev.view(np.uint32)
np.savez(sys.argv[2], **out)
"""


def run_form(tmp_path, name, env_extra):
    env = dict(os.environ, **env_extra)
    path = str(tmp_path / f"{name}.npz")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return np.load(path)


def test_role_kernel_and_one_wavefront_kernel_are_bit_identical(tmp_path):
    roles = run_form(tmp_path, "roles", {})
    single = run_form(tmp_path, "single", {"OMX_WAVEFORM_SINGLE": "1"})
    assert sorted(roles.files) == sorted(single.files) and len(roles.files) > 100
    produced = 0
    for key in roles.files:
        assert np.array_equal(roles[key], single[key]), key
        produced += int(roles[key].size > 0 and key.endswith("_cols") and roles[key].shape[0] > 0)
    assert produced > 40   # the sequences do emit columns
