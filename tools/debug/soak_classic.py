"""which columns of a soak sequence trip the classic loud-bin bar, and how loud their neighbours are"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import openmeters_amd
from openmeters_amd import capi
import parity
import test_gpu_state_machine as t
seed = int(sys.argv[1])
omx = openmeters_amd.api(); oracle = capi.Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
orig = parity.check_classic
def spy(got, want, **kw):
    tops = [float(o.astype(np.float64).max()) * (156.0 / 65535.0) - 144.0 if len(o) else -144.0 for o in want]
    for i, (h, o) in enumerate(zip(got, want)):
        m = parity.classic_column_metrics(h, o)
        if m["loud_code_diff"] > 1:
            print(f"column {i} of {len(want)}: loud_code_diff {m['loud_code_diff']}, own top {tops[i]:.1f} dB, neighbours {[round(x,1) for x in tops[max(i-2,0):i+3]]}, n_diff {m['n_diff']}")
    return orig(got, want, **kw)
t.check_classic = spy
parity.EXEMPTIONS_ALLOWED = True
try:
    t.test_spectrogram_random_operation_sequences(omx, oracle, seed)
    print("sequence passed")
except AssertionError as e:
    print("FAILED:", str(e)[:300])
